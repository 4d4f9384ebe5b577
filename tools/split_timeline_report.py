import csv,glob,sys
d=sys.argv[1]
rows=[]
for f in glob.glob(d+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-40:],r.get("Queue_Id","?")))
rows.sort()
q=[i for i,r in enumerate(rows) if "DecodeQueueKernel" in r[2]]
# last step = last two queue kernels (long + main)
a=q[-2]; 
# find the FeatKernel before
i0=max(i for i in range(a) if "FeatKernel" in rows[i][2])
t0=rows[i0][0]
last=None
for s,e,n,qid in rows[i0:q[-1]+1]:
    if "Decode" in n or "Feat" in n or "Pull" in n or (e-s)>1.5e6:
        print("%9.3f - %9.3f q%s %s"%((s-t0)/1e6,(e-t0)/1e6,qid,n))
g=[(s,e) for s,e,n,qid in rows[i0:q[-1]+1] if "Gemm" in n]
print("gemm kernels",len(g),"busy %.2f ms"%(sum(e-s for s,e in g)/1e6),"span %.2f-%.2f"%((g[0][0]-t0)/1e6,(g[-1][1]-t0)/1e6))
