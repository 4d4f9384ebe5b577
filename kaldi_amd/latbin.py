"""The lattice end of local/score.sh on the host (pure Python, no device work): reading and writing Kaldi lattice
archives of either kind, lattice-scale, lattice-add-penalty, lattice-best-path, compute-wer.

  lat/kaldi-lattice.cc:62-130,321-420    text / binary forms (OpenFst vector FST over lattice4 / compactlattice44 arcs)
  latbin/lattice-scale.cc:40-110         ScaleLattice with {{lm, acoustic2lm}, {lm2acoustic, acoustic}}
  latbin/lattice-add-penalty.cc:30-80    AddWordInsPenToCompactLattice (lat/lattice-functions.cc): + penalty on arcs with a word
  latbin/lattice-best-path.cc:30-140     scale, CompactLatticeShortestPath, words / alignment / total weight
  bin/compute-wer.cc:30-140, util/edit-distance-inl.h:79-127   %WER / %SER lines with the reference's tie-breaking of ins / del / sub

A lattice here is the CompactLattice view: states with final (graph, acoustic, transition-ids) and arcs
(src, dst, word, graph, acoustic, transition-ids); a raw Lattice is converted on reading (ConvertLattice:
the arc's transition-id becomes a one-element string)."""
import struct
import sys

import numpy as np

from . import table
from ._lib import KamdError

INF = float("inf")


class Lat:
    def __init__(self, start=-1):
        self.start = start
        self.final = []          # per state: None or (graph, acoustic, [tids])
        self.arcs = []           # per state: [(dst, word, graph, acoustic, [tids])]

    def add_state(self):
        self.final.append(None); self.arcs.append([])
        return len(self.final) - 1


def _parse_binary(buf, pos):
    def i32():
        nonlocal pos
        v = struct.unpack_from("<i", buf, pos)[0]; pos += 4
        return v

    def i64():
        nonlocal pos
        v = struct.unpack_from("<q", buf, pos)[0]; pos += 8
        return v

    def f32():
        nonlocal pos
        v = struct.unpack_from("<f", buf, pos)[0]; pos += 4
        return v

    def s():
        nonlocal pos
        n = i32(); v = buf[pos:pos + n].decode(); pos += n
        return v
    if i32() != 2125659606:
        raise KamdError("not an OpenFst binary FST")
    fsttype, arctype = s(), s()
    i32(); flags = i32(); pos += 8
    start, numstates, _ = i64(), i64(), i64()
    if fsttype != "vector" or arctype not in ("lattice4", "compactlattice44") or flags & 3:
        raise KamdError("expected a vector FST over lattice4 / compactlattice44 arcs without symbol tables, got %s / %s" % (fsttype, arctype))
    compact = arctype == "compactlattice44"
    lat = Lat(start)
    for st in range(numstates):
        lat.add_state()
        g, a = f32(), f32()
        tids = [i32() for _ in range(i32())] if compact else []
        if not (g == INF and a == INF):
            lat.final[st] = (g, a, tids)
        for _ in range(i64()):
            il, ol = i32(), i32()
            g, a = f32(), f32()
            if compact:
                tids = [i32() for _ in range(i32())]
                word = il
            else:
                tids = [il] if il != 0 else []
                word = ol
            lat.arcs[st].append((i32(), word, g, a, tids))
    return lat, pos


def _num(x):
    return INF if x == "Infinity" else (-INF if x == "-Infinity" else float(x))


def _parse_text(buf, pos):
    """after "key" came a newline: FstPrinter lines until an empty line"""
    lat = Lat()
    first = True
    while True:
        e = buf.find(b"\n", pos)
        line = buf[pos:e if e >= 0 else len(buf)].decode()
        pos = e + 1 if e >= 0 else len(buf)
        col = line.split()
        if not col:
            break

        def weight(w):
            p = w.split(",")
            tids = [int(x) for x in p[2].split("_")] if len(p) > 2 and p[2] else []
            return _num(p[0]), _num(p[1]), tids, len(p) > 2
        s = int(col[0])
        while len(lat.final) <= s:
            lat.add_state()
        if first:
            lat.start = s; first = False
        if len(col) <= 2:                                   # final state
            g, a, tids, _ = weight(col[1]) if len(col) == 2 else (0.0, 0.0, [], True)
            lat.final[s] = None if (g == INF and a == INF) else (g, a, tids)
        else:
            d = int(col[1])
            while len(lat.final) <= d:
                lat.add_state()
            if len(col) >= 4 and "," not in col[3]:         # Lattice: src dst ilabel olabel [w]   (a weight always has a comma)
                il, ol = int(col[2]), int(col[3])
                g, a, _, _ = weight(col[4]) if len(col) > 4 else (0.0, 0.0, [], False)
                lat.arcs[s].append((d, ol, g, a, [il] if il != 0 else []))
            else:                                           # CompactLattice (acceptor): src dst label [w]
                g, a, tids, _ = weight(col[3]) if len(col) > 3 else (0.0, 0.0, [], True)
                lat.arcs[s].append((d, int(col[2]), g, a, tids))
    return lat, pos


def read_lattices(rspecifier):
    """yields (key, Lat) for an archive of Lattices or CompactLattices, binary or text"""
    kind, rx, _ = table.classify_rspecifier(rspecifier)
    if kind != table.ARCHIVE:
        raise KamdError("lattice tables are read from archives (ark:...), got " + rspecifier)
    with table.Input(rx) as (path, off):
        buf = open(path, "rb").read()
    pos = off
    while True:
        while pos < len(buf) and buf[pos:pos + 1] in (b" ", b"\n"):
            pos += 1
        if pos >= len(buf):
            return
        e = pos
        while e < len(buf) and buf[e:e + 1] not in (b" ", b"\n"):
            e += 1
        key = buf[pos:e].decode()
        if buf[e:e + 1] == b" " and buf[e + 1] == 214:
            lat, pos = _parse_binary(buf, e + 1)
        else:
            nl = buf.index(b"\n", e)
            lat, pos = _parse_text(buf, nl + 1)
        yield key, lat


def _fmt(x):
    return "Infinity" if x == INF else ("-Infinity" if x == -INF else ("%.9g" % np.float32(x)).replace("e+", "e+").rstrip())


def compact_bytes(lat, binary=True):
    """the archive entry after "key ": CompactLattice in binary or text form (kaldi-lattice.cc:62-130, 388-420)"""
    S = len(lat.final)
    if binary:
        b = bytearray()

        def s(x):
            b.extend(struct.pack("<i", len(x))); b.extend(x.encode())
        b.extend(struct.pack("<i", 2125659606)); s("vector"); s("compactlattice44")
        b.extend(struct.pack("<iiQqqq", 2, 0, 3, lat.start if S else -1, S, 0))
        for st in range(S):
            f = lat.final[st]
            if f is None:
                b.extend(struct.pack("<ffi", INF, INF, 0))
            else:
                b.extend(struct.pack("<ffi", f[0], f[1], len(f[2]))); b.extend(struct.pack("<%di" % len(f[2]), *f[2]))
            b.extend(struct.pack("<q", len(lat.arcs[st])))
            for d, w, g, a, tids in lat.arcs[st]:
                b.extend(struct.pack("<iiffi", w, w, g, a, len(tids))); b.extend(struct.pack("<%di" % len(tids), *tids))
                b.extend(struct.pack("<i", d))
        return bytes(b)
    out = ["\n"]

    def state(st):
        wrote = False
        for d, w, g, a, tids in lat.arcs[st]:
            line = "%d\t%d\t%d" % (st, d, w)
            if not (g == 0.0 and a == 0.0 and not tids):
                line += "\t%s,%s,%s" % (_fmt(g), _fmt(a), "_".join(str(t) for t in tids))
            out.append(line + "\n"); wrote = True
        f = lat.final[st]
        if f is not None or not wrote:
            if f is None:
                out.append("%d\tInfinity,Infinity,\n" % st)
            elif f[0] == 0.0 and f[1] == 0.0 and not f[2]:
                out.append("%d\n" % st)
            else:
                out.append("%d\t%s,%s,%s\n" % (st, _fmt(f[0]), _fmt(f[1]), "_".join(str(t) for t in f[2])))
    if S and lat.start >= 0:
        state(lat.start)
        for st in range(S):
            if st != lat.start:
                state(st)
    out.append("\n")
    return "".join(out).encode()


def scale(lat, lm_scale=1.0, acoustic_scale=1.0, acoustic2lm_scale=0.0, lm2acoustic_scale=0.0):
    def w(g, a):
        if g == INF and a == INF:
            return g, a
        return (np.float32(lm_scale * g + acoustic2lm_scale * a), np.float32(lm2acoustic_scale * g + acoustic_scale * a))
    for st in range(len(lat.final)):
        if lat.final[st] is not None:
            g, a, t = lat.final[st]
            lat.final[st] = w(g, a) + (t,)
        lat.arcs[st] = [(d, wd) + w(g, a) + (t,) for d, wd, g, a, t in lat.arcs[st]]
    return lat


def add_penalty(lat, word_ins_penalty):
    for st in range(len(lat.final)):
        lat.arcs[st] = [(d, wd, np.float32(g + word_ins_penalty) if wd != 0 else g, a, t) for d, wd, g, a, t in lat.arcs[st]]
    return lat


def best_path(lat):
    """CompactLatticeShortestPath: (words, alignment, graph, acoustic) of the path minimising graph + acoustic, or None"""
    S = len(lat.final)
    if S == 0 or lat.start < 0:
        return None
    indeg = [0] * S
    for st in range(S):
        for d, *_ in lat.arcs[st]:
            indeg[d] += 1
    order, stack = [], [st for st in range(S) if indeg[st] == 0]
    while stack:
        st = stack.pop(); order.append(st)
        for d, *_ in lat.arcs[st]:
            indeg[d] -= 1
            if indeg[d] == 0:
                stack.append(d)
    if len(order) != S:
        raise KamdError("lattice has cycles")
    best = [INF] * S; back = [None] * S
    best[lat.start] = 0.0
    for st in order:
        if best[st] == INF:
            continue
        for k, (d, wd, g, a, t) in enumerate(lat.arcs[st]):
            c = best[st] + float(g) + float(a)
            if c < best[d]:
                best[d] = c; back[d] = (st, k)
    end, tot = None, INF
    for st in range(S):
        if lat.final[st] is not None and best[st] + float(lat.final[st][0]) + float(lat.final[st][1]) < tot:
            tot = best[st] + float(lat.final[st][0]) + float(lat.final[st][1]); end = st
    if end is None:
        return None
    words, ali = [], list(lat.final[end][2])
    g_sum, a_sum = float(lat.final[end][0]), float(lat.final[end][1])
    st = end
    path = []
    while back[st] is not None:
        p, k = back[st]; path.append(lat.arcs[p][k]); st = p
    ali_parts = []
    for d, wd, g, a, t in reversed(path):
        if wd != 0:
            words.append(wd)
        ali_parts += t
        g_sum += float(g); a_sum += float(a)
    return words, ali_parts + ali, g_sum, a_sum


def _topo_order(lat):
    S = len(lat.final)
    indeg = [0] * S
    for st in range(S):
        for d, *_ in lat.arcs[st]:
            indeg[d] += 1
    order, stack = [], [st for st in range(S) if indeg[st] == 0]
    while stack:
        st = stack.pop(); order.append(st)
        for d, *_ in lat.arcs[st]:
            indeg[d] -= 1
            if indeg[d] == 0:
                stack.append(d)
    if len(order) != S:
        raise KamdError("lattice has cycles")
    return order


def nbest(lat, n):
    """lattice-to-nbest (latbin/lattice-to-nbest.cc -> fst::ShortestPath with n paths, lat/lattice-functions): the n
    lowest-cost paths of an acyclic compact lattice as [(words, graph + acoustic cost)], best first.  Best-first search
    with the exact cost-to-go as the heuristic, so the paths come out in order."""
    import heapq
    S = len(lat.final)
    if S == 0 or lat.start < 0 or n <= 0:
        return []
    togo = [INF] * S
    for st in reversed(_topo_order(lat)):
        best = INF if lat.final[st] is None else float(lat.final[st][0]) + float(lat.final[st][1])
        for d, wd, g, a, t in lat.arcs[st]:
            best = min(best, float(g) + float(a) + togo[d])
        togo[st] = best
    if togo[lat.start] == INF:
        return []
    out, tie = [], 0
    heap = [(togo[lat.start], 0, 0.0, lat.start, ())]        # (estimate, tie-break, cost so far, state or -1 = finished, words)
    while heap and len(out) < n:
        est, _, cost, st, words = heapq.heappop(heap)
        if st < 0:
            out.append((list(words), cost))
            continue
        if lat.final[st] is not None:
            tie += 1
            c = cost + float(lat.final[st][0]) + float(lat.final[st][1])
            heapq.heappush(heap, (c, tie, c, -1, words))
        for d, wd, g, a, t in lat.arcs[st]:
            if togo[d] == INF:
                continue
            tie += 1
            c = cost + float(g) + float(a)
            heapq.heappush(heap, (c + togo[d], tie, c, d, words + (wd,) if wd != 0 else words))
    return out


def oracle_errors(lat, ref):
    """lattice-oracle (latbin/lattice-oracle.cc): the smallest edit distance between `ref` and the word sequence of ANY
    path through the lattice (the lattice's oracle error count), by dynamic programming over (state, reference position)."""
    S, R = len(lat.final), len(ref)
    if S == 0 or lat.start < 0:
        return R
    cost = [[INF] * (R + 1) for _ in range(S)]
    row = cost[lat.start]
    for j in range(R + 1):
        row[j] = j                                            # j reference words deleted before the first arc
    best = INF
    for st in _topo_order(lat):
        row = cost[st]
        for j in range(R):                                    # deletions inside this state
            if row[j] + 1 < row[j + 1]:
                row[j + 1] = row[j] + 1
        if row[0] == INF and min(row) == INF:
            continue
        if lat.final[st] is not None:
            best = min(best, row[R])
        for d, wd, g, a, t in lat.arcs[st]:
            nxt = cost[d]
            if wd == 0:
                for j in range(R + 1):
                    if row[j] < nxt[j]:
                        nxt[j] = row[j]
            else:
                for j in range(R + 1):
                    c = row[j]
                    if c == INF:
                        continue
                    if c + 1 < nxt[j]:
                        nxt[j] = c + 1                        # the arc's word is an insertion
                    if j < R:
                        e = c + (0 if ref[j] == wd else 1)
                        if e < nxt[j + 1]:
                            nxt[j + 1] = e
    return R if best == INF else best


def edit_distance(ref, hyp):
    """LevenshteinEditDistance with counts, the reference's recursion and tie-breaking (util/edit-distance-inl.h:79-127)"""
    e = [(i, 0, i, 0) for i in range(len(ref) + 1)]             # (total, ins, del, sub)
    for h in hyp:
        cur = [(e[0][0] + 1, e[0][1] + 1, e[0][2], e[0][3])]
        for r in range(1, len(ref) + 1):
            ins_err, del_err = e[r][0] + 1, cur[r - 1][0] + 1
            diff = h != ref[r - 1]
            sub_err = e[r - 1][0] + (1 if diff else 0)
            if sub_err < ins_err and sub_err < del_err:
                t = e[r - 1]; cur.append((sub_err, t[1], t[2], t[3] + (1 if diff else 0)))
            elif del_err < ins_err:
                t = cur[r - 1]; cur.append((del_err, t[1], t[2] + 1, t[3]))
            else:
                t = e[r]; cur.append((ins_err, t[1] + 1, t[2], t[3]))
        e = cur
    return e[-1]


def compute_wer(ref, hyp, mode="strict"):
    """ref / hyp: {key: [tokens]} (ref ordered).  Returns the three lines compute-wer prints."""
    if mode not in ("strict", "present", "all"):
        raise KamdError('--mode option invalid: expected "present"|"all"|"strict", got ' + mode)
    nw = we = ns = se = ni = nd = nsub = absent = 0
    for key, r in ref.items():
        if key not in hyp:
            if mode == "strict":
                raise KamdError("No hypothesis for key %s and strict mode specifier." % key)
            absent += 1
            if mode == "present":
                continue
            h = []
        else:
            h = hyp[key]
        nw += len(r)
        tot, i, d, s = edit_distance(r, h)
        we += tot; ni += i; nd += d; nsub += s; ns += 1; se += int(r != h)
    wer = 100.0 * np.float32(we) / np.float32(nw) if nw else float("nan")
    ser = 100.0 * np.float32(se) / np.float32(ns) if ns else float("nan")
    return ("%%WER %.2f [ %d / %d, %d ins, %d del, %d sub ]%s" % (wer, we, nw, ni, nd, nsub, " [PARTIAL]" if absent else ""),
            "%%SER %.2f [ %d / %d ]" % (ser, se, ns), "Scored %d sentences, %d not present in hyp." % (ns, absent))


# ---- command lines ---------------------------------------------------------------------------
def main(prog, argv):
    po = table.ParseOptions(prog)
    if prog == "lattice-scale":
        po.register("write-compact", bool, True, "(lattices are always written in compact form)")
        po.register("acoustic-scale", float, 1.0); po.register("inv-acoustic-scale", float, 1.0)
        po.register("lm-scale", float, 1.0); po.register("acoustic2lm-scale", float, 0.0); po.register("lm2acoustic-scale", float, 0.0)
        a = po.read(argv)
        if len(a) != 2:
            po.print_usage(); return 1
        if po["inv-acoustic-scale"] != 1.0 and po["acoustic-scale"] != 1.0:
            raise KamdError("You cannot specify both --acoustic-scale and --inv-acoustic-scale")
        ac = 1.0 / po["inv-acoustic-scale"] if po["inv-acoustic-scale"] != 1.0 else po["acoustic-scale"]
        n = 0
        with table.TableWriter(a[1], "raw") as w:
            for key, lat in read_lattices(a[0]):
                w.write(key, compact_bytes(scale(lat, po["lm-scale"], ac, po["acoustic2lm-scale"], po["lm2acoustic-scale"]), w.opts["binary"])); n += 1
        print("LOG Done %d lattices." % n, file=sys.stderr)
        return 0 if n else 1
    if prog == "lattice-add-penalty":
        po.register("word-ins-penalty", float, 0.0, "Word insertion penalty")
        a = po.read(argv)
        if len(a) != 2:
            po.print_usage(); return 1
        n = 0
        with table.TableWriter(a[1], "raw") as w:
            for key, lat in read_lattices(a[0]):
                w.write(key, compact_bytes(add_penalty(lat, po["word-ins-penalty"]), w.opts["binary"])); n += 1
        print("LOG Done adding word insertion penalty to %d lattices." % n, file=sys.stderr)
        return 0 if n else 1
    if prog == "lattice-best-path":
        po.register("acoustic-scale", float, 1.0); po.register("lm-scale", float, 1.0); po.register("word-symbol-table", str, "")
        a = po.read(argv)
        if not 1 <= len(a) <= 3:
            po.print_usage(); return 1
        ww = table.TableWriter(a[1], "int32") if len(a) > 1 and a[1] else None
        aw = table.TableWriter(a[2], "int32") if len(a) > 2 and a[2] else None
        n_done = n_fail = 0
        tot_w = 0.0; frames = 0
        for key, lat in read_lattices(a[0]):
            bp = best_path(scale(lat, po["lm-scale"], po["acoustic-scale"]))
            if bp is None:
                print("WARNING Best-path failed for key " + key, file=sys.stderr); n_fail += 1
                continue
            words, ali, g, ac = bp
            if ww:
                ww.write(key, words)
            if aw:
                aw.write(key, ali)
            n_done += 1; tot_w += g + ac; frames += len(ali)
            print("LOG For utterance %s, best cost %g + %g = %g over %d frames." % (key, g, ac, g + ac, len(ali)), file=sys.stderr)
        for w in (ww, aw):
            if w:
                w.close()
        print("LOG Overall cost per frame is %g over %d frames; done %d lattices, failed for %d" % (tot_w / max(frames, 1), frames, n_done, n_fail), file=sys.stderr)
        return 0 if n_done else 1
    if prog == "compute-wer":
        po.register("mode", str, "strict"); po.register("text", bool, False, "Deprecated option! Keeping for compatibility reasons.")
        a = po.read(argv)
        if len(a) != 2:
            po.print_usage(); return 1
        ref = dict(table.SequentialTableReader(a[0], "tokens"))
        hyp = dict(table.SequentialTableReader(a[1], "tokens"))
        for line in compute_wer(ref, hyp, po["mode"]):
            print(line)
        return 0
    raise KamdError("unknown program " + prog)


def run(prog):
    try:
        sys.exit(main(prog, sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
