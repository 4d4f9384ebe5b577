# drive the host-only entry points through ctypes with the sanitized library
import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'libhost_asan.so'))
from kaldi_amd import abi, synth
from oracle import orc
import tempfile
tmp = tempfile.mkdtemp()
g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=1)
arcs = np.ascontiguousarray(g.arcs); off = np.ascontiguousarray(g.arc_off, np.int64); fin = np.ascontiguousarray(g.final, np.float32)
for t, al in ((0, 0), (1, 0), (1, 1)):
    p = os.path.join(tmp, "g%d%d.fst" % (t, al)).encode()
    assert L.kamd_openfst_write(p, t, al, g.num_states, g.start, off.ctypes.data_as(C.c_void_p), arcs.ctypes.data_as(C.c_void_p), fin.ctypes.data_as(C.c_void_p)) == 0
    n, st = C.c_int32(), C.c_int32(); po, pa, pf = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.kamd_openfst_read(p, C.byref(n), C.byref(st), C.byref(po), C.byref(pa), C.byref(pf)) == 0
    assert n.value == g.num_states
    for q in (po, pa, pf): L.kamd_host_free(q)
    open(p, "ab").truncate(os.path.getsize(p) - 9)
    assert L.kamd_openfst_read(p, C.byref(n), C.byref(st), C.byref(po), C.byref(pa), C.byref(pf)) != 0
ll, _, _ = synth.sample_utterance(g, n_words=3, seed=2, peak=2.0)
cfg = abi.decoder_config_recipe(); cfg.lattice_beam = 6.0
d = orc.Decoder(g, cfg, 1); d.Decode(ll); lat = d.GetRawLattice()
S = lat.frame.size
f2 = np.full(2 * S, np.inf, np.float32); m = np.isfinite(lat.final); f2[0::2][m] = lat.final[m]; f2[1::2][m] = 0
la = np.ascontiguousarray(lat.arcs)
for binary in (0, 1):
    p = os.path.join(tmp, "lat%d" % binary).encode()
    assert L.kamd_lattice_write(p, 0, b"k1", binary, S, int(lat.start), f2.ctypes.data_as(C.c_void_p), la.ctypes.data_as(C.c_void_p), la.size) == 0
    assert L.kamd_lattice_write(p, 1, b"k2", binary, S, int(lat.start), f2.ctypes.data_as(C.c_void_p), la.ctypes.data_as(C.c_void_p), la.size) == 0
    offp = C.c_int64(0); key = C.create_string_buffer(64)
    cnt = 0
    while True:
        n, st, na = C.c_int32(), C.c_int32(), C.c_int32(); pf, pa = C.c_void_p(), C.c_void_p()
        rc = L.kamd_lattice_read(p, C.byref(offp), key, 64, C.byref(n), C.byref(st), C.byref(pf), C.byref(pa), C.byref(na))
        if rc == 1: break
        assert rc == 0 and n.value == S and na.value == la.size
        L.kamd_host_free(pf); L.kamd_host_free(pa); cnt += 1
    assert cnt == 2
tp = np.zeros(g.tid2pdf.size, np.int32); tp[1::2] = np.arange(1, (g.tid2pdf.size - 1) // 2 + 1)
L.kamd_lattice_determinize_phone_pruned.restype = C.c_void_p
for beam in (1e30, 3.0, 0.5):
    h = L.kamd_lattice_determinize_phone_pruned(S, int(lat.start), f2.ctypes.data_as(C.c_void_p), la.ctypes.data_as(C.c_void_p), la.size, tp.ctypes.data_as(C.c_void_p), tp.size - 1, C.c_double(beam), None)
    assert h
    for binary in (0, 1):
        assert L.kamd_compact_lattice_write(os.path.join(tmp, "c%d" % binary).encode(), 0, b"u", binary, C.c_void_p(h), C.c_float(0.5)) == 0
    L.kamd_compact_lattice_destroy(C.c_void_p(h))
# wave + archives
import struct, wave
wp = os.path.join(tmp, "a.wav")
x = np.random.default_rng(0).integers(-30000, 30000, (777, 2)).astype("<i2")
with wave.open(wp, "wb") as w:
    w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
sf, nc, ns = C.c_float(), C.c_int32(), C.c_int64(); pw = C.c_void_p()
assert L.kamd_wave_read(wp.encode(), C.byref(sf), C.byref(nc), C.byref(ns), C.byref(pw)) == 0 and ns.value == 777
L.kamd_host_free(pw)
head = open(wp, "rb").read()[:60]
open(wp, "wb").write(head)
assert L.kamd_wave_read(wp.encode(), C.byref(sf), C.byref(nc), C.byref(ns), C.byref(pw)) == 0     # truncated data: warning only
L.kamd_host_free(pw)
m = np.random.default_rng(1).standard_normal((9, 4)).astype(np.float32)
for binary in (0, 1):
    ap = os.path.join(tmp, "m%d.ark" % binary).encode()
    assert L.kamd_ark_write_matrix(ap, 0, b"a", binary, 9, 4, m.ctypes.data_as(C.c_void_p)) == 0
    assert L.kamd_ark_write_matrix(ap, 1, b"b", binary, 0, 0, m.ctypes.data_as(C.c_void_p)) == 0
    offp = C.c_int64(0); key = C.create_string_buffer(64); cnt = 0
    while True:
        r, c = C.c_int32(), C.c_int32(); pm = C.c_void_p()
        rc = L.kamd_ark_read_matrix(ap, C.byref(offp), key, 64, C.byref(r), C.byref(c), C.byref(pm))
        if rc == 1: break
        assert rc == 0; L.kamd_host_free(pm); cnt += 1
    assert cnt == 2
# extended filenames, specifiers, scp offsets, pipes
for name in ("", "-", "a", "a ", " a", "b|", "|b", "a b c:123", "x:", "ark,s,cs:a b c", "scp:a", "a|b", ":", "1", ":1"):
    L.kamd_classify_rxfilename(name.encode()); L.kamd_classify_wxfilename(name.encode())
buf, buf2 = C.create_string_buffer(8), C.create_string_buffer(8)
o = C.c_int()
for spec in ("ark:foo|", "b,ark:foo|", "scp,scp,b:foo|", "s,scp,no:foo|", "", "scp", "ark:foo ", " t,ark:boo", "t,ark,scp:a b,c,d", "b,ark,scp:,", "ark:a-very-long-name-that-does-not-fit"):
    L.kamd_classify_rspecifier(spec.encode(), buf, 8, C.byref(o))
    L.kamd_classify_wspecifier(spec.encode(), buf, 8, buf2, 8, C.byref(o))
    L.kamd_classify_rspecifier(spec.encode(), None, 0, None)
path = C.create_string_buffer(4096); off64 = C.c_int64(); tmpf = C.c_int()
ap = os.path.join(tmp, "m1.ark")
assert L.kamd_rx_materialize(("cat %s |" % ap).encode(), path, 4096, C.byref(off64), C.byref(tmpf)) == 0 and tmpf.value == 1
r, c = C.c_int32(), C.c_int32(); pm = C.c_void_p(); key = C.create_string_buffer(64); offp = C.c_int64(0)
assert L.kamd_ark_read_matrix(path.value, C.byref(offp), key, 64, C.byref(r), C.byref(c), C.byref(pm)) == 0 and r.value == 9
L.kamd_host_free(pm); os.unlink(path.value)
assert L.kamd_rx_materialize(b"false |", path, 4096, C.byref(off64), C.byref(tmpf)) != 0
assert L.kamd_rx_materialize(("%s:2" % ap).encode(), path, 4096, C.byref(off64), C.byref(tmpf)) == 0 and off64.value == 2
offp = C.c_int64(2)
assert L.kamd_ark_read_matrix(path.value, C.byref(offp), None, 0, C.byref(r), C.byref(c), C.byref(pm)) == 0 and (r.value, c.value) == (9, 4)
L.kamd_host_free(pm)
offp = C.c_int64(10 ** 6)
assert L.kamd_ark_read_matrix(path.value, C.byref(offp), None, 0, C.byref(r), C.byref(c), C.byref(pm)) != 0
assert L.kamd_rx_materialize(ap.encode(), path, 4, C.byref(off64), C.byref(tmpf)) != 0          # buffer too small
print("sanitized host paths ok; raw lattice", S, la.size)
