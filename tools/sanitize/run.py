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
print("sanitized host paths ok; raw lattice", S, la.size)
