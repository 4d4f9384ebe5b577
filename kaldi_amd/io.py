"""On-disk formats either side of the decode path (host side, no GPU needed):
OpenFst binary HCLG files (fstext/kaldi-fst-io.cc:44-89 ReadFstKaldiGeneric) and Kaldi
lattice archives (lat/kaldi-lattice.cc:62-130, util/kaldi-table TableWriter entries).
Thin wrappers over the C-ABI (kaldi_amd/csrc/fst_io.cc)."""
import ctypes as C

import numpy as np

from . import abi
from ._lib import check, lib


def write_openfst(path, g, fst_type="const", align=False):
    """fst::Fst::Write of an HCLG held as CSR arrays (kaldi_amd.synth.Hclg or anything with
    num_states / start / arc_off / arcs / final)."""
    arcs = np.ascontiguousarray(g.arcs)
    off = np.ascontiguousarray(g.arc_off, np.int64)
    fin = np.ascontiguousarray(g.final, np.float32)
    check(lib().kamd_openfst_write(str(path).encode(), {"vector": 0, "const": 1}[fst_type], int(align),
                                   g.num_states, g.start, abi.iptr(off, C.c_int64),
                                   arcs.ctypes.data_as(C.c_void_p), abi.fptr(fin)))


class FstArrays:
    """CSR view of an FST read from disk (same field names as synth.Hclg)."""

    def __init__(self, num_states, start, arc_off, arcs, final):
        self.num_states, self.start, self.arc_off, self.arcs, self.final = num_states, start, arc_off, arcs, final

    @property
    def num_arcs(self):
        return int(self.arc_off[-1])


def read_openfst(path):
    n, st = C.c_int32(), C.c_int32()
    off, fin = C.POINTER(C.c_int64)(), C.POINTER(C.c_float)()
    arcs = C.c_void_p()
    check(lib().kamd_openfst_read(str(path).encode(), C.byref(n), C.byref(st), C.byref(off), C.byref(arcs),
                                  C.byref(fin)))
    try:
        S = n.value
        a_off = np.ctypeslib.as_array(off, (S + 1,)).copy()
        A = int(a_off[-1])
        a = np.zeros(A, abi.ARC_DTYPE)
        if A:
            C.memmove(a.ctypes.data, arcs.value, A * abi.ARC_DTYPE.itemsize)
        f = np.ctypeslib.as_array(fin, (max(S, 1),)).copy()[:S]
    finally:
        lib().kamd_host_free(C.cast(off, C.c_void_p))
        lib().kamd_host_free(arcs)
        lib().kamd_host_free(C.cast(fin, C.c_void_p))
    return FstArrays(S, st.value, a_off, a, f)


def lattice_arrays(lat, acoustic_scale=1.0):
    """(start, final[2S], arcs) of a decoder Lattice (kaldi_amd.decoder.Lattice / oracle
    Lattice).  acoustic_scale != 1 divides the acoustic costs by it, the
    ScaleLattice(AcousticLatticeScale(1/acoustic_scale)) step of
    decoder/decoder-wrappers.cc:251-256."""
    S = lat.frame.size
    fin = np.full(2 * S, np.inf, np.float32)
    is_final = np.isfinite(lat.final)
    fin[0::2][is_final] = lat.final[is_final]
    fin[1::2][is_final] = 0.0
    arcs = lat.arcs.copy()
    if acoustic_scale != 1.0:
        arcs["acoustic_cost"] = arcs["acoustic_cost"] / np.float32(acoustic_scale)
    return int(lat.start), fin, arcs


def write_lattice(path, key, lat, binary=True, append=True, acoustic_scale=1.0):
    start, fin, arcs = lattice_arrays(lat, acoustic_scale)
    check(lib().kamd_lattice_write(str(path).encode(), int(append), key.encode(), int(binary), lat.frame.size, start,
                                   abi.fptr(fin), arcs.ctypes.data_as(C.c_void_p), arcs.size))


def read_lattices(path):
    """Yields (key, start, final[2S], arcs) for every entry of a lattice archive."""
    off = C.c_int64(0)
    key = C.create_string_buffer(4096)
    while True:
        n, st, m = C.c_int32(), C.c_int32(), C.c_int32()
        fin = C.POINTER(C.c_float)()
        arcs = C.c_void_p()
        rc = lib().kamd_lattice_read(str(path).encode(), C.byref(off), key, 4096, C.byref(n), C.byref(st), C.byref(fin),
                                     C.byref(arcs), C.byref(m))
        if rc == 1:
            return
        check(rc)
        try:
            f = np.ctypeslib.as_array(fin, (max(2 * n.value, 1),)).copy()[:2 * n.value]
            a = np.zeros(m.value, abi.LAT_ARC_DTYPE)
            if m.value:
                C.memmove(a.ctypes.data, arcs.value, m.value * abi.LAT_ARC_DTYPE.itemsize)
        finally:
            lib().kamd_host_free(C.cast(fin, C.c_void_p))
            lib().kamd_host_free(arcs)
        yield key.value.decode(), st.value, f, a


class CompactLattice:
    """Determinized word lattice (kaldi::CompactLattice): acceptor on word labels, every arc
    and final weight carries (graph, acoustic) costs and a transition-id string."""

    def __init__(self, handle, owned=True):
        self._h, self._owned = handle, owned         # owned=False: borrowed from a batch decoder
        n, m, k, st, ok = (C.c_int32() for _ in range(5))
        check(lib().kamd_compact_lattice_sizes(handle, C.byref(n), C.byref(m), C.byref(k), C.byref(st), C.byref(ok)))
        self.num_states, self.start, self.reached_beam = n.value, st.value, bool(ok.value)
        self.final = np.zeros(2 * n.value, np.float32)
        self.final_str_begin = np.zeros(n.value, np.int32)
        self.final_str_len = np.zeros(n.value, np.int32)
        self.arcs = np.zeros(m.value, abi.CLAT_ARC_DTYPE)
        self.strings = np.zeros(k.value, np.int32)
        check(lib().kamd_compact_lattice_get(handle, abi.fptr(self.final), abi.iptr(self.final_str_begin),
                                             abi.iptr(self.final_str_len), self.arcs.ctypes.data_as(C.c_void_p),
                                             abi.iptr(self.strings)))

    def arc_string(self, i):
        a = self.arcs[i]
        return self.strings[a["str_begin"]:a["str_begin"] + a["str_len"]]

    def final_string(self, s):
        return self.strings[self.final_str_begin[s]:self.final_str_begin[s] + self.final_str_len[s]]

    def write(self, path, key, binary=True, append=True, acoustic_scale=1.0):
        check(lib().kamd_compact_lattice_write(str(path).encode(), int(append), key.encode(), int(binary), self._h,
                                               float(acoustic_scale)))

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_owned", True):
            lib().kamd_compact_lattice_destroy(self._h)
        self._h = None


def determinize_opts_default():
    o = abi.DeterminizeOpts()
    lib().kamd_determinize_opts_default(C.byref(o))
    return o


def determinize_lattice(lat, beam, tid_phone=None, opts=None):
    """DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1484-1509) of a
    raw lattice.  tid_phone[tid] = phone entered by that transition-id, 0 otherwise."""
    from ._lib import KamdError
    start, fin, arcs = lattice_arrays(lat)
    o = opts or determinize_opts_default()
    if tid_phone is None:
        o.phone_determinize = 0
        tp, nt = None, 0
    else:
        tp = np.ascontiguousarray(tid_phone, np.int32)
        nt = tp.size - 1
    h = lib().kamd_lattice_determinize_phone_pruned(lat.frame.size, start, abi.fptr(fin), arcs.ctypes.data_as(C.c_void_p),
                                                    arcs.size, abi.iptr(tp) if tp is not None else None, nt,
                                                    float(beam), C.byref(o))
    if not h:
        raise KamdError(lib().kamd_last_error().decode())
    return CompactLattice(h)


def read_wave(path):
    """WaveData::Read (feat/wave-reader.cc:272-318): (samp_freq, data[num_channels, num_samples])
    with samples in int16 range."""
    sf, nc, ns = C.c_float(), C.c_int32(), C.c_int64()
    p = C.POINTER(C.c_float)()
    check(lib().kamd_wave_read(str(path).encode(), C.byref(sf), C.byref(nc), C.byref(ns), C.byref(p)))
    try:
        a = np.ctypeslib.as_array(p, (nc.value * ns.value,)).copy().reshape(nc.value, ns.value)
    finally:
        lib().kamd_host_free(C.cast(p, C.c_void_p))
    return sf.value, a


def read_matrix_ark(path):
    """Yields (key, matrix) for every entry of a Kaldi float-matrix archive (binary, compressed
    or text)."""
    off = C.c_int64(0)
    key = C.create_string_buffer(4096)
    while True:
        r, c = C.c_int32(), C.c_int32()
        p = C.POINTER(C.c_float)()
        rc = lib().kamd_ark_read_matrix(str(path).encode(), C.byref(off), key, 4096, C.byref(r), C.byref(c), C.byref(p))
        if rc == 1:
            return
        check(rc)
        try:
            n = r.value * c.value
            a = np.ctypeslib.as_array(p, (max(n, 1),)).copy()[:n].reshape(r.value, c.value)
        finally:
            lib().kamd_host_free(C.cast(p, C.c_void_p))
        yield key.value.decode(), a


def write_matrix_ark(path, key, matrix, binary=True, append=True):
    m = np.ascontiguousarray(matrix, np.float32)
    check(lib().kamd_ark_write_matrix(str(path).encode(), int(append), key.encode(), int(binary), m.shape[0], m.shape[1],
                                      abi.fptr(m)))


def read_int32_vector_ark(path):
    off = C.c_int64(0)
    key = C.create_string_buffer(4096)
    while True:
        n = C.c_int32()
        p = C.POINTER(C.c_int32)()
        rc = lib().kamd_ark_read_int32_vector(str(path).encode(), C.byref(off), key, 4096, C.byref(n), C.byref(p))
        if rc == 1:
            return
        check(rc)
        try:
            a = np.ctypeslib.as_array(p, (max(n.value, 1),)).copy()[:n.value]
        finally:
            lib().kamd_host_free(C.cast(p, C.c_void_p))
        yield key.value.decode(), a
