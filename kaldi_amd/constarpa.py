"""ConstArpaLm (lm/const-arpa-lm.h:211-352) and lattice rescoring with it
(latbin/lattice-lmrescore-const-arpa.cc), over kamd_const_arpa_* (host code, no GPU)."""
import ctypes as C

import numpy as np

from . import abi, latbin
from ._lib import KamdError, check, lib

FLT_MIN = float(np.finfo(np.float32).tiny)


class ConstArpaLm:
    def __init__(self, handle):
        if not handle:
            raise KamdError(lib().kamd_last_error().decode())
        self._h = handle
        v = [C.c_int32() for _ in range(5)]
        n = C.c_int64()
        check(lib().kamd_const_arpa_info(handle, *[C.byref(x) for x in v], C.byref(n)))
        self.bos, self.eos, self.unk, self.order, self.num_words = (x.value for x in v)
        self.lm_states_size = n.value

    @classmethod
    def build(cls, arpa_path, bos, eos, unk=-1, words_txt=None):
        """arpa-to-const-arpa --bos-symbol --eos-symbol --unk-symbol (BuildConstArpaLm)"""
        return cls(lib().kamd_const_arpa_build(str(arpa_path).encode(), int(bos), int(eos), int(unk),
                                               None if words_txt is None else str(words_txt).encode()))

    @classmethod
    def read(cls, path):
        return cls(lib().kamd_const_arpa_read(str(path).encode()))

    def write(self, path):
        check(lib().kamd_const_arpa_write(self._h, str(path).encode()))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_const_arpa_destroy(self._h)
            self._h = None

    def GetNgramLogprob(self, word, hist):
        h = np.ascontiguousarray(hist, np.int32)
        return lib().kamd_const_arpa_ngram_logprob(self._h, int(word), abi.iptr(h) if h.size else None, int(h.size))

    def sentence_cost(self, words):
        """-log P(<s> words </s>): what composing a sentence acceptor with the LM FST gives
        (ConstArpaLmDeterministicFst: arcs -logprob, final -logprob(</s>))"""
        hist, cost = [self.bos], 0.0
        for w in list(words) + [self.eos]:
            lp = self.GetNgramLogprob(w, hist)
            if lp == FLT_MIN:
                return float("inf")
            cost -= lp
            hist.append(w)
        return cost

    def rescore(self, lat, lm_scale=1.0):
        """lattice-lmrescore-const-arpa on a latbin.Lat (compact lattice); returns a new latbin.Lat or None when the
        composition is empty ("Empty lattice ... (incompatible LM?)")"""
        if lm_scale == 0.0:
            return lat
        S = len(lat.final)
        fin = np.full(2 * S, np.inf, np.float32)
        fb, fl = np.zeros(S, np.int32), np.zeros(S, np.int32)
        strings, arcs = [], []
        for s in range(S):
            if lat.final[s] is not None:
                g, a, t = lat.final[s]
                fin[2 * s], fin[2 * s + 1] = g, a
                fb[s], fl[s] = len(strings), len(t)
                strings.extend(t)
            for d, wd, g, a, t in lat.arcs[s]:
                arcs.append((s, d, wd, g, a, len(strings), len(t)))
                strings.extend(t)
        A = np.zeros(len(arcs), abi.CLAT_ARC_DTYPE)
        for k, r in enumerate(arcs):
            A[k] = r
        st = np.ascontiguousarray(strings if strings else [0], np.int32)
        h = lib().kamd_compact_lattice_lmrescore_const_arpa(S, lat.start, abi.fptr(fin), abi.iptr(fb), abi.iptr(fl),
                                                            A.ctypes.data_as(C.c_void_p), A.size, abi.iptr(st), self._h, float(lm_scale))
        if not h:
            msg = lib().kamd_last_error().decode()
            if "Empty lattice" in msg:
                return None
            raise KamdError(msg)
        from .io import CompactLattice
        cl = CompactLattice(h)
        out = latbin.Lat(cl.start)
        for s in range(cl.num_states):
            out.add_state()
            if np.isfinite(cl.final[2 * s]):
                out.final[s] = (cl.final[2 * s], cl.final[2 * s + 1], cl.final_string(s).tolist())
        for k in range(cl.arcs.size):
            a = cl.arcs[k]
            out.arcs[int(a["src"])].append((int(a["dst"]), int(a["label"]), a["graph_cost"], a["acoustic_cost"], cl.arc_string(k).tolist()))
        return out


def parse_arpa(path, words_txt=None, cap=100000):
    """ArpaFileParser::Read: (ngram counts, [(line, words, logprob, backoff)]) with natural-log values"""
    counts = np.zeros(16, np.int32)
    nc, n = C.c_int32(), C.c_int32()
    lines, orders = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    words = np.zeros((cap, 8), np.int32)
    lp, bo = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    check(lib().kamd_arpa_parse(str(path).encode(), None if words_txt is None else str(words_txt).encode(), abi.iptr(counts), 16,
                                C.byref(nc), abi.iptr(lines), abi.iptr(orders), abi.iptr(words), abi.fptr(lp), abi.fptr(bo), cap, C.byref(n)))
    return counts[:nc.value].tolist(), [(int(lines[i]), words[i, :orders[i]].tolist(), float(lp[i]), float(bo[i])) for i in range(n.value)]
