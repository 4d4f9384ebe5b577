"""Host-side mirror of LatticeFasterDecoder (decoder/lattice-faster-decoder.h:226-343),
DecodableMatrixMapped (decoder/decodable-matrix.h:98-136) and the nnet / graph handles,
over the C-ABI.  Same method names and argument meaning as the reference classes."""
import ctypes as C

import numpy as np

from . import abi
from ._lib import KamdError, check, lib


class Graph:
    """HCLG on device (replaces the fst::Fst<StdArc> handed to the decoder ctor)."""

    def __init__(self, hclg):
        self.hclg = hclg
        arcs = np.ascontiguousarray(hclg.arcs)
        off = np.ascontiguousarray(hclg.arc_off, np.int64)
        fin = np.ascontiguousarray(hclg.final, np.float32)
        self._h = lib().kamd_graph_create(hclg.num_states, hclg.start, abi.iptr(off, C.c_int64),
                                          arcs.ctypes.data_as(C.c_void_p), abi.fptr(fin))
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    @classmethod
    def from_file(cls, path):
        """ReadFstKaldiGeneric (fstext/kaldi-fst-io.cc:44-89): HCLG.fst ("vector" or "const"
        OpenFst binary over StdArc) straight into HBM."""
        self = cls.__new__(cls)
        self.hclg = None
        self._h = lib().kamd_graph_read_openfst(str(path).encode())
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())
        return self

    def num_states(self):
        return lib().kamd_graph_num_states(self._h)

    def num_arcs(self):
        return lib().kamd_graph_num_arcs(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_graph_destroy(self._h)
            self._h = None


class Nnet:
    """The collapsed acoustic model on device (AmNnetSimple after CollapseModel)."""

    def __init__(self, model):
        self.model = model
        self._h = lib().kamd_nnet_create(model.descs(), len(model.layers), model.input_dim,
                                         model.subsampling)
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_nnet_destroy(self._h)
            self._h = None

    def OutputDim(self):
        return lib().kamd_nnet_output_dim(self._h)

    def Context(self):
        return lib().kamd_nnet_left_context(self._h), lib().kamd_nnet_right_context(self._h)

    def Forward(self, feats, ivector=None):
        """All of DecodableNnetSimple::GetOutputForFrame for one utterance."""
        feats = np.ascontiguousarray(feats, np.float32)
        T = feats.shape[0]
        n_out = lib().kamd_nnet_num_output_frames(self._h, T)
        out = np.zeros((n_out, self.OutputDim()), np.float32)
        iv = None if ivector is None else np.ascontiguousarray(ivector, np.float32)
        check(lib().kamd_nnet_forward(self._h, abi.fptr(feats), T, abi.fptr(iv), abi.fptr(out), n_out))
        return out


    def ForwardSlots(self, feats, slot_table, slot_first, period, slices=None):
        """The looped decodable's i-vectors (Round(ivector, period)): first-layer row at time t reads
        slot_table[floor(t / period) - slot_first] (clamped).  slices: [(first input frame, frames)] items of one
        batched forward, each treated like a separate utterance at its edges (default: the whole matrix);
        returns the output rows of each item."""
        ld = (self.model.input_dim + 15) // 16 * 16
        f = np.zeros((feats.shape[0], ld), np.float32)
        f[:, :feats.shape[1]] = feats
        slices = slices or [(0, feats.shape[0])]
        n = len(slices)
        tab = np.ascontiguousarray(slot_table, np.float32)
        in_start = np.asarray([a for a, _ in slices], np.int64)
        in_len = np.asarray([b for _, b in slices], np.int32)
        out_off = np.zeros(n + 1, np.int64)
        for i, (_, b) in enumerate(slices):
            out_off[i + 1] = out_off[i] + lib().kamd_nnet_num_output_frames(self._h, int(b))
        base = np.zeros(n, np.int32); first = np.full(n, slot_first, np.int32); cnt = np.full(n, tab.shape[0], np.int32)
        t0 = np.asarray([a for a, _ in slices], np.int32)
        P = self.OutputDim()
        d_f, d_t, d_o = DeviceMatrix(f), DeviceMatrix(tab), DeviceMatrix(np.zeros((int(out_off[-1]), P), np.float32))
        check(lib().kamd_nnet_forward_slices_slots_device(self._h, d_f.ptr(0), abi.iptr(in_start, C.c_int64), abi.iptr(in_len), ld,
                                                          d_t.ptr(0), tab.shape[0], period, abi.iptr(base), abi.iptr(first),
                                                          abi.iptr(cnt), abi.iptr(t0), n, d_o.ptr(0),
                                                          abi.iptr(out_off, C.c_int64), P, None))
        out = d_o.download()
        return [out[out_off[i]:out_off[i + 1]] for i in range(n)]

    def ForwardInferenceTasks(self, feats_list, ivector_table, tasks):
        """One minibatch of NnetInferenceTasks (kamd_nnet_forward_inference_tasks_device): tasks = [(utterance index, first output
        frame, number of output frames, row of ivector_table or -1)]; returns one [num_output_frames x P] array per task."""
        ld = (self.model.input_dim + 15) // 16 * 16
        in_off = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats_list])]).astype(np.int64)
        feats = np.zeros((int(in_off[-1]), ld), np.float32)
        for u, f in enumerate(feats_list):
            feats[in_off[u]:in_off[u + 1], :f.shape[1]] = f
        arr = (abi.InferenceTask * len(tasks))()
        for i, (u, t, n, r) in enumerate(tasks):
            arr[i] = abi.InferenceTask(int(in_off[u]), int(feats_list[u].shape[0]), int(t), int(n), int(r))
        P = self.OutputDim()
        rows = sum(int(n) for _, _, n, _ in tasks)
        d_f, d_o = DeviceMatrix(feats), DeviceMatrix(np.zeros((rows, P), np.float32))
        d_iv = None if ivector_table is None else DeviceMatrix(np.ascontiguousarray(ivector_table, np.float32))
        check(lib().kamd_nnet_forward_inference_tasks_device(self._h, d_f.ptr(0), ld, d_iv.ptr(0) if d_iv is not None else None,
                                                             0 if d_iv is None else int(np.asarray(ivector_table).shape[1]), arr, len(tasks),
                                                             d_o.ptr(0), P, None))
        out = d_o.download()
        cut = np.concatenate([[0], np.cumsum([int(n) for _, _, n, _ in tasks])])
        return [out[cut[i]:cut[i + 1]] for i in range(len(tasks))]

    def ForwardChunked(self, feats_list, online_ivectors_list, ivector_period=10, frames_per_chunk=50, batch_computer=False):
        """DecodableNnetSimple with online ivectors for a batch of utterances: one ivector per
        chunk (nnet3/nnet-am-decodable-simple.cc:93-214).  Returns one [n_out x P] array each.
        batch_computer: NnetBatchComputer's tasks instead (SplitUtteranceIntoTasks + Compute + MergeTaskOutput,
        nnet3/nnet-batch-compute.cc:774-870: kamd_nnet_forward_tasks_device)."""
        ld = (self.model.input_dim + 15) // 16 * 16
        n = len(feats_list)
        in_off = np.zeros(n + 1, np.int64); iv_off = np.zeros(n + 1, np.int64); out_off = np.zeros(n + 1, np.int64)
        for u, (f, iv) in enumerate(zip(feats_list, online_ivectors_list)):
            in_off[u + 1] = in_off[u] + f.shape[0]
            iv_off[u + 1] = iv_off[u] + iv.shape[0]
            out_off[u + 1] = out_off[u] + lib().kamd_nnet_num_output_frames(self._h, f.shape[0])
        feats = np.zeros((int(in_off[-1]), ld), np.float32)
        for u, f in enumerate(feats_list):
            feats[in_off[u]:in_off[u + 1], :f.shape[1]] = f
        ivs = np.ascontiguousarray(np.concatenate(online_ivectors_list), np.float32)
        P = self.OutputDim()
        d_f, d_iv, d_o = DeviceMatrix(feats), DeviceMatrix(ivs), DeviceMatrix(np.zeros((int(out_off[-1]), P), np.float32))
        fwd = lib().kamd_nnet_forward_tasks_device if batch_computer else lib().kamd_nnet_forward_chunked_device
        check(fwd(self._h, d_f.ptr(0), abi.iptr(in_off, C.c_int64), ld, d_iv.ptr(0), abi.iptr(iv_off, C.c_int64), ivs.shape[1], ivector_period,
                  frames_per_chunk, n, d_o.ptr(0), abi.iptr(out_off, C.c_int64), P, None))
        out = d_o.download()
        return [out[out_off[u]:out_off[u + 1]] for u in range(n)]


class Lattice:
    """Raw lattice (kaldi::Lattice equivalent) in canonical numbering."""

    def __init__(self, start, frame, hclg, cost, final, arcs, num_frames):
        self.start, self.frame, self.hclg, self.cost = start, frame, hclg, cost
        self.final, self.arcs, self.num_frames = final, arcs, num_frames


def lattices_equal(a, b):
    """Exact (bit-level) equality of two canonical raw lattices."""
    if a is None or b is None:
        return a is b
    ok = (a.start == b.start and a.num_frames == b.num_frames
          and np.array_equal(a.frame, b.frame) and np.array_equal(a.hclg, b.hclg)
          and np.array_equal(a.cost.view(np.uint32), b.cost.view(np.uint32))
          and np.array_equal(a.final.view(np.uint32), b.final.view(np.uint32))
          and a.arcs.size == b.arcs.size and a.arcs.tobytes() == b.arcs.tobytes())
    return ok


def lattice_diff(a, b):
    """Human-readable summary of where two canonical lattices differ."""
    out = ["states %d vs %d, arcs %d vs %d, start %d vs %d" % (
        a.frame.size, b.frame.size, a.arcs.size, b.arcs.size, a.start, b.start)]
    ka = set(zip(a.frame.tolist(), a.hclg.tolist()))
    kb = set(zip(b.frame.tolist(), b.hclg.tolist()))
    out.append("tokens only in A: %s" % sorted(ka - kb)[:10])
    out.append("tokens only in B: %s" % sorted(kb - ka)[:10])
    if ka == kb:
        bad = np.nonzero(a.cost.view(np.uint32) != b.cost.view(np.uint32))[0]
        out.append("cost mismatches: %d %s" % (bad.size, [(int(i), float(a.cost[i]), float(b.cost[i])) for i in bad[:5]]))
        sa = set(map(bytes, a.arcs.view("V24")))
        sb = set(map(bytes, b.arcs.view("V24")))
        out.append("arcs only in A: %d, only in B: %d" % (len(sa - sb), len(sb - sa)))
    return "\n".join(out)


class Component:
    """nnet3::Component::Propagate (nnet3/nnet-component-itf.h:130-132) for one fused layer (a `kaldi_amd.nnet.Layer`):
    the compatibility entry for a host that keeps nnet3's own computation.  TdnnComponent::Propagate's shape: `x` holds
    consecutive time steps, the result has x.rows - (max offset - min offset) rows."""

    def __init__(self, layer):
        from .nnet import Model
        self._m = Model([layer], layer.in_dim, 0, 1, layer.out_dim)
        self._h = lib().kamd_component_create(self._m.descs())
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_component_destroy(self._h)
            self._h = None

    def Propagate(self, x):
        x = np.ascontiguousarray(x, np.float32)
        n_out = lib().kamd_component_output_rows(self._h, x.shape[0])
        d_in, d_out = DeviceMatrix(x), DeviceMatrix(np.zeros((max(n_out, 1), self._m.layers[0].out_dim), np.float32))
        got = lib().kamd_component_propagate(self._h, d_in.ptr(0), x.shape[0], x.shape[1], d_out.ptr(0), d_out.cols, None)
        if got < 0:
            raise KamdError(lib().kamd_last_error().decode())
        return d_out.download()[:got]


class DeviceMatrix:
    """A host float32 matrix uploaded to HBM (DecodableMatrixMapped's 'likes' matrix)."""

    def __init__(self, loglikes):
        a = np.ascontiguousarray(loglikes, np.float32)
        self.rows, self.cols = a.shape
        self._d = lib().kamd_malloc(max(a.nbytes, 16))
        if not self._d:
            raise KamdError(lib().kamd_last_error().decode())
        if a.nbytes:
            check(lib().kamd_memcpy_h2d(self._d, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def __del__(self):
        if getattr(self, "_d", None):
            lib().kamd_free(self._d)
            self._d = None

    def ptr(self, row=0):
        return self._d + row * self.cols * 4

    def download(self):
        out = np.zeros((self.rows, self.cols), np.float32)
        if out.nbytes:
            check(lib().kamd_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self._d, out.nbytes))
        return out


class LatticeFasterDecoder:
    """One lane of the device decoder behind the reference's per-utterance API."""

    def __init__(self, graph, config=None, sizes=None, lane=0, _shared=None, tid2pdf=None):
        """tid2pdf: TransitionModel::id2pdf_id_ (index 0 unused); defaults to the graph's own
        table when the graph came from kaldi_amd.synth."""
        self.graph = graph
        self.config = config or abi.decoder_config_default()
        if _shared is not None:
            self._dec, self._own = _shared, False
        else:
            s = sizes
            if s is None:
                s = abi.DecoderSizes()
                lib().kamd_decoder_sizes_default(C.byref(s))
                s.max_lanes = 1
            self.sizes = s
            t2p = np.ascontiguousarray(tid2pdf if tid2pdf is not None else graph.hclg.tid2pdf, np.int32)
            self._t2p = t2p
            self._dec = lib().kamd_decoder_create(graph._h, C.byref(self.config), C.byref(s),
                                                  abi.iptr(t2p), t2p.size - 1)
            if not self._dec:
                raise KamdError(lib().kamd_last_error().decode())
            self._own = True
        self.lane = lane
        self._keep = []

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "_dec", None):
            lib().kamd_decoder_destroy(self._dec)
            self._dec = None

    def SetOptions(self, config):
        self.config = config
        check(lib().kamd_decoder_set_options(self._dec, C.byref(config)))

    def SetSearchMode(self, mode):
        """1 = canonical (default), 2 = canonical-loose (kamd_decoder_set_search_mode)."""
        check(lib().kamd_decoder_set_search_mode(self._dec, int(mode)))

    def SetLevel1Table(self, words):
        """Words of the level-1 (LDS) table region, 0 = none (kamd_decoder_set_level1_table): a test knob, results do not depend on it."""
        check(lib().kamd_decoder_set_level1_table(self._dec, int(words)))

    def InitDecoding(self):
        lanes = np.asarray([self.lane], np.int32)
        check(lib().kamd_decoder_init(self._dec, abi.iptr(lanes), 1, None))
        check(lib().kamd_decoder_sync(self._dec))

    def AdvanceDecoding(self, decodable, max_num_frames=-1):
        """`decodable`: DeviceMatrix or host [frames x pdfs] array of log-likelihoods not yet
        decoded (NumFramesReady - NumFramesDecoded rows)."""
        dm = decodable if isinstance(decodable, DeviceMatrix) else DeviceMatrix(decodable)
        self._keep.append(dm)
        n = dm.rows if max_num_frames < 0 else min(dm.rows, max_num_frames)
        task = abi.DecodeTask(self.lane, n, dm.ptr(0), dm.cols, 0)
        check(lib().kamd_decoder_advance(self._dec, C.byref(task), 1, None))
        check(lib().kamd_decoder_sync(self._dec))
        self._keep.clear()

    def FinalizeDecoding(self):
        lanes = np.asarray([self.lane], np.int32)
        check(lib().kamd_decoder_finalize(self._dec, abi.iptr(lanes), 1, None))
        check(lib().kamd_decoder_sync(self._dec))

    def PruneActiveTokens(self):
        """lattice-faster-decoder.cc:519-546 as a compaction of the lane's arenas (kamd_decoder_compact): what the final
        sweep would drop is dropped now; decoding goes on; the final lattice does not change."""
        compact(self._dec, [self.lane])
        check(lib().kamd_decoder_sync(self._dec))

    def usage(self):
        """(tokens in use, token capacity, links in use, link capacity) of the lane's arenas"""
        return lane_usage(self._dec, self.lane)

    def Decode(self, decodable):
        self.InitDecoding()
        self.AdvanceDecoding(decodable)
        self.FinalizeDecoding()
        return True

    def NumFramesDecoded(self):
        return lib().kamd_decoder_num_frames_decoded(self._dec, self.lane)

    def FinalRelativeCost(self):
        return lib().kamd_decoder_final_relative_cost(self._dec, self.lane)

    def ReachedFinal(self):
        return bool(lib().kamd_decoder_reached_final(self._dec, self.lane))

    def GetRawLattice(self, use_final_probs=True):
        """lattice-faster-decoder.cc:113-196.  After FinalizeDecoding: the pruned raw lattice (use_final_probs must be true,
        as in the reference).  Before it: every token and link the live decoder holds, final costs computed on the spot
        (use_final_probs False: every token of the last frame is final with weight One)."""
        if lib().kamd_decoder_lattice_size(self._dec, self.lane, C.byref(abi.LatticeSize())) == 0:
            if not use_final_probs:
                raise KamdError("You cannot call FinalizeDecoding() and then call GetRawLattice() with use_final_probs == false")
            return get_raw_lattice(self._dec, self.lane)
        return get_live_raw_lattice(self._dec, self.lane, use_final_probs)

    def GetRawLatticePruned(self, use_final_probs, beam):
        """LatticeFasterOnlineDecoder::GetRawLatticePruned (lattice-faster-online-decoder.cc:168-265): the live raw lattice
        restricted to the paths within `beam` of the best one (exact pruning; the reference uses the extra costs of its
        last periodic PruneActiveTokens)."""
        return prune_lattice(self.GetRawLattice(use_final_probs), beam)

    def GetBestPath(self, use_final_probs=True):
        """After FinalizeDecoding: ShortestPath of the raw lattice.  Before it (streaming):
        LatticeFasterOnlineDecoder::GetBestPath = BestPathEnd + TraceBackBestPath."""
        if lib().kamd_decoder_lattice_size(self._dec, self.lane, C.byref(abi.LatticeSize())) == 0:
            return best_path(self._dec, self.lane)
        return partial_best_path(self._dec, self.lane, use_final_probs)

    def trace(self):
        return get_trace(self._dec, self.lane)

    def counters(self):
        c = np.zeros(8, np.int64)
        check(lib().kamd_decoder_get_counters(self._dec, self.lane, abi.iptr(c, C.c_int64)))
        return c


def get_raw_lattice(dec, lane):
    sz = abi.LatticeSize()
    check(lib().kamd_decoder_lattice_size(dec, lane, C.byref(sz)))
    n, m = sz.num_states, sz.num_arcs
    if n == 0:
        return None
    fr, hc = np.zeros(n, np.int32), np.zeros(n, np.int32)
    co, fi = np.zeros(n, np.float32), np.zeros(n, np.float32)
    arcs = np.zeros(m, abi.LAT_ARC_DTYPE)
    check(lib().kamd_decoder_get_raw_lattice(dec, lane, abi.iptr(fr), abi.iptr(hc), abi.fptr(co),
                                             abi.fptr(fi), arcs.ctypes.data_as(C.c_void_p)))
    return Lattice(sz.start, fr, hc, co, fi, arcs, sz.num_frames)


def get_live_raw_lattice(dec, lane, use_final_probs=True):
    sz = abi.LatticeSize()
    check(lib().kamd_decoder_live_lattice_size(dec, lane, int(bool(use_final_probs)), C.byref(sz)))
    n, m = sz.num_states, sz.num_arcs
    if n == 0:
        return None
    fr, hc = np.zeros(n, np.int32), np.zeros(n, np.int32)
    co, fi = np.zeros(n, np.float32), np.zeros(n, np.float32)
    arcs = np.zeros(m, abi.LAT_ARC_DTYPE)
    check(lib().kamd_decoder_get_live_raw_lattice(dec, lane, int(bool(use_final_probs)), abi.iptr(fr), abi.iptr(hc), abi.fptr(co),
                                                  abi.fptr(fi), arcs.ctypes.data_as(C.c_void_p)))
    return Lattice(sz.start, fr, hc, co, fi, arcs, sz.num_frames)


def prune_lattice(lat, beam):
    """PruneLattice(beam, &lat) on a raw Lattice (kamd_lattice_prune); canonical numbering is kept."""
    if lat is None:
        return None
    n, m = lat.frame.size, lat.arcs.size
    smap, keep = np.zeros(n, np.int32), np.zeros(max(m, 1), np.uint8)
    n_out, m_out = C.c_int32(), C.c_int32()
    arcs = np.ascontiguousarray(lat.arcs)
    check(lib().kamd_lattice_prune(n, lat.start, abi.fptr(np.ascontiguousarray(lat.final, np.float32)), arcs.ctypes.data_as(C.c_void_p), m,
                                   float(beam), abi.iptr(smap), keep.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(n_out), C.byref(m_out)))
    if n_out.value == 0:
        return None
    ks = smap >= 0
    a = arcs[keep[:m].astype(bool)].copy()
    a["src"], a["dst"] = smap[a["src"]], smap[a["dst"]]
    return Lattice(int(smap[lat.start]), lat.frame[ks], lat.hclg[ks], lat.cost[ks], lat.final[ks], a, lat.num_frames)


def queue_fetch_lattice(dec, utt, copy_stream=None):
    """Raw lattice of a finished work-queue utterance (None: the reference would produce no lattice)."""
    ns, na, st = C.c_int32(), C.c_int32(), C.c_int32()
    ip, fp = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    p_fr, p_hc, p_co, p_fi, p_arcs = ip(), ip(), fp(), fp(), C.c_void_p()
    check(lib().kamd_decoder_queue_fetch_lattice(dec, utt, copy_stream, C.byref(ns), C.byref(na), C.byref(st), C.byref(p_fr),
                                                 C.byref(p_hc), C.byref(p_co), C.byref(p_fi), C.byref(p_arcs)))
    n, m = ns.value, na.value
    try:
        if n == 0:
            return None
        fr = np.ctypeslib.as_array(p_fr, (n,)).copy(); hc = np.ctypeslib.as_array(p_hc, (n,)).copy()
        co = np.ctypeslib.as_array(p_co, (n,)).copy(); fi = np.ctypeslib.as_array(p_fi, (n,)).copy()
        arcs = np.zeros(m, abi.LAT_ARC_DTYPE)
        if m:
            C.memmove(arcs.ctypes.data, p_arcs, m * abi.LAT_ARC_DTYPE.itemsize)
    finally:
        for p in (p_fr, p_hc, p_co, p_fi):
            lib().kamd_host_free(C.cast(p, C.c_void_p))
        lib().kamd_host_free(p_arcs)
    return Lattice(st.value, fr, hc, co, fi, arcs, int(fr.max()) if n else 0)


def lattice_best_path(lat):
    """GetBestPath of a raw Lattice held on the host (kamd_lattice_best_path)."""
    if lat is None:
        return None
    cap = max(int(lat.arcs.size), 1)
    ali, words = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    na, nw = C.c_int(), C.c_int()
    g, a = C.c_float(), C.c_float()
    fin = np.ascontiguousarray(lat.final, np.float32)
    arcs = np.ascontiguousarray(lat.arcs)
    rc = lib().kamd_lattice_best_path(int(lat.frame.size), int(lat.start), abi.fptr(fin), arcs.ctypes.data_as(C.c_void_p), int(arcs.size),
                                      abi.iptr(ali), cap, C.byref(na), abi.iptr(words), cap, C.byref(nw), C.byref(g), C.byref(a))
    if rc != 0:
        return None
    return dict(alignment=ali[:na.value].copy(), words=words[:nw.value].copy(), graph_cost=g.value, acoustic_cost=a.value)


def best_path(dec, lane):
    sz = abi.LatticeSize()
    check(lib().kamd_decoder_lattice_size(dec, lane, C.byref(sz)))
    cap = max(sz.num_arcs, 1)
    ali, words = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    na, nw = C.c_int(), C.c_int()
    g, a = C.c_float(), C.c_float()
    rc = lib().kamd_decoder_best_path(dec, lane, abi.iptr(ali), cap, C.byref(na), abi.iptr(words), cap,
                                      C.byref(nw), C.byref(g), C.byref(a))
    if rc != 0:
        return None
    return dict(alignment=ali[:na.value].copy(), words=words[:nw.value].copy(),
                graph_cost=g.value, acoustic_cost=a.value)


def partial_best_path(dec, lane, use_final_probs=True):
    n = lib().kamd_decoder_num_frames_decoded(dec, lane)
    cap = 4 * (n + 2) + 1024
    ali, words = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    na, nw = C.c_int(), C.c_int()
    g, a = C.c_float(), C.c_float()
    check(lib().kamd_decoder_partial_best_path(dec, lane, int(use_final_probs), abi.iptr(ali), cap, C.byref(na),
                                               abi.iptr(words), cap, C.byref(nw), C.byref(g), C.byref(a)))
    return dict(alignment=ali[:na.value].copy(), words=words[:nw.value].copy(), graph_cost=g.value,
                acoustic_cost=a.value)


def partial_best_paths(dec, lanes, use_final_probs=False, incremental=False):
    """partial_best_path of several un-finalized lanes in one launch -> list of dicts (None: no token alive).
    incremental: kamd_decoder_partial_best_paths_incremental -- the decoder keeps every lane's previous answer and walks
    back only to the first frame whose token is unchanged (for a host that asks after every tick; no final-probs)."""
    ln = np.ascontiguousarray(lanes, np.int32)
    if ln.size == 0:
        return []
    nmax = max(lib().kamd_decoder_num_frames_decoded(dec, int(l)) for l in ln)
    cap = 4 * (nmax + 2) + 1024
    ali, words = np.zeros((ln.size, cap), np.int32), np.zeros((ln.size, cap), np.int32)
    na, nw = np.zeros(ln.size, np.int32), np.zeros(ln.size, np.int32)
    g, a = np.zeros(ln.size, np.float32), np.zeros(ln.size, np.float32)
    if incremental:
        if use_final_probs:
            raise KamdError("incremental partial best paths are without final-probs")
        check(lib().kamd_decoder_partial_best_paths_incremental(dec, abi.iptr(ln), ln.size, abi.iptr(ali), cap, abi.iptr(na),
                                                                abi.iptr(words), cap, abi.iptr(nw), abi.fptr(g), abi.fptr(a)))
    else:
        check(lib().kamd_decoder_partial_best_paths(dec, abi.iptr(ln), ln.size, int(use_final_probs), abi.iptr(ali), cap, abi.iptr(na),
                                                    abi.iptr(words), cap, abi.iptr(nw), abi.fptr(g), abi.fptr(a)))
    return [None if na[i] < 0 else dict(alignment=ali[i, :na[i]].copy(), words=words[i, :nw[i]].copy(), graph_cost=float(g[i]),
                                        acoustic_cost=float(a[i])) for i in range(ln.size)]


def compact(dec, lanes):
    ln = np.ascontiguousarray(lanes, np.int32)
    check(lib().kamd_decoder_compact(dec, abi.iptr(ln), ln.size, None))


def lane_usage(dec, lane):
    v = [C.c_int32() for _ in range(4)]
    check(lib().kamd_decoder_lane_usage(dec, int(lane), *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def frame_tracebacks(dec, lanes, incremental=False):
    """What OnlineSilenceWeighting::ComputeCurrentTraceback reads off the decoder, for several un-finalized lanes in
    one launch -> per lane (tids, tokens), newest frame first (None: no token alive).  incremental: only down to the
    first frame whose token is the one the previous incremental call reported (that entry included) ->
    (tids, tokens, frames decoded)."""
    ln = np.ascontiguousarray(lanes, np.int32)
    if ln.size == 0:
        return []
    cap = max(1, max(lib().kamd_decoder_num_frames_decoded(dec, int(l)) for l in ln))
    tids, toks = np.zeros((ln.size, cap), np.int32), np.zeros((ln.size, cap), np.int32)
    cnt = np.zeros(ln.size, np.int32)
    if incremental:
        m = np.zeros(ln.size, np.int32)
        check(lib().kamd_decoder_frame_tracebacks_incremental(dec, abi.iptr(ln), ln.size, abi.iptr(tids), abi.iptr(toks), cap, abi.iptr(cnt),
                                                              abi.iptr(m)))
        return [None if cnt[i] < 0 else (tids[i, :m[i]].copy(), toks[i, :m[i]].copy(), int(cnt[i])) for i in range(ln.size)]
    check(lib().kamd_decoder_frame_tracebacks(dec, abi.iptr(ln), ln.size, abi.iptr(tids), abi.iptr(toks), cap, abi.iptr(cnt)))
    return [None if cnt[i] < 0 else (tids[i, :cnt[i]].copy(), toks[i, :cnt[i]].copy()) for i in range(ln.size)]


def endpoint_config_default():
    """OnlineEndpointConfig() (online2/online-endpoint.h:149-154)"""
    c = abi.EndpointConfig()
    lib().kamd_endpoint_config_default(C.byref(c))
    return c


def endpoint_detected_from(config, num_frames_decoded, trailing_silence_frames, frame_shift_in_seconds, final_relative_cost):
    """EndpointDetected on plain numbers (online2/online-endpoint.cc:46-68)"""
    rc = lib().kamd_endpoint_detected(C.byref(config), int(num_frames_decoded), int(trailing_silence_frames),
                                      float(frame_shift_in_seconds), float(final_relative_cost))
    check(rc)
    return bool(rc)


def set_silence_phones(dec, tid2phone, silence_phones):
    """tid2phone[tid] for tid in 1..num_tids (index 0 unused) = TransitionIdToPhone"""
    tp = np.ascontiguousarray(tid2phone, np.int32)
    sp = np.ascontiguousarray(list(silence_phones), np.int32)
    check(lib().kamd_decoder_set_silence_phones(dec, abi.iptr(tp), tp.size - 1, abi.iptr(sp), sp.size))


def trailing_silence_frames(dec, lanes):
    """TrailingSilenceLength of these un-finalized lanes (online2/online-endpoint.cc:71-102), one launch"""
    ln = np.ascontiguousarray(lanes, np.int32)
    out = np.zeros(ln.size, np.int32)
    check(lib().kamd_decoder_trailing_silence_frames(dec, abi.iptr(ln), ln.size, abi.iptr(out)))
    return out


def endpoint_detected(dec, config, lanes, frame_shift_in_seconds):
    """EndpointDetected(config, tmodel, frame_shift, decoder) for these lanes -> (flags, trailing silence frames)"""
    ln = np.ascontiguousarray(lanes, np.int32)
    det, sil = np.zeros(ln.size, np.int32), np.zeros(ln.size, np.int32)
    check(lib().kamd_decoder_endpoint_detected(dec, C.byref(config), abi.iptr(ln), ln.size, float(frame_shift_in_seconds),
                                               abi.iptr(det), abi.iptr(sil)))
    return det.astype(bool), sil


def get_trace(dec, lane):
    n = lib().kamd_decoder_num_frames_decoded(dec, lane)
    nt, cu, of = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
    k = check(lib().kamd_decoder_get_trace(dec, lane, abi.iptr(nt), abi.fptr(cu), abi.fptr(of), n))
    return nt[:k], cu[:k], of[:k]


class BatchDecoder:
    """N lanes decoded in one launch (NnetBatchDecoder's role, nnet3/nnet-batch-compute.h:606)."""

    def __init__(self, graph, config, sizes):
        self.graph, self.config, self.sizes = graph, config, sizes
        t2p = np.ascontiguousarray(graph.hclg.tid2pdf, np.int32)
        self._t2p = t2p
        self._dec = lib().kamd_decoder_create(graph._h, C.byref(config), C.byref(sizes), abi.iptr(t2p),
                                              t2p.size - 1)
        if not self._dec:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_dec", None):
            lib().kamd_decoder_destroy(self._dec)
            self._dec = None

    def SetSearchMode(self, mode):
        check(lib().kamd_decoder_set_search_mode(self._dec, int(mode)))

    def SetTokenPreselection(self, on):
        """kamd_decoder_set_token_preselection: off = every token the reference creates is created and counted."""
        check(lib().kamd_decoder_set_token_preselection(self._dec, int(bool(on))))

    def SetLevel1Table(self, words):
        check(lib().kamd_decoder_set_level1_table(self._dec, int(words)))

    def decode(self, matrices):
        """matrices: list of DeviceMatrix / host arrays, one per lane."""
        n = len(matrices)
        dms = [m if isinstance(m, DeviceMatrix) else DeviceMatrix(m) for m in matrices]
        lanes = np.arange(n, dtype=np.int32)
        tasks = (abi.DecodeTask * n)()
        for i, dm in enumerate(dms):
            tasks[i] = abi.DecodeTask(i, dm.rows, dm.ptr(0), dm.cols, 0)
        check(lib().kamd_decoder_init(self._dec, abi.iptr(lanes), n, None))
        check(lib().kamd_decoder_advance(self._dec, tasks, n, None))
        check(lib().kamd_decoder_finalize(self._dec, abi.iptr(lanes), n, None))
        check(lib().kamd_decoder_sync(self._dec))
        return [get_raw_lattice(self._dec, i) for i in range(n)]

    def decode_queue(self, matrices, resident_lanes=0, order=None, pool_bytes=0):
        """Work-queue decoding (kamd_decoder_queue_*): `resident_lanes` persistent lanes pull the
        utterances (longest first unless `order` says otherwise); returns (lattices, records, kernel ms)."""
        n = len(matrices)
        dms = [m if isinstance(m, DeviceMatrix) else DeviceMatrix(m) for m in matrices]
        if order is None:
            order = sorted(range(n), key=lambda i: -dms[i].rows)
        tasks = (abi.QueueTask * n)()
        for k, i in enumerate(order):
            tasks[k] = abi.QueueTask(dms[i].ptr(0), dms[i].cols, dms[i].rows, i, 0)
        if pool_bytes:
            check(lib().kamd_decoder_queue_configure(self._dec, pool_bytes))
        check(lib().kamd_decoder_queue_launch(self._dec, tasks, n, resident_lanes, None))
        ms, lanes = C.c_float(), C.c_int32()
        check(lib().kamd_decoder_queue_wait(self._dec, C.byref(ms), C.byref(lanes)))
        done = np.zeros(n, np.int32)
        got = lib().kamd_decoder_queue_poll(self._dec, abi.iptr(done), n)
        if got != n:
            raise KamdError("queue kernel ended with %d of %d utterances published" % (got, n))
        lats, recs = [], []
        for i in range(n):
            r = abi.QueueResult()
            check(lib().kamd_decoder_queue_result(self._dec, i, C.byref(r)))
            recs.append(r)
            lats.append(queue_fetch_lattice(self._dec, i))
        self.queue_order = done
        return lats, recs, ms.value

    def best_path(self, lane):
        return best_path(self._dec, lane)

    def counters(self, lane):
        c = np.zeros(8, np.int64)
        check(lib().kamd_decoder_get_counters(self._dec, lane, abi.iptr(c, C.c_int64)))
        return c

    def phase_cycles(self, lane):
        c = np.zeros(16, np.uint64)
        check(lib().kamd_decoder_get_phase_cycles(self._dec, lane, c.ctypes.data_as(C.POINTER(C.c_uint64))))
        return c

    def last_advance_ms(self):
        return lib().kamd_decoder_last_advance_ms(self._dec)
