"""A whole test set (or one GPU's shard of it) through the device: the Python mirror of
NnetBatchDecoder (nnet3/nnet-batch-compute.h:606-833) over kamd_batch_decoder_*.

    bd = NnetBatchDecoder(mfcc_opts, model, hclg, cfg, max_seconds=36)
    bd.load(waves)                     # AcceptInput for every utterance; resident in HBM
    stats = bd.run()                   # Finished(): features -> nnet -> work-queue search -> host tail
    bd.output(u)                       # GetOutput: words, alignment, costs, record
"""
import ctypes as C

import numpy as np

from . import abi, decoder, pipeline
from ._lib import KamdError, check, lib


class NnetBatchDecoder:
    def __init__(self, mfcc_opts, model, hclg, cfg, max_seconds=36.0, resident_lanes=0, host_threads=8, determinize=True,
                 keep_raw_lattices=False, tid_phone=None, sizes=None, nnet_pass_frames=1000000, lattice_pool_bytes=1 << 30,
                 hash_capacity=None, tokens_per_frame=None, search_mode=2, det=None, long_lanes=0, first_pass_frames=None, hbm_fraction=0.5,
                 nnet=None):
        featmod = __import__("kaldi_amd.feat", fromlist=["Mfcc"])
        # mfcc_opts = None: no feature stage, the caller hands over feature matrices (load_features), as the
        # reference's AcceptInput does (nnet-batch-compute.h:665)
        if mfcc_opts is None:
            self.feat = None
        else:
            self.feat = featmod.Fbank(mfcc_opts) if isinstance(mfcc_opts, abi.FbankOpts) else featmod.Mfcc(mfcc_opts)
        self.model, self.cfg = model, cfg
        self.nnet = nnet if nnet is not None else decoder.Nnet(model)      # (one set of weights in HBM can serve several decoders)
        self.graph = hclg if isinstance(hclg, decoder.Graph) else decoder.Graph(hclg)
        fps = 100.0 if mfcc_opts is None else 1000.0 / mfcc_opts.frame.frame_shift_ms
        max_out = int(max_seconds * fps / model.subsampling) + 2
        lanes = resident_lanes or lib().kamd_device_num_cus() * lib().kamd_decoder_lanes_per_cu()
        # every lane must hold the longest utterance: uniform arenas (avg = max)
        self.sizes = sizes or pipeline.default_sizes(cfg, lanes, max_out, max_out, hash_capacity=hash_capacity,
                                                     tokens_per_frame=tokens_per_frame, hbm_fraction=hbm_fraction)
        self.dec = decoder.BatchDecoder(self.graph, cfg, self.sizes)
        self.dec.SetSearchMode(search_mode)
        o = abi.BatchOpts()
        lib().kamd_batch_opts_default(C.byref(o))
        o.resident_lanes, o.host_threads = int(resident_lanes), int(host_threads)
        o.determinize, o.keep_raw_lattices = int(bool(determinize)), int(bool(keep_raw_lattices))
        o.nnet_pass_frames, o.lattice_pool_bytes = int(nnet_pass_frames), int(lattice_pool_bytes)
        o.lattice_beam = cfg.lattice_beam
        if first_pass_frames is not None:
            o.first_pass_frames = int(first_pass_frames)
        for k, v in (det or {}).items():        # DeterminizeLatticePhonePrunedOptions: delta, phone_determinize, word_determinize, ...
            setattr(o.det, k, v)
        self.opts = o
        tp = None if tid_phone is None else np.ascontiguousarray(tid_phone, np.int32)
        self._h = lib().kamd_batch_decoder_create(self.feat._h if self.feat is not None else None, self.nnet._h, self.dec._dec, C.byref(o),
                                                  abi.iptr(tp) if tp is not None else None, 0 if tp is None else tp.size - 1)
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())
        self.n_utts = 0
        self.stats = None
        # long_lanes > 0: a second decoder object of that many lanes, on which run() searches the longest utterances beside
        # the acoustic model of the rest when the shard is small enough for one utterance's chain of frames to bound it
        # (kamd_batch_decoder_set_long_decoder)
        self.dec_long = None
        if long_lanes > 0:
            sz = pipeline.default_sizes(cfg, int(long_lanes), max_out, max_out, hash_capacity=hash_capacity, tokens_per_frame=tokens_per_frame)
            self.dec_long = decoder.BatchDecoder(self.graph, cfg, sz)
            self.dec_long.SetSearchMode(search_mode)
            check(lib().kamd_batch_decoder_set_long_decoder(self._h, self.dec_long._dec, int(long_lanes)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_batch_decoder_destroy(self._h)
            self._h = None

    def set_ivector_extractor(self, extractor, frames_per_chunk=50):
        """--online-ivectors of the recipe (steps/nnet3/decode.sh:105-107), estimated on the device from every pass's own
        features; the model then runs chunk by chunk like DecodableNnetSimple.  Before load() / load_host(); None removes it."""
        self._ie = extractor
        check(lib().kamd_batch_decoder_set_ivector_extractor(self._h, None if extractor is None else extractor._h, int(frames_per_chunk)))

    def set_chunk_rule(self, rule):
        """0 / "simple": DecodableNnetSimple's chunks (nnet3-latgen-faster; the default); 1 / "batch_computer": NnetBatchComputer's
        tasks (nnet3-latgen-faster-batch, nnet3/nnet-batch-compute.cc:774-829).  kamd_batch_decoder_set_chunk_rule."""
        rules = {"simple": 0, "batch_computer": 1, 0: 0, 1: 1}
        if isinstance(rule, bool) or rule not in rules:
            raise ValueError("set_chunk_rule: %r is not one of 0 / 'simple', 1 / 'batch_computer'" % (rule,))
        check(lib().kamd_batch_decoder_set_chunk_rule(self._h, rules[rule]))

    def load(self, waves):
        waves = [np.asarray(w, np.float32) for w in waves]
        off = np.concatenate([[0], np.cumsum([w.size for w in waves])]).astype(np.int64)
        flat = np.ascontiguousarray(np.concatenate(waves), np.float32)
        check(lib().kamd_batch_decoder_load(self._h, abi.fptr(flat), abi.iptr(off, C.c_int64), len(waves)))
        self.n_utts = len(waves)
        self.audio_seconds = float(flat.size) / self.feat.opts.frame.samp_freq

    def load_host(self, waves):
        """AcceptInput with the samples left in host memory: every run() uploads them itself, pass by pass, overlapped
        with the features and the acoustic model (kamd_batch_decoder_load_host).  The buffer is kept alive here."""
        waves = [np.asarray(w, np.float32) for w in waves]
        off = np.concatenate([[0], np.cumsum([w.size for w in waves])]).astype(np.int64)
        flat = np.ascontiguousarray(np.concatenate(waves), np.float32)
        # the library page-locks `flat` in place and unlocks the PREVIOUS buffer inside this call: the previous array must
        # outlive it (rebinding self._host_waves first would free memory that is still registered)
        previous = getattr(self, "_host_waves", None)
        check(lib().kamd_batch_decoder_load_host(self._h, abi.fptr(flat), abi.iptr(off, C.c_int64), len(waves)))
        self._host_waves, self._host_off = flat, off
        del previous
        self.n_utts = len(waves)
        self.audio_seconds = float(self._host_waves.size) / self.feat.opts.frame.samp_freq

    def unload_host(self):
        """Releases the buffer of load_host() (page lock undone first, then the array); outputs stay readable."""
        check(lib().kamd_batch_decoder_unload_host(self._h))
        self._host_waves = None

    def output_frames(self):
        """Output frames of every loaded utterance (0: too short for a frame)."""
        fr = np.zeros(max(self.n_utts, 1), np.int32)
        lib().kamd_batch_decoder_output_frames(self._h, abi.iptr(fr), self.n_utts)
        return fr[:self.n_utts]

    def set_loglike_override(self, device_ptr):
        """Bench workload synthesis: the search reads this device matrix ([sum of output_frames() x P], load order)
        instead of the acoustic model's output (still computed).  None switches it off."""
        check(lib().kamd_batch_decoder_set_loglike_override(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def load_features(self, feats, ivectors=None):
        """AcceptInput(utterance_id, input, ivector, ...) for every utterance: feats = list of [T x dim] matrices,
        ivectors = one row per utterance when the model has an ivector node."""
        feats = [np.ascontiguousarray(f, np.float32) for f in feats]
        if not feats:
            raise KamdError("empty test set")
        off = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
        flat = np.ascontiguousarray(np.concatenate(feats, axis=0), np.float32)
        iv = None if ivectors is None else np.ascontiguousarray(ivectors, np.float32).reshape(len(feats), -1)
        check(lib().kamd_batch_decoder_load_features(self._h, abi.fptr(flat), abi.iptr(off, C.c_int64), flat.shape[1],
                                                     abi.fptr(iv) if iv is not None else None, 0 if iv is None else iv.shape[1], len(feats)))
        self.n_utts = len(feats)
        self.audio_seconds = float(flat.shape[0]) / 100.0

    def run(self):
        st = abi.BatchStats()
        check(lib().kamd_batch_decoder_run(self._h, C.byref(st)))
        self.stats = st
        return st

    def output(self, u, cap=None):
        """-> dict(words, alignment, graph_cost, acoustic_cost, record) or None when this utterance failed
        (record.error says why) / produced an empty lattice."""
        rec = abi.QueueResult()
        nw, na = C.c_int(), C.c_int()
        g, a = C.c_float(), C.c_float()
        rc = lib().kamd_batch_decoder_get_output(self._h, u, None, 0, C.byref(nw), None, 0, C.byref(na), C.byref(g), C.byref(a), C.byref(rec))
        if rc != 0:
            return None
        words, ali = np.zeros(max(nw.value, 1), np.int32), np.zeros(max(na.value, 1), np.int32)
        check(lib().kamd_batch_decoder_get_output(self._h, u, abi.iptr(words), words.size, C.byref(nw), abi.iptr(ali), ali.size,
                                                  C.byref(na), C.byref(g), C.byref(a), C.byref(rec)))
        return dict(words=words[:nw.value], alignment=ali[:na.value], graph_cost=g.value, acoustic_cost=a.value, record=rec)

    def record(self, u):
        rec = abi.QueueResult()
        lib().kamd_batch_decoder_get_output(self._h, u, None, 0, None, None, 0, None, None, None, C.byref(rec))
        return rec

    def raw_lattice(self, u):
        ns, na, st = C.c_int32(), C.c_int32(), C.c_int32()
        ip, fp = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        p_fr, p_hc, p_co, p_fi, p_arcs = ip(), ip(), fp(), fp(), C.c_void_p()
        check(lib().kamd_batch_decoder_get_raw_lattice(self._h, u, C.byref(ns), C.byref(na), C.byref(st), C.byref(p_fr), C.byref(p_hc),
                                                       C.byref(p_co), C.byref(p_fi), C.byref(p_arcs)))
        n, m = ns.value, na.value
        if n == 0:
            return None
        fr = np.ctypeslib.as_array(p_fr, (n,)).copy(); hc = np.ctypeslib.as_array(p_hc, (n,)).copy()
        co = np.ctypeslib.as_array(p_co, (n,)).copy(); fi = np.ctypeslib.as_array(p_fi, (n,)).copy()
        arcs = np.zeros(m, abi.LAT_ARC_DTYPE)
        if m:
            C.memmove(arcs.ctypes.data, p_arcs, m * abi.LAT_ARC_DTYPE.itemsize)
        return decoder.Lattice(st.value, fr, hc, co, fi, arcs, int(fr.max()))

    def compact_lattice(self, u):
        from .io import CompactLattice
        h = lib().kamd_batch_decoder_get_compact_lattice(self._h, u)
        return CompactLattice(h, owned=False) if h else None

    def loglikes(self, u):
        rows, cols = C.c_int32(), C.c_int32()
        lib().kamd_batch_decoder_get_loglikes(self._h, u, None, 0, C.byref(rows), C.byref(cols))
        buf = np.zeros((rows.value, cols.value), np.float32)
        check(lib().kamd_batch_decoder_get_loglikes(self._h, u, abi.fptr(buf), rows.value, C.byref(rows), C.byref(cols)))
        return buf
