"""wav -> MFCC -> TDNN-F -> LatticeFasterDecoder for a batch of utterances, device resident
(nnet3-latgen-faster-batch's job, nnet3bin/nnet3-latgen-faster-batch.cc:170-214)."""
import ctypes as C

import numpy as np

from . import abi, decoder
from ._lib import KamdError, check, lib


def default_sizes(cfg, max_utts, max_out_frames, avg_out_frames=None, hash_capacity=None,
                  tokens_per_frame=None, links_per_frame=None, hbm_fraction=0.5):
    """Device sizing: `max_utts` lanes, each at most `max_out_frames` decoded frames; the
    token / link pools hold `avg_out_frames` frames per lane on average (the pools are split
    between lanes in proportion to utterance length, kamd_decoder_reserve)."""
    out = abi.DecoderSizes()                      # the arithmetic lives in the library (kamd_decoder_sizes_suggest): C / C++ hosts size the same way
    check(lib().kamd_decoder_sizes_suggest(C.byref(cfg), int(max_utts), int(max_out_frames), int(avg_out_frames or 0), int(hash_capacity or 0),
                                           int(tokens_per_frame or 0), int(links_per_frame or 0), float(hbm_fraction), C.byref(out)))
    return out


class Pipeline:
    def __init__(self, mfcc_opts, model, hclg, cfg, max_utts=64, max_seconds=36.0, avg_seconds=None,
                 sizes=None):
        featmod = __import__("kaldi_amd.feat", fromlist=["Mfcc"])
        # MfccOptions or FbankOptions (--feature-type of the online2 binaries; compute-fbank-feats recipes)
        self.feat = featmod.Fbank(mfcc_opts) if isinstance(mfcc_opts, abi.FbankOpts) else featmod.Mfcc(mfcc_opts)
        self.model = model
        self.nnet = decoder.Nnet(model)
        self.graph = decoder.Graph(hclg)
        self.cfg = cfg
        fps = 1000.0 / mfcc_opts.frame.frame_shift_ms
        max_out = int(max_seconds * fps / model.subsampling) + 2
        avg_out = None if avg_seconds is None else int(avg_seconds * fps / model.subsampling) + 2
        self.sizes = sizes or default_sizes(cfg, max_utts, max_out, avg_out)
        self.dec = decoder.BatchDecoder(self.graph, cfg, self.sizes)
        self._h = lib().kamd_pipeline_create(self.feat._h, self.nnet._h, self.dec._dec)
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())
        self.n_utts = 0
        self.last_stage_ms = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_pipeline_destroy(self._h)
            self._h = None

    def load(self, waves):
        """Utterances too short for a single frame are skipped with a None result, like the
        reference's "Zero-length utterance" warning (nnet3bin/nnet3-latgen-faster.cc:148-152)."""
        self._lane_of = []
        keep = []
        for w in waves:
            if self.feat.NumFrames(np.asarray(w).size) > 0:
                self._lane_of.append(len(keep))
                keep.append(np.asarray(w, np.float32))
            else:
                self._lane_of.append(-1)
        self.n_utts = len(keep)
        self.n_input = len(waves)
        self.audio_seconds = float(sum(np.asarray(w).size for w in waves)) / self.feat.opts.frame.samp_freq
        if not keep:
            return
        off = np.concatenate([[0], np.cumsum([w.size for w in keep])]).astype(np.int64)
        flat = np.ascontiguousarray(np.concatenate(keep), np.float32)
        check(lib().kamd_pipeline_load_batch(self._h, abi.fptr(flat), abi.iptr(off, C.c_int64), len(keep)))
        self._loaded = ("waves", flat, off, len(keep))     # kept for grow(): the reference never runs out of room
        self._iv_loaded = None

    def load_features(self, feats):
        """Feature matrices computed elsewhere (the features-rspecifier of nnet3-latgen-faster)
        instead of waveforms; empty matrices are skipped with a None result."""
        self._lane_of, keep = [], []
        for f in feats:
            f = np.asarray(f, np.float32)
            if f.ndim != 2:
                raise KamdError("features must be [frames x dim] matrices")
            if f.shape[0] > 0:
                self._lane_of.append(len(keep))
                keep.append(f)
            else:
                self._lane_of.append(-1)
        self.n_utts, self.n_input = len(keep), len(feats)
        shift_s = self.feat.opts.frame.frame_shift_ms * 1e-3
        self.audio_seconds = float(sum(f.shape[0] for f in keep)) * shift_s
        if not keep:
            return
        off = np.concatenate([[0], np.cumsum([f.shape[0] for f in keep])]).astype(np.int64)
        flat = np.ascontiguousarray(np.concatenate(keep), np.float32)
        check(lib().kamd_pipeline_load_features(self._h, abi.fptr(flat), abi.iptr(off, C.c_int64), len(keep), flat.shape[1]))
        self._loaded = ("feats", flat, off, len(keep))
        self._iv_loaded = None

    def set_ivectors(self, ivectors):
        """One ivector per (non-skipped) utterance of the loaded batch."""
        if ivectors is None:
            check(lib().kamd_pipeline_set_ivectors(self._h, None, 0))
            self._iv_loaded = None
            return
        iv = np.ascontiguousarray([v for v, lane in zip(ivectors, self._lane_of) if lane >= 0], np.float32)
        check(lib().kamd_pipeline_set_ivectors(self._h, abi.fptr(iv), iv.shape[1]))
        self._iv_loaded = ("utt", iv)

    def set_online_ivectors(self, ivector_matrices, ivector_period=10, frames_per_chunk=50):
        """--online-ivectors / --online-ivector-period: one [rows x dim] matrix per utterance of the
        loaded batch (a row per `ivector_period` frames); the nnet stage then runs chunk by chunk."""
        if ivector_matrices is None:
            check(lib().kamd_pipeline_set_online_ivectors(self._h, None, None, 0, 0, 0))
            self._iv_loaded = None
            return
        mats = [np.ascontiguousarray(v, np.float32) for v, lane in zip(ivector_matrices, self._lane_of) if lane >= 0]
        off = np.zeros(len(mats) + 1, np.int64)
        off[1:] = np.cumsum([m.shape[0] for m in mats])
        iv = np.ascontiguousarray(np.concatenate(mats), np.float32)
        check(lib().kamd_pipeline_set_online_ivectors(self._h, abi.fptr(iv), abi.iptr(off, C.c_int64), iv.shape[1],
                                                      ivector_period, frames_per_chunk))
        self._iv_loaded = ("online", iv, off, ivector_period, frames_per_chunk)

    def set_ivector_extractor(self, extractor, frames_per_chunk=50):
        """i-vectors estimated on the device from the batch's own features (kaldi_amd.ivector.IvectorExtractor),
        fed to the nnet chunk by chunk like --online-ivectors; None = off."""
        check(lib().kamd_pipeline_set_ivector_extractor(self._h, extractor._h if extractor is not None else None, frames_per_chunk))
        self._ie = (extractor, frames_per_chunk)

    def set_overlap(self, bounds):
        """Cuts the nnet stage in time at these output-frame indices and overlaps every later slice's
        forward with the search over the slice before it (kamd_pipeline_set_overlap); [] = off."""
        b = np.ascontiguousarray(bounds, np.int32)
        check(lib().kamd_pipeline_set_overlap(self._h, abi.iptr(b) if b.size else None, int(b.size)))
        self._overlap = b.tolist()

    def grow(self, factor=2):
        """Rebuilds the decoder with `factor` x the frame table and arenas and reloads the batch
        (the reference's HashList and token lists simply grow; here a lane that overflows reports
        KAMD_ERR_CAPACITY and the caller decides)."""
        s = self.sizes
        self.sizes = abi.DecoderSizes(s.max_lanes, s.hash_capacity * factor, s.arena_tokens * factor,
                                      s.arena_links * factor, s.max_frames)
        lib().kamd_pipeline_destroy(self._h)
        self._h = None
        self.dec = decoder.BatchDecoder(self.graph, self.cfg, self.sizes)
        self._h = lib().kamd_pipeline_create(self.feat._h, self.nnet._h, self.dec._dec)
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())
        if getattr(self, "_loaded", None) is not None and self.n_utts > 0:
            kind, flat, off, n = self._loaded
            if kind == "waves":
                check(lib().kamd_pipeline_load_batch(self._h, abi.fptr(flat), abi.iptr(off, C.c_int64), n))
            else:
                check(lib().kamd_pipeline_load_features(self._h, abi.fptr(flat), abi.iptr(off, C.c_int64), n, flat.shape[1]))
            if getattr(self, "_overlap", None):
                self.set_overlap(self._overlap)
            if getattr(self, "_ie", None) and self._ie[0] is not None:
                self.set_ivector_extractor(*self._ie)
            iv = getattr(self, "_iv_loaded", None)
            if iv is not None and iv[0] == "utt":
                check(lib().kamd_pipeline_set_ivectors(self._h, abi.fptr(iv[1]), iv[1].shape[1]))
            elif iv is not None:
                check(lib().kamd_pipeline_set_online_ivectors(self._h, abi.fptr(iv[1]), abi.iptr(iv[2], C.c_int64), iv[1].shape[1],
                                                              iv[3], iv[4]))

    def run(self, auto_grow=0):
        """auto_grow = how many times a capacity overflow may be answered by grow() + rerun
        (the batch and its ivectors are reloaded)."""
        ms = np.zeros(4, np.float32)
        if self.n_utts == 0:
            self.last_stage_ms = ms.tolist()
            return self.last_stage_ms
        while True:
            try:
                check(lib().kamd_pipeline_run(self._h, abi.fptr(ms)))
                break
            except KamdError as e:
                if auto_grow <= 0 or "capacity" not in str(e):
                    raise
                auto_grow -= 1
                self.grow()
        self.last_stage_ms = ms.tolist()
        return self.last_stage_ms

    def results(self, lattices=True):
        out = []
        for lane in getattr(self, "_lane_of", range(self.n_utts)):
            if lane < 0:
                out.append(None)
                continue
            u = lane
            bp = decoder.best_path(self.dec._dec, u)
            r = dict(words=bp["words"] if bp else np.zeros(0, np.int32), best=bp)
            if lattices:
                r["lattice"] = decoder.get_raw_lattice(self.dec._dec, u)
            out.append(r)
        return out

    def decode(self, waves, lattices=True):
        self.load(waves)
        self.run()
        return self.results(lattices)

    def _get(self, fn, u):
        rows, cols = C.c_int32(), C.c_int32()
        cap = 1 << 16
        buf = np.zeros((cap, 1), np.float32)
        # query the shape with a zero-capacity call first
        fn(self._h, u, abi.fptr(buf), 0, C.byref(rows), C.byref(cols))
        buf = np.zeros((rows.value, cols.value), np.float32)
        check(fn(self._h, u, abi.fptr(buf), rows.value, C.byref(rows), C.byref(cols)))
        return buf

    def features(self, u):
        return self._get(lib().kamd_pipeline_get_features, u)

    def loglikes(self, u):
        return self._get(lib().kamd_pipeline_get_loglikes, u)
