"""Register() functions of the option structs on the decode path, on kaldi_amd.table.ParseOptions:
same option names, defaults and meaning as the reference, so that conf/mfcc_hires.conf,
conf/decode.config and the command lines steps/nnet3/decode.sh builds are accepted as they are.
  FrameExtractionOptions::Register   feat/feature-window.h:68-100
  MelBanksOptions::Register          feat/mel-computations.h:59-77
  MfccOptions::Register              feat/feature-mfcc.h:58-75
  LatticeFasterDecoderConfig::Register  decoder/lattice-faster-decoder.h:65-83
  NnetSimpleComputationOptions::Register nnet3/nnet-am-decodable-simple.h:68-105
Options that steer parts of the reference that do not exist here (computation optimisation,
debugging) are accepted and ignored, and listed in IGNORED."""
from . import abi
from ._lib import KamdError

IGNORED = ("debug-computation", "extra-left-context", "extra-right-context", "extra-left-context-initial",
           "extra-right-context-final", "allow-downsample", "max-feature-vectors", "debug-mel", "minimize", "max-mem",
           "hash-ratio", "prune-scale")


def register_mfcc(po):
    d = abi.mfcc_opts_default()
    _register_frame_mel(po, d.frame, d.mel)
    po.register("num-ceps", int, d.num_ceps, "Number of cepstra in MFCC computation (including C0)")
    po.register("use-energy", bool, bool(d.use_energy), "Use energy (not C0) in MFCC computation")
    po.register("energy-floor", float, d.energy_floor, "Floor on energy (absolute, not relative) in MFCC computation")
    po.register("raw-energy", bool, bool(d.raw_energy), "If true, compute energy before preemphasis and windowing")
    po.register("cepstral-lifter", float, d.cepstral_lifter, "Constant that controls scaling of MFCCs")
    po.register("htk-compat", bool, bool(d.htk_compat), "If true, put energy or C0 last")


def mfcc_opts(po):
    o = abi.mfcc_opts_default()
    _frame_mel_from(po, o)
    o.num_ceps, o.use_energy, o.energy_floor = po["num-ceps"], int(po["use-energy"]), po["energy-floor"]
    o.raw_energy, o.cepstral_lifter, o.htk_compat = int(po["raw-energy"]), po["cepstral-lifter"], int(po["htk-compat"])
    return o


def _register_frame_mel(po, f, m):
    po.register("sample-frequency", float, f.samp_freq, "Waveform data sample frequency")
    po.register("frame-length", float, f.frame_length_ms, "Frame length in milliseconds")
    po.register("frame-shift", float, f.frame_shift_ms, "Frame shift in milliseconds")
    po.register("preemphasis-coefficient", float, f.preemph_coeff, "Coefficient for use in signal preemphasis")
    po.register("remove-dc-offset", bool, bool(f.remove_dc_offset), "Subtract mean from waveform on each frame")
    po.register("dither", float, 0.0, "Dithering constant; only 0 is supported")
    po.register("window-type", str, "povey", "Type of window (hamming|hanning|povey|rectangular|blackman)")
    po.register("blackman-coeff", float, f.blackman_coeff, "Constant coefficient for generalized Blackman window.")
    po.register("round-to-power-of-two", bool, True, "Round window size to power of two by zero-padding")
    po.register("snip-edges", bool, True, "Only output frames that completely fit in the file")
    po.register("allow-downsample", bool, False, "(ignored)")
    po.register("max-feature-vectors", int, -1, "(ignored)")
    po.register("num-mel-bins", int, m.num_bins, "Number of triangular mel-frequency bins")
    po.register("low-freq", float, m.low_freq, "Low cutoff frequency for mel bins")
    po.register("high-freq", float, m.high_freq, "High cutoff frequency for mel bins (if <= 0, offset from Nyquist)")
    po.register("vtln-low", float, m.vtln_low, "Low inflection point in piecewise linear VTLN warping function")
    po.register("vtln-high", float, m.vtln_high, "High inflection point in piecewise linear VTLN warping function")
    po.register("debug-mel", bool, False, "(ignored)")


def _frame_mel_from(po, o):
    if po["dither"] != 0.0:
        raise KamdError("--dither=%g: only --dither=0 is supported (dithering is random in the reference)" % po["dither"])
    if not po.was_given("dither"):
        import sys
        print("WARNING --dither not given: this implementation computes with --dither=0 (Kaldi's default is 1.0, random); "
              "pass --dither=0 to make that explicit", file=sys.stderr)
    if po["window-type"] not in abi.KAMD_WIN:
        raise KamdError("Invalid window type " + po["window-type"])
    f, m = o.frame, o.mel
    f.samp_freq, f.frame_length_ms, f.frame_shift_ms = po["sample-frequency"], po["frame-length"], po["frame-shift"]
    f.preemph_coeff, f.remove_dc_offset, f.dither = po["preemphasis-coefficient"], int(po["remove-dc-offset"]), 0.0
    f.window_type, f.blackman_coeff = abi.KAMD_WIN[po["window-type"]], po["blackman-coeff"]
    f.round_to_power_of_two, f.snip_edges = int(po["round-to-power-of-two"]), int(po["snip-edges"])
    m.num_bins, m.low_freq, m.high_freq, m.vtln_low, m.vtln_high = (po["num-mel-bins"], po["low-freq"], po["high-freq"],
                                                                   po["vtln-low"], po["vtln-high"])


def register_fbank(po):
    """FbankOptions::Register (feat/feature-fbank.h:60-80)"""
    d = abi.fbank_opts_default()
    _register_frame_mel(po, d.frame, d.mel)
    po.register("use-energy", bool, bool(d.use_energy), "Add an extra dimension with energy to the FBANK output.")
    po.register("energy-floor", float, d.energy_floor, "Floor on energy (absolute, not relative) in FBANK computation")
    po.register("raw-energy", bool, bool(d.raw_energy), "If true, compute energy before preemphasis and windowing")
    po.register("htk-compat", bool, bool(d.htk_compat), "If true, put energy last.")
    po.register("use-log-fbank", bool, bool(d.use_log_fbank), "If true, produce log-filterbank, else produce linear.")
    po.register("use-power", bool, bool(d.use_power), "If true, use power, else use magnitude.")


def fbank_opts(po):
    o = abi.fbank_opts_default()
    _frame_mel_from(po, o)
    o.use_energy, o.energy_floor, o.raw_energy = int(po["use-energy"]), po["energy-floor"], int(po["raw-energy"])
    o.htk_compat, o.use_log_fbank, o.use_power = int(po["htk-compat"]), int(po["use-log-fbank"]), int(po["use-power"])
    return o


def register_decoder(po):
    d = abi.decoder_config_default()
    po.register("beam", float, d.beam, "Decoding beam.  Larger->slower, more accurate.")
    po.register("max-active", int, d.max_active, "Decoder max active states.  Larger->slower; more accurate")
    po.register("min-active", int, d.min_active, "Decoder minimum #active states.")
    po.register("lattice-beam", float, d.lattice_beam, "Lattice generation beam.  Larger->slower, and deeper lattices")
    po.register("prune-interval", int, d.prune_interval, "Interval (in frames) at which to prune tokens")
    po.register("determinize-lattice", bool, True, "If true, determinize the lattice (lattice-determinization, keeping "
                "only best pdf-sequence for each word-sequence).")
    po.register("beam-delta", float, d.beam_delta, "Increment used in decoding-- this parameter is obscure and relates "
                "to a speedup in the way the max-active constraint is applied.  Larger is more accurate.")
    po.register("hash-ratio", float, d.hash_ratio, "(ignored: the device token table is sized by the arena options)")
    po.register("prune-scale", float, d.prune_scale, "(ignored: lattice pruning is exact here)")
    po.register("delta", float, 0.000976562, "Tolerance used in determinization")
    po.register("max-mem", int, 50000000, "(ignored)")
    po.register("phone-determinize", bool, True, "If true, do an initial pass of determinization on both phones and words")
    po.register("word-determinize", bool, True, "If true, do a second pass of determinization on words only")
    po.register("minimize", bool, False, "(ignored: as in the reference's default, lattices are not minimized)")


def decoder_config(po):
    c = abi.decoder_config_default()
    c.beam, c.max_active, c.min_active = po["beam"], po["max-active"], po["min-active"]
    c.lattice_beam, c.prune_interval, c.beam_delta = po["lattice-beam"], po["prune-interval"], po["beam-delta"]
    if c.beam <= 0 or c.max_active <= 1 or c.lattice_beam <= 0 or c.min_active > c.max_active or c.prune_interval <= 0:
        raise KamdError("LatticeFasterDecoderConfig::Check failed")      # lattice-faster-decoder.h:84-89
    return c


def register_nnet_simple(po):
    po.register("extra-left-context", int, 0, "(ignored: TDNN models need no extra context)")
    po.register("extra-right-context", int, 0, "(ignored)")
    po.register("extra-left-context-initial", int, -1, "(ignored)")
    po.register("extra-right-context-final", int, -1, "(ignored)")
    po.register("frame-subsampling-factor", int, 1, "Required if the frame-rate of the output (e.g. in 'chain' models) "
                "is less than the frame-rate of the original alignment.")
    po.register("acoustic-scale", float, 0.1, "Scaling factor for acoustic log-likelihoods")
    po.register("frames-per-chunk", int, 50, "Number of frames in each chunk that is separately evaluated by the neural "
                "net (only matters with online ivectors here: without them whole utterances are batched, which gives the "
                "same numbers).")
    po.register("debug-computation", bool, False, "(ignored)")
