"""compute-mfcc-feats / compute-fbank-feats (featbin/compute-mfcc-feats.cc:30-200, compute-fbank-feats.cc) and
nnet3-compute (nnet3bin/nnet3-compute.cc:35-250) over the device kernels, batches of utterances per launch."""
import ctypes as C
import sys

import numpy as np

from . import abi, decoder, feat, options, table
from ._lib import KamdError, check, lib


def compute_feats(prog, argv):
    fbank = prog == "compute-fbank-feats"
    po = table.ParseOptions(prog)
    (options.register_fbank if fbank else options.register_mfcc)(po)
    po.register("output-format", str, "kaldi", "Format of the output files [kaldi]")
    po.register("subtract-mean", bool, False, "Subtract mean of each feature file [CMS]; not recommended to do it this way.")
    po.register("vtln-warp", float, 1.0, "Vtln warp factor (only applicable if vtln-map not specified)")
    po.register("vtln-map", str, "", "(not supported)")
    po.register("utt2spk", str, "", "(only used with --vtln-map)")
    po.register("channel", int, -1, "Channel to extract (-1 -> expect mono, 0 -> left, 1 -> right)")
    po.register("min-duration", float, 0.0, "Minimum duration of segments to process (in seconds).")
    po.register("batch", int, 128, "utterances per device pass")
    a = po.read(argv)
    if len(a) != 2:
        po.print_usage()
        return 1
    if po["output-format"] != "kaldi" or po["vtln-map"]:
        raise KamdError("--output-format=htk and --vtln-map are not supported")
    opts = options.fbank_opts(po) if fbank else options.mfcc_opts(po)
    F = (feat.Fbank if fbank else feat.Mfcc)(opts, vtln_warp=po["vtln-warp"])
    n_utts = n_ok = 0
    with table.TableWriter(a[1], "matrix") as w:
        batch = []

        def flush():
            nonlocal n_ok
            for key, wave in batch:                          # one launch per utterance here; the pipeline batches them
                m = F.ComputeFeatures(wave)
                if po["subtract-mean"] and m.shape[0]:
                    m = m - m.astype(np.float64).mean(0).astype(np.float32)   # features.AddVecToRows(-1.0, mean) with a float mean
                w.write(key, m); n_ok += 1
            batch.clear()
        for key, (sf, data) in table.SequentialTableReader(a[0], "wave"):
            n_utts += 1
            if data.shape[1] / sf < po["min-duration"]:
                print("WARNING File: %s is too short (%g sec): producing no output." % (key, data.shape[1] / sf), file=sys.stderr)
                continue
            ch = po["channel"]
            if ch == -1:
                ch = 0
                if data.shape[0] != 1:
                    print("WARNING Channel not specified but you have data with %d channels; defaulting to zero" % data.shape[0], file=sys.stderr)
            elif ch >= data.shape[0]:
                print("WARNING File with id %s has %d channels but you specified channel %d, producing no output." % (key, data.shape[0], ch), file=sys.stderr)
                continue
            if sf != opts.frame.samp_freq:
                print("WARNING Failed to compute features for utterance %s (sampling rate %g, expected %g)" % (key, sf, opts.frame.samp_freq), file=sys.stderr)
                continue
            batch.append((key, data[ch]))
            if len(batch) == po["batch"]:
                flush()
        flush()
    print("LOG  Done %d out of %d utterances." % (n_ok, n_utts), file=sys.stderr)
    return 0 if n_ok else 1


def nnet3_compute(argv):
    """nnet3-compute [options] <nnet-in> <features-rspecifier> <matrix-wspecifier>: final.mdl (the raw nnet of an
    AmNnetSimple) -> per-utterance output matrices.  --use-priors / --apply-exp as in the reference."""
    from . import mdl
    po = table.ParseOptions("nnet3-compute")
    options.register_nnet_simple(po)
    po.register("ivectors", str, ""); po.register("utt2spk", str, "")
    po.register("online-ivectors", str, ""); po.register("online-ivector-period", int, 0)
    po.register("apply-exp", bool, False, "If true, apply exp function to output")
    po.register("use-gpu", str, "yes", "(always)")
    po.register("use-priors", bool, False, "If true, subtract the logs of the priors stored with the model")
    a = po.read(argv)
    if len(a) != 3:
        po.print_usage()
        return 1
    if po["ivectors"] and po["online-ivectors"]:
        raise KamdError("--ivectors and --online-ivectors exclude each other")
    model, _, _ = mdl.read_mdl(a[0], 1.0, po["frame-subsampling-factor"])
    if not po["use-priors"]:
        model.layers[-1].post_offset = None
    model.layers[-1].post_scale = 1.0
    N = decoder.Nnet(model)
    ivecs = table.RandomAccessTableReader(po["ivectors"], "vector") if po["ivectors"] else None
    oivecs = table.RandomAccessTableReader(po["online-ivectors"], "matrix") if po["online-ivectors"] else None
    utt2spk = {k: v[0] for k, v in table.SequentialTableReader(po["utt2spk"], "tokens")} if po["utt2spk"] else None
    n_ok = n_fail = 0
    frames = 0
    with table.TableWriter(a[2], "matrix") as w:
        for key, x in table.SequentialTableReader(a[1], "matrix"):
            if x.shape[0] == 0:
                print("WARNING Zero-length utterance: " + key, file=sys.stderr); n_fail += 1
                continue
            if oivecs is not None:
                if key not in oivecs:
                    print("WARNING No online iVector available for utterance " + key, file=sys.stderr); n_fail += 1
                    continue
                y = N.ForwardChunked([x], [oivecs[key]], po["online-ivector-period"], po["frames-per-chunk"])[0]
            elif ivecs is not None:
                sk = utt2spk.get(key, key) if utt2spk else key
                if sk not in ivecs:
                    print("WARNING No iVector available for utterance " + key, file=sys.stderr); n_fail += 1
                    continue
                y = N.Forward(x, ivector=ivecs[sk])
            else:
                y = N.Forward(x)
            if po["apply-exp"]:
                y = np.exp(y)
            w.write(key, y); n_ok += 1; frames += y.shape[0]
    print("LOG Done %d utterances, failed for %d; %d output frames" % (n_ok, n_fail, frames), file=sys.stderr)
    return 0 if n_ok else 1


def run(prog):
    try:
        sys.exit(nnet3_compute(sys.argv) if prog == "nnet3-compute" else compute_feats(prog, sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
