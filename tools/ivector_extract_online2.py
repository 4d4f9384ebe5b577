"""ivector-extract-online2 (online2bin/ivector-extract-online2.cc) on the device:

  ivector_extract_online2.py [options] <spk2utt-rspecifier> <feature-rspecifier> <ivector-wspecifier>
  e.g.  ivector_extract_online2.py --config=conf/ivector_extractor.conf ark:data/test/spk2utt scp:data/test/feats.scp \\
            "ark:| copy-feats --compress=true ark:- ark,scp:ivector_online.ark,ivector_online.scp"

Options as in the reference (OnlineIvectorExtractionConfig::Register plus --repeat); the adaptation
state is carried from one utterance of a speaker to the next.  Speakers are independent, so the
k-th utterances of up to --batch speakers are extracted together.  --frame-weights-rspecifier is not
supported."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from kaldi_amd import abi, ivector, table
from kaldi_amd._lib import KamdError, check, lib


def main(argv):
    po = table.ParseOptions(__doc__)
    for name in ("lda-matrix", "global-cmvn-stats", "cmvn-config", "splice-config", "diag-ubm", "ivector-extractor"):
        po.register(name, str, "")
    po.register("ivector-period", int, 10); po.register("num-gselect", int, 5); po.register("min-post", float, 0.025)
    po.register("posterior-scale", float, 0.1); po.register("max-count", float, 0.0)
    po.register("use-most-recent-ivector", bool, True, "(set to false by the binary, as here)")
    po.register("greedy-ivector-extractor", bool, False, "(ignored)")
    po.register("max-remembered-frames", float, 1000.0)
    po.register("num-threads", int, 8, "(ignored)")
    po.register("repeat", bool, False, "If true, output the same number of iVectors as input frames (including repeated data).")
    po.register("frame-weights-rspecifier", str, "", "(not supported)")
    po.register("length-tolerance", int, 0, "(ignored)")
    po.register("batch", int, 64, "Speakers processed together")
    args = po.read(argv)
    if len(args) != 3:
        po.print_usage()
        return 1
    if po["frame-weights-rspecifier"]:
        raise KamdError("--frame-weights-rspecifier is not supported")
    # the options this binary shares with the config-file reader
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".conf", delete=False) as f:
        for name in ("lda-matrix", "global-cmvn-stats", "cmvn-config", "splice-config", "diag-ubm", "ivector-extractor"):
            f.write("--%s=%s\n" % (name, po[name]))
        for name in ("ivector-period", "num-gselect", "min-post", "posterior-scale", "max-count"):
            f.write("--%s=%s\n" % (name, po[name]))
        tmp = f.name
    try:
        info = ivector.IvectorExtractionInfo.from_config(tmp)
    finally:
        os.unlink(tmp)
    ie = ivector.IvectorExtractor(info)
    kind, rx, _ = table.classify_rspecifier(args[0])
    if kind != table.ARCHIVE:
        raise KamdError("the spk2utt rspecifier must be a text archive (ark:data/.../spk2utt)")
    spk2utt = [(spk, rest.split()) for spk, rest in table.read_script_file(rx)]     # "spk utt1 utt2 ...": same line format
    feats = table.RandomAccessTableReader(args[1], "matrix")
    writer = table.TableWriter(args[2], "matrix")
    n_done = n_err = 0
    tot_t = tot_len_end = 0.0
    out = {}
    for b0 in range(0, len(spk2utt), po["batch"]):
        group = spk2utt[b0:b0 + po["batch"]]
        states = {spk: None for spk, _ in group}
        for k in range(max(len(u) for _, u in group)):
            for spk, utts in group:                      # one utterance per speaker and round; rounds are sequential
                if k >= len(utts):
                    continue
                utt = utts[k]
                if utt not in feats:
                    print("WARNING Did not find audio for utterance " + utt, file=sys.stderr)
                    n_err += 1
                    continue
                x = feats[utt]
                iv, states[spk] = ie.extract_online(x, state=states[spk], return_state=True,
                                                    max_remembered_frames=po["max-remembered-frames"])
                if po["repeat"]:
                    iv = np.repeat(iv, info.ivector_period, axis=0)[:x.shape[0]]
                out[utt] = iv
                tot_t += x.shape[0]
                tot_len_end += x.shape[0] * float(np.linalg.norm(iv[-1]))
                n_done += 1
        for spk, utts in group:                          # written in spk2utt order, as the reference does
            for utt in utts:
                if utt in out:
                    writer.write(utt, out.pop(utt))
    writer.close()
    print("LOG Estimated iVectors for %d files, %d with errors." % (n_done, n_err), file=sys.stderr)
    if tot_t:
        print("LOG Average iVector length at utterance-end was %g over %d frames; expected length is %g" %
              (tot_len_end / tot_t, tot_t, np.sqrt(info.ivector_dim)), file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
