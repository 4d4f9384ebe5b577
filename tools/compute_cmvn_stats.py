"""compute-cmvn-stats (featbin/compute-cmvn-stats.cc) on the device:
  compute_cmvn_stats.py [--spk2utt=<rspecifier>] <feats-rspecifier> (<stats-wspecifier>|<stats-wxfilename>)
Per utterance by default, per speaker with --spk2utt; a plain output filename gets the statistics
summed over all utterances (the global stats the i-vector extractor reads).  --weights is not supported."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import cmvn, ivector, table
from kaldi_amd._lib import KamdError


def main(argv):
    po = table.ParseOptions(__doc__)
    po.register("spk2utt", str, "", "rspecifier for speaker to utterance-list map")
    po.register("binary", bool, True, "write in binary mode (applies only to global CMN/CVN)")
    po.register("weights", str, "", "rspecifier for a vector of floats for each utterance, that's a per-frame weight.")
    po.register("batch", int, 256, "utterances per device pass")
    args = po.read(argv)
    if len(args) != 2:
        po.print_usage()
        return 1
    feats = list(table.SequentialTableReader(args[0], "matrix"))
    weights = table.RandomAccessTableReader(po["weights"], "vector") if po["weights"] else None
    n_err = 0
    if weights is not None:                                  # AccCmvnStatsWrapper (compute-cmvn-stats.cc:27-48)
        kept = []
        for k, m in feats:
            if k not in weights:
                print("WARNING No weights available for utterance " + k, file=sys.stderr); n_err += 1
            elif np.asarray(weights[k]).size != m.shape[0]:
                print("WARNING Weights for utterance %s have wrong dimension %d vs. %d" % (k, np.asarray(weights[k]).size, m.shape[0]), file=sys.stderr)
                n_err += 1
            else:
                kept.append((k, m))
        feats = kept
    stats = {}
    for b0 in range(0, len(feats), po["batch"]):
        chunk = feats[b0:b0 + po["batch"]]
        w = None if weights is None else [weights[k] for k, _ in chunk]
        for (k, _), st in zip(chunk, cmvn.acc_stats([m for _, m in chunk], weights=w)):
            stats[k] = st
    n_done = len(stats)
    if table.classify_wspecifier(args[1])[0] == table.NO_SPECIFIER:          # global statistics to a file
        tot = sum(stats.values())
        if not po["binary"]:
            raise KamdError("text-mode output of a single matrix is not supported")
        ivector.write_kaldi_matrix(args[1], np.asarray(tot, np.float64))
        print("LOG Wrote global CMVN stats to " + args[1], file=sys.stderr)
    else:
        with table.TableWriter(args[1], "dmatrix") as w:                  # DoubleMatrixWriter (compute-cmvn-stats.cc:93)
            if po["spk2utt"]:
                for spk, utts in table.SequentialTableReader(po["spk2utt"], "tokens"):
                    have = [stats[u] for u in utts if u in stats]
                    if not have:
                        print("WARNING No stats accumulated for speaker " + spk, file=sys.stderr)
                        continue
                    w.write(spk, sum(have))
            else:
                for k, _ in feats:
                    w.write(k, stats[k])
    print("LOG Done accumulating CMVN stats for %d utterances; %d had errors." % (n_done, n_err), file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
