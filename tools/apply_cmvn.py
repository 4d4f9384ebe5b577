"""apply-cmvn (featbin/apply-cmvn.cc) on the device:
  apply_cmvn.py [--utt2spk=<rspecifier>] [--norm-means=true] [--norm-vars=false] [--reverse=false] [--skip-dims=0:1:2]
                (<cmvn-stats-rspecifier>|<cmvn-stats-rxfilename>) <feats-rspecifier> <feats-wspecifier>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import cmvn, ivector, table
from kaldi_amd._lib import KamdError


def main(argv):
    po = table.ParseOptions(__doc__)
    po.register("utt2spk", str, "", "rspecifier for utterance to speaker map")
    po.register("norm-vars", bool, False, "If true, normalize variances.")
    po.register("norm-means", bool, True, "You can set this to false to turn off mean normalization.")
    po.register("skip-dims", str, "", "Dimensions for which to skip normalization: colon-separated list of integers, e.g. 13:14:15")
    po.register("reverse", bool, False, "If true, apply CMVN in a reverse sense, so as to transform zero-mean, unit-variance input into data "
                "with the given mean and variance.")
    po.register("batch", int, 256, "utterances per device pass")
    args = po.read(argv)
    if len(args) != 3:
        po.print_usage()
        return 1
    try:
        skip = [int(x) for x in po["skip-dims"].split(":") if x != ""] if po["skip-dims"] else []
    except ValueError:
        raise KamdError("Bad --skip-dims option (should be colon-separated list of integers)")
    if po["norm-vars"] and not po["norm-means"]:
        raise KamdError("You cannot normalize the variance but not the mean.")
    glob = None
    if table.classify_rspecifier(args[0])[0] == table.NO_SPECIFIER:
        glob = ivector.read_kaldi_matrix(args[0], np.float64)
        reader = None
    else:
        reader = table.RandomAccessTableReader(args[0], "dmatrix")       # RandomAccessDoubleMatrixReaderMapped
    utt2spk = {k: v[0] for k, v in table.SequentialTableReader(po["utt2spk"], "tokens")} if po["utt2spk"] else None
    n_done = n_err = 0
    with table.TableWriter(args[2], "matrix") as w:
        batch = []

        def flush():
            nonlocal n_done
            if batch:
                out = cmvn.apply([m for _, m, _ in batch], [s for _, _, s in batch], po["norm-means"], po["norm-vars"],
                                 reverse=po["reverse"], skip_dims=skip)
                for (k, _, _), o in zip(batch, out):
                    w.write(k, o)
                n_done += len(batch)
                batch.clear()

        for key, m in table.SequentialTableReader(args[1], "matrix"):
            if glob is not None:
                st = glob
            else:
                if utt2spk is not None and key not in utt2spk:
                    # RandomAccessTableReaderMapped::HasKey / Value: KALDI_ERR, not a silent per-utterance fallback
                    raise KamdError("Attempting to read key %s, which is not present in utt2spk map or its wxfilename" % key)
                sk = utt2spk[key] if utt2spk is not None else key
                if sk not in reader:
                    print("WARNING No normalization statistics available for key " + key + ", producing no output for this utterance",
                          file=sys.stderr)
                    n_err += 1
                    continue
                st = reader[sk]
            batch.append((key, m, np.asarray(st, np.float64)))
            if len(batch) == po["batch"]:
                flush()
        flush()
    print("LOG Applied cepstral mean %snormalization to %d utterances, errors on %d" % ("and variance " if po["norm-vars"] else "", n_done, n_err),
          file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
