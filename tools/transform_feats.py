"""transform-feats (featbin/transform-feats.cc) on the device:
  transform_feats.py [--utt2spk=<rspecifier>] (<transform-rspecifier>|<transform-rxfilename>) <feats-rspecifier> <feats-wspecifier>
A plain filename is one matrix for everything (LDA+MLLT final.mat); an rspecifier gives per-utterance or, with --utt2spk,
per-speaker matrices (fMLLR)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import cmvn, ivector, table
from kaldi_amd._lib import KamdError


def main(argv):
    po = table.ParseOptions(__doc__)
    po.register("utt2spk", str, "", "rspecifier for utterance to speaker map")
    a = po.read(argv)
    if len(a) != 3:
        po.print_usage()
        return 1
    one = reader = None
    if table.classify_rspecifier(a[0])[0] == table.NO_SPECIFIER:
        one = ivector.read_kaldi_matrix(a[0])
    else:
        reader = table.RandomAccessTableReader(a[0], "matrix")
    utt2spk = {k: v[0] for k, v in table.SequentialTableReader(po["utt2spk"], "tokens")} if po["utt2spk"] else None
    n_done = n_err = 0
    with table.TableWriter(a[2], "matrix") as w:
        for key, m in table.SequentialTableReader(a[1], "matrix"):
            xf = one
            if xf is None:
                sk = utt2spk.get(key, key) if utt2spk else key
                if sk not in reader:
                    print("WARNING No fMLLR transform available for utterance %s, producing no output for this utterance" % key, file=sys.stderr)
                    n_err += 1
                    continue
                xf = reader[sk]
            w.write(key, cmvn.splice_transform([m], 0, 0, transforms=xf)[0]); n_done += 1
    print("LOG Applied transform to %d utterances; %d had errors." % (n_done, n_err), file=sys.stderr)
    return 0 if n_done else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
