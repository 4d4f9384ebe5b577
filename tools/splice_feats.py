"""splice-feats (featbin/splice-feats.cc) on the device:  splice_feats.py [--left-context=4 --right-context=4] in-rspecifier out-wspecifier"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import cmvn, table
from kaldi_amd._lib import KamdError


def main(argv):
    po = table.ParseOptions(__doc__)
    po.register("left-context", int, 4, "Number of frames of left context")
    po.register("right-context", int, 4, "Number of frames of right context")
    a = po.read(argv)
    if len(a) != 2:
        po.print_usage()
        return 1
    n = 0
    with table.TableWriter(a[1], "matrix") as w:
        for key, m in table.SequentialTableReader(a[0], "matrix"):
            w.write(key, cmvn.splice_transform([m], po["left-context"], po["right-context"])[0]); n += 1
    print("LOG Spliced %d feature matrices." % n, file=sys.stderr)
    return 0 if n else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
