"""add-deltas (featbin/add-deltas.cc) on the device:  add_deltas.py [--delta-order=2 --delta-window=2] in-rspecifier out-wspecifier"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import cmvn, table
from kaldi_amd._lib import KamdError


def main(argv):
    po = table.ParseOptions(__doc__)
    po.register("delta-order", int, 2, "Order of delta computation")
    po.register("delta-window", int, 2, "Parameter controlling window for delta computation (actual window size for each delta order is 1 + 2*delta-window-size)")
    po.register("truncate", int, 0, "If nonzero, first truncate features to this dimension.")
    po.register("batch", int, 256, "utterances per device pass")
    a = po.read(argv)
    if len(a) != 2:
        po.print_usage()
        return 1
    n = 0
    with table.TableWriter(a[1], "matrix") as w:
        batch = []

        def flush():
            nonlocal n
            if batch:
                for (k, _), o in zip(batch, cmvn.add_deltas([m for _, m in batch], po["delta-order"], po["delta-window"])):
                    w.write(k, o); n += 1
                batch.clear()
        for key, m in table.SequentialTableReader(a[0], "matrix"):
            if m.shape[0] == 0:
                print("WARNING Empty feature matrix for key " + key, file=sys.stderr)
                continue
            if po["truncate"]:
                if po["truncate"] > m.shape[1]:
                    raise KamdError("Cannot truncate features as dimension %d is smaller than truncation dimension." % m.shape[1])
                m = m[:, :po["truncate"]]
            batch.append((key, m))
            if len(batch) == po["batch"]:
                flush()
        flush()
    print("LOG Done %d utterances." % n, file=sys.stderr)
    return 0 if n else 1


if __name__ == "__main__":
    try:
        sys.exit(main(sys.argv))
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        sys.exit(255)
