#!/usr/bin/env python3
"""lattice-lmrescore-const-arpa (latbin/lattice-lmrescore-const-arpa.cc):
lattice-lmrescore-const-arpa [--lm-scale=1.0] <lattice-rspecifier> <const-arpa-in> <lattice-wspecifier>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import constarpa, latbin, table  # noqa: E402
from kaldi_amd._lib import KamdError  # noqa: E402


def main(argv):
    po = table.ParseOptions("Rescores lattice with the ConstArpaLm format language model.\n"
                            "Usage: lattice-lmrescore-const-arpa [options] lattice-rspecifier const-arpa-in lattice-wspecifier")
    po.register("lm-scale", float, 1.0, "Scaling factor for language model costs; frequently 1.0 or -1.0")
    try:
        args = po.read(argv)
        if len(args) != 3:
            po.print_usage()
            return 1
        lm = constarpa.ConstArpaLm.read(args[1])
        n_done = n_fail = 0
        with table.TableWriter(args[2], "raw") as w:
            for key, lat in latbin.read_lattices(args[0]):
                out = lm.rescore(lat, po["lm-scale"])
                if out is None:
                    print("WARNING Empty lattice for utterance %s (incompatible LM?)" % key, file=sys.stderr)
                    n_fail += 1
                else:
                    w.write(key, latbin.compact_bytes(out, w.opts["binary"]))
                    n_done += 1
        print("LOG Done %d lattices, failed for %d" % (n_done, n_fail), file=sys.stderr)
        return 0 if n_done else 1
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        return 255


if __name__ == "__main__":
    sys.exit(main(sys.argv))
