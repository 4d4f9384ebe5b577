#!/usr/bin/env python3
"""arpa-to-const-arpa (lmbin/arpa-to-const-arpa.cc): arpa-to-const-arpa --bos-symbol=B --eos-symbol=E [--unk-symbol=U] <arpa-rxfilename> <const-arpa-wxfilename>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import constarpa, table  # noqa: E402
from kaldi_amd._lib import KamdError  # noqa: E402


def main(argv):
    po = table.ParseOptions("Converts an Arpa format language model into ConstArpaLm format.\n"
                            "Usage: arpa-to-const-arpa [opts] <input-arpa> <const-arpa>")
    po.register("bos-symbol", int, -1, "Integer corresponds to <s>. You must set this to your actual BOS integer.")
    po.register("eos-symbol", int, -1, "Integer corresponds to </s>. You must set this to your actual EOS integer.")
    po.register("unk-symbol", int, -1, "Integer corresponds to unknown-word in language model. -1 if no such word is provided.")
    po.register("read-symbol-table", str, "", "Use this file as the symbol table, or empty if the LM already holds integers")
    try:
        args = po.read(argv)
        if len(args) != 2:
            po.print_usage()
            return 1
        with table.Input(args[0]) as (path, off):
            if off:
                raise KamdError("arpa input with an offset is not supported")
            lm = constarpa.ConstArpaLm.build(path, po["bos-symbol"], po["eos-symbol"], po["unk-symbol"], po["read-symbol-table"] or None)
        lm.write(args[1])
        print("LOG Wrote %s: order %d, %d words, %d ints of LM states" % (args[1], lm.order, lm.num_words, lm.lm_states_size), file=sys.stderr)
        return 0
    except KamdError as e:
        print("ERROR " + str(e), file=sys.stderr)
        return 255


if __name__ == "__main__":
    sys.exit(main(sys.argv))
