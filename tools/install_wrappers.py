"""Writes one-line shell wrappers named like Kaldi's binaries into a directory that can be put in front of the
Kaldi bin directories in path.sh:  install_wrappers.py <dir>
(device tools get --device=${KAMD_DEVICE:-0} where they take it)."""
import os
import stat
import sys

TOOLS = {"nnet3-latgen-faster": ("nnet3_latgen_faster.py", True), "nnet3-latgen-faster-parallel": ("nnet3_latgen_faster.py", True),
         "nnet3-latgen-faster-batch": ("nnet3_latgen_faster.py", True),
         "online2-wav-nnet3-latgen-faster": ("online2_wav_nnet3_latgen_faster.py", False), "latgen-faster-mapped": ("latgen_faster_mapped.py", False),
         "gmm-latgen-faster": ("gmm_latgen_faster.py", False), "nnet3-compute": ("nnet3_compute.py", False),
         "compute-mfcc-feats": ("compute_mfcc_feats.py", False), "compute-fbank-feats": ("compute_fbank_feats.py", False),
         "compute-cmvn-stats": ("compute_cmvn_stats.py", False), "apply-cmvn": ("apply_cmvn.py", False), "add-deltas": ("add_deltas.py", False),
         "splice-feats": ("splice_feats.py", False), "transform-feats": ("transform_feats.py", False),
         "ivector-extract-online2": ("ivector_extract_online2.py", False), "lattice-scale": ("lattice_scale.py", False),
         "lattice-add-penalty": ("lattice_add_penalty.py", False), "lattice-best-path": ("lattice_best_path.py", False),
         "compute-wer": ("compute_wer.py", False)}


def main():
    if len(sys.argv) != 2:
        print(__doc__, file=sys.stderr)
        return 1
    out = sys.argv[1]
    here = os.path.dirname(os.path.abspath(__file__))
    os.makedirs(out, exist_ok=True)
    for name, (script, has_device) in TOOLS.items():
        p = os.path.join(out, name)
        with open(p, "w") as f:
            f.write("#!/bin/sh\nexec %s %s%s \"$@\"\n" % (sys.executable, os.path.join(here, script), " --device=${KAMD_DEVICE:-0}" if has_device else ""))
        os.chmod(p, os.stat(p).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    print("wrote %d wrappers to %s" % (len(TOOLS), out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
