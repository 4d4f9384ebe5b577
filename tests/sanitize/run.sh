#!/bin/bash
# AddressSanitizer + UBSan over the host-only sources (OpenFst / lattice I/O, determinization),
# built with g++ against a stub of the device-facing helpers.  GPU sanitizers are not available
# on the pool; the HIP sources are covered by the bit-exact parity tests instead.
set -e
D=$(cd "$(dirname "$0")" && pwd)
mkdir -p "$D/_build"
# the sources include "common.h" (HIP headers): build copies next to a HIP-free stand-in of it
cp "$D/../../kaldi_amd/csrc/fst_io.cc" "$D/../../kaldi_amd/csrc/determinize.cc" "$D/../../kaldi_amd/csrc/kaldi_io.cc" "$D/../../kaldi_amd/csrc/table.cc" "$D/_build/"
sed 's#"../../include/kaldi_amd.h"#"'"$D"'/../../include/kaldi_amd.h"#' "$D/common.h" > "$D/_build/common.h"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC \
    -o "$D/_build/libhost_asan.so" "$D/stub.cc" "$D/_build/fst_io.cc" "$D/_build/determinize.cc" "$D/_build/kaldi_io.cc" "$D/_build/table.cc"
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python3 "$D/run.py"
