#!/usr/bin/env python3
"""One fused layer of a given shape, timed alone on the device: TFLOP/s of the GEMM kernels in isolation.
  python tools/gemm_probe.py [rows]            (run under rocprofv3 --pmc for the counters)"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from kaldi_amd import abi, decoder, nnet  # noqa: E402
from kaldi_amd._lib import check, lib  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
rng = np.random.default_rng(0)
shapes = [("linear 2x1536 -> 160", 1536, 160, [-1, 0]), ("affine 2x160 -> 1536", 160, 1536, [0, 1]),
          ("output 256 -> 6000", 256, 6000, [0]), ("tdnn1 3x... 224 -> 1536", 224, 1536, [0])]
for name, din, dout, offs in shapes:
    W = (rng.standard_normal((dout, len(offs) * din)) / np.sqrt(len(offs) * din)).astype(np.float32)
    L = nnet.Layer("l", din, dout, offs, -1, W, bias=np.zeros(dout, np.float32), relu=True)
    m = nnet.Model([L], din, 0, subsampling=1, num_pdfs=dout)
    N = decoder.Nnet(m)
    ld = (din + 15) // 16 * 16
    x = rng.standard_normal((rows, ld)).astype(np.float32)
    d_x = decoder.DeviceMatrix(x)
    d_o = decoder.DeviceMatrix(np.zeros((rows, dout), np.float32))
    in_off = np.asarray([0, rows], np.int64)
    out_off = np.asarray([0], np.int64)
    for rep in range(3):
        check(lib().kamd_device_synchronize())
        t0 = time.time()
        check(lib().kamd_nnet_forward_batch_device(N._h, d_x.ptr(0), abi.iptr(in_off, C.c_int64), ld, None, 1, d_o.ptr(0),
                                                   abi.iptr(out_off, C.c_int64), dout, None))
        check(lib().kamd_device_synchronize())
        dt = time.time() - t0
    fl = lib().kamd_nnet_last_flops(N._h)
    # numerics of this shape's kernel: rows away from the edges against float64 numpy
    got = d_o.download()
    sel = np.asarray([5, 77, rows // 2, rows - 9])
    want = np.zeros((sel.size, dout))
    for j, o in enumerate(offs):
        want += x[sel + o, :din].astype(np.float64) @ W[:, j * din:(j + 1) * din].astype(np.float64).T
    err = float(np.abs(got[sel] - np.maximum(want, 0.0)).max())
    print("%-28s rows %d: %.2f ms  %.1f TFLOP/s  max|err| %.2e" % (name, rows, dt * 1e3, fl / dt / 1e12, err), flush=True)
