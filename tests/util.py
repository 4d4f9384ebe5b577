"""Shared test helpers (readers for the reference's golden data formats)."""
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_wav(path):
    """RIFF PCM16 mono reader == WaveData::Read (feat/wave-reader.cc:272): samples are
    kept at int16 scale as floats."""
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", b[pos + 8:pos + 24])
        elif cid == b"data":
            data = b[pos + 8:pos + 8 + sz]
            break
        pos += 8 + sz + (sz & 1)
    assert fmt[0] == 1 and fmt[1] == 1 and fmt[5] == 16
    return np.frombuffer(data, "<i2").astype(np.float32), fmt[2]


def read_htk(path):
    """ReadHtk (matrix/kaldi-matrix.cc:2257-2330): 12-byte big-endian header + BE floats."""
    b = open(path, "rb").read()
    n, period, size, kind = struct.unpack(">iihh", b[:12])
    cols = size // 4
    return np.frombuffer(b[12:12 + n * cols * 4], ">f4").reshape(n, cols).astype(np.float32), kind


from kaldi_amd.decoder import lattice_diff, lattices_equal  # noqa: E402,F401  (the package holds them: bench.py uses them too)


def assert_work_counters(record, oracle_counters, err_msg=""):
    """The seven work counters of a work-queue utterance against the oracle's.  All seven, bit for bit -- except that an
    utterance with pre-selected frames (kamd_decoder_set_token_preselection; record.n_preselected > 0) counts in N_tok
    (counters[5]) the tokens the lane inserted: never more than the oracle created, and every other counter unchanged."""
    got, want = np.asarray(record.counters[:7]), np.asarray(oracle_counters[:7])
    if getattr(record, "n_preselected", 0) > 0:
        np.testing.assert_array_equal(np.delete(got, 5), np.delete(want, 5), err_msg=err_msg)
        assert 0 < got[5] <= want[5], (err_msg, got[5], want[5])
    else:
        np.testing.assert_array_equal(got, want, err_msg=err_msg)
