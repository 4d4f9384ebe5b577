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


def lattices_equal(a, b):
    """Exact (bit-level) equality of two canonical raw lattices."""
    if a is None or b is None:
        return a is b
    ok = (a.start == b.start and a.num_frames == b.num_frames
          and np.array_equal(a.frame, b.frame) and np.array_equal(a.hclg, b.hclg)
          and np.array_equal(a.cost.view(np.uint32), b.cost.view(np.uint32))
          and np.array_equal(a.final.view(np.uint32), b.final.view(np.uint32))
          and a.arcs.size == b.arcs.size and a.arcs.tobytes() == b.arcs.tobytes())
    return ok


def lattice_diff(a, b):
    """Human-readable summary of where two canonical lattices differ."""
    out = ["states %d vs %d, arcs %d vs %d, start %d vs %d" % (
        a.frame.size, b.frame.size, a.arcs.size, b.arcs.size, a.start, b.start)]
    ka = set(zip(a.frame.tolist(), a.hclg.tolist()))
    kb = set(zip(b.frame.tolist(), b.hclg.tolist()))
    out.append("tokens only in A: %s" % sorted(ka - kb)[:10])
    out.append("tokens only in B: %s" % sorted(kb - ka)[:10])
    if ka == kb:
        bad = np.nonzero(a.cost.view(np.uint32) != b.cost.view(np.uint32))[0]
        out.append("cost mismatches: %d %s" % (bad.size, [(int(i), float(a.cost[i]), float(b.cost[i])) for i in bad[:5]]))
        sa = set(map(bytes, a.arcs.view("V24")))
        sb = set(map(bytes, b.arcs.view("V24")))
        out.append("arcs only in A: %d, only in B: %d" % (len(sa - sb), len(sb - sa)))
    return "\n".join(out)
