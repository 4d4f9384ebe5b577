"""Input-side formats: RIFF waveforms (pinned by Python's stdlib `wave` writer and by
hand-built headers for the cases feat/wave-reader.cc handles) and Kaldi matrix / int-vector
archives (layout from matrix/kaldi-matrix.cc, matrix/compressed-matrix.cc, base/io-funcs-inl.h)."""
import struct
import wave

import numpy as np
import pytest

from kaldi_amd import io as kio
from kaldi_amd._lib import KamdError


def pcm(n, ch, seed=0):
    return np.random.default_rng(seed).integers(-32768, 32767, (n, ch)).astype("<i2")


@pytest.mark.parametrize("ch,rate", [(1, 16000), (2, 8000)])
def test_wave_written_by_stdlib(tmp_path, ch, rate):
    x = pcm(1234, ch, 1)
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(ch); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(x.tobytes())
    sf, data = kio.read_wave(p)
    assert sf == rate and data.shape == (ch, 1234)
    np.testing.assert_array_equal(data, x.T.astype(np.float32))     # int16 range floats, row per channel


def riff(fmt_chunk, data, pre=b"", mid=b"", riff_size=None, data_size=None, tag=b"RIFF"):
    e = ">" if tag == b"RIFX" else "<"
    body = b"WAVE" + pre + b"fmt " + struct.pack(e + "I", len(fmt_chunk)) + fmt_chunk + mid
    body += b"data" + struct.pack(e + "I", len(data) if data_size is None else data_size) + data
    return tag + struct.pack(e + "I", len(body) if riff_size is None else riff_size) + body


def test_wave_header_variants(tmp_path):
    x = pcm(500, 1, 2)
    fmt = struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16)
    p = tmp_path / "v.wav"
    # filler chunk before fmt (Apple's JUNK), fact + LIST chunks before data
    junk = b"JUNK" + struct.pack("<I", 6) + b"\0" * 6
    fact = b"fact" + struct.pack("<I", 4) + struct.pack("<I", 500) + b"LIST" + struct.pack("<I", 4) + b"abcd"
    p.write_bytes(riff(fmt, x.tobytes(), pre=junk, mid=fact))
    np.testing.assert_array_equal(kio.read_wave(p)[1][0], x[:, 0].astype(np.float32))
    # WAVE_FORMAT_EXTENSIBLE with the PCM GUID
    ext = struct.pack("<HHIIHH", 0xFFFE, 1, 16000, 32000, 2, 16) + struct.pack("<HHI", 22, 16, 4)
    ext += struct.pack("<IIII", 0x00000001, 0x00100000, 0xAA000080, 0x719B3800)
    p.write_bytes(riff(ext, x.tobytes()))
    np.testing.assert_array_equal(kio.read_wave(p)[1][0], x[:, 0].astype(np.float32))
    # streamed sizes (SoX writes 0x7FFFF000): read to the end of the file
    p.write_bytes(riff(fmt, x.tobytes(), riff_size=0xFFFFFFFF, data_size=0x7FFFF000))
    assert kio.read_wave(p)[1].shape == (1, 500)
    # big-endian container
    fmtb = struct.pack(">HHIIHH", 1, 1, 16000, 32000, 2, 16)
    p.write_bytes(riff(fmtb, x.astype(">i2").tobytes(), tag=b"RIFX"))
    np.testing.assert_array_equal(kio.read_wave(p)[1][0], x[:, 0].astype(np.float32))
    # rejections (feat/wave-reader.cc:166-207)
    p.write_bytes(riff(struct.pack("<HHIIHH", 3, 1, 16000, 64000, 4, 32), b"\0" * 16))
    with pytest.raises(KamdError, match="only PCM"):
        kio.read_wave(p)
    p.write_bytes(riff(struct.pack("<HHIIHH", 1, 1, 16000, 16000, 1, 8), b"\0" * 16))
    with pytest.raises(KamdError, match="bits_per_sample"):
        kio.read_wave(p)
    p.write_bytes(b"RIFFxxxxWAVX")
    with pytest.raises(KamdError, match="WAVE"):
        kio.read_wave(p)


@pytest.mark.parametrize("binary", [True, False])
def test_matrix_archive_round_trip(tmp_path, binary):
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((7, 5)).astype(np.float32), rng.standard_normal((1, 3)).astype(np.float32)
    p = tmp_path / "m.ark"
    kio.write_matrix_ark(p, "utt1", a, binary=binary, append=False)
    kio.write_matrix_ark(p, "utt2", b, binary=binary)
    kio.write_matrix_ark(p, "empty", np.zeros((0, 0), np.float32), binary=binary)
    got = list(kio.read_matrix_ark(p))
    assert [k for k, _ in got] == ["utt1", "utt2", "empty"]
    np.testing.assert_array_equal(got[0][1], a)
    np.testing.assert_array_equal(got[1][1], b)
    assert got[2][1].size == 0
    if binary:
        raw = p.read_bytes()
        assert raw.startswith(b"utt1 \0BFM \x04" + struct.pack("<i", 7) + b"\x04" + struct.pack("<i", 5))
    else:
        assert p.read_text().startswith("utt1  [\n  ")


def test_double_and_compressed_matrices(tmp_path):
    p = tmp_path / "c.ark"
    d = np.arange(6, dtype="<f8").reshape(2, 3) / 7
    # DM
    p.write_bytes(b"k \0BDM \x04" + struct.pack("<i", 2) + b"\x04" + struct.pack("<i", 3) + d.tobytes())
    (k, m), = list(kio.read_matrix_ark(p))
    np.testing.assert_allclose(m, d, rtol=1e-7)
    # CM2: uint16, value = min + range * v / 65535 (compressed-matrix.cc:371-377)
    v = np.array([[0, 65535, 100], [3, 4, 5]], "<u2")
    p.write_bytes(b"k \0BCM2 " + struct.pack("<ffii", -1.0, 2.0, 2, 3) + v.tobytes())
    (k, m), = list(kio.read_matrix_ark(p))
    np.testing.assert_allclose(m, -1.0 + 2.0 * v / 65535.0, rtol=1e-6, atol=1e-6)
    # CM3: uint8
    v8 = np.array([[0, 255, 7]], "u1")
    p.write_bytes(b"k \0BCM3 " + struct.pack("<ffii", 0.5, 4.0, 1, 3) + v8.tobytes())
    (k, m), = list(kio.read_matrix_ark(p))
    np.testing.assert_allclose(m, 0.5 + 4.0 * v8 / 255.0, rtol=1e-6)
    # CM: per-column percentile headers, column-major bytes, piecewise-linear decode (:490-500)
    hdr = np.array([[0, 16384, 49151, 65535], [0, 100, 200, 300]], "<u2")          # [cols][4]
    by = np.array([[0, 64, 192, 255], [10, 100, 200, 250]], "u1")                 # [cols][rows]
    p.write_bytes(b"k \0BCM " + struct.pack("<ffii", 0.0, 65535.0, 4, 2) + hdr.tobytes() + by.tobytes())
    (k, m), = list(kio.read_matrix_ark(p))
    def dec(h, b):
        p0, p25, p75, p100 = [float(x) for x in h]
        return p0 + (p25 - p0) * b / 64 if b <= 64 else (p25 + (p75 - p25) * (b - 64) / 128 if b <= 192 else p75 + (p100 - p75) * (b - 192) / 63)
    want = np.array([[dec(hdr[c], by[c, r]) for c in range(2)] for r in range(4)])
    np.testing.assert_allclose(m, want, rtol=1e-5, atol=1e-3)


def test_int32_vector_archives(tmp_path):
    p = tmp_path / "ali.ark"
    # BasicVectorHolder<int32> (util/kaldi-holder-inl.h:230-250): every integer carries its size byte
    def i32(x):
        return b"\x04" + struct.pack("<i", x)
    p.write_bytes(b"u1 \0B" + i32(3) + i32(5) + i32(6) + i32(7) + b"u2 \0B" + i32(0))
    got = list(kio.read_int32_vector_ark(p))
    assert got[0][0] == "u1" and got[0][1].tolist() == [5, 6, 7] and got[1][1].size == 0
    p.write_text("u1 1 2 3 \nu2 \nu3 42\n")
    got = list(kio.read_int32_vector_ark(p))
    assert [(k, v.tolist()) for k, v in got] == [("u1", [1, 2, 3]), ("u2", []), ("u3", [42])]


def _wav_bytes(riff_len, channels, hz, data_len, samples):
    """the byte layout of the reference's wave-reader-test.cc cases: fmt chunk of 18 bytes (WAVEFORMATEX with cbSize)"""
    import struct
    byps = hz * channels * 2
    return (b"RIFF" + struct.pack("<I", riff_len & 0xFFFFFFFF) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHHH", 18, 1, channels, hz, byps, 2 * channels, 16, 0)
            + b"data" + struct.pack("<I", data_len & 0xFFFFFFFF) + struct.pack("<%dh" % len(samples), *samples))


@pytest.mark.parametrize("name,args,hz,want", [
    # feat/wave-reader-test.cc:32-80 UnitTestStereo8K: interleaved L R L R L R -> two rows
    ("stereo8k", (50, 2, 8000, 12, [0, -1, -32768, 0, 32767, 1]), 8000, [[0, -32768, 32767], [-1, 0, 1]]),
    # :82-125 UnitTestMono22K
    ("mono22k", (48, 1, 22050, 10, [0, -1, -32768, 32767, 1]), 22050, [[0, -1, -32768, 32767, 1]]),
    # :127-162 UnitTestEndless1: RIFF and data lengths 0 = "unknown", read to the end of the stream
    ("endless1", (0, 1, 8000, 0, [1, 2, 3]), 8000, [[1, 2, 3]]),
    # :164-199 UnitTestEndless2: lengths 0xFFFFFFFF
    ("endless2", (-1, 1, 8000, -1, [1, 2, 3]), 8000, [[1, 2, 3]]),
])
def test_wave_reader_reference_known_answers(tmp_path, name, args, hz, want):
    """WaveData::Read pinned to the four byte-level cases of the reference's own unit test (restated as data)."""
    p = tmp_path / (name + ".wav")
    p.write_bytes(_wav_bytes(*args))
    sf, data = kio.read_wave(p)
    assert sf == hz
    np.testing.assert_array_equal(data, np.asarray(want, np.float32))
    assert abs(data.shape[1] / sf - len(want[0]) / hz) < 1e-6           # Duration()


def test_corrupt_counts_fail_before_anything_is_allocated(tmp_path):
    """A header that announces more data than the file holds is an error message, not a multi-gigabyte allocation (and
    no exception crosses the C ABI)."""
    p = tmp_path / "huge.ark"
    p.write_bytes(b"utt \0BFM \x04" + struct.pack("<i", 2000000000) + b"\x04" + struct.pack("<i", 2000000000) + b"\0" * 64)
    with pytest.raises(KamdError, match="exceeds the file size"):
        list(kio.read_matrix_ark(p))
    p.write_bytes(b"utt \0BDV \x04" + struct.pack("<i", 2000000000) + b"\0" * 64)
    with pytest.raises(KamdError, match="exceeds the file size"):
        list(kio.read_matrix_ark(p))
    p.write_bytes(b"utt \0BCM2 " + struct.pack("<ffii", 0.0, 1.0, 1000000, 1000000) + b"\0" * 64)
    with pytest.raises(KamdError, match="exceeds the file size"):
        list(kio.read_matrix_ark(p))
    p.write_bytes(b"utt \0B\x04" + struct.pack("<i", 2000000000) + b"\x04" + struct.pack("<i", 7))
    with pytest.raises(KamdError, match="exceeds the file size"):
        list(kio.read_int32_vector_ark(p))
