"""GPU parity: HIP MFCC/fbank (through the C-ABI) vs the CPU oracle and vs the
reference's HTK golden vectors.  Tolerance (fp32, different FFT/summation order):
|delta| <= 2e-3 absolute on log-domain features (values span roughly [-50, 30])."""
import os

import numpy as np
import pytest

from kaldi_amd import abi, feat, synth
from oracle import orc
from tests.test_oracle_feat import FEAT, fbank_htk_cases, mfcc_htk_cases
from tests.util import read_htk, read_wav

pytestmark = pytest.mark.gpu
TOL = 2e-3


@pytest.fixture(scope="module")
def wave():
    return read_wav(os.path.join(FEAT, "test.wav"))[0]


@pytest.mark.parametrize("case,op,warp", mfcc_htk_cases())
def test_mfcc_vs_oracle_and_htk(case, op, warp, wave):
    got = feat.Mfcc(op, warp).ComputeFeatures(wave, 16000)
    ref = orc.mfcc(op, wave, warp)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < TOL
    htk, _ = read_htk(os.path.join(FEAT, "test.wav.fea_htk.%d" % case))
    assert np.abs(got[10:-10] - htk[10:-10, :got.shape[1]]).max() < 0.1


@pytest.mark.parametrize("case,op,warp", fbank_htk_cases())
def test_fbank_vs_oracle_and_htk(case, op, warp, wave):
    got = feat.Fbank(op, warp).ComputeFeatures(wave, 16000)
    ref = orc.fbank(op, wave, warp)
    assert np.abs(got - ref).max() < TOL
    htk, _ = read_htk(os.path.join(FEAT, "test.wav.fbank_htk.%d" % case))
    d = np.abs(got[10:-10] - htk[10:-10])
    if case == 3:
        d = d[:, :20]
    assert d.max() < (0.01 if case == 4 else 0.001) + TOL


@pytest.mark.parametrize("snip", [1, 0])
@pytest.mark.parametrize("seconds", [0.03, 0.5, 3.7])
def test_hires_mfcc_synthetic(snip, seconds):
    op = abi.mfcc_opts_hires()
    op.frame.snip_edges = snip
    w = synth.make_wave(seconds, seed=int(seconds * 100))
    got = feat.Mfcc(op).ComputeFeatures(w)
    ref = orc.mfcc(op, w)
    assert got.shape == ref.shape and got.shape[1] == 40
    if got.size:
        assert np.abs(got - ref).max() < TOL


def test_too_short_wave_gives_zero_frames():
    got = feat.Mfcc(abi.mfcc_opts_hires()).ComputeFeatures(np.zeros(399, np.float32))
    assert got.shape == (0, 40)


def test_energy_variants():
    w = synth.make_wave(0.8, seed=4)
    for raw, floor in ((1, 0.0), (0, 0.0), (1, 1e9)):
        op = abi.mfcc_opts_default()
        op.raw_energy, op.energy_floor = raw, floor
        got = feat.Mfcc(op).ComputeFeatures(w)
        assert np.abs(got - orc.mfcc(op, w)).max() < TOL
    fo = abi.fbank_opts_default()
    fo.use_energy, fo.use_power, fo.mel.num_bins = 1, 0, 40
    assert np.abs(feat.Fbank(fo).ComputeFeatures(w) - orc.fbank(fo, w)).max() < TOL


def test_cmvn_stats_and_apply_match_the_oracle(tmp_path):
    """compute-cmvn-stats / apply-cmvn on the device: statistics 1e-12 relative (fp64, other order of summation),
    normalised features bit-equal given the same statistics (same float operations), per utterance and per speaker;
    the two command-line tools chained through files."""
    import subprocess
    import sys
    from kaldi_amd import cmvn, table
    rng = np.random.default_rng(0)
    mats = [(rng.standard_normal((T, 13)) * np.linspace(0.5, 4, 13) + 2).astype(np.float32) for T in (1, 37, 300, 1000)]
    st = cmvn.acc_stats(mats)
    for m, s in zip(mats, st):
        want = orc.cmvn_acc_stats(m)
        np.testing.assert_allclose(s, want, rtol=1e-12, atol=1e-12)
    for nv in (False, True):
        got = cmvn.apply(mats[1:], [orc.cmvn_acc_stats(m) for m in mats[1:]], norm_vars=nv)
        for m, g in zip(mats[1:], got):
            np.testing.assert_array_equal(g, orc.cmvn_apply(m, orc.cmvn_acc_stats(m), nv))
    same = cmvn.apply(mats, st, norm_means=False)
    for m, g in zip(mats, same):
        np.testing.assert_array_equal(m, g)
    with pytest.raises(Exception):
        cmvn.apply(mats[:1], np.zeros((1, 2, 14)))                      # "Insufficient stats"
    with pytest.raises(Exception):
        cmvn.apply(mats[:1], st[:1], norm_means=False, norm_vars=True)
    # running statistics: a speaker's second utterance added to the first
    both = cmvn.acc_stats([mats[2]], stats=cmvn.acc_stats([mats[1]]))
    np.testing.assert_allclose(both[0], orc.cmvn_acc_stats(mats[2], orc.cmvn_acc_stats(mats[1])), rtol=1e-12)
    # tools: per-speaker statistics, then apply with utt2spk
    with table.TableWriter("ark:%s" % (tmp_path / "f.ark"), "matrix") as w:
        for i, m in enumerate(mats):
            w.write("u%d" % i, m)
    (tmp_path / "spk2utt").write_text("A u0 u1\nB u2 u3\n")
    (tmp_path / "utt2spk").write_text("u0 A\nu1 A\nu2 B\nu3 B\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cmd in ([root + "/tools/compute_cmvn_stats.py", "--spk2utt=ark:%s" % (tmp_path / "spk2utt"), "ark:%s" % (tmp_path / "f.ark"),
                 "ark:%s" % (tmp_path / "cmvn.ark")],
                [root + "/tools/apply_cmvn.py", "--norm-vars=true", "--utt2spk=ark:%s" % (tmp_path / "utt2spk"), "ark:%s" % (tmp_path / "cmvn.ark"),
                 "ark:%s" % (tmp_path / "f.ark"), "ark:%s" % (tmp_path / "out.ark")]):
        r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
    out = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "out.ark"), "matrix"))
    spkB = np.concatenate(mats[2:])
    np.testing.assert_allclose(np.concatenate([out["u2"], out["u3"]]).mean(0), 0, atol=2e-5)
    np.testing.assert_allclose(np.concatenate([out["u2"], out["u3"]]).std(0), 1, atol=1e-4)
    assert abs(float(spkB.mean())) > 1                                   # it was not normalised before
