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
