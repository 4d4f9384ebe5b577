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
    # --weights: per-frame weights (zeros skipped, count = sum of the weights)
    wts = [np.where(rng.random(m.shape[0]) < 0.3, 0.0, rng.random(m.shape[0]) * 2).astype(np.float32) for m in mats]
    for s1, m, w in zip(cmvn.acc_stats(mats, weights=wts), mats, wts):
        want = orc.cmvn_acc_stats(m, weights=w)
        np.testing.assert_allclose(s1, want, rtol=1e-12, atol=1e-12)
        assert abs(s1[0, -1] - w.astype(np.float64).sum()) < 1e-9
    with pytest.raises(Exception):
        cmvn.acc_stats(mats[:1], weights=[np.ones(5, np.float32)])
    # --reverse (ApplyCmvnReverse) and --skip-dims (FakeStatsForSomeDims): bit-equal to the oracle; skipped dimensions
    # come out untouched without variance normalisation, and reverse undoes forward up to rounding
    for nv in (False, True):
        stats = [orc.cmvn_acc_stats(m) for m in mats[1:]]
        rev = cmvn.apply(mats[1:], stats, norm_vars=nv, reverse=True)
        for m, s1, g in zip(mats[1:], stats, rev):
            np.testing.assert_array_equal(g, orc.cmvn_apply(m, s1, nv, reverse=True))
        back = cmvn.apply(cmvn.apply(mats[1:], stats, norm_vars=nv), stats, norm_vars=nv, reverse=True)
        for m, b in zip(mats[1:], back):
            np.testing.assert_allclose(b, m, atol=2e-5 * np.abs(m).max())
        sk = cmvn.apply(mats[1:], stats, norm_vars=nv, skip_dims=[0, 5, 12])
        for m, s1, g in zip(mats[1:], stats, sk):
            np.testing.assert_array_equal(g, orc.cmvn_apply(m, cmvn.fake_stats_for_some_dims(s1, [0, 5, 12]), nv))
            if not nv:
                np.testing.assert_array_equal(g[:, [0, 5, 12]], m[:, [0, 5, 12]])
            assert np.abs(g[:, 1] - m[:, 1]).max() > 0.1
    with pytest.raises(Exception):
        cmvn.apply(mats[1:2], [orc.cmvn_acc_stats(mats[1])], skip_dims=[13])
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
    # compute-cmvn-stats --weights through the command line (text archive of per-frame weights; u3 has none -> skipped)
    with open(tmp_path / "w.ark", "w") as f:
        for i in range(3):
            f.write("u%d  [ %s ]\n" % (i, " ".join("%.9g" % x for x in wts[i])))
    r = subprocess.run([sys.executable, root + "/tools/compute_cmvn_stats.py", "--weights=ark:%s" % (tmp_path / "w.ark"), "ark:%s" % (tmp_path / "f.ark"),
                        "ark:%s" % (tmp_path / "cmvn_w.ark")], capture_output=True, text=True)
    assert r.returncode == 0 and "No weights available for utterance u3" in r.stderr and "3 utterances; 1 had errors" in r.stderr, r.stderr[-1500:]
    got_w = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "cmvn_w.ark"), "dmatrix"))      # statistics are double matrices
    assert sorted(got_w) == ["u0", "u1", "u2"]
    for i in range(3):
        np.testing.assert_allclose(got_w["u%d" % i], orc.cmvn_acc_stats(mats[i], weights=wts[i]), rtol=1e-12, atol=1e-12)
    r = subprocess.run([sys.executable, root + "/tools/apply_cmvn.py", "--norm-vars=true", "--reverse=true", "--skip-dims=1:2",
                        "--utt2spk=ark:%s" % (tmp_path / "utt2spk"), "ark:%s" % (tmp_path / "cmvn.ark"), "ark:%s" % (tmp_path / "out.ark"),
                        "ark:%s" % (tmp_path / "back.ark")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    back = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "back.ark"), "matrix"))
    for i, m in enumerate(mats):
        keep = [d for d in range(13) if d not in (1, 2)]
        np.testing.assert_allclose(back["u%d" % i][:, keep], m[:, keep], atol=2e-5 * np.abs(m).max())     # reverse undoes forward
        np.testing.assert_array_equal(back["u%d" % i][:, [1, 2]], out["u%d" % i][:, [1, 2]])               # skipped: left alone
    spkB = np.concatenate(mats[2:])
    np.testing.assert_allclose(np.concatenate([out["u2"], out["u3"]]).mean(0), 0, atol=2e-5)
    np.testing.assert_allclose(np.concatenate([out["u2"], out["u3"]]).std(0), 1, atol=1e-4)
    assert abs(float(spkB.mean())) > 1                                   # it was not normalised before


def test_featbin_and_nnet3_compute_tools(tmp_path):
    """compute-mfcc-feats / compute-fbank-feats with --config files over a wav.scp (file and pipe entries, a stereo file
    with --channel) and nnet3-compute over the resulting archive: the same matrices as the in-process calls."""
    import subprocess
    import sys
    import wave
    from kaldi_amd import decoder, nnet, table
    from tests.mdl_writer import write_mdl
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    waves = {"u1": np.round(synth.make_wave(1.3, seed=1)).astype(np.float32), "u2": np.round(synth.make_wave(0.7, seed=2)).astype(np.float32)}
    with open(tmp_path / "wav.scp", "w") as scp:
        for i, (k, w) in enumerate(waves.items()):
            with wave.open(str(tmp_path / (k + ".wav")), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(w.astype("<i2").tobytes())
            scp.write("%s %s\n" % (k, ("cat %s |" if i else "%s") % (tmp_path / (k + ".wav"))))
    (tmp_path / "mfcc.conf").write_text("--use-energy=false   # hires\n--num-mel-bins=40\n--num-ceps=40\n--low-freq=20\n--high-freq=-400\n")
    (tmp_path / "fbank.conf").write_text("--num-mel-bins=40\n--use-energy=true\n")
    for tool, conf, F in (("compute_mfcc_feats.py", "mfcc.conf", feat.Mfcc(abi.mfcc_opts_hires())), ("compute_fbank_feats.py", "fbank.conf", None)):
        out = tmp_path / (tool + ".ark")
        r = subprocess.run([sys.executable, root + "/tools/" + tool, "--config=%s" % (tmp_path / conf), "scp:%s" % (tmp_path / "wav.scp"), "ark:%s" % out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        assert "Done 2 out of 2 utterances." in r.stderr
        got = dict(table.SequentialTableReader("ark:%s" % out, "matrix"))
        if F is None:
            fo = abi.fbank_opts_default(); fo.mel.num_bins = 40; fo.use_energy = 1
            F = feat.Fbank(fo)
            assert got["u1"].shape[1] == 41
        for k, w in waves.items():
            np.testing.assert_array_equal(got[k], F.ComputeFeatures(w))
    # --subtract-mean and a stereo file
    st = np.stack([waves["u1"][:8000], waves["u2"][:8000]], 1).astype("<i2")
    with wave.open(str(tmp_path / "st.wav"), "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(16000); f.writeframes(st.tobytes())
    (tmp_path / "st.scp").write_text("s %s\n" % (tmp_path / "st.wav"))
    r = subprocess.run([sys.executable, root + "/tools/compute_mfcc_feats.py", "--config=%s" % (tmp_path / "mfcc.conf"), "--channel=1", "--subtract-mean=true",
                        "scp:%s" % (tmp_path / "st.scp"), "ark:%s" % (tmp_path / "st.ark")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    s = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "st.ark"), "matrix"))["s"]
    ref = feat.Mfcc(abi.mfcc_opts_hires()).ComputeFeatures(waves["u2"][:8000])
    np.testing.assert_allclose(s, ref - ref.mean(0), atol=2e-5)
    # nnet3-compute on the MFCC archive
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, 50, input_dim=40, seed=12, output_scale=3.0)
    write_mdl(tmp_path / "final.mdl", m, num_units=25)
    for flag, keep_priors in (("--use-priors=false", False), ("--use-priors=true", True)):
        r = subprocess.run([sys.executable, root + "/tools/nnet3_compute.py", flag, "--frame-subsampling-factor=3", str(tmp_path / "final.mdl"),
                            "ark:%s" % (tmp_path / "compute_mfcc_feats.py.ark"), "ark:%s" % (tmp_path / "out.ark")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        got = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "out.ark"), "matrix"))
        m2 = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, 50, input_dim=40, seed=12, output_scale=3.0)
        m2.layers[-1].post_scale = 1.0
        if not keep_priors:
            m2.layers[-1].post_offset = None
        N = decoder.Nnet(m2)
        mf = feat.Mfcc(abi.mfcc_opts_hires())
        for k, w in waves.items():
            want = N.Forward(mf.ComputeFeatures(w))
            np.testing.assert_allclose(got[k], want, rtol=0, atol=1e-4 * max(1.0, np.abs(want).max()))


def test_add_deltas_matches_the_oracle_and_the_closed_form(tmp_path):
    from kaldi_amd import cmvn
    rng = np.random.default_rng(2)
    mats = [rng.standard_normal((T, 13)).astype(np.float32) for T in (1, 3, 50, 400)]
    for order, window in ((2, 2), (1, 3), (0, 2), (3, 1)):
        got = cmvn.add_deltas(mats, order, window)
        for m, g in zip(mats, got):
            np.testing.assert_allclose(g, orc.add_deltas(m, order, window), rtol=0, atol=1e-6 * max(1.0, np.abs(g).max()))
    # first-order delta, window 2: sum_j j x[t+j] / 10 with the ends clamped
    x = mats[2].astype(np.float64)
    pad = np.concatenate([x[:1], x[:1], x, x[-1:], x[-1:]])
    d1 = sum(j * pad[2 + j:2 + j + x.shape[0]] for j in (-2, -1, 1, 2)) / 10.0
    np.testing.assert_allclose(cmvn.add_deltas([mats[2]], 2, 2)[0][:, 13:26], d1, atol=1e-5)


def test_splice_and_transform_feats(tmp_path):
    """splice-feats | transform-feats: clamped context, linear and affine transforms, one transform per speaker"""
    import subprocess
    import sys
    from kaldi_amd import cmvn, ivector, table
    rng = np.random.default_rng(4)
    mats = [rng.standard_normal((T, 5)).astype(np.float32) for T in (1, 2, 40)]
    sp = cmvn.splice_transform(mats, 2, 1)
    for m, s in zip(mats, sp):
        T = m.shape[0]
        want = np.concatenate([m[np.clip(np.arange(T) + o, 0, T - 1)] for o in (-2, -1, 0, 1)], 1)
        np.testing.assert_array_equal(s, want)
    lin, aff = rng.standard_normal((7, 20)).astype(np.float32), rng.standard_normal((7, 21)).astype(np.float32)
    for M in (lin, aff):
        got = cmvn.splice_transform(mats, 2, 1, transforms=M)
        for s, g in zip(sp, got):
            want = s.astype(np.float64) @ M[:, :20].T.astype(np.float64) + (M[:, 20] if M.shape[1] == 21 else 0)
            np.testing.assert_allclose(g, want, rtol=0, atol=1e-5 * max(1.0, np.abs(want).max()))
    two = cmvn.splice_transform(mats, 2, 1, transforms=[lin, aff[:, :20]], utt_transform=[0, 1, 0])
    np.testing.assert_array_equal(two[1], cmvn.splice_transform([mats[1]], 2, 1, transforms=aff[:, :20])[0])
    with pytest.raises(Exception):
        cmvn.splice_transform(mats, 2, 1, transforms=lin[:, :19])
    # the tools, chained: splice-feats | transform-feats with final.mat
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with table.TableWriter("ark:%s" % (tmp_path / "f.ark"), "matrix") as w:
        for i, m in enumerate(mats):
            w.write("u%d" % i, m)
    ivector.write_kaldi_matrix(tmp_path / "final.mat", aff)
    rspec = "ark:%s %s/tools/splice_feats.py --left-context=2 --right-context=1 ark:%s ark:- |" % (sys.executable, root, tmp_path / "f.ark")
    r = subprocess.run([sys.executable, root + "/tools/transform_feats.py", str(tmp_path / "final.mat"), rspec, "ark:%s" % (tmp_path / "o.ark")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    out = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "o.ark"), "matrix"))
    for i, m in enumerate(mats):
        np.testing.assert_array_equal(out["u%d" % i], cmvn.splice_transform([m], 2, 1, transforms=aff)[0])
