"""The i-vector oracle against closed forms (no extractor, UBM or golden i-vectors exist offline:
parity unpinned, see oracle/orc_ivector.cc)."""
import numpy as np
import pytest

from kaldi_amd import ivector
from oracle import orc


def small_info(**kw):
    return ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=16, ivector_dim=10, seed=3, splice_left=2, splice_right=1, **kw)


def test_online_cmvn_window_and_smoothing():
    """frame t is normalised with the sum over the last cmn_window frames up to t, topped up with
    at most global_frames frames' worth of the global mean (feat/online-feature.cc:325-407)."""
    info = small_info(cmn_window=20, speaker_frames=20, global_frames=5)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((70, 8)) * 3 + 1).astype(np.float32)
    got = orc.online_cmvn(info, x)
    gmean = info.global_cmvn_stats[0, :8] / info.global_cmvn_stats[0, 8]
    for t in (0, 1, 4, 14, 15, 19, 20, 33, 69):
        lo = max(0, t - 20 + 1)
        n = t - lo + 1
        s = x[lo:t + 1].astype(np.float64).sum(0)
        extra = min(20 - n, 5) if n < 20 else 0
        mean = (s + extra * gmean) / (n + extra)
        np.testing.assert_allclose(got[t], x[t] - mean, rtol=0, atol=2e-6)


def test_online_cmvn_variance_normalisation():
    """--norm-vars=true (feat/online-feature.cc:339-350 accumulates the squares in the sliding window, :437 ->
    transform/cmvn.cc:92-114): x * scale + offset with scale = 1 / sqrt(E[x^2] - mean^2) and offset = -mean * scale over the
    window, both rows of the statistics topped up with the global ones."""
    info = small_info(cmn_window=20, speaker_frames=20, global_frames=5, normalize_variance=True)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((70, 8)) * 3 + 1).astype(np.float32)
    got = orc.online_cmvn(info, x)
    g = info.global_cmvn_stats
    gmean, gsq = g[0, :8] / g[0, 8], g[1, :8] / g[0, 8]
    for t in (0, 1, 4, 14, 15, 19, 20, 33, 69):
        lo = max(0, t - 20 + 1)
        n = t - lo + 1
        w = x[lo:t + 1].astype(np.float64)
        extra = min(20 - n, 5) if n < 20 else 0
        mean = (w.sum(0) + extra * gmean) / (n + extra)
        var = ((w * w).sum(0) + extra * gsq) / (n + extra) - mean * mean
        want = x[t] / np.sqrt(var) - mean / np.sqrt(var)
        np.testing.assert_allclose(got[t], want, rtol=0, atol=5e-6 * max(1.0, np.abs(want).max()))
    assert np.abs(got - orc.online_cmvn(small_info(cmn_window=20, speaker_frames=20, global_frames=5), x)).max() > 0.1


def test_posterior_entry_pruning_and_normalisation():
    ll = np.log(np.array([0.5, 0.3, 0.1, 0.06, 0.03, 0.006, 0.004], np.float32))
    tot_ll, g, p = orc.posterior_entry(ll, 5, 0.025)
    # 0.006 and 0.004 are below 0.025 of the maximum's scale (0.5 * 0.025 = 0.0125); top 5 kept; 0.03 / 0.99 > 0.025 stays
    assert g.tolist() == [0, 1, 2, 3, 4]
    np.testing.assert_allclose(p, np.array([0.5, 0.3, 0.1, 0.06, 0.03]) / 0.99, rtol=1e-5)
    assert abs(tot_ll - np.log(0.99)) < 1e-5
    # num_gselect cuts first, then entries below min_post of the kept mass are dropped from the back
    _, g, p = orc.posterior_entry(ll, 3, 0.2)
    assert g.tolist() == [0, 1] and abs(p.sum() - 1) < 1e-6
    _, g, p = orc.posterior_entry(np.array([-3.0, -1000.0, -3.0], np.float32), 5, 0.025)
    assert g.tolist() == [0, 2] and np.allclose(p, 0.5)                 # ties: smaller index first
    _, g, p = orc.posterior_entry(np.array([-1.0, -2.0], np.float32), 5, 0.0)
    assert g.tolist() == [0, 1]


def test_linear_cgd_against_direct_solve():
    rng = np.random.default_rng(1)
    n = 12
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    b = rng.standard_normal(n)
    packed = A[np.tril_indices(n)]
    x, k = orc.linear_cgd(packed, b, np.zeros(n), -1)
    np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    assert 0 < k <= n + 5
    x3, k3 = orc.linear_cgd(packed, b, np.zeros(n), 3)                  # max_iters binds
    assert k3 == 3 and np.linalg.norm(A @ x3 - b) < np.linalg.norm(b)
    # CG from the solution does not move
    x0 = np.linalg.solve(A, b)
    x4, _ = orc.linear_cgd(packed, b, x0, 15)
    np.testing.assert_allclose(x4, x0, rtol=1e-9)


def test_ivectors_solve_the_accumulated_system():
    """Row i = (prior + sum of posterior-weighted U_g)^-1 (prior_offset e_0 + sum of Sigma_inv_M_g^T x)
    over frames 0 .. i * period, up to the 15 warm-started CG steps; dimension 0 has the prior offset removed."""
    info = small_info(ivector_period=4, num_cg_iters=50)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((37, 8)).astype(np.float32)
    iv, dg = orc.ivector_extract_online(info, x, diagnostics=True)
    assert iv.shape == (10, 10) and dg["cg_got_worse"] == 0
    D, I = 6, 10
    tri = np.tril_indices(D)
    quad = np.eye(I)
    lin = np.zeros(I)
    lin[0] = info.prior_offset
    want = []
    for t in range(37):
        for g, w in zip(dg["post_gauss"][t], dg["post_weight"][t]):
            if g < 0:
                continue
            S = np.zeros((D, D))
            S[tri] = info.sigma_inv[g]
            S = S + S.T - np.diag(np.diag(S))
            M = info.M[g]
            quad += float(w) * (M.T @ S @ M)
            lin += float(w) * (M.T @ S @ dg["raw_lda"][t].astype(np.float64))
        if t % 4 == 0:
            sol = np.linalg.solve(quad, lin)
            sol[0] -= info.prior_offset
            want.append(sol)
    np.testing.assert_allclose(iv, np.array(want), rtol=2e-4, atol=2e-5)
    # posteriors: at most num_gselect per frame, scaled by posterior_scale, descending
    w = dg["post_weight"]
    assert np.allclose(w.sum(1), info.posterior_scale, atol=1e-6)
    assert (np.diff(w, axis=1) <= 1e-9).all()
    # the first i-vector comes from one frame only: close to the prior (zero after the offset is removed)
    assert np.abs(iv[0]).max() < np.abs(iv[-1]).max() + 1.0


def test_max_count_scales_the_prior():
    a = orc.ivector_extract_online(small_info(ivector_period=5), np.random.default_rng(5).standard_normal((200, 8)).astype(np.float32))
    b = orc.ivector_extract_online(small_info(ivector_period=5, max_count=2.0), np.random.default_rng(5).standard_normal((200, 8)).astype(np.float32))
    # counts are 0.1 per frame: below max_count = 2.0 (20 frames) nothing changes; later the prior is scaled up
    np.testing.assert_allclose(a[:4], b[:4], rtol=1e-6, atol=1e-7)
    assert np.linalg.norm(b[-1]) < np.linalg.norm(a[-1])


def test_files_round_trip(tmp_path):
    info = small_info(ivector_period=7, max_count=100.0)
    conf = ivector.write_config_dir(tmp_path / "ivector_extractor", info)
    back = ivector.IvectorExtractionInfo.from_config(conf)
    for k in ("lda", "global_cmvn_stats", "ubm_weights", "ubm_means_invvars", "ubm_inv_vars", "ubm_gconsts", "M", "sigma_inv"):
        np.testing.assert_array_equal(getattr(back, k), getattr(info, k))
    assert (back.prior_offset, back.ivector_period, back.max_count, back.splice_left, back.splice_right) == (2.0, 7, 100.0, 2, 1)
    x = np.random.default_rng(0).standard_normal((30, 8)).astype(np.float32)
    np.testing.assert_array_equal(orc.ivector_extract_online(back, x), orc.ivector_extract_online(info, x))
