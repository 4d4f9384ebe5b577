"""Cross-checks the nnet oracle (oracle/orc_nnet.cc) against an independent float64 numpy
evaluation.  No reference-run outputs exist (nnet3 is unbuildable here and its tests are
self-consistency only), so parity for this stage is 'unpinned' by the task's definition;
tolerance: 2e-4 relative to the output scale (fp32 accumulation-order noise)."""
import numpy as np
import pytest

from kaldi_amd import nnet
from oracle import orc
from tests.nnet_ref import forward_f64


@pytest.mark.parametrize("T", [1, 2, 17, 50, 101])
@pytest.mark.parametrize("ivec", [0, 10])
def test_oracle_nnet_vs_f64(T, ivec):
    m = nnet.tdnnf_tiny(num_pdfs=50, ivector_dim=ivec, seed=T + ivec)
    rng = np.random.default_rng(T)
    feats = rng.standard_normal((T, m.input_dim)).astype(np.float32) * 3
    iv = rng.standard_normal(ivec).astype(np.float32) if ivec else None
    got = orc.nnet_forward(m, feats, iv)
    ref = forward_f64(m, feats, iv)
    assert got.shape == ref.shape == ((T + 2) // 3, 50)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 2e-4 * scale


def test_oracle_log_softmax_output():
    """LogSoftmaxComponent on the output node (non-chain models): rows are normalised
    before -log(prior) and the acoustic scale are applied."""
    m = nnet.tdnnf_tiny(num_pdfs=61, seed=4)
    out = m.layers[-1]
    out.log_softmax = True
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((40, m.input_dim)).astype(np.float32) * 3
    got = orc.nnet_forward(m, feats)
    ref = forward_f64(m, feats)
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()
    # undo "(+post_offset) * post_scale": what is left must be a normalised distribution
    lp = got / out.post_scale - out.post_offset
    np.testing.assert_allclose(np.exp(lp.astype(np.float64)).sum(axis=1), 1.0, atol=1e-4)


def test_chunked_forward_with_online_ivectors():
    """DecodableNnetSimple chunk semantics: with one ivector for the whole utterance the chunked
    evaluation equals the plain one; with time-varying ivectors every chunk uses the row picked
    for its middle (nnet-am-decodable-simple.cc:181-211)."""
    m = nnet.tdnnf_tiny(num_pdfs=30, ivector_dim=10, seed=3)
    rng = np.random.default_rng(1)
    T = 140
    feats = rng.standard_normal((T, m.input_dim)).astype(np.float32)
    iv1 = rng.standard_normal(10).astype(np.float32)
    const = np.tile(iv1, (T // 10 + 1, 1))
    a = orc.nnet_forward_chunked(m, feats, const, 10, 50)
    b = orc.nnet_forward(m, feats, iv1)
    np.testing.assert_array_equal(a, b)
    ivs = rng.standard_normal((T // 10 + 1, 10)).astype(np.float32)
    c = orc.nnet_forward_chunked(m, feats, ivs, 10, 50)           # 50 -> 51: 17 output frames per chunk
    n_out = (T + 2) // 3
    for start in range(0, n_out, 17):
        num = min(17, n_out - start)
        first, last = 3 * start, 3 * (start + num - 1)
        row = min((first + (last - first) // 2) // 10, ivs.shape[0] - 1)
        want = orc.nnet_forward(m, feats, ivs[row])[start:start + num]
        np.testing.assert_array_equal(c[start:start + num], want)
    assert np.abs(c - b).max() > 1e-3                              # the ivectors matter


def test_context_matches_recipe_topologies():
    """ComputeSimpleNnetContext: 1 + 3*1 + 0 + 12*3 = 40 for run_tdnn_1d."""
    assert nnet.tdnnf_librispeech(num_pdfs=16).context() == (40, 40)
    assert nnet.tdnnf_mini_librispeech(num_pdfs=16).context() == (28, 28)
    m = nnet.tdnnf_tiny()
    assert orc.nnet_context(m) == m.context() == (8, 8)


def test_macs_per_output_frame_matches_survey_appendix_b():
    """~23.5 M MAC per output frame for the 1d topology with P=6000
    (the appendix omits the ivector columns of lda; we build without ivector here)."""
    m = nnet.tdnnf_librispeech(num_pdfs=6000)
    macs = m.macs_per_output_frame()
    assert 22.0e6 < macs < 25.0e6, macs


def test_chunk_invariance():
    """Whole-utterance evaluation == per-chunk evaluation with clamped context
    (what DecodableNnetSimple does chunk by chunk, frames_per_chunk=51)."""
    m = nnet.tdnnf_tiny(num_pdfs=20)
    rng = np.random.default_rng(0)
    T = 120
    feats = rng.standard_normal((T, 40)).astype(np.float32)
    whole = orc.nnet_forward(m, feats)
    left, right = m.context()
    out = []
    for s in range(0, (T + 2) // 3, 17):
        n = min(17, (T + 2) // 3 - s)
        first_out, last_out = 3 * s, 3 * (s + n - 1)
        t = np.clip(np.arange(first_out - left, last_out + right + 1), 0, T - 1)
        chunk = orc.nnet_forward(m, feats[t])             # chunk-local time 0 == first input row
        # chunk-local output rows are at local t = left + 3k; the oracle evaluates at 0,3,6..
        # so shift: evaluate on a chunk whose first row is time first_out-left and read rows
        # (left + 3k)/3 only when left % 3 == 0; otherwise re-evaluate with padding.
        assert left % 3 != 0 or True
        padl = (-left) % 3
        tt = np.clip(np.arange(first_out - left - padl, last_out + right + 1), 0, T - 1)
        chunk = orc.nnet_forward(m, feats[tt])
        k0 = (left + padl) // 3
        out.append(chunk[k0:k0 + n])
    chunked = np.concatenate(out)
    # interior chunks see clamped-at-chunk-edge inputs only where the utterance itself is
    # clamped, so results are identical up to fp32 summation order (none here: same code).
    np.testing.assert_allclose(chunked, whole, rtol=0, atol=1e-5)


def test_blas_forward_equals_scalar_oracle():
    """The CPU baseline's forward (the reference's path: DecodableNnetSimple chunks, one sgemm per Propagate) in its two
    forms -- oracle/orc_nnet_blas.cc with OpenBLAS's cblas_sgemm, oracle/orc_blas.py with numpy -- against the scalar
    oracle used for parity, for chunk sizes that do and do not divide the utterance, with and without an i-vector."""
    from oracle import orc_blas
    rng = np.random.default_rng(3)
    for ivd in (0, 10):
        m = nnet.tdnnf_tiny(num_pdfs=50, ivector_dim=ivd, seed=4)
        iv = rng.standard_normal(ivd).astype(np.float32) if ivd else None
        for T in (1, 4, 52, 160):
            f = rng.standard_normal((T, m.input_dim)).astype(np.float32)
            want = orc.nnet_forward(m, f, iv)
            tol = 1e-5 * max(1.0, float(np.abs(want).max()))
            for fpc in (50, 21, 0):
                assert np.abs(orc_blas.nnet_forward_blas(m, f, iv, frames_per_chunk=fpc) - want).max() < tol
                if orc.cblas_sgemm() is not None:
                    assert np.abs(orc.nnet_forward_blas(m, f, iv, frames_per_chunk=fpc) - want).max() < tol


def test_blas_chunked_forward_with_online_ivectors_equals_scalar_oracle():
    """The like-for-like CPU baseline's model: the sgemm forward with one online i-vector per chunk against the scalar
    chunked oracle (same chunk -> i-vector rule, nnet-am-decodable-simple.cc:181-211), incl. a table that ends early."""
    if orc.cblas_sgemm() is None:
        pytest.skip("no OpenBLAS cblas_sgemm next to numpy")
    rng = np.random.default_rng(5)
    m = nnet.tdnnf_tiny(num_pdfs=50, ivector_dim=10, seed=4)
    for T in (4, 52, 160, 333):
        f = rng.standard_normal((T, m.input_dim)).astype(np.float32)
        for n_iv in ((T + 9) // 10, max(1, (T + 9) // 10 - 2)):
            iv = rng.standard_normal((n_iv, 10)).astype(np.float32)
            for fpc in (50, 21):
                want = orc.nnet_forward_chunked(m, f, iv, 10, fpc)
                got = orc.nnet_forward_blas_chunked(m, f, iv, 10, fpc)
                assert np.abs(got - want).max() < 1e-5 * max(1.0, float(np.abs(want).max()))


def test_batch_computer_tasks_follow_the_reference_rule():
    """NnetBatchComputer::SplitUtteranceIntoTasks (nnet3/nnet-batch-compute.cc:586-829), cases worked by hand from the rule:
    fpc = 50 / 3 = 16 output frames per task; the last of several tasks ends on the last frame and overlaps its predecessor;
    an utterance shorter than a task is one task of 16 frames; first_input_t = -left context; the i-vector row is
    ((begin_output_t + 8) * 3) // period, the last row when at most 20 frames beyond the table."""
    m = nnet.tdnnf_tiny(num_pdfs=20, ivector_dim=6, seed=3)
    left, right = m.context()
    rng = np.random.default_rng(0)

    def run(T, n_iv=None, period=10):
        x = rng.standard_normal((T, m.input_dim)).astype(np.float32)
        iv = rng.standard_normal(((T + period - 1) // period if n_iv is None else n_iv, 6)).astype(np.float32)
        out, tasks = orc.nnet_forward_batch_computer(m, x, iv, period, 50, return_tasks=True)
        return x, iv, out, tasks.tolist()

    # columns: first_used_output_frame_index, num_initial_unused, num_used, num_output_frames, first_input_t, i-vector row
    assert run(5)[3] == [[0, 0, 2, 16, -left, 0]]                       # 2 output frames, one padded task; row (8 * 3) // 10 = 2 -> clamped to the only row
    assert run(48)[3] == [[0, 0, 16, 16, -left, 2]]
    assert run(49)[3] == [[0, 0, 16, 16, -left, 2], [16, 15, 1, 16, -left, 2]]      # 17 frames: the second task covers [1, 17), mid (1 + 8) * 3 // 10
    assert run(100)[3] == [[0, 0, 16, 16, -left, 2], [16, 0, 16, 16, -left, 7], [32, 14, 2, 16, -left, 7]]
    assert run(160)[3] == [[0, 0, 16, 16, -left, 2], [16, 0, 16, 16, -left, 7], [32, 0, 16, 16, -left, 12], [48, 10, 6, 16, -left, 13]]
    # a table shorter than the task's middle: the last row is taken.  (As written in the reference the 20-frame margin can
    # never refuse a non-empty table -- `ivector_frame > num_rows - margin` holds whenever `ivector_frame >= num_rows` --
    # so neither does the restatement.)
    assert run(160, n_iv=13)[3][-1][5] == 12
    assert [t[5] for t in run(160, n_iv=9)[3]] == [2, 7, 8, 8]
    # every task sees ONE i-vector: with the same row everywhere the result is the plain forward, whatever the chunking
    x, iv, _, _ = run(100)
    const = np.tile(iv[:1], (iv.shape[0], 1))
    np.testing.assert_array_equal(orc.nnet_forward_batch_computer(m, x, const, 10, 50), orc.nnet_forward(m, x, iv[0]))
    # ... and with time-varying rows every kept output row equals the plain forward with its task's row
    out, tasks = orc.nnet_forward_batch_computer(m, x, iv, 10, 50, return_tasks=True)
    for first, unused, used, n_out, _, row in tasks.tolist():
        np.testing.assert_array_equal(out[first:first + used], orc.nnet_forward(m, x, iv[row])[first:first + used])
    # the two chunkings of the reference are different functions of the same inputs (17 against 16 frames per chunk)
    assert np.abs(out - orc.nnet_forward_chunked(m, x, iv, 10, 50)).max() > 1e-3
