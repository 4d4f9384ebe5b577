"""Online i-vector extraction on the device against the CPU oracle (oracle/orc_ivector.cc).
Tolerances: UBM posteriors 2e-6 absolute (same float operation order, device expf / 1.0f/x differ in
the last place); i-vectors 1e-4 relative to the largest component (the statistics are accumulated in
fp64 on both sides, in a different order, and 15 warm-started CG steps amplify that slightly)."""
import numpy as np
import pytest

from kaldi_amd import abi, decoder, feat, ivector, nnet, pipeline, synth
from kaldi_amd._lib import KamdError
from oracle import orc
from tests.util import lattices_equal

pytestmark = pytest.mark.gpu


def check(info, ie, x):
    got = ie.extract_online(x)
    want, dg = orc.ivector_extract_online(info, x, diagnostics=True)
    assert dg["cg_got_worse"] == 0
    g, w = ie.last_posteriors(x.shape[0])
    np.testing.assert_array_equal(g, dg["post_gauss"])
    np.testing.assert_allclose(w, dg["post_weight"], rtol=0, atol=2e-6)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * max(1.0, np.abs(want).max()))
    return got


@pytest.mark.parametrize("opts", [dict(), dict(ivector_period=4, max_count=1.5), dict(cmn_window=25, speaker_frames=25, global_frames=10),
                                  dict(num_gselect=2, min_post=0.2), dict(normalize_mean=False, num_cg_iters=3),
                                  dict(normalize_variance=True, cmn_window=25, speaker_frames=25, global_frames=10)])
def test_small_extractor_matches_oracle(opts):
    info = ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=70, ivector_dim=10, seed=3, splice_left=2, splice_right=1, **opts)
    ie = ivector.IvectorExtractor(info)
    rng = np.random.default_rng(1)
    for T in (1, 2, 9, 10, 11, 53, 160):
        check(info, ie, (rng.standard_normal((T, 8)) * 1.5 + 0.3).astype(np.float32))


@pytest.mark.parametrize("shape", [dict(feat_dim=13, lda_dim=20, num_gauss=130, ivector_dim=128, num_gselect=8, ivector_period=3),
                                   dict(feat_dim=80, lda_dim=24, num_gauss=64, ivector_dim=65, splice_left=0, splice_right=0),
                                   dict(feat_dim=5, lda_dim=5, num_gauss=3, ivector_dim=1, num_gselect=5, min_post=0.0)])
def test_extractor_shapes_at_the_limits(shape):
    """the largest i-vector dimension (two rows per lane), more Gaussians than lanes but not a multiple of 64,
    num_gselect 8 and more than there are Gaussians, no splicing, a 1-dimensional i-vector, min_post 0"""
    info = ivector.make_synthetic(seed=8, **shape)
    ie = ivector.IvectorExtractor(info)
    rng = np.random.default_rng(3)
    for T in (1, 7, 64):
        check(info, ie, (rng.standard_normal((T, shape["feat_dim"])) * 1.3).astype(np.float32))
    with pytest.raises(Exception):
        ivector.IvectorExtractor(ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=8, ivector_dim=129))
    with pytest.raises(Exception):
        ivector.IvectorExtractor(ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=8, ivector_dim=10, normalize_variance=True,
                                                        normalize_mean=False))      # "You cannot normalize the variance but not the mean."


def test_recipe_size_extractor_on_mfcc_features():
    """hires MFCC 40 -> splice +-3 -> LDA 40 -> 512 Gaussians -> 100-dim i-vectors, period 10."""
    op = abi.mfcc_opts_hires()
    feats = [feat.Mfcc(op).ComputeFeatures(synth.make_wave(d, seed=40 + i)) for i, d in enumerate((2.3, 0.4, 7.1))]
    allf = np.concatenate(feats)
    info = ivector.make_synthetic(seed=5, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=100.0)
    ie = ivector.IvectorExtractor(info)
    solo = [check(info, ie, f) for f in feats]
    assert solo[2].shape == (71, 100)
    # the i-vector moves away from the prior as frames come in, and it is not constant
    assert np.linalg.norm(solo[2][-1]) > np.linalg.norm(solo[2][0])
    assert np.abs(np.diff(solo[2], axis=0)).max() > 1e-3


def test_pipeline_with_the_extractor_equals_precomputed_online_ivectors():
    """features -> i-vectors -> chunked nnet -> search in one run(), against --online-ivectors with the
    matrices the extractor returns for the same features, and against the oracle's matrices."""
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=20, seed=12, output_scale=3.0)
    cfg = abi.decoder_config_recipe()
    waves = [synth.make_wave(d, seed=70 + i) for i, d in enumerate((1.7, 3.2, 0.9))]
    op = abi.mfcc_opts_hires()
    feats = [feat.Mfcc(op).ComputeFeatures(w) for w in waves]
    allf = np.concatenate(feats)
    info = ivector.make_synthetic(num_gauss=64, ivector_dim=20, seed=6, feat_mean=allf.mean(0), feat_std=allf.std(0))
    ie = ivector.IvectorExtractor(info)
    pipe = pipeline.Pipeline(op, m, g, cfg, max_utts=3, max_seconds=4.0)
    pipe.load(waves)
    pipe.set_ivector_extractor(ie, frames_per_chunk=50)
    pipe.run()
    got = pipe.results()
    got_ll = [pipe.loglikes(u).copy() for u in range(3)]
    mats = [ie.extract_online(f) for f in feats]
    pipe.set_ivector_extractor(None)
    pipe.load(waves)
    pipe.set_online_ivectors(mats, info.ivector_period, 50)
    pipe.run()
    ref = pipe.results()
    for u in range(3):
        np.testing.assert_array_equal(pipe.loglikes(u), got_ll[u])        # same kernels, same inputs
        assert got[u]["words"].tolist() == ref[u]["words"].tolist()
        want = orc.nnet_forward_chunked(m, feats[u], orc.ivector_extract_online(info, feats[u]), info.ivector_period, 50)
        np.testing.assert_allclose(got_ll[u], want, rtol=0, atol=2e-3)


def test_batch_decoder_with_the_extractor_equals_the_pipeline_and_the_oracle():
    """kamd_batch_decoder_set_ivector_extractor (the work-queue path of the bench with the recipe's online i-vectors): per
    pass features -> i-vectors -> chunked forward, through load() and load_host(), several passes, an utterance too short
    for a frame in the middle.  Log-likelihoods: the oracle's DecodableNnetSimple with the oracle's own i-vector
    matrices; lattices: the three-launch pipeline's."""
    from kaldi_amd import batch
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=20, seed=12, output_scale=3.0)
    cfg = abi.decoder_config_recipe()
    durs = (1.7, 3.2, 0.9, 2.4, 0.6, 1.1, 2.9)
    waves = [synth.make_wave(d, seed=70 + i) for i, d in enumerate(durs)]
    op = abi.mfcc_opts_hires()
    feats = [feat.Mfcc(op).ComputeFeatures(w) for w in waves]
    allf = np.concatenate(feats)
    info = ivector.make_synthetic(num_gauss=64, ivector_dim=20, seed=6, feat_mean=allf.mean(0), feat_std=allf.std(0))
    ie = ivector.IvectorExtractor(info)
    pipe = pipeline.Pipeline(op, m, g, cfg, max_utts=len(waves), max_seconds=4.0)
    pipe.load(waves)
    pipe.set_ivector_extractor(ie, frames_per_chunk=50)
    pipe.run()
    ref = pipe.results()
    ref_ll = [pipe.loglikes(u).copy() for u in range(len(waves))]
    with_short = waves[:3] + [np.zeros(120, np.float32)] + waves[3:]
    idx = [0, 1, 2, 4, 5, 6, 7]
    for loader in ("load", "load_host"):
        bd = batch.NnetBatchDecoder(op, m, g, cfg, max_seconds=4.0, resident_lanes=3, host_threads=2, determinize=True, keep_raw_lattices=True,
                                    search_mode=1, nnet_pass_frames=500, first_pass_frames=200)
        with pytest.raises(KamdError, match="ivector input"):
            bd.load(waves)
        bd.set_ivector_extractor(ie, 50)
        getattr(bd, loader)(with_short)
        st = bd.run()
        assert st.nnet_passes >= 3 and st.n_failed == 1 and st.ivector_ms > 0
        assert bd.output(3) is None
        for u, k in enumerate(idx):
            np.testing.assert_array_equal(bd.loglikes(k), ref_ll[u])          # same kernels on the same rows, pass by pass
            want = orc.nnet_forward_chunked(m, feats[u], orc.ivector_extract_online(info, feats[u]), info.ivector_period, 50)
            np.testing.assert_allclose(bd.loglikes(k), want, rtol=0, atol=2e-3)
            assert bd.output(k)["words"].tolist() == ref[u]["words"].tolist()
            assert bd.raw_lattice(k).arcs.tobytes() == ref[u]["lattice"].arcs.tobytes()


def test_recipe_sized_model_with_online_ivectors_through_the_batch_decoder():
    """BASELINE configs[2] as the recipe runs it (run_tdnn_1d.sh:220 `input dim=100 name=ivector`, steps/nnet3/decode.sh:
    105-107): the LibriSpeech TDNN-F topology (1536 / 160, 17 layers, P = 6000) with 100-dim online i-vectors from an
    extractor of the recipe's shape (512 Gaussians), through kamd_batch_decoder_*: log-likelihoods against the oracle's
    DecodableNnetSimple (chunks of 50 -> 51 frames, context recomputed, GetCurrentIvector's row) fed with the ORACLE's
    i-vector matrices.  Tolerance: 2e-3 of the largest |log-likelihood| (fp32 GEMMs with K up to 3072 in another
    summation order, on i-vectors that themselves agree to 1e-4)."""
    from kaldi_amd import batch
    g = synth.make_hclg(num_units=3000, vocab=300, n_hist=40, seed=2)
    m = nnet.tdnnf_librispeech(num_pdfs=g.num_pdfs, ivector_dim=100, output_scale=1.0)
    assert m.ivector_dim == 100 and len(m.layers) > 30
    cfg = abi.decoder_config_recipe()
    waves = [synth.make_wave(d, seed=170 + i) for i, d in enumerate((2.2, 1.1))]
    op = abi.mfcc_opts_hires()
    feats = [feat.Mfcc(op).ComputeFeatures(w) for w in waves]
    allf = np.concatenate(feats)
    info = ivector.make_synthetic(seed=11, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=100.0)
    ie = ivector.IvectorExtractor(info)
    bd = batch.NnetBatchDecoder(op, m, g, cfg, max_seconds=3.0, resident_lanes=2, host_threads=2, determinize=False, search_mode=1)
    bd.set_ivector_extractor(ie, 50)
    bd.load_host(waves)
    st = bd.run()
    assert st.ivector_ms > 0
    # 17 output frames a chunk; what the chunks cost beyond the model's algorithmic work is the recomputed context
    alg = 2.0 * m.macs_per_output_frame() * sum((f.shape[0] + 2) // 3 for f in feats)
    assert 1.2 < st.nnet_flops / alg < 3.0
    for u in range(2):
        iv = orc.ivector_extract_online(info, feats[u])
        want = orc.nnet_forward_chunked(m, feats[u], iv, info.ivector_period, 50)
        got = bd.loglikes(u)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 2e-3 * np.abs(want).max()
        # and the i-vectors matter: the same model on one constant i-vector gives other numbers
        assert np.abs(orc.nnet_forward_chunked(m, feats[u], np.repeat(iv[:1], iv.shape[0], 0), info.ivector_period, 50) - want).max() > \
            20 * np.abs(got - want).max()


def test_latgen_tool_with_device_ivectors_and_with_online_ivector_archives(tmp_path):
    """tools/nnet3_latgen_faster.py: --ivector-extraction-config (estimated on the device) gives the
    same word sequences as --online-ivectors with the matrices dumped to an archive, and as the
    in-memory pipeline; the extractor is read back from final.ie / final.dubm / final.mat / conf files."""
    import subprocess
    import sys
    import wave
    import os
    from kaldi_amd import io as kio
    from kaldi_amd import table
    from tests.mdl_writer import write_mdl
    g = synth.make_hclg(num_units=25, vocab=60, n_hist=12, seed=6)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=20, seed=12, output_scale=3.0)
    write_mdl(tmp_path / "final.mdl", m, num_units=25)
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    waves = [np.round(synth.make_wave(d, seed=80 + i)).astype(np.float32) for i, d in enumerate((1.4, 2.6))]
    op = abi.mfcc_opts_hires()
    feats = [feat.Mfcc(op).ComputeFeatures(w) for w in waves]
    allf = np.concatenate(feats)
    info = ivector.make_synthetic(num_gauss=64, ivector_dim=20, seed=6, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=100.0)
    conf = ivector.write_config_dir(tmp_path / "ivector_extractor", info)
    with open(tmp_path / "wav.scp", "w") as scp:
        for i, w in enumerate(waves):
            with wave.open(str(tmp_path / ("u%d.wav" % i)), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
                f.writeframes(w.astype("<i2").tobytes())
            scp.write("utt%d %s\n" % (i, tmp_path / ("u%d.wav" % i)))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = [sys.executable, root + "/tools/nnet3_latgen_faster.py", "--beam=15", "--max-active=7000", "--lattice-beam=8",
            "--acoustic-scale=1.0", "--frame-subsampling-factor=3", "--wav", str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst"),
            "scp:%s" % (tmp_path / "wav.scp")]
    r = subprocess.run(tool[:6] + ["--ivector-extraction-config=%s" % conf] + tool[6:] +
                       ["ark:%s" % (tmp_path / "lat1.ark"), "ark,t:%s" % (tmp_path / "w1.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    ie = ivector.IvectorExtractor(ivector.IvectorExtractionInfo.from_config(conf))
    with table.TableWriter("ark:%s" % (tmp_path / "ivector_online.ark"), "matrix") as w:
        for i, f in enumerate(feats):
            w.write("utt%d" % i, ie.extract_online(f))
    r = subprocess.run(tool[:6] + ["--online-ivectors=ark:%s" % (tmp_path / "ivector_online.ark"), "--online-ivector-period=10"] + tool[6:] +
                       ["ark:%s" % (tmp_path / "lat2.ark"), "ark,t:%s" % (tmp_path / "w2.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(tmp_path / "w1.txt").read() == open(tmp_path / "w2.txt").read()
    assert open(tmp_path / "lat1.ark", "rb").read() == open(tmp_path / "lat2.ark", "rb").read()
    # --ivectors + --utt2spk: one constant i-vector per speaker (RandomAccessBaseFloatVectorReaderMapped)
    spk_iv = np.linspace(-1, 1, 20).astype(np.float32)
    (tmp_path / "spk_iv.ark").write_text("spkA  [ " + " ".join("%.9g" % x for x in spk_iv) + " ]\n")
    (tmp_path / "utt2spk").write_text("utt0 spkA\nutt1 spkA\n")
    r = subprocess.run(tool[:6] + ["--ivectors=ark:%s" % (tmp_path / "spk_iv.ark"), "--utt2spk=ark:%s" % (tmp_path / "utt2spk")] + tool[6:] +
                       ["ark:%s" % (tmp_path / "lat3.ark"), "ark,t:%s" % (tmp_path / "w3.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    g.tid2pdf = np.concatenate([[-1], np.stack([2 * np.arange(25) + 1, 2 * np.arange(25)], 1).reshape(-1)]).astype(np.int32)
    pipe = pipeline.Pipeline(op, m, g, abi.decoder_config_recipe(), max_utts=2, max_seconds=3.0)
    pipe.load(waves)
    pipe.set_ivectors([spk_iv, spk_iv])
    pipe.run()
    ref3 = pipe.results()
    got3 = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(tmp_path / "w3.txt")}
    for i in range(2):
        assert got3["utt%d" % i] == ref3[i]["words"].tolist()
    pipe.set_ivectors(None)
    pipe.set_ivector_extractor(ie, 50)
    ref = pipe.decode(waves)
    got = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(tmp_path / "w1.txt")}
    for i in range(2):
        assert got["utt%d" % i] == ref[i]["words"].tolist()


@pytest.mark.parametrize("max_count", [0.0, 3.0])
def test_speaker_adaptation_state_carried_across_utterances(max_count, norm_vars=False):
    """ivector-extract-online2 with a spk2utt that groups utterances: the second and third utterance
    start from the CMVN speaker statistics and the i-vector statistics the earlier ones left
    (LimitFrames applied in between), on the device as in the oracle."""
    info = ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=40, ivector_dim=10, seed=4, splice_left=2, splice_right=1,
                                  cmn_window=60, speaker_frames=40, global_frames=10, max_count=max_count, ivector_period=5,
                                  normalize_variance=norm_vars)
    ie = ivector.IvectorExtractor(info)
    rng = np.random.default_rng(7)
    utts = [(rng.standard_normal((T, 8)) * 1.2 + 0.5).astype(np.float32) for T in (90, 31, 140)]
    st_d, st_o = None, None
    for i, x in enumerate(utts):
        got, st_d = ie.extract_online(x, state=st_d, return_state=True, max_remembered_frames=100.0)
        want, st_o = orc.ivector_extract_online(info, x, state=st_o, return_state=True, max_remembered_frames=100.0)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * max(1.0, np.abs(want).max()))
        np.testing.assert_allclose(st_d, st_o, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(st_o).max()))
        if i > 0:   # the carried state matters: a fresh start gives something else
            fresh = ie.extract_online(x)
            assert np.abs(fresh[0] - got[0]).max() > 1e-3
    # LimitFrames: the speaker's CMVN count is scaled back to max_remembered_frames (float arithmetic, as there)
    assert abs(st_d[8] - 100.0) < 1e-3


def test_speaker_adaptation_state_with_variance_normalisation():
    """OnlineCmvn --norm-vars=true: the speaker statistics' second row (sums of squares) smooths the window as well."""
    test_speaker_adaptation_state_carried_across_utterances(0.0, norm_vars=True)


def test_ivector_extract_online2_tool(tmp_path):
    """spk2utt with two speakers (two and one utterances), features from an scp: the archive written equals
    the oracle run utterance by utterance with the speaker's state carried over."""
    import subprocess
    import sys
    import os
    from kaldi_amd import table
    info = ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=40, ivector_dim=10, seed=4, splice_left=2, splice_right=1, max_count=5.0)
    conf = ivector.write_config_dir(tmp_path / "ie", info)
    rng = np.random.default_rng(9)
    feats = {"a1": 47, "a2": 130, "b1": 80}
    feats = {k: (rng.standard_normal((T, 8)) + 0.2).astype(np.float32) for k, T in feats.items()}
    with table.TableWriter("ark,scp:%s,%s" % (tmp_path / "f.ark", tmp_path / "f.scp"), "matrix") as w:
        for k in ("a1", "a2", "b1"):
            w.write(k, feats[k])
    (tmp_path / "spk2utt").write_text("spkA a1 a2\nspkB b1\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, root + "/tools/ivector_extract_online2.py", "--config=%s" % conf, "--max-remembered-frames=60",
                        "ark:%s" % (tmp_path / "spk2utt"), "scp:%s" % (tmp_path / "f.scp"), "ark:%s" % (tmp_path / "iv.ark")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Estimated iVectors for 3 files, 0 with errors." in r.stderr
    got = list(table.SequentialTableReader("ark:%s" % (tmp_path / "iv.ark"), "matrix"))
    assert [k for k, _ in got] == ["a1", "a2", "b1"]
    got = dict(got)
    w1, st = orc.ivector_extract_online(info, feats["a1"], return_state=True, max_remembered_frames=60.0)
    w2 = orc.ivector_extract_online(info, feats["a2"], state=st)
    w3 = orc.ivector_extract_online(info, feats["b1"])
    for k, w in (("a1", w1), ("a2", w2), ("b1", w3)):
        np.testing.assert_allclose(got[k], w, rtol=0, atol=1e-4 * max(1.0, np.abs(w).max()))


def test_streaming_updates_match_the_oracle_schedule():
    """kamd_ivector_stream_update_device: two streams with different arrival schedules (frames in uneven batches, one
    batch longer than the i-vector period, a call without new frames in between is simply not made); every estimate
    equals the oracle's GetFrame sequence, and the record's statistics equal the oracle's at the end."""
    import ctypes as C
    from kaldi_amd import cmvn
    from kaldi_amd._lib import check, lib
    from kaldi_amd.decoder import DeviceMatrix
    info = ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=40, ivector_dim=10, seed=4, splice_left=2, splice_right=1,
                                  max_count=2.0, cmn_window=50, speaker_frames=50, global_frames=20)
    ie = ivector.IvectorExtractor(info)
    rng = np.random.default_rng(5)
    T = [97, 140]
    max_frames = 160
    feats = np.zeros((2 * max_frames, 8), np.float32)
    x = [(rng.standard_normal((t, 8)) * 1.1 + 0.4).astype(np.float32) for t in T]
    for s in range(2):
        feats[s * max_frames:s * max_frames + T[s]] = x[s]
    # base frames ready at each tick; frames that may enter the statistics = ready - splice_right until the end
    sched = [[5, 23, 24, 60, 97], [30, 31, 95, 120, 140]]
    RS = lib().kamd_ivector_stream_record_size(ie._h)
    rec = np.zeros((2, RS), np.float64)
    dp = C.POINTER(C.c_double)
    for s in range(2):
        check(lib().kamd_ivector_stream_record_init(ie._h, None, rec[s].ctypes.data_as(dp)))
    d_feats, d_rec, d_out = DeviceMatrix(feats), cmvn._Dev(rec), DeviceMatrix(np.zeros((2, 10), np.float32))
    done = [0, 0]
    got = [[], []]
    for tick in range(5):
        items = []
        for s in range(2):
            ready = sched[s][tick]
            upto = ready if ready == T[s] else max(0, ready - info.splice_right)
            if upto > done[s]:
                items.append((s, ready, done[s], upto))
        fr = np.asarray([s * max_frames for s, _, _, _ in items], np.int64)
        nb = np.asarray([r for _, r, _, _ in items], np.int32)
        nd = np.asarray([d for _, _, d, _ in items], np.int32)
        nu = np.asarray([u for _, _, _, u in items], np.int32)
        ri = np.asarray([s for s, _, _, _ in items], np.int32)
        check(lib().kamd_ivector_stream_update_device(ie._h, d_feats.ptr(0), 8, 2 * max_frames, abi.iptr(fr, C.c_int64), abi.iptr(nb), abi.iptr(nd),
                                                      abi.iptr(nu), abi.iptr(ri), len(items), d_rec.p, d_out.ptr(0), None))
        out = d_out.download()
        for k, (s, _, _, u) in enumerate(items):
            got[s].append((u, out[k].copy()))
            done[s] = u
    d_rec.download(rec)
    for s in range(2):
        uptos = [u for u, _ in got[s]]
        want, st = orc.ivector_extract_streaming(info, x[s], uptos)
        for (u, g), w in zip(got[s], want):
            np.testing.assert_allclose(g, w, rtol=0, atol=1e-4 * max(1.0, np.abs(w).max()), err_msg="stream %d upto %d" % (s, u))
        n0 = 2 * 9
        np.testing.assert_allclose(rec[s][n0:info.state_size()], st[n0:], rtol=1e-9, atol=1e-9 * np.abs(st).max())
    assert len(got[0]) == 5 and len(got[1]) == 5


def test_long_utterance_split_with_an_extractor_and_a_loglike_override(monkeypatch):
    """Round 4: kamd_batch_decoder_set_long_decoder also when the model takes online i-vectors from the device extractor and
    when the search reads a log-likelihood override (bench.py's headline on ranks of 4 and more): the long utterances are
    stored first, scored by a pass of their own (its own i-vectors and chunks) and searched on the second decoder object.
    Nothing an utterance returns may depend on the way it went: log-likelihoods, lattices, best paths equal the plain run."""
    from kaldi_amd import batch
    monkeypatch.setenv("KAMD_BATCH_SPLIT", "1")
    g = synth.make_hclg(num_units=40, vocab=120, n_hist=20, seed=4)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, ivector_dim=100, output_scale=3.0)
    cfg = abi.decoder_config_recipe()
    durs = [0.9, 2.6, 1.1, 0.8, 3.1, 1.3, 0.7, 1.6, 2.9, 1.0]
    waves = [synth.make_wave(d, seed=300 + i) for i, d in enumerate(durs)]
    op = abi.mfcc_opts_hires()
    allf = feat.Mfcc(op).ComputeFeatures(waves[1])
    ie = ivector.IvectorExtractor(ivector.make_synthetic(seed=11, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=100.0))
    kw = dict(max_seconds=3.5, resident_lanes=4, host_threads=2, determinize=True, keep_raw_lattices=True, search_mode=2, nnet_pass_frames=400,
              first_pass_frames=150)
    plain = batch.NnetBatchDecoder(op, m, g, cfg, **kw)
    plain.set_ivector_extractor(ie, 50)
    plain.load_host(waves)
    st0 = plain.run()
    assert st0.n_failed == 0 and st0.long_utterances == 0
    split = batch.NnetBatchDecoder(op, m, g, cfg, long_lanes=3, **kw)
    split.set_ivector_extractor(ie, 50)
    split.load_host(waves)
    st = split.run()
    assert st.n_failed == 0 and st.long_utterances == 3 and st.ivector_ms > 0
    for u in range(len(waves)):
        np.testing.assert_array_equal(split.loglikes(u), plain.loglikes(u))
        assert lattices_equal(split.raw_lattice(u), plain.raw_lattice(u))
        assert split.output(u)["words"].tolist() == plain.output(u)["words"].tolist()
    # the search on planted rows (load order = the caller's order: the library maps them to where it stored the utterances)
    fr = split.output_frames()
    planted = [synth.random_loglikes(int(t), g.num_pdfs, seed=70 + u, scale=2.0) for u, t in enumerate(fr)]
    dev = decoder.DeviceMatrix(np.concatenate(planted, axis=0))
    split.set_loglike_override(dev.ptr(0))
    st = split.run()
    assert st.n_failed == 0 and st.long_utterances == 3
    for u in range(len(waves)):
        o = orc.Decoder(g, cfg, 2)
        o.Decode(planted[u])
        assert lattices_equal(split.raw_lattice(u), o.GetRawLattice()), u


def test_statistics_pass_by_pass_then_one_solver_launch_equals_the_one_call_form():
    """kamd_ivector_online_reserve_steps / _stats_device (per pass) / _solve_device (once, over every pass's utterances) against
    kamd_ivector_extract_online_device on the same device features: the same kernels on the same rows, bit for bit -- also
    with the utterances handed to the passes out of order and with another extraction in between (the step statistics of a
    stats / solve sequence live in rows of their own)."""
    import ctypes as C
    from kaldi_amd._lib import check as ck, lib
    info = ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=70, ivector_dim=10, seed=3, splice_left=2, splice_right=1)
    ie = ivector.IvectorExtractor(info)
    rng = np.random.default_rng(11)
    lens = [53, 7, 160, 1, 29, 88, 10, 11]
    feats = [(rng.standard_normal((T, 8)) * 1.5 + 0.3).astype(np.float32) for T in lens]
    dm = decoder.DeviceMatrix(np.concatenate(feats, axis=0))
    row = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    niv = [ie.num_ivectors(T) for T in lens]
    orow = np.concatenate([[0], np.cumsum(niv)]).astype(np.int64)
    out_a = decoder.DeviceMatrix(np.zeros((int(orow[-1]), 10), np.float32))
    out_b = decoder.DeviceMatrix(np.full((int(orow[-1]), 10), 7.0, np.float32))
    ck(lib().kamd_ivector_extract_online_device(ie._h, dm.ptr(0), abi.iptr(row, C.c_int64), 8, len(lens), out_a.ptr(0), abi.iptr(orow, C.c_int64), None))
    ck(lib().kamd_device_synchronize())
    want = out_a.download()
    for u, f in enumerate(feats):
        np.testing.assert_array_equal(want[orow[u]:orow[u + 1]], ie.extract_online(f))
    ck(lib().kamd_ivector_online_reserve_steps(ie._h, int(orow[-1])))
    for u0, u1 in ((5, 8), (0, 2), (2, 5)):                   # three "passes", not in order
        ck(lib().kamd_ivector_online_stats_device(ie._h, dm.ptr(0), abi.iptr(np.ascontiguousarray(row[u0:u1 + 1]), C.c_int64), 8, u1 - u0,
                                                  abi.iptr(np.ascontiguousarray(orow[u0:u1 + 1]), C.c_int64), None))
    ck(lib().kamd_ivector_online_solve_device(ie._h, abi.iptr(row, C.c_int64), len(lens), out_b.ptr(0), abi.iptr(orow, C.c_int64), None))
    ck(lib().kamd_device_synchronize())
    np.testing.assert_array_equal(out_b.download(), want)
    # rows outside what was reserved are refused
    bad = orow + 5
    assert lib().kamd_ivector_online_stats_device(ie._h, dm.ptr(0), abi.iptr(row, C.c_int64), 8, len(lens), abi.iptr(bad, C.c_int64), None) < 0
