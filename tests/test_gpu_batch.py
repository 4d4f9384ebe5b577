"""kamd_batch_decoder_* (the NnetBatchDecoder mirror): a shard of utterances through features,
nnet passes, ONE work-queue launch of the search and the threaded host tail.  Per utterance the
result must be what the oracle gives for the device's own log-likelihoods, bit for bit, and what
the batch pipeline (three launches) gives for the same waveforms."""
import numpy as np
import pytest

from kaldi_amd import abi, batch, io as kio, nnet, pipeline, synth
from oracle import orc
from tests.util import assert_work_counters, lattice_diff, lattices_equal

pytestmark = pytest.mark.gpu


def _setup(n=23, seed=3):
    g = synth.make_hclg(num_units=32, vocab=120, n_hist=20, seed=8)
    model = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=4.0)
    cfg = abi.decoder_config_recipe()
    rng = np.random.default_rng(seed)
    waves = [synth.make_wave(float(s), seed=100 + i) for i, s in enumerate(rng.uniform(0.3, 3.0, n))]
    return g, model, cfg, waves


@pytest.mark.parametrize("mode", [1, 2])
def test_batch_decoder_equals_oracle_and_pipeline(mode):
    g, model, cfg, waves = _setup()
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, max_seconds=4.0, resident_lanes=4, host_threads=3,
                                determinize=True, keep_raw_lattices=True, nnet_pass_frames=900, search_mode=mode)
    bd.load(waves)
    st = bd.run()
    assert st.n_failed == 0 and st.lanes == 4 and st.nnet_passes > 1
    assert st.total_ms > 0 and st.decode_ms > 0 and st.nnet_flops > 0
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), model, g, cfg, max_utts=len(waves), max_seconds=4.0)
    ref = pipe.decode(waves)
    for u in range(len(waves)):
        ll = bd.loglikes(u)
        np.testing.assert_array_equal(ll, pipe.loglikes(u))        # sub-batching the nnet changes nothing (k-ordered chains)
        o = orc.Decoder(g, cfg, mode)
        o.Decode(ll)
        lo = o.GetRawLattice()
        lat = bd.raw_lattice(u)
        assert lattices_equal(lat, lo), "utt %d: %s" % (u, lattice_diff(lat, lo))
        if mode == 1:                                   # the three-launch pipeline runs the default search mode 1
            assert lattices_equal(lat, ref[u]["lattice"])
        out = bd.output(u)
        bo = lo.best_path()
        assert out["words"].tolist() == bo["words"].tolist() == ref[u]["words"].tolist()
        assert out["alignment"].tolist() == bo["alignment"].tolist()
        assert out["graph_cost"] == bo["graph_cost"] and out["acoustic_cost"] == bo["acoustic_cost"]
        assert out["record"].n_frames == ll.shape[0] and out["record"].error == 0
        assert_work_counters(out["record"], o.counters())
        cl, cr = bd.compact_lattice(u), kio.determinize_lattice(lat, cfg.lattice_beam)
        assert cl.num_states == cr.num_states and cl.arcs.tobytes() == cr.arcs.tobytes()
        assert cl.strings.tobytes() == cr.strings.tobytes() and cl.final.tobytes() == cr.final.tobytes()
    # a second run of the same shard, and a different shard through the same object
    keep = {u: bd.raw_lattice(u) for u in (0, 5, len(waves) - 1)}
    st2 = bd.run()
    assert st2.n_failed == 0
    for u, l in keep.items():
        assert lattices_equal(bd.raw_lattice(u), l)
    first = {u: bd.raw_lattice(u) for u in range(len(waves))}
    bd.load(waves[3:9])
    bd.run()
    for k, u in enumerate(range(3, 9)):
        assert lattices_equal(bd.raw_lattice(k), first[u])


def test_batch_decoder_without_raw_lattices_or_determinization():
    g, model, cfg, waves = _setup(n=7, seed=5)
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, max_seconds=4.0, resident_lanes=2, host_threads=1,
                                determinize=False, keep_raw_lattices=False)
    bd.load(waves)
    bd.run()
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), model, g, cfg, max_utts=len(waves), max_seconds=4.0)
    ref = pipe.decode(waves)
    for u in range(len(waves)):
        assert bd.output(u)["words"].tolist() == ref[u]["words"].tolist()
        assert bd.compact_lattice(u) is None
        with pytest.raises(Exception):
            bd.raw_lattice(u)


def test_batch_decoder_accepts_feature_matrices_and_ivectors():
    """AcceptInput as the reference declares it (nnet-batch-compute.h:665): feature matrices in, plus one
    i-vector per utterance.  Without i-vectors the result equals the waveform-in run; with them, the oracle's
    DecodableNnetSimple + decoder on the same matrices."""
    from kaldi_amd import feat
    g, model, cfg, waves = _setup(n=9, seed=11)
    mf = feat.Mfcc(abi.mfcc_opts_hires())
    feats = [mf.ComputeFeatures(w) for w in waves]
    bw = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, max_seconds=4.0, resident_lanes=3, host_threads=2,
                                keep_raw_lattices=True)
    bw.load(waves)
    bw.run()
    bf = batch.NnetBatchDecoder(None, model, g, cfg, max_seconds=4.0, resident_lanes=3, host_threads=2, keep_raw_lattices=True)
    with pytest.raises(Exception):
        bf.load(waves)                                # no feature stage
    bf.load_features(feats)
    st = bf.run()
    assert st.n_failed == 0
    for u in range(len(waves)):
        np.testing.assert_array_equal(bf.loglikes(u), bw.loglikes(u))
        assert lattices_equal(bf.raw_lattice(u), bw.raw_lattice(u))
        assert bf.output(u)["words"].tolist() == bw.output(u)["words"].tolist()
    with pytest.raises(Exception):
        bf.load_features([f[:, :13] for f in feats])  # wrong feature dim
    with pytest.raises(Exception):
        bf.load_features(feats, ivectors=np.zeros((len(feats), 10), np.float32))   # the model has no ivector node
    # a model with an ivector node
    mi = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, ivector_dim=10, output_scale=3.0)
    ivs = np.random.default_rng(4).standard_normal((len(feats), 10)).astype(np.float32)
    bi = batch.NnetBatchDecoder(None, mi, g, cfg, max_seconds=4.0, resident_lanes=2, host_threads=2, keep_raw_lattices=True,
                                nnet_pass_frames=700)
    with pytest.raises(Exception):
        bi.load_features(feats)                       # missing i-vectors
    bi.load_features(feats, ivectors=ivs)
    st = bi.run()
    assert st.n_failed == 0 and st.nnet_passes > 1
    for u in range(len(feats)):
        want = orc.nnet_forward(mi, feats[u], ivector=ivs[u])
        ll = bi.loglikes(u)
        np.testing.assert_allclose(ll, want, rtol=2e-4, atol=2e-4)
        o = orc.Decoder(g, cfg, 2)
        o.Decode(ll)
        assert lattices_equal(bi.raw_lattice(u), o.GetRawLattice())


def test_utterances_too_short_for_a_frame_fail_alone():
    """nnet3-latgen-faster-batch.cc:184-188 warns "Zero-length utterance" and goes on: an utterance with no frame must not
    take the shard down; the others decode as if it were not there."""
    from kaldi_amd import feat
    g, model, cfg, waves = _setup(n=6, seed=13)
    short = [np.zeros(0, np.float32), waves[0][:100]]                   # no samples at all; 100 samples < one 25 ms window
    mixed = [waves[0], short[0], waves[1], waves[2], short[1], waves[3], waves[4], waves[5]]
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, max_seconds=4.0, resident_lanes=3, host_threads=2, keep_raw_lattices=True)
    bd.load(waves)
    bd.run()
    want = [bd.raw_lattice(u) for u in range(len(waves))]
    bd.load(mixed)
    st = bd.run()
    assert st.n_failed == 2
    pos = [0, 2, 3, 5, 6, 7]
    for k, u in enumerate(pos):
        assert lattices_equal(bd.raw_lattice(u), want[k])
        np.testing.assert_array_equal(bd.loglikes(u).shape[1], g.num_pdfs)
    for u in (1, 4):
        assert bd.output(u) is None and bd.loglikes(u).shape[0] == 0
        with pytest.raises(Exception):
            bd.raw_lattice(u)
    # the same through the feature-matrix entry
    mf = feat.Mfcc(abi.mfcc_opts_hires())
    feats = [mf.ComputeFeatures(w) for w in waves]
    mixed_f = [feats[0], np.zeros((0, 40), np.float32), feats[1], feats[2], feats[3], np.zeros((0, 40), np.float32), feats[4], feats[5]]
    bf = batch.NnetBatchDecoder(None, model, g, cfg, max_seconds=4.0, resident_lanes=3, host_threads=2, keep_raw_lattices=True)
    bf.load_features(mixed_f)
    st = bf.run()
    assert st.n_failed == 2
    for k, u in enumerate([0, 2, 3, 4, 6, 7]):
        assert lattices_equal(bf.raw_lattice(u), want[k])
    # nothing but short utterances: a run that decodes nothing and says so
    bd.load(short)
    st = bd.run()
    assert st.n_failed == 2 and bd.output(0) is None


def test_lattice_pool_too_small_fails_the_utterances_that_do_not_fit_alone(monkeypatch):
    """The finished lattices wait in a bump-allocated pool of page-locked host memory (kamd_decoder_queue_configure).  When
    it is exhausted the utterances that no longer fit report flag 64 (lattice pool) and no output; those that did fit are
    what they are with a pool of the default size; the next run of the same object starts with an empty pool.  That is the
    first search; by default run() then gives the failed ones a second chance (fewer utterances per launch: each finds room)."""
    monkeypatch.setenv("KAMD_BATCH_RETRY", "0")
    g, model, cfg, waves = _setup(n=12, seed=5)
    ref = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, max_seconds=4.0, resident_lanes=3, host_threads=2, keep_raw_lattices=True)
    ref.load(waves)
    assert ref.run().n_failed == 0
    want = [ref.raw_lattice(u) for u in range(len(waves))]
    blob = [ref.record(u).blob_bytes for u in range(len(waves))]
    pool = int(max(sum(sorted(blob)[:5]), max(blob)) + 64)       # room for a handful of lattices (and for any single one), not for all twelve
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, max_seconds=4.0, resident_lanes=3, host_threads=2,
                                keep_raw_lattices=True, lattice_pool_bytes=max(pool, 4096))
    bd.load(waves)
    for _ in range(2):                                           # twice: the pool is reset by every launch
        st = bd.run()
        assert 0 < st.n_failed < len(waves)
        n_ok = 0
        for u in range(len(waves)):
            rec = bd.record(u)
            if bd.output(u) is None:
                assert rec.error & 64 and rec.blob_bytes == 0
                with pytest.raises(Exception, match="lattice-pool"):
                    bd.raw_lattice(u)
            else:
                assert rec.error == 0 and lattices_equal(bd.raw_lattice(u), want[u])
                n_ok += 1
        assert n_ok == len(waves) - st.n_failed and n_ok >= 1, (n_ok, st.n_failed, pool, sorted(blob))
    monkeypatch.delenv("KAMD_BATCH_RETRY")
    st = bd.run()                                                # second chance on: nobody fails, and nothing changes for anybody
    assert st.n_failed == 0 and st.n_retried > 0
    for u in range(len(waves)):
        assert bd.record(u).error == 0 and lattices_equal(bd.raw_lattice(u), want[u])


def test_token_arena_too_small_gets_a_second_chance_on_a_wider_lane(monkeypatch):
    """An utterance whose lane runs out of token / link arena fails alone (flags 2 / 4) in the first search; the second
    chance searches it on a lane that owns 8x (then 64x) the arena.  Same lattices as a run with ample arenas."""
    g, model, cfg, waves = _setup(n=10, seed=21)
    kw = dict(max_seconds=4.0, resident_lanes=8, host_threads=2, keep_raw_lattices=True)
    ref = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, **kw)
    ref.load(waves)
    assert ref.run().n_failed == 0
    want = [ref.raw_lattice(u) for u in range(len(waves))]
    tpf = max(4, int(max(ref.record(u).counters[5] / max(ref.record(u).n_frames, 1) for u in range(len(waves))) / 3))
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, tokens_per_frame=tpf, **kw)      # a third of what the densest needs
    bd.load(waves)
    monkeypatch.setenv("KAMD_BATCH_RETRY", "0")
    st = bd.run()
    assert st.n_failed > 0 and st.n_retried == 0
    assert all(bd.record(u).error & (2 | 4) for u in range(len(waves)) if bd.output(u) is None)
    monkeypatch.delenv("KAMD_BATCH_RETRY")
    st = bd.run()
    assert st.n_failed == 0 and st.n_retried > 0
    for u in range(len(waves)):
        assert lattices_equal(bd.raw_lattice(u), want[u])


def test_long_utterances_searched_beside_the_acoustic_model_give_the_same_results(monkeypatch):
    """kamd_batch_decoder_set_long_decoder: a shard whose search is bound by its longest utterance has that utterance (the
    `long_lanes` longest) scored first and searched on a second decoder object while the model of the others runs.  Nothing
    an utterance returns may depend on which way it went: log-likelihoods, raw and determinized lattices, best paths and
    records equal those of the plain run; the split happens by itself for a chain-bound shard and not for a balanced one."""
    g, model, cfg, _ = _setup(n=1)
    rng = np.random.default_rng(11)
    durs = [6.0] + list(rng.uniform(0.3, 0.9, 30))                      # one long utterance over 4 lanes' worth of short ones
    waves = [synth.make_wave(float(s), seed=300 + i) for i, s in enumerate(durs)]
    order = rng.permutation(len(waves))
    waves = [waves[i] for i in order]
    kw = dict(max_seconds=7.0, resident_lanes=16, host_threads=3, determinize=True, keep_raw_lattices=True, nnet_pass_frames=1200, search_mode=2)
    plain = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, **kw)
    plain.load(waves)
    st0 = plain.run()
    assert st0.n_failed == 0 and st0.long_utterances == 0
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, long_lanes=2, **kw)
    bd.load(waves)
    for forced in (None, "1", "0"):
        if forced is None:
            monkeypatch.delenv("KAMD_BATCH_SPLIT", raising=False)
        else:
            monkeypatch.setenv("KAMD_BATCH_SPLIT", forced)
        st = bd.run()
        assert st.n_failed == 0
        assert st.long_utterances == (0 if forced == "0" else 2)        # 0.7 x 200 frames > 31 utterances' frames / 16 lanes
        for u in range(len(waves)):
            np.testing.assert_array_equal(bd.loglikes(u), plain.loglikes(u))
            assert lattices_equal(bd.raw_lattice(u), plain.raw_lattice(u))
            a, c = bd.output(u), plain.output(u)
            assert a["words"].tolist() == c["words"].tolist() and a["alignment"].tolist() == c["alignment"].tolist()
            assert a["graph_cost"] == c["graph_cost"] and a["acoustic_cost"] == c["acoustic_cost"]
            assert a["record"].n_frames == c["record"].n_frames and list(a["record"].counters[:7]) == list(c["record"].counters[:7])
            x, y = bd.compact_lattice(u), plain.compact_lattice(u)
            assert x.num_states == y.num_states and x.arcs.tobytes() == y.arcs.tobytes() and x.strings.tobytes() == y.strings.tobytes()
    # the same with the waveforms uploaded inside run() (load_host stores the long utterances first, as a pass of their own)
    monkeypatch.delenv("KAMD_BATCH_SPLIT", raising=False)
    bh = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, long_lanes=2, **kw)
    bh.load_host(waves)
    for rep in range(2):
        st = bh.run()
        assert st.n_failed == 0 and st.long_utterances == 2 and st.upload_passes == st.nnet_passes > 2
        for u in range(len(waves)):
            np.testing.assert_array_equal(bh.loglikes(u), plain.loglikes(u))
            assert lattices_equal(bh.raw_lattice(u), plain.raw_lattice(u))
            a, c = bh.output(u), plain.output(u)
            assert a["words"].tolist() == c["words"].tolist() and a["graph_cost"] == c["graph_cost"] and a["acoustic_cost"] == c["acoustic_cost"]
            x, y = bh.compact_lattice(u), plain.compact_lattice(u)
            assert x.num_states == y.num_states and x.arcs.tobytes() == y.arcs.tobytes()
    # ... and a planted matrix in LOAD order still reaches the right utterances, the long ones on the second decoder object
    # included (round 4: the override's rows follow the caller's order, the stored order has the long ones first)
    fr = bh.output_frames()
    planted = [synth.random_loglikes(int(t), g.num_pdfs, seed=70 + u, scale=2.0) for u, t in enumerate(fr)]
    from kaldi_amd import decoder
    dev = decoder.DeviceMatrix(np.concatenate(planted, axis=0))
    bh.set_loglike_override(dev.ptr(0))
    assert bh.run().long_utterances == 2
    for u in (0, int(np.argmax(fr)), int(np.argsort(fr)[-2]), len(waves) - 1):
        o = orc.Decoder(g, cfg, 2)
        o.Decode(planted[u])
        assert lattices_equal(bh.raw_lattice(u), o.GetRawLattice())
    bh.set_loglike_override(None)
    # a balanced shard (many utterances per lane) is left alone
    monkeypatch.delenv("KAMD_BATCH_SPLIT", raising=False)
    many = [synth.make_wave(float(s), seed=500 + i) for i, s in enumerate(rng.uniform(0.5, 1.0, 40))]
    b2 = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, long_lanes=2, max_seconds=7.0, resident_lanes=2, host_threads=2)
    b2.load(many)
    assert b2.run().long_utterances == 0


def test_nnet3_latgen_faster_batch_tool(tmp_path):
    """tools/nnet3_latgen_faster_batch.py = nnet3-latgen-faster-batch's command line over the work-queue path: the same
    lattices as tools/nnet3_latgen_faster.py (three launches per batch) writes for the same files, in input order, with a
    zero-length utterance failing alone, several sets per run, and --wav (features on the device)."""
    import os
    import subprocess
    import sys
    import wave
    from kaldi_amd import feat, latbin, table
    from kaldi_amd import io as kio
    from tests.mdl_writer import write_mdl
    g = synth.make_hclg(num_units=25, vocab=60, n_hist=12, seed=6)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, seed=12, output_scale=3.0)
    write_mdl(tmp_path / "final.mdl", m, num_units=25)
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    waves = [np.round(synth.make_wave(d, seed=50 + i)).astype(np.float32) for i, d in enumerate((1.2, 2.0, 0.8, 1.6, 1.1))]
    mf = feat.Mfcc(abi.mfcc_opts_hires())
    with table.TableWriter("ark,scp:%s,%s" % (tmp_path / "feats.ark", tmp_path / "feats.scp"), "matrix") as w:
        for i, wv in enumerate(waves):
            w.write("utt%d" % i, mf.ComputeFeatures(wv))
            if i == 1:
                w.write("empty", np.zeros((0, 40), np.float32))
    with open(tmp_path / "wav.scp", "w") as scp:
        for i, wv in enumerate(waves):
            with wave.open(str(tmp_path / ("u%d.wav" % i)), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(wv.astype("<i2").tobytes())
            scp.write("utt%d %s\n" % (i, tmp_path / ("u%d.wav" % i)))
    (tmp_path / "words.txt").write_text("".join("w%d %d\n" % (k, k) for k in range(0, 61)))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--beam=15", "--max-active=7000", "--lattice-beam=8", "--acoustic-scale=1.0", "--frame-subsampling-factor=3"]
    mdl_, fst_ = str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst")
    r = subprocess.run([sys.executable, root + "/tools/nnet3_latgen_faster_batch.py"] + common +
                       ["--num-threads=3", "--set-frames=250", "--search-mode=1", "--word-symbol-table=%s" % (tmp_path / "words.txt"),
                        mdl_, fst_, "scp:%s" % (tmp_path / "feats.scp"), "ark:%s" % (tmp_path / "batch.lat")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Zero-length utterance: empty" in r.stderr and "Decoded 6 utterances, 1 with errors." in r.stderr
    r2 = subprocess.run([sys.executable, root + "/tools/nnet3_latgen_faster.py"] + common +
                        [mdl_, fst_, "scp:%s" % (tmp_path / "feats.scp"), "ark:%s" % (tmp_path / "plain.lat")], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert open(tmp_path / "batch.lat", "rb").read() == open(tmp_path / "plain.lat", "rb").read()
    got = [k for k, _ in latbin.read_lattices("ark:%s" % (tmp_path / "batch.lat"))]
    assert got == ["utt%d" % i for i in range(5)]                  # input order, across three sets
    assert any(l.startswith("utt0 w") for l in r.stderr.splitlines())      # the sentence of every utterance with --word-symbol-table
    # waveforms in, raw lattices out
    r3 = subprocess.run([sys.executable, root + "/tools/nnet3_latgen_faster_batch.py"] + common +
                        ["--wav", "--determinize-lattice=false", "--search-mode=1", mdl_, fst_, "scp:%s" % (tmp_path / "wav.scp"),
                         "ark:%s" % (tmp_path / "raw.lat")], capture_output=True, text=True)
    assert r3.returncode == 0, r3.stderr[-2000:]
    r4 = subprocess.run([sys.executable, root + "/tools/nnet3_latgen_faster.py"] + common +
                        ["--wav", "--determinize-lattice=false", mdl_, fst_, "scp:%s" % (tmp_path / "wav.scp"), "ark:%s" % (tmp_path / "raw2.lat")],
                        capture_output=True, text=True)
    assert r4.returncode == 0, r4.stderr[-2000:]
    a = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "raw.lat"), "lattice"))
    b = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "raw2.lat"), "lattice"))
    assert list(a) == list(b) and all(a[k][2].size == b[k][2].size for k in a)
    bad = subprocess.run([sys.executable, root + "/tools/nnet3_latgen_faster_batch.py", "--online-ivectors=ark:x", mdl_, fst_, "scp:a", "ark:b"],
                         capture_output=True, text=True)
    assert bad.returncode == 255 and "not supported" in bad.stderr


def test_load_host_uploads_inside_run_and_changes_nothing():
    """kamd_batch_decoder_load_host: the samples stay in host memory, every run() uploads them pass by pass behind the
    features and the model of the passes before (small first pass).  Same log-likelihoods and lattices as load(), bit for
    bit, also with an utterance that is too short for a frame in the middle of the buffer; the loglike override feeds the
    search a planted matrix while the model still runs."""
    g, model, cfg, waves = _setup(n=19, seed=11)
    waves.insert(7, np.zeros(100, np.float32))                 # < one frame: skipped, fails alone
    kw = dict(max_seconds=4.0, resident_lanes=4, host_threads=3, determinize=True, keep_raw_lattices=True, nnet_pass_frames=700,
              search_mode=2)
    ref = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, **kw)
    ref.load(waves)
    ref.run()
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, first_pass_frames=150, **kw)
    bd.load_host(waves)
    for rep in range(2):
        st = bd.run()
        assert st.n_failed == 1 and st.upload_passes == st.nnet_passes > 2 and st.upload_ms > 0 and st.first_pass_start_ms > 0
        assert st.feat_ms > 0 and st.nnet_ms > 0
        for u in range(len(waves)):
            if u == 7:
                assert bd.output(u) is None and bd.record(u).n_frames == 0
                continue
            np.testing.assert_array_equal(bd.loglikes(u), ref.loglikes(u))
            assert lattices_equal(bd.raw_lattice(u), ref.raw_lattice(u)), lattice_diff(bd.raw_lattice(u), ref.raw_lattice(u))
            assert bd.output(u)["words"].tolist() == ref.output(u)["words"].tolist()
    fr = bd.output_frames()
    assert fr[7] == 0 and all(fr[u] == ref.loglikes(u).shape[0] for u in range(len(waves)))
    # planted log-likelihoods: the search must see THEM (oracle on the planted matrix), the model's output is untouched
    from kaldi_amd import decoder
    P = g.num_pdfs
    planted = [synth.random_loglikes(int(t), P, seed=50 + u, scale=2.0) for u, t in enumerate(fr) if t > 0]
    dev = decoder.DeviceMatrix(np.concatenate(planted, axis=0))
    bd.set_loglike_override(dev.ptr(0))
    bd.run()
    k = 0
    for u in range(len(waves)):
        if u == 7:
            continue
        o = orc.Decoder(g, cfg, 2)
        o.Decode(planted[k])
        k += 1
        assert lattices_equal(bd.raw_lattice(u), o.GetRawLattice())
        np.testing.assert_array_equal(bd.loglikes(u), ref.loglikes(u))
    bd.set_loglike_override(None)
    bd.run()
    assert lattices_equal(bd.raw_lattice(3), ref.raw_lattice(3))
    # kamd_batch_decoder_unload_host: the page lock goes before the memory does; the last run's outputs stay readable,
    # another run() needs a new load; loading again (which unlocks the old buffer inside the call) works as before
    bd.unload_host()
    assert lattices_equal(bd.raw_lattice(3), ref.raw_lattice(3)) and bd.output(3)["words"].tolist() == ref.output(3)["words"].tolist()
    with pytest.raises(Exception, match="released"):
        bd.run()
    bd.load_host(waves)
    bd.load_host(list(waves))                                   # a second buffer while the first is still registered
    assert bd.run().n_failed == 1
    assert lattices_equal(bd.raw_lattice(5), ref.raw_lattice(5))
