"""Streaming (config 5) parity: chunked on-device features == offline features, streaming
nnet rows == offline forward, chunk-wise decoding == one-shot decoding (bit-exact lattice),
partial best path == oracle's best path over the un-finalized lattice."""
import numpy as np
import pytest

from kaldi_amd import abi, decoder, feat, nnet, online, synth
from oracle import orc
from tests.util import lattice_diff, lattices_equal

pytestmark = pytest.mark.gpu
CHUNK = int(0.18 * 16000)      # online2-wav-nnet3-latgen-faster.cc:107 chunk-length 0.18 s


@pytest.mark.parametrize("snip", [1, 0])
def test_online_features_equal_offline(snip):
    op = abi.mfcc_opts_hires()
    op.frame.snip_edges = snip
    w = synth.make_wave(2.37, seed=3)
    off = feat.Mfcc(op).ComputeFeatures(w)
    on = online.OnlineMfcc(op)
    ready = []
    for i in range(0, w.size, CHUNK):
        on.AcceptWaveform(16000, w[i:i + CHUNK])
        ready.append(on.NumFramesReady())
        # NumFrames(total, opts, flush=false) (feat/feature-window.cc:41-87)
        assert ready[-1] == orc.lib().orc_feat_num_frames(op.frame, min(i + CHUNK, w.size)) - (
            0 if snip else _unflushed(op, min(i + CHUNK, w.size)))
    on.InputFinished()
    assert on.NumFramesReady() == off.shape[0] and on.IsLastFrame(off.shape[0] - 1)
    np.testing.assert_array_equal(on.GetFrames(0, off.shape[0]), off)   # same kernel => bit-equal
    assert ready == sorted(ready)
    with pytest.raises(Exception):
        on.AcceptWaveform(16000, w[:10])      # online-feature.cc:126: no waveform after InputFinished


def _unflushed(op, n):
    """frames whose window would run past the end when not flushing (snip_edges=false)."""
    shift, length = 160, 400
    nf = (n + shift // 2) // shift
    end = (shift * (nf - 1) + shift // 2 - length // 2) + length
    k = 0
    while nf - k > 0 and end > n:
        k += 1
        end -= shift
    return k


def test_streaming_nnet_rows_equal_offline():
    m = nnet.tdnnf_tiny(num_pdfs=40)
    N = decoder.Nnet(m)
    op = abi.mfcc_opts_hires()
    w = synth.make_wave(3.1, seed=5)
    feats = feat.Mfcc(op).ComputeFeatures(w)
    whole = N.Forward(feats)
    g = synth.make_hclg(num_units=20, vocab=30, n_hist=6, seed=1)
    cfg = abi.decoder_config_recipe()
    s = online.SingleUtteranceNnet3Decoder(op, N, decoder.Graph(g), cfg, sizes=abi.DecoderSizes(1, 1 << 14, 1 << 18, 1 << 19, 512))
    s.record_loglikes()
    for i in range(0, w.size, CHUNK):
        s.AcceptWaveform(16000, w[i:i + CHUNK])
        s.AdvanceDecoding()
        assert s.NumFramesDecoded() == s.NumFramesReady() <= whole.shape[0]
    s.InputFinished()
    s.AdvanceDecoding()
    got = s.loglikes()
    assert got.shape == whole.shape
    # same kernels, same k order; the per-row result does not depend on the slice
    np.testing.assert_array_equal(got, whole)


def test_streaming_decode_equals_offline_and_partial_results():
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    sz = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512)
    w = synth.make_wave(2.9, seed=9)
    s = online.SingleUtteranceNnet3Decoder(op, N, G, cfg, sizes=sz)
    s.record_loglikes()
    partial_checked = 0
    for i in range(0, w.size, CHUNK):
        s.AcceptWaveform(16000, w[i:i + CHUNK])
        if s.AdvanceDecoding() and s.NumFramesDecoded() > 3:
            # partial result (end_of_utterance=false): compare with the oracle decoding the
            # same rows so far, best path over its un-finalized lattice
            o = orc.Decoder(g, cfg, 1)
            o.InitDecoding()
            o.AdvanceDecoding(s.loglikes())
            for use_final in (False, True):
                bp = s.GetBestPath(end_of_utterance=use_final)
                lat = o.GetRawLattice()
                if not use_final:
                    lat.final[:] = np.where(lat.frame == lat.num_frames, 0.0, np.inf).astype(np.float32)
                ob = lat.best_path()
                assert bp["words"].tolist() == ob["words"].tolist()
                assert bp["alignment"].tolist() == ob["alignment"].tolist()
                assert abs((bp["graph_cost"] + bp["acoustic_cost"]) - (ob["graph_cost"] + ob["acoustic_cost"])) < 1e-3
            partial_checked += 1
    s.InputFinished()
    s.AdvanceDecoding()
    s.FinalizeDecoding()
    assert partial_checked >= 3
    ll = s.loglikes()
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    assert lattices_equal(s.GetRawLattice(), o.GetRawLattice()), lattice_diff(s.GetRawLattice(), o.GetRawLattice())
    # and equal to the offline pipeline on the whole waveform
    off = decoder.LatticeFasterDecoder(G, cfg, sz)
    off.Decode(N.Forward(feat.Mfcc(op).ComputeFeatures(w)))
    assert lattices_equal(s.GetRawLattice(), off.GetRawLattice())


def test_batched_streams_equal_offline():
    """Five streams of different lengths fed in unequal, unaligned chunks and decoded together:
    every stream's lattice equals the offline decode of its whole waveform, and a stream slot
    can be reused for a new utterance."""
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    S = 5
    sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=6.0, sizes=abi.DecoderSizes(S, 1 << 14, 1 << 19, 1 << 20, 512))
    rng = np.random.default_rng(0)
    durs = [2.9, 1.3, 4.1, 0.7, 3.3]
    waves = [synth.make_wave(d, seed=20 + i) for i, d in enumerate(durs)]
    off_sz = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512)

    def offline(w):
        d = decoder.LatticeFasterDecoder(G, cfg, off_sz)
        d.Decode(N.Forward(feat.Mfcc(op).ComputeFeatures(w)))
        return d.GetRawLattice()

    sb.start(np.arange(S))
    pos = [0] * S
    done = [False] * S
    ticks = 0
    while not all(done):
        live = [s for s in range(S) if not done[s]]
        pieces = []
        for s in live:
            n = int(rng.integers(800, 6000))                 # 50 .. 375 ms, different per stream and tick
            chunk = waves[s][pos[s]:pos[s] + n]
            pos[s] += chunk.size
            pieces.append(chunk)
        if ticks % 2:                                        # every other tick: one upload for all streams
            sb.accept_many(live, pieces, [pos[s] >= waves[s].size for s in live])
        else:
            for s, chunk in zip(live, pieces):
                sb.accept(s, chunk, input_finished=pos[s] >= waves[s].size)
        before = [int(x) for x in sb.advance(live)]
        for s, nd in zip(live, before):
            if pos[s] >= waves[s].size:
                done[s] = True
        ticks += 1
    assert ticks > 5
    sb.finalize(np.arange(S))
    for s in range(S):
        want = offline(waves[s])
        got = sb.raw_lattice(s)
        assert lattices_equal(got, want), (s, lattice_diff(got, want))
    with pytest.raises(Exception):                           # a stream may appear once per call
        sb.accept_many([0, 0], [np.zeros(4, np.float32)] * 2)
    # reuse slot 1 for another utterance while the others keep their results
    w2 = synth.make_wave(1.9, seed=77)
    sb.start([1])
    for i in range(0, w2.size, CHUNK):
        sb.accept(1, w2[i:i + CHUNK], input_finished=i + CHUNK >= w2.size)
        sb.advance([1])
        if i > 3 * CHUNK:
            assert sb.partial_best_path(1) is not None
    sb.finalize([1])
    assert lattices_equal(sb.raw_lattice(1), offline(w2))
    assert lattices_equal(sb.raw_lattice(2), offline(waves[2]))


def test_streams_pruned_every_prune_interval_frames_end_on_the_offline_lattice():
    """kamd_stream_batch_set_prune_interval = LatticeFasterDecoderConfig::prune_interval (lattice-faster-decoder.cc:617-619): a
    stream is compacted (PruneActiveTokens) whenever it has decoded that many frames since its last compaction, so that the
    FinalizeDecoding at the end of the utterance finds all but the last frames pruned.  The final raw lattice of every stream
    is still the offline decode of its whole waveform bit for bit, partial best paths in between stay those of the full
    walk, and a restarted stream counts its frames from zero again."""
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    S = 4
    sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=6.0, sizes=abi.DecoderSizes(S, 1 << 14, 1 << 19, 1 << 20, 512))
    sb.set_compaction(0.0)                                    # (only the interval compacts)
    sb.set_prune_interval(25)
    waves = [synth.make_wave(d, seed=20 + i) for i, d in enumerate([3.9, 1.3, 5.1, 2.2])]
    off_sz = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512)

    def offline(w):
        d = decoder.LatticeFasterDecoder(G, cfg, off_sz)
        d.Decode(N.Forward(feat.Mfcc(op).ComputeFeatures(w)))
        return d.GetRawLattice(), d.GetBestPath()

    sb.start(np.arange(S))
    pos = [0] * S
    rng = np.random.default_rng(5)
    while any(pos[s] < waves[s].size for s in range(S)):
        live = [s for s in range(S) if pos[s] < waves[s].size]
        for s in live:
            n = int(rng.integers(1500, 5000))
            sb.accept(s, waves[s][pos[s]:pos[s] + n], input_finished=pos[s] + n >= waves[s].size)
            pos[s] += n
        nd = sb.advance(live)
        cand = [s for s, d in zip(live, nd) if d > 0]
        if cand:
            for a, b in zip(sb.partial_best_paths(cand, incremental=True), sb.partial_best_paths(cand)):
                assert (a is None) == (b is None)
                if a is not None:
                    assert a["words"].tolist() == b["words"].tolist() and a["graph_cost"] == b["graph_cost"] and a["acoustic_cost"] == b["acoustic_cost"]
    n_comp = sb.num_compactions()
    assert n_comp >= 3 + 1 + 4 + 2                              # output frames // (25 + a tick's worth) per stream, at least
    sb.finalize(np.arange(S))
    for s in range(S):
        want, bp = offline(waves[s])
        got = sb.raw_lattice(s)
        assert lattices_equal(got, want), (s, lattice_diff(got, want))
        assert sb.best_path(s)["words"].tolist() == bp["words"].tolist()
    w2 = synth.make_wave(2.9, seed=77)                        # a slot reused: its frame count starts over
    sb.start([2])
    for i in range(0, w2.size, CHUNK):
        sb.accept(2, w2[i:i + CHUNK], input_finished=i + CHUNK >= w2.size)
        sb.advance([2])
    sb.finalize([2])
    assert lattices_equal(sb.raw_lattice(2), offline(w2)[0]) and sb.num_compactions() > n_comp


def test_batched_streams_with_online_ivectors():
    """online2-wav-nnet3-latgen-faster with an i-vector extractor, two streams fed in different chunkings:
    (1) the slots' i-vectors equal the oracle's GetFrame sequence under the reference's chunk schedule
        (chunk k computable once (k+1)*20 + right context frames exist; estimate advanced once per tick to
        frames_ready - splice_right; slots floor(t/20) of the new chunks' input ranges get it);
    (2) the lattice equals an offline decode of the network evaluated with exactly those slots (bit-equal);
    (3) the adaptation state left for the speaker's next utterance equals the oracle's."""
    from kaldi_amd import ivector
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=16, seed=12, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    waves = [synth.make_wave(d, seed=60 + i) for i, d in enumerate((3.0, 1.7))]
    feats = [feat.Mfcc(op).ComputeFeatures(w) for w in waves]
    allf = np.concatenate(feats)
    info = ivector.make_synthetic(num_gauss=64, ivector_dim=16, seed=9, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=10.0)
    ie = ivector.IvectorExtractor(info)
    S, C_ = 2, 20
    sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=5.0, sizes=abi.DecoderSizes(S, 1 << 14, 1 << 19, 1 << 20, 512))
    sb.set_ivector_extractor(ie, C_)
    C_ = sb.frames_per_chunk
    assert C_ == 21                       # --frames-per-chunk=20 rounded up to a multiple of the subsampling factor
    sb.start([0, 1])
    L, R = N.Context()
    step = [int(0.18 * 16000), int(0.31 * 16000)]
    pos = [0, 0]
    calls = [[], []]                      # oracle schedule: upto of every estimate update
    slot_src = [[], []]                   # per assigned slot: index into calls (or -1 = zero vector)
    chunks_done, iv_done = [0, 0], [0, 0]
    done = [False, False]
    sub = m.subsampling
    while not all(done):
        live = [s for s in range(S) if not done[s]]
        for s in live:
            chunk = waves[s][pos[s]:pos[s] + step[s]]
            pos[s] += chunk.size
            sb.accept(s, chunk, input_finished=pos[s] >= waves[s].size)
        sb.advance(live)
        for s in live:
            fin = pos[s] >= waves[s].size
            F = sb.num_frames_ready(s)
            n_out_total = (F + sub - 1) // sub
            k = chunks_done[s]
            while F > 0 and ((k * C_ < n_out_total * sub) if fin else ((k + 1) * C_ + R <= F)):
                k += 1
            if k > chunks_done[s]:
                iv_ready = F if fin else max(0, F - info.splice_right)
                if iv_ready > iv_done[s]:
                    calls[s].append(iv_ready); iv_done[s] = iv_ready
                src = len(calls[s]) - 1
                slot_first = -((L + C_ - 1) // C_)
                last = (k * C_ + R - 1) // C_ - slot_first
                slot_src[s] += [src] * (last + 1 - len(slot_src[s]))
                chunks_done[s] = k
            if fin:
                done[s] = True
    sb.finalize([0, 1])
    for s in range(S):
        assert feats[s].shape[0] == sb.num_frames_ready(s)
        first, slots = sb.ivector_slots(s)
        assert first == -((L + C_ - 1) // C_) and slots.shape[0] == len(slot_src[s])
        want_iv, want_state = orc.ivector_extract_streaming(info, feats[s], calls[s])
        for j, src in enumerate(slot_src[s]):
            w = want_iv[src] if src >= 0 else np.zeros(16, np.float32)
            np.testing.assert_allclose(slots[j], w, rtol=0, atol=1e-4 * max(1.0, np.abs(w).max()), err_msg="stream %d slot %d" % (s, j))
        assert len(set(slot_src[s])) >= 4                              # the estimate did move between chunks
        ll = N.ForwardSlots(feats[s], slots, first, C_)[0]
        off = decoder.LatticeFasterDecoder(G, cfg, abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512))
        off.Decode(ll)
        got = sb.raw_lattice(s)
        assert lattices_equal(got, off.GetRawLattice()), (s, lattice_diff(got, off.GetRawLattice()))
        # the network really saw different i-vectors over time, close to the oracle's evaluation of the same slots
        np.testing.assert_allclose(ll, orc.nnet_forward_slots(m, feats[s], slots, first, C_), rtol=0, atol=2e-3)
        st = sb.adaptation_state(s, max_remembered_frames=1e9)
        np.testing.assert_allclose(st, want_state, rtol=1e-8, atol=1e-8 * np.abs(want_state).max())
    # the speaker's next utterance starts from that state
    st0 = sb.adaptation_state(0, max_remembered_frames=100.0)
    sb.start([0], states=[st0])
    sb.accept(0, waves[1], input_finished=True)
    sb.advance([0])
    sb.finalize([0])
    _, slots2 = sb.ivector_slots(0)
    want2, _ = orc.ivector_extract_streaming(info, feats[1], [feats[1].shape[0]], state=st0)
    np.testing.assert_allclose(slots2[-1], want2[0], rtol=0, atol=1e-4 * max(1.0, np.abs(want2[0]).max()))


def test_online2_wav_nnet3_latgen_faster_tool(tmp_path):
    """final.mdl with an i-vector input + online.conf (mfcc config, i-vector extraction config) + spk2utt + wav.scp ->
    CompactLattice archive.  Two speakers; streams are independent, so decoding them one at a time or together gives
    the same archive; the first utterance equals a direct StreamBatch run; the second utterance of a speaker starts
    from the first one's adaptation state (it differs from a fresh start)."""
    import subprocess
    import sys
    import wave
    import os
    from kaldi_amd import io as kio
    from kaldi_amd import ivector, latbin
    from tests.mdl_writer import write_mdl
    g = synth.make_hclg(num_units=25, vocab=60, n_hist=12, seed=6)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=16, seed=12, output_scale=3.0)
    write_mdl(tmp_path / "final.mdl", m, num_units=25)
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    waves = {"a1": 2.1, "a2": 1.4, "b1": 1.8}
    waves = {k: np.round(synth.make_wave(d, seed=90 + i)).astype(np.float32) for i, (k, d) in enumerate(waves.items())}
    op = abi.mfcc_opts_hires()
    allf = np.concatenate([feat.Mfcc(op).ComputeFeatures(w) for w in waves.values()])
    info = ivector.make_synthetic(num_gauss=64, ivector_dim=16, seed=9, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=10.0)
    iconf = ivector.write_config_dir(tmp_path / "ivector_extractor", info)
    (tmp_path / "mfcc.conf").write_text("--use-energy=false\n--num-mel-bins=40\n--num-ceps=40\n--low-freq=20\n--high-freq=-400\n")
    (tmp_path / "online.conf").write_text("--feature-type=mfcc\n--mfcc-config=%s\n--ivector-extraction-config=%s\n--endpoint.silence-phones=1:2\n" %
                                          (tmp_path / "mfcc.conf", iconf))
    with open(tmp_path / "wav.scp", "w") as scp:
        for k, w in waves.items():
            with wave.open(str(tmp_path / (k + ".wav")), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(w.astype("<i2").tobytes())
            scp.write("%s %s\n" % (k, tmp_path / (k + ".wav")))
    (tmp_path / "spk2utt").write_text("spkA a1 a2\nspkB b1\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, root + "/tools/online2_wav_nnet3_latgen_faster.py", "--config=%s" % (tmp_path / "online.conf"), "--beam=15",
            "--max-active=7000", "--lattice-beam=8", "--acoustic-scale=1.0", "--frame-subsampling-factor=3", "--max-seconds=3",
            "--chunk-length=0.18"]
    outs = []
    for batch in (1, 2):
        lat = tmp_path / ("lat%d.ark" % batch)
        r = subprocess.run(base + ["--batch=%d" % batch, str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst"), "ark:%s" % (tmp_path / "spk2utt"),
                                   "scp:%s" % (tmp_path / "wav.scp"), "ark:%s" % lat], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Decoded 3 utterances, 0 with errors." in r.stderr
        outs.append(open(lat, "rb").read())
    assert outs[0] == outs[1]
    got = {k: latbin.best_path(l)[0] for k, l in latbin.read_lattices("ark:%s" % (tmp_path / "lat1.ark"))}
    assert list(got) == ["a1", "a2", "b1"]
    # --do-endpointing: the decision is per stream (one traceback launch per tick for the whole batch), so the
    # batch size does not matter; with most phones called silence some utterance ends before its audio does
    frames_full = {k: len(latbin.best_path(l)[1]) for k, l in latbin.read_lattices("ark:%s" % (tmp_path / "lat1.ark"))}
    ends = []
    for batch in (1, 3):
        lat = tmp_path / ("lat_ep%d.ark" % batch)
        r = subprocess.run(base + ["--batch=%d" % batch, "--do-endpointing=true", "--endpoint.silence-phones=" + ":".join(str(p) for p in range(1, 21)),
                                   "--endpoint.rule3.min-trailing-silence=0.06", "--endpoint.rule3.max-relative-cost=inf",
                                   str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst"), "ark:%s" % (tmp_path / "spk2utt"),
                                   "scp:%s" % (tmp_path / "wav.scp"), "ark:%s" % lat], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Decoded 3 utterances, 0 with errors." in r.stderr
        ends.append({k: len(latbin.best_path(l)[1]) for k, l in latbin.read_lattices("ark:%s" % lat)})
    assert ends[0] == ends[1]
    assert all(ends[0][k] <= frames_full[k] for k in frames_full) and any(ends[0][k] < frames_full[k] for k in frames_full), (ends[0], frames_full)
    # --ivector-silence-weighting.*: per stream again (same archive whatever the batch size); the statistics change, so
    # the lattices' acoustic costs differ from the unweighted run's
    sw_outs = []
    for batch in (1, 3):
        lat = tmp_path / ("lat_sw%d.ark" % batch)
        r = subprocess.run(base + ["--batch=%d" % batch, "--ivector-silence-weighting.silence-phones=" + ":".join(str(p) for p in range(1, 26, 2)),
                                   "--ivector-silence-weighting.silence-weight=0.001", "--ivector-silence-weighting.max-state-duration=5",
                                   str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst"), "ark:%s" % (tmp_path / "spk2utt"),
                                   "scp:%s" % (tmp_path / "wav.scp"), "ark:%s" % lat], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Decoded 3 utterances, 0 with errors." in r.stderr
        sw_outs.append(open(lat, "rb").read())
    assert sw_outs[0] == sw_outs[1] and sw_outs[0] != outs[0]
    # direct run of the first utterance, same chunking
    g.tid2pdf = np.concatenate([[-1], np.stack([2 * np.arange(25) + 1, 2 * np.arange(25)], 1).reshape(-1)]).astype(np.int32)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    ie = ivector.IvectorExtractor(ivector.IvectorExtractionInfo.from_config(iconf))
    sb = online.StreamBatch(op, N, G, abi.decoder_config_recipe(), 1, max_seconds=3.0)
    sb.set_ivector_extractor(ie, 20)

    def run(w, state=None):
        sb.start([0], states=None if state is None else [state])
        step = int(0.18 * 16000)
        for i in range(0, w.size, step):
            sb.accept(0, w[i:i + step], input_finished=i + step >= w.size)
            sb.advance([0])
        sb.finalize([0])
        return sb.best_path(0)["words"].tolist(), sb.adaptation_state(0, 1000.0), sb.ivector_slots(0)[1]
    w1, st1, _ = run(waves["a1"])
    assert got["a1"] == w1
    w2, _, slots_adapted = run(waves["a2"], st1)
    assert got["a2"] == w2
    _, _, slots_fresh = run(waves["a2"])
    assert np.abs(slots_adapted[-1] - slots_fresh[-1]).max() > 1e-3


def test_endpointing_on_the_device_equals_the_oracle():
    """EndpointDetected / TrailingSilenceLength (online2/online-endpoint.cc:71-121).  Single stream: after every
    chunk the device's trailing-silence count and decision equal the oracle's (its decoder fed the same rows: best
    path without final-probs, walked back; FinalRelativeCost; the five rules).  Batch: one launch for all streams
    gives what each stream's own partial best path gives."""
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    num_tids = len(g.tid2pdf) - 1
    tid2phone = np.concatenate([[0], (np.arange(num_tids) // 2) + 1]).astype(np.int32)   # self-loop + forward tid per unit
    n_phones = int(tid2phone.max())
    sil = [p for p in range(1, n_phones + 1) if p % 3 != 0]            # 2/3 of the units count as silence
    ep = online.OnlineEndpointConfig()
    ep.rule2.min_trailing_silence = 0.09; ep.rule2.max_relative_cost = 30.0      # reachable within 3 s of noise
    ep.rule3.min_trailing_silence = 0.06; ep.rule3.max_relative_cost = float("inf")
    rules = [[float(r.must_contain_nonsilence), r.min_trailing_silence, r.max_relative_cost, r.min_utterance_length]
             for r in (ep.rule1, ep.rule2, ep.rule3, ep.rule4, ep.rule5)]
    sz = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512)
    w = synth.make_wave(2.9, seed=9)
    s = online.SingleUtteranceNnet3Decoder(op, N, G, cfg, sizes=sz)
    s.record_loglikes()
    assert not s.EndpointDetected(ep, tid2phone, sil)                  # nothing decoded yet (:110)
    seen, fired = set(), 0
    for i in range(0, w.size, CHUNK):
        s.AcceptWaveform(16000, w[i:i + CHUNK])
        if not s.AdvanceDecoding():
            continue
        o = orc.Decoder(g, cfg, 1)
        o.InitDecoding()
        o.AdvanceDecoding(s.loglikes())
        lat = o.GetRawLattice()
        lat.final[:] = np.where(lat.frame == lat.num_frames, 0.0, np.inf).astype(np.float32)   # use_final_probs = false
        want_sil = orc.trailing_silence_length(lat.best_path()["alignment"], tid2phone, sil)
        got_sil = s.TrailingSilenceLength(tid2phone, sil)
        assert got_sil == want_sil, (i, got_sil, want_sil)
        want = orc.endpoint_detected(rules, o.NumFramesDecoded(), want_sil, 0.03, o.FinalRelativeCost())
        assert s.EndpointDetected(ep, tid2phone, sil) == want
        seen.add(want_sil); fired += int(want)
    assert len(seen) >= 3 and 0 < fired                                # the counts vary and some rule fires
    s.InputFinished(); s.AdvanceDecoding(); s.FinalizeDecoding()
    with pytest.raises(Exception):                                     # BestPathEnd: finalized && !use_final_probs is an error
        s.TrailingSilenceLength(tid2phone, sil)

    S = 4
    sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=6.0, sizes=abi.DecoderSizes(S, 1 << 14, 1 << 19, 1 << 20, 512))
    waves = [synth.make_wave(d, seed=40 + k) for k, d in enumerate([2.1, 1.2, 2.6, 0.9])]
    sb.start(np.arange(S))
    pos = [0] * S
    rng = np.random.default_rng(3)
    checked = 0
    while any(pos[k] < waves[k].size for k in range(S)):
        live = [k for k in range(S) if pos[k] < waves[k].size]
        for k in live:
            n = int(rng.integers(1500, 5000))
            sb.accept(k, waves[k][pos[k]:pos[k] + n], input_finished=False)
            pos[k] += n
        nd = sb.advance(live)
        cand = [k for k, d in zip(live, nd) if d > 0]
        if not cand:
            continue
        flags, sil_frames = sb.endpoint_detected(ep, cand, tid2phone, sil)
        for use_final in (False, True):                                 # all streams' partial results in one launch
            many = sb.partial_best_paths(cand, use_final_probs=use_final)
            for k, got in zip(cand, many):
                one = sb.partial_best_path(k, use_final_probs=use_final)
                assert got["alignment"].tolist() == one["alignment"].tolist() and got["words"].tolist() == one["words"].tolist()
                assert got["graph_cost"] == one["graph_cost"] and got["acoustic_cost"] == one["acoustic_cost"]
        for k, f, t in zip(cand, flags, sil_frames):
            bp = sb.partial_best_path(k, use_final_probs=False)
            want_sil = orc.trailing_silence_length(bp["alignment"], tid2phone, sil)
            assert t == want_sil
            frc = decoder.lib().kamd_decoder_final_relative_cost(sb.dec._dec, k)
            assert bool(f) == orc.endpoint_detected(rules, len(bp["alignment"]), want_sil, 0.03, frc)
            checked += 1
    assert checked >= 8


def test_incremental_partial_best_paths_equal_the_full_walk():
    """kamd_decoder_partial_best_paths_incremental keeps every stream's previous answer on the device and walks back only
    to the first frame whose best-path token is unchanged.  After every tick -- also when a stream was not asked for a
    while, with arena compaction at a low threshold, and across a restart of a stream with another utterance -- it
    returns what the full walk returns: alignment, words and both costs, bit for bit."""
    g = synth.make_hclg(num_units=25, vocab=60, n_hist=12, seed=6)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    S = 5
    sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=8.0, sizes=abi.DecoderSizes(S, 1 << 14, S * 60000, S * 120000, 512))
    sb.set_compaction(0.05)                                   # (3000 tokens: a stream compacts every few ticks)
    rng = np.random.default_rng(17)
    waves = [synth.make_wave(d, seed=70 + k) for k, d in enumerate([3.1, 1.4, 4.2, 2.0, 0.8])]
    again = [synth.make_wave(d, seed=170 + k) for k, d in enumerate([1.1, 2.3, 0.9, 1.7, 2.9])]
    sb.start(np.arange(S))
    pos, second = [0] * S, [False] * S
    compared = shorter_walks = 0
    for tick in range(400):
        live = [k for k in range(S) if pos[k] < waves[k].size]
        if not live:
            break
        for k in live:
            n = int(rng.integers(1200, 6000))
            sb.accept(k, waves[k][pos[k]:pos[k] + n], input_finished=False)
            pos[k] += n
        nd = sb.advance(live)
        cand = [k for k, d in zip(live, nd) if d > 0 and rng.random() < 0.8]       # (a stream is skipped now and then)
        if cand:
            inc = sb.partial_best_paths(cand, incremental=True)
            full = sb.partial_best_paths(cand)
            for k, a, b in zip(cand, inc, full):
                assert (a is None) == (b is None)
                if a is None:
                    continue
                assert a["alignment"].tolist() == b["alignment"].tolist() and a["words"].tolist() == b["words"].tolist(), (tick, k)
                assert a["graph_cost"] == b["graph_cost"] and a["acoustic_cost"] == b["acoustic_cost"], (tick, k)
                compared += 1
        for k in live:                                                            # a finished stream starts another utterance once
            if pos[k] >= waves[k].size and not second[k]:
                sb.finalize([k])
                sb.start([k])
                waves[k], pos[k], second[k] = again[k], 0, True
    assert compared >= 40 and sb.num_compactions() > 0


def test_one_overflowing_stream_does_not_stop_the_others():
    """A stream that outgrows its lane's arena fails alone: the tick reports it, the other streams of the tick have
    advanced and keep decoding to the offline result, the failed stream refuses further ticks until it is restarted,
    and a bad entry in a tick's stream list changes nothing (validation comes before any state change)."""
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    waves = [synth.make_wave(d, seed=90 + i) for i, d in enumerate((4.0, 0.9, 1.1))]
    big = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512)

    def offline(w):
        d = decoder.LatticeFasterDecoder(G, cfg, big)
        d.Decode(N.Forward(feat.Mfcc(op).ComputeFeatures(w)))
        return d.GetRawLattice(), d.counters()

    # an arena that holds the two short utterances but not the long one
    need = [int(offline(w)[1][5]) for w in waves]
    cap = max(need[1], need[2]) + 64
    assert need[0] > 2 * cap
    S = 3
    sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=5.0, sizes=abi.DecoderSizes(S, 1 << 14, cap, 4 * cap, 512))
    sb.set_compaction(0.0)                                   # (with the default compaction the small lane would not overflow)
    sb.start([0, 1, 2])
    with pytest.raises(Exception):
        sb.advance([0, 0])                                   # duplicates are rejected ...
    with pytest.raises(Exception):
        sb.advance([1, 7])                                   # ... and so is an unknown stream, before anything happens
    CH = 4000
    pos, failed_at = [0, 0, 0], None
    for tick in range(40):
        live = [s for s in range(S) if pos[s] < waves[s].size and sb.status([s])[0] == 0]
        if not live:
            break
        pieces = []
        for s in live:
            pieces.append(waves[s][pos[s]:pos[s] + CH]); pos[s] += pieces[-1].size
        sb.accept_many(live, pieces, [pos[s] >= waves[s].size for s in live])
        try:
            sb.advance(live)
        except Exception as e:
            assert "capacity" in str(e)
            failed_at = tick
    assert failed_at is not None
    st = sb.status([0, 1, 2])
    assert st[0] != 0 and st[1] == 0 and st[2] == 0
    with pytest.raises(Exception):
        sb.advance([0])                                      # out of service until restarted
    sb.finalize([1, 2])
    for s in (1, 2):
        assert lattices_equal(sb.raw_lattice(s), offline(waves[s])[0])
    sb.start([0])                                            # restart the failed slot with an utterance that fits
    assert sb.status([0])[0] == 0
    sb.accept(0, waves[1], input_finished=True)
    sb.advance([0])
    sb.finalize([0])
    assert lattices_equal(sb.raw_lattice(0), offline(waves[1])[0])


def test_frame_tracebacks_equal_the_oracle_traceback():
    """kamd_decoder_frame_tracebacks (what OnlineSilenceWeighting::ComputeCurrentTraceback reads off the decoder): after
    every chunk, (transition-id, token) per decoded frame == the oracle decoder's best path without final-probs over
    the same rows, with a token named by its HCLG state."""
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    w = synth.make_wave(2.4, seed=21)
    s = online.SingleUtteranceNnet3Decoder(op, N, G, cfg, sizes=abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512))
    s.record_loglikes()
    prev, rewrites = None, 0
    remembered, cut = {}, 0
    for i in range(0, w.size, CHUNK):
        s.AcceptWaveform(16000, w[i:i + CHUNK])
        if not s.AdvanceDecoding():
            continue
        (tids, toks), = decoder.frame_tracebacks(s.decoder._dec, [s.decoder.lane])
        o = orc.Decoder(g, cfg, 1)
        o.InitDecoding()
        o.AdvanceDecoding(s.loglikes())
        lat = o.GetRawLattice()
        lat.final[:] = np.where(lat.frame == lat.num_frames, 0.0, np.inf).astype(np.float32)   # use_final_probs = false
        want = lat.best_path_frames()
        assert len(want) == s.NumFramesDecoded() == tids.size
        assert [(int(a), int(b)) for a, b in zip(tids, toks)] == want
        # the incremental form: the same walk, cut at the first frame whose token the previous incremental call reported
        (itids, itoks, n_dec), = decoder.frame_tracebacks(s.decoder._dec, [s.decoder.lane], incremental=True)
        assert n_dec == len(want)
        expect = []
        for k, (t, tok) in enumerate(want):
            frame = n_dec - 1 - k
            expect.append((t, tok))
            if frame in remembered and remembered[frame] == tok:
                break
        assert [(int(a), int(b)) for a, b in zip(itids, itoks)] == expect
        cut += int(len(expect) < len(want))
        for k, (t, tok) in enumerate(expect):
            remembered[n_dec - 1 - k] = tok
        if prev is not None:
            old = prev[::-1]
            new = want[::-1][:len(old)]
            rewrites += int(old != new)
        prev = want
    assert rewrites > 0                     # the best path did change its mind about earlier frames along the way
    assert cut > 0                          # ... and most ticks the incremental walk stopped early


def test_streams_with_silence_weighting():
    """--ivector-silence-weighting.* in the streaming batch (online2-wav-nnet3-latgen-faster.cc:214-216, 258-266).  Per tick and
    stream the device's own traceback (read with frame_tracebacks right before the tick, the data the tick itself uses) drives
    the ORACLE's OnlineSilenceWeighting + delta-weight queue; the oracle's weighted OnlineIvectorFeature then has to reproduce
    every i-vector the device put into the network's slots, the lattice has to equal an offline decode with exactly those slots,
    and the adaptation state the oracle's.  The weighting must really act: negative deltas occur, and the i-vectors differ from
    the unweighted run's."""
    from kaldi_amd import ivector
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=16, seed=12, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    waves = [synth.make_wave(d, seed=60 + i) for i, d in enumerate((3.0, 1.7))]
    feats = [feat.Mfcc(op).ComputeFeatures(w) for w in waves]
    allf = np.concatenate(feats)
    info = ivector.make_synthetic(num_gauss=64, ivector_dim=16, seed=9, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=10.0)
    ie = ivector.IvectorExtractor(info)
    num_tids = len(g.tid2pdf) - 1
    tid2phone = np.concatenate([[0], (np.arange(num_tids) // 2) + 1]).astype(np.int32)
    sil = [p for p in range(1, int(tid2phone.max()) + 1) if p % 2 == 0]
    swc = online.OnlineSilenceWeightingConfig(":".join(map(str, sil)), 0.1, 4.0)
    assert swc.Active() and swc.silence_phones() == sil
    S, sub = 2, m.subsampling
    L, R = N.Context()
    sizes = abi.DecoderSizes(S, 1 << 14, 1 << 19, 1 << 20, 512)

    def run(weighted):
        sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=5.0, sizes=sizes)
        sb.set_ivector_extractor(ie, 20)
        if weighted:
            sb.set_silence_weighting(swc, tid2phone)
        C_ = sb.frames_per_chunk
        sb.start([0, 1])
        step = [int(0.18 * 16000), int(0.31 * 16000)]
        pos, done = [0, 0], [False, False]
        chunks_done, iv_done = [0, 0], [0, 0]
        calls, lists, slot_src = [[], []], [[], []], [[], []]
        osw = [orc.OnlineSilenceWeighting(tid2phone, sil, swc.silence_weight, swc.max_state_duration, sub) for _ in range(S)]
        oq = [orc.DeltaWeightQueue() for _ in range(S)]
        n_neg = 0
        while not all(done):
            live = [s for s in range(S) if not done[s]]
            for s in live:
                chunk = waves[s][pos[s]:pos[s] + step[s]]
                pos[s] += chunk.size
                sb.accept(s, chunk, input_finished=pos[s] >= waves[s].size)
            tb = sb.frame_tracebacks(live)                      # the decoder's state as the tick will find it
            n_dec_before = {s: (0 if tb[i] is None else tb[i][0].size) for i, s in enumerate(live)}
            sb.advance(live)
            for i, s in enumerate(live):
                fin = pos[s] >= waves[s].size
                F = sb.num_frames_ready(s)
                iv_ready = F if fin else max(0, F - info.splice_right)
                if weighted:
                    if n_dec_before[s] > 0:
                        osw[s].ComputeCurrentTraceback(n_dec_before[s], zip(tb[i][0].tolist(), tb[i][1].tolist()))
                    oq[s].UpdateFrameWeights(osw[s].GetDeltaWeights(iv_ready))
                n_out_total = (F + sub - 1) // sub
                k = chunks_done[s]
                while F > 0 and ((k * C_ < n_out_total * sub) if fin else ((k + 1) * C_ + R <= F)):
                    k += 1
                if k > chunks_done[s]:
                    if iv_ready > iv_done[s]:
                        calls[s].append(iv_ready); iv_done[s] = iv_ready
                        if weighted:
                            lists[s].append(oq[s].pop_until(iv_ready - 1))
                            n_neg += sum(1 for _, x in lists[s][-1] if x < 0)
                    src = len(calls[s]) - 1
                    slot_first = -((L + C_ - 1) // C_)
                    last = (k * C_ + R - 1) // C_ - slot_first
                    slot_src[s] += [src] * (last + 1 - len(slot_src[s]))
                    chunks_done[s] = k
                if fin:
                    done[s] = True
        sb.finalize([0, 1])
        return sb, C_, calls, lists, slot_src, n_neg

    sb, C_, calls, lists, slot_src, n_neg = run(True)
    assert n_neg > 0
    plain, _, calls0, _, slot_src0, _ = run(False)
    for s in range(S):
        first, slots = sb.ivector_slots(s)
        want_iv, want_state = orc.ivector_extract_streaming_weighted(info, feats[s], calls[s], lists[s])
        assert slots.shape[0] == len(slot_src[s])
        for j, src in enumerate(slot_src[s]):
            w = want_iv[src] if src >= 0 else np.zeros(16, np.float32)
            np.testing.assert_allclose(slots[j], w, rtol=0, atol=1e-4 * max(1.0, np.abs(w).max()), err_msg="stream %d slot %d" % (s, j))
        ll = N.ForwardSlots(feats[s], slots, first, C_)[0]
        off = decoder.LatticeFasterDecoder(G, cfg, abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512))
        off.Decode(ll)
        assert lattices_equal(sb.raw_lattice(s), off.GetRawLattice())
        st = sb.adaptation_state(s, max_remembered_frames=1e9)
        np.testing.assert_allclose(st, want_state, rtol=1e-7, atol=1e-7 * np.abs(want_state).max())
        _, slots0 = plain.ivector_slots(s)
        assert calls0[s] == calls[s] and slot_src0[s] == slot_src[s]               # same schedule
        assert np.abs(slots0 - slots).max() > 1e-2                                 # different statistics
        # the unweighted run is the weighted machinery with every frame at weight 1, once
        unit, _ = orc.ivector_extract_streaming_weighted(info, feats[s], calls[s], [[(t, 1.0) for t in range(a, b)]
                                                                                    for a, b in zip([0] + calls[s][:-1], calls[s])])
        np.testing.assert_allclose(unit, orc.ivector_extract_streaming(info, feats[s], calls[s])[0], rtol=0, atol=1e-6)


def test_shared_extractor_between_stream_batches_and_offline_calls():
    """Round-2 advisor finding: the silence-weighted update re-reads the LDA rows of frames it processed on EARLIER ticks;
    those rows lived in workspaces of the EXTRACTOR, which a second stream batch sharing it, or an offline extraction
    between two ticks, overwrote.  Each stream batch now owns its rows (kamd_ivector_workspace, bound per call): two batches
    on one extractor, their ticks interleaved with each other and with offline extractions, must give what each gives
    alone -- i-vector slots, adaptation states and lattices bit for bit."""
    from kaldi_amd import ivector
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=16, seed=12, output_scale=3.0)
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    waves = [synth.make_wave(d, seed=70 + i) for i, d in enumerate((2.6, 1.9))]
    feats = [feat.Mfcc(op).ComputeFeatures(w) for w in waves]
    allf = np.concatenate(feats)
    info = ivector.make_synthetic(num_gauss=64, ivector_dim=16, seed=9, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=10.0)
    ie = ivector.IvectorExtractor(info)
    num_tids = len(g.tid2pdf) - 1
    tid2phone = np.concatenate([[0], (np.arange(num_tids) // 2) + 1]).astype(np.int32)
    sil = [p for p in range(1, int(tid2phone.max()) + 1) if p % 2 == 0]
    swc = online.OnlineSilenceWeightingConfig(":".join(map(str, sil)), 0.1, 4.0)
    sizes = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512)
    step = [int(0.21 * 16000), int(0.17 * 16000)]

    def make():
        sb = online.StreamBatch(op, N, G, cfg, 1, max_seconds=4.0, sizes=sizes)
        sb.set_ivector_extractor(ie, 20)
        sb.set_silence_weighting(swc, tid2phone)
        sb.start([0])
        return sb

    def tick(sb, w, pos):
        chunk = waves[w][pos:pos + step[w]]
        sb.accept(0, chunk, input_finished=pos + chunk.size >= waves[w].size)
        sb.advance([0])
        return pos + chunk.size

    alone = []
    for w in range(2):
        sb, pos = make(), 0
        while pos < waves[w].size:
            pos = tick(sb, w, pos)
        sb.finalize([0])
        alone.append((sb.ivector_slots(0), sb.adaptation_state(0, max_remembered_frames=1e9), sb.raw_lattice(0)))
    sbs, pos, k = [make(), make()], [0, 0], 0
    while any(pos[w] < waves[w].size for w in range(2)):
        for w in range(2):
            if pos[w] < waves[w].size:
                pos[w] = tick(sbs[w], w, pos[w])
        ie.extract_online(feats[k % 2])                # an offline extraction on the same extractor between the ticks
        k += 1
    for w in range(2):
        sbs[w].finalize([0])
        (first, slots), state, lat = alone[w]
        first2, slots2 = sbs[w].ivector_slots(0)
        assert first2 == first
        np.testing.assert_array_equal(slots2, slots)
        np.testing.assert_array_equal(sbs[w].adaptation_state(0, max_remembered_frames=1e9), state)
        assert lattices_equal(sbs[w].raw_lattice(0), lat)


@pytest.mark.parametrize("mode", [1, 2])
def test_raw_lattice_of_a_live_decoder_after_every_chunk(mode):
    """GetRawLattice before FinalizeDecoding (lattice-faster-decoder.cc:113-196 with !decoding_finalized_; what
    SingleUtteranceNnet3Decoder::GetLattice(end_of_utterance = false) reads, online-nnet3-decoding.cc:66-79): after every
    chunk of AdvanceDecoding the device's live lattice -- every token and link, final costs computed on the spot, both
    values of use_final_probs -- equals the oracle's in the same search mode bit for bit; reading it changes nothing
    (the final lattice equals the uninterrupted decode's); GetRawLatticePruned keeps exactly the states and arcs on paths
    within the beam (checked against a brute-force forward-backward); GetLattice = its pruned determinization."""
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=10, seed=6)
    cfg = abi.decoder_config_recipe()
    cfg.max_active, cfg.min_active = 60, 20               # max-active binds: the two search modes really differ
    G = decoder.Graph(g)
    ll, words, _ = synth.sample_utterance(g, n_words=7, seed=31, peak=2.5, noise=1.2)
    sizes = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512)
    d = decoder.LatticeFasterDecoder(G, cfg, sizes)
    d.SetSearchMode(mode)
    o = orc.Decoder(g, cfg, mode)
    d.InitDecoding(); o.InitDecoding()
    pos, saw_final_difference = 0, False
    for n in (1, 2, 5, 3, 9, 4, 1000):
        chunk = ll[pos:pos + n]
        if chunk.shape[0] == 0:
            break
        pos += chunk.shape[0]
        d.AdvanceDecoding(decoder.DeviceMatrix(chunk)); o.AdvanceDecoding(chunk)
        for ufp in (True, False):
            got, want = d.GetRawLattice(use_final_probs=ufp), o.GetRawLattice(use_final_probs=ufp)
            assert lattices_equal(got, want), (pos, ufp, lattice_diff(got, want))
        a, b = d.GetRawLattice(True), d.GetRawLattice(False)
        saw_final_difference |= not np.array_equal(a.final, b.final)
        # GetRawLatticePruned: the states / arcs on paths within the beam, by a brute-force forward-backward on the live lattice
        beam = 3.0
        pr = d.GetRawLatticePruned(False, beam)
        n_st = b.frame.size
        w = b.arcs["graph_cost"].astype(np.float64) + b.arcs["acoustic_cost"].astype(np.float64)
        fwd, bwd = np.full(n_st, np.inf), np.where(np.isfinite(b.final), b.final.astype(np.float64), np.inf)
        fwd[b.start] = 0.0
        for _ in range(n_st):                             # Bellman-Ford sweeps (a small acyclic lattice)
            f2, b2 = fwd.copy(), bwd.copy()
            np.minimum.at(f2, b.arcs["dst"], fwd[b.arcs["src"]] + w)
            np.minimum.at(b2, b.arcs["src"], bwd[b.arcs["dst"]] + w)
            if np.array_equal(f2, fwd) and np.array_equal(b2, bwd):
                break
            fwd, bwd = f2, b2
        keep = fwd + bwd <= bwd[b.start] + beam
        assert pr.frame.size == int(keep.sum()) and np.array_equal(pr.hclg, b.hclg[keep]) and np.array_equal(pr.frame, b.frame[keep])
        assert pr.arcs.size == int((fwd[b.arcs["src"]] + w + bwd[b.arcs["dst"]] <= bwd[b.start] + beam).sum())
    assert saw_final_difference
    d.FinalizeDecoding(); o.FinalizeDecoding()
    assert lattices_equal(d.GetRawLattice(), o.GetRawLattice())
    with pytest.raises(Exception):
        d.GetRawLattice(use_final_probs=False)            # :117-120: not after FinalizeDecoding
    ref = decoder.LatticeFasterDecoder(G, cfg, sizes)
    ref.SetSearchMode(mode)
    ref.Decode(decoder.DeviceMatrix(ll))
    assert lattices_equal(d.GetRawLattice(), ref.GetRawLattice())
