"""The C++ host mirror (include/kaldi_amd.hpp) driven like Kaldi code, compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from kaldi_amd import abi, synth
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_cxx(tmp):
    exe = os.path.join(tmp, "host_api_test")
    lib = os.path.join(ROOT, "kaldi_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "host_api_test.cc"), "-o", exe,
                           "-L", lib, "-lkaldi_amd", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def vec(f, a):
    a = np.ascontiguousarray(a)
    f.write(np.int64(a.size).tobytes())
    f.write(a.tobytes())


def test_cxx_host_api(tmp_path):
    g = synth.make_hclg(num_units=40, vocab=150, n_hist=25, seed=3)
    ll, words, _ = synth.sample_utterance(g, n_words=6, seed=5, peak=7.0)
    fx = tmp_path / "fixture.bin"
    with open(fx, "wb") as f:
        vec(f, np.asarray([g.num_states, g.start, ll.shape[0], ll.shape[1]], np.int64))
        vec(f, g.arc_off.astype(np.int64))
        vec(f, g.arcs)
        vec(f, g.final.astype(np.float32))
        vec(f, g.tid2pdf.astype(np.int32))
        vec(f, ll)
    # second fixture: a small TDNN-F model over this graph's pdfs + a waveform, for the streaming mirror
    from kaldi_amd import decoder, feat, nnet
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    wave = synth.make_wave(2.3, seed=11)
    mfx = tmp_path / "model.bin"
    with open(mfx, "wb") as f:
        vec(f, np.asarray([len(m.layers), m.input_dim, m.subsampling], np.int64))
        for l in m.layers:
            offs = list(l.offsets) + [0] * (8 - len(l.offsets))
            vec(f, np.asarray([l.in_dim, l.out_dim, len(l.offsets)] + offs + [l.input_layer, l.ivector_dim, l.bypass_layer,
                               int(l.relu), int(l.log_softmax)], np.int32))
            vec(f, np.asarray([l.bypass_scale, l.post_scale], np.float32))
            for nm in ("W", "bias", "bn_scale", "bn_offset", "post_offset"):
                a = getattr(l, nm)
                vec(f, np.zeros(0, np.float32) if a is None else np.ascontiguousarray(a, np.float32))
        vec(f, wave.astype(np.float32))
    # third fixture: a small i-vector extractor and two utterances of one speaker
    from kaldi_amd import ivector
    info = ivector.make_synthetic(feat_dim=8, lda_dim=6, num_gauss=40, ivector_dim=10, seed=4, splice_left=2, splice_right=1, max_count=5.0,
                                  cmn_window=60, speaker_frames=40, global_frames=10)
    rng = np.random.default_rng(9)
    f1, f2 = [(rng.standard_normal((T, 8)) + 0.2).astype(np.float32) for T in (47, 95)]
    xfx = tmp_path / "ivector.bin"
    with open(xfx, "wb") as f:
        vec(f, np.asarray([8, 2, 1, info.lda.shape[0], info.lda.shape[1], 40, 10, info.ivector_period, info.num_gselect, info.num_cg_iters,
                           47, 95, 60, 40, 10], np.int32))
        for a in (info.lda, info.ubm_gconsts, info.ubm_means_invvars, info.ubm_inv_vars):
            vec(f, np.ascontiguousarray(a, np.float32))
        for a in (info.global_cmvn_stats, info.M, info.sigma_inv, np.asarray([info.prior_offset, info.min_post, info.posterior_scale, info.max_count])):
            vec(f, np.ascontiguousarray(a, np.float64))
        vec(f, f1); vec(f, f2)
    # an integer-id bigram-less LM over the graph's words for the const-ARPA mirror
    vocab = int(g.arcs["olabel"].max())
    lrng = np.random.default_rng(17)
    lp = -lrng.uniform(0.5, 3.0, vocab + 1)
    with open(tmp_path / "G.arpa", "w") as f:
        f.write("\\data\\\nngram 1=%d\nngram 2=%d\n\n\\1-grams:\n" % (vocab + 2, vocab))
        f.write("-99\t100001\t-0.3\n-1.1\t100002\n")
        for w in range(1, vocab + 1):
            f.write("%.4f\t%d\t%.4f\n" % (lp[w], w, -0.1 * (w % 5)))
        f.write("\n\\2-grams:\n")
        for w in range(1, vocab + 1):
            f.write("%.4f\t100001 %d\n" % (lp[w] * 0.5, w))
        f.write("\n\\end\\\n")
    exe = build_cxx(str(tmp_path))
    out = subprocess.check_output([exe, str(fx), str(tmp_path), str(mfx), str(xfx), "batch"], text=True, stderr=subprocess.DEVNULL).strip().splitlines()
    # OnlineStreamBatch with online i-vectors: the same two streams through the Python mirror
    from kaldi_amd import online
    N2, G2 = decoder.Nnet(m), decoder.Graph(g)
    cfgb = abi.decoder_config_recipe()
    sb = online.StreamBatch(abi.mfcc_opts_hires(), N2, G2, cfgb, 2, max_seconds=4.0, sizes=abi.DecoderSizes(2, 1 << 14, 1 << 18, 1 << 19, 1024))
    ieb = ivector.IvectorExtractor(info)
    sb.set_ivector_extractor(ieb, 20)
    sb.start([0, 1])
    wv = wave.astype(np.float32)
    lens, steps, pos = [wv.size, wv.size * 2 // 3], [2880, 4960], [0, 0]
    while pos[0] < lens[0] or pos[1] < lens[1]:
        live = []
        for s_ in range(2):
            if pos[s_] >= lens[s_]:
                continue
            n_ = min(steps[s_], lens[s_] - pos[s_])
            sb.accept(s_, wv[pos[s_]:pos[s_] + n_], input_finished=pos[s_] + n_ >= lens[s_])
            pos[s_] += n_; live.append(s_)
        sb.advance(live)
    sb.finalize([0, 1])
    batch_lines = [l for l in out if l.startswith("batch ")]
    assert len(batch_lines) == 2
    for s_ in range(2):
        bp = sb.best_path(s_)
        st_ = sb.adaptation_state(s_, max_remembered_frames=60.0)
        want_line = "batch stream=%d ok=1 frames=%d cost=%.9g count=%.9g lin1=%.9g words=%s" % (
            s_, len(bp["alignment"]), np.float32(bp["graph_cost"]) + np.float32(bp["acoustic_cost"]), st_[8], st_[2 * 9 + 55 + 1],
            ",".join(str(w) for w in bp["words"]))
        assert batch_lines[s_] == want_line
    out = [l for l in out if not l.startswith("batch ")]
    # NnetBatchDecoder (offline batch) against the Python pipeline
    from kaldi_amd import pipeline
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, cfgb, max_utts=2, max_seconds=4.0, sizes=abi.DecoderSizes(2, 1 << 14, 1 << 18, 1 << 19, 1024))
    res = pipe.decode([wv, wv[:wv.size * 2 // 3]])
    off_lines = [l for l in out if l.startswith("offline ")]
    assert len(off_lines) == 2
    for u_ in range(2):
        bp = res[u_]["best"]
        assert off_lines[u_] == "offline utt=%d ok=1 frames=%d cost=%.9g words=%s" % (
            u_, len(bp["alignment"]), np.float32(bp["graph_cost"]) + np.float32(bp["acoustic_cost"]), ",".join(str(w) for w in bp["words"]))
    out = [l for l in out if not l.startswith("offline ")]
    # ConstArpaLm / LatticeLmrescoreConstArpa through the C++ mirror == the Python mirror on the lattice the C++ side wrote
    from kaldi_amd import constarpa, latbin
    lm_ = constarpa.ConstArpaLm.read(tmp_path / "G.carpa")
    (_, lin), = list(latbin.read_lattices("ark:%s" % (tmp_path / "carpa_in.ark")))
    (_, lout), = list(latbin.read_lattices("ark:%s" % (tmp_path / "carpa_out.ark")))
    want_l = lm_.rescore(lin, 1.0)
    crow = [l for l in out if l.startswith("carpa ")][0]
    assert crow == "carpa ok=1 order=2 bos=100001 eos=100002 states=%d copy_states=0 p=%.9g" % (
        len(want_l.final), lm_.GetNgramLogprob(1, [100001]))
    assert latbin.compact_bytes(lout) == latbin.compact_bytes(want_l)
    bi, bo_ = latbin.best_path(lin), latbin.best_path(lout)
    assert bo_ is not None and bi is not None
    out = [l for l in out if not l.startswith("carpa ")]
    ie = ivector.IvectorExtractor(info)
    _, st = ie.extract_online(f1, return_state=True, max_remembered_frames=60.0)
    want_iv = ie.extract_online(f2, state=st)
    iv_line = [l for l in out if l.startswith("ivector ")][0].split()
    assert iv_line[1:3] == ["rows=%d" % want_iv.shape[0], "dim=10"]
    np.testing.assert_array_equal(np.asarray([float(x) for x in iv_line[3:]], np.float32), want_iv[-1])
    out = [l for l in out if not l.startswith("ivector ")]
    o = orc.Decoder(g, abi.decoder_config_recipe(), 1)
    o.Decode(ll)
    lat = o.GetRawLattice()
    bp = lat.best_path()
    want = "ok=1 frames=%d reached_final=1 states=%d arcs=%d graph=%.9g acoustic=%.9g words=%s" % (
        ll.shape[0], lat.frame.size, lat.arcs.size, bp["graph_cost"], bp["acoustic_cost"],
        ",".join(str(w) for w in bp["words"]))
    assert out[0] == "mapped " + want
    assert out[1] == "chunked " + want
    assert out[2] == "generic " + want
    assert out[-1] == "badconfig threw"
    # DecodeUtteranceLatticeFaster: words / alignment / lattice archives
    like = -(bp["graph_cost"] + bp["acoustic_cost"])
    assert out[3].startswith("wrapper ok=1") and out[4].startswith("wrapper ok=1")
    assert abs(float(out[3].split("like=")[1]) - like) < 1e-3 * abs(like)
    wl = open(tmp_path / "words.txt").read().splitlines()
    assert wl[0].split() == ["utt-det"] + [str(w) for w in words] and wl[1].split()[0] == "utt-raw"
    raw = open(tmp_path / "ali.ark", "rb").read()
    assert raw.startswith(b"utt-det \0B\x04") and int.from_bytes(raw[11:15], "little") == ll.shape[0]
    from kaldi_amd import io as kio2
    alis = dict(kio2.read_int32_vector_ark(tmp_path / "ali.ark"))
    assert alis["utt-det"].tolist() == bp["alignment"].tolist()
    from kaldi_amd import io as kio
    (key, st, fin, arcs), = list(kio.read_lattices(tmp_path / "lat.txt"))
    assert key == "utt-raw" and arcs.size == lat.arcs.size
    assert np.allclose(arcs["acoustic_cost"], lat.arcs["acoustic_cost"] / np.float32(0.5), rtol=1e-5)
    # streaming mirror == offline decode of the same waveform through the Python mirror
    off = decoder.LatticeFasterDecoder(decoder.Graph(g), abi.decoder_config_recipe())
    off.Decode(decoder.Nnet(m).Forward(feat.Mfcc(abi.mfcc_opts_hires()).ComputeFeatures(wave)))
    ob = off.GetBestPath()
    assert "live lattices=1" in out           # GetLattice(end_of_utterance = false) / GetRawLattice / GetRawLatticePruned on the live decoder
    srow = [l for l in out if l.startswith("streaming ")][0]
    assert srow.startswith("streaming ok=1 frames=%d partials=1 " % off.NumFramesDecoded())
    assert srow.endswith("words=" + ",".join(str(w) for w in ob["words"]))
    assert abs(float(srow.split("graph=")[1].split()[0]) - ob["graph_cost"]) < 1e-3
    # endpointing through the C++ mirror == the Python mirror on the same chunking
    from kaldi_amd import online
    ep = online.OnlineEndpointConfig()
    ep.rule3.min_trailing_silence = 0.03; ep.rule3.max_relative_cost = float("inf"); ep.rule2.must_contain_nonsilence = False
    tid2phone = np.concatenate([[0], np.arange(len(g.tid2pdf) - 1) // 2 + 1]).astype(np.int32)
    sil = list(range(1, int(tid2phone.max()) + 1))
    ep.rule2.max_relative_cost = float("inf")
    sd = online.SingleUtteranceNnet3Decoder(abi.mfcc_opts_hires(), decoder.Nnet(m), decoder.Graph(g), abi.decoder_config_recipe())
    flags, sils = [], []
    for i in range(0, wave.size, 2880):
        sd.AcceptWaveform(16000, wave[i:i + 2880])
        if i + 2880 >= wave.size:
            sd.InputFinished()
        sd.AdvanceDecoding()
        flags.append(int(sd.EndpointDetected(ep, tid2phone, sil)))
        sils.append(sd.TrailingSilenceLength(tid2phone, sil) if sd.NumFramesDecoded() else 0)
    erow = [l for l in out if l.startswith("endpoint ")][0]
    assert erow == "endpoint flags=%s silence=%s plain=%d%d" % (",".join(map(str, flags)), ",".join(map(str, sils)),
                                                               online.endpoint_detected(ep, 100, 17, 0.03, 1.9), online.endpoint_detected(ep, 100, 1, 0.03, 9.0))
    assert 0 < sum(flags) and len(set(sils)) > 1, (flags, sils)
    clat = open(tmp_path / "clat.ark", "rb").read()
    assert clat.startswith(b"utt-det ") and clat[8] == 214 and b"compactlattice44" in clat[:64]
    assert bp["words"].tolist() == words


def test_nnet3_latgen_faster_example_writes_the_python_tools_archive(tmp_path):
    """examples/nnet3_latgen_faster.cc -- final.mdl read by kamd_model_read, HCLG.fst by kamd_graph_read_openfst, the sets
    run by the hpp's NnetBatchDecoder -- against tools/nnet3_latgen_faster_batch.py on the same files: the lattice archive
    byte for byte (features + per-speaker i-vectors, compact lattices through an output pipe; waveforms in, raw lattices
    out), the reference's warnings for a zero-length utterance and a missing i-vector, words / alignments tables."""
    import struct
    import sys
    import wave
    from kaldi_amd import feat, latbin, nnet, table
    from kaldi_amd import io as kio
    from tests.mdl_writer import write_mdl
    lib = os.path.join(ROOT, "kaldi_amd", "lib")
    exe = str(tmp_path / "nnet3-latgen-faster-amd")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "nnet3_latgen_faster.cc"),
                           "-o", exe, "-L", lib, "-lkaldi_amd", "-lpthread", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"])
    g = synth.make_hclg(num_units=25, vocab=60, n_hist=12, seed=6)
    m_iv = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=10, seed=12, output_scale=3.0)
    m_plain = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, seed=12, output_scale=3.0)
    write_mdl(tmp_path / "final_iv.mdl", m_iv, num_units=25)
    write_mdl(tmp_path / "final.mdl", m_plain, num_units=25)
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    waves = [np.round(synth.make_wave(d, seed=50 + i)).astype(np.float32) for i, d in enumerate((1.2, 2.0, 0.8, 1.6, 1.1, 0.9))]
    mf = feat.Mfcc(abi.mfcc_opts_hires())
    with table.TableWriter("ark,scp:%s,%s" % (tmp_path / "feats.ark", tmp_path / "feats.scp"), "matrix") as w:
        for i, wv in enumerate(waves):
            w.write("utt%d" % i, mf.ComputeFeatures(wv))
            if i == 1:
                w.write("empty", np.zeros((0, 40), np.float32))
    rng = np.random.default_rng(3)
    with open(tmp_path / "ivectors.ark", "wb") as f:            # BaseFloatVectorWriter entries: key, "\0B", "FV ", size, data
        for spk in ("spkA", "spkB"):
            v = rng.standard_normal(10).astype(np.float32)
            f.write(spk.encode() + b" \0BFV \x04" + struct.pack("<i", 10) + v.tobytes())
    (tmp_path / "utt2spk").write_text("".join("utt%d %s\n" % (i, ("spkA", "spkB", "spkC")[i % 3]) for i in range(6)) + "empty spkA\n")
    with open(tmp_path / "wav.scp", "w") as scp:
        for i, wv in enumerate(waves):
            with wave.open(str(tmp_path / ("u%d.wav" % i)), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(wv.astype("<i2").tobytes())
            scp.write("utt%d %s\n" % (i, tmp_path / ("u%d.wav" % i)))
    (tmp_path / "words.txt").write_text("".join("w%d %d\n" % (k, k) for k in range(0, 61)))
    common = ["--beam=15", "--max-active=7000", "--lattice-beam=8", "--acoustic-scale=1.0", "--frame-subsampling-factor=3", "--search-mode=1"]
    fst_ = str(tmp_path / "HCLG.fst")
    iv = ["--ivectors=ark:%s" % (tmp_path / "ivectors.ark"), "--utt2spk=ark:%s" % (tmp_path / "utt2spk"), "--set-frames=250", "--num-threads=3",
          "--word-symbol-table=%s" % (tmp_path / "words.txt"), str(tmp_path / "final_iv.mdl"), fst_, "scp:%s" % (tmp_path / "feats.scp")]
    r = subprocess.run([exe] + common + iv + ["ark:| cat > %s" % (tmp_path / "cxx.lat"), "ark:%s" % (tmp_path / "words.ark"),
                                              "ark,t:%s" % (tmp_path / "ali.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    p = subprocess.run([sys.executable, ROOT + "/tools/nnet3_latgen_faster_batch.py"] + common + iv + ["ark:%s" % (tmp_path / "py.lat")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    assert open(tmp_path / "cxx.lat", "rb").read() == open(tmp_path / "py.lat", "rb").read()
    lats = list(latbin.read_lattices("ark:%s" % (tmp_path / "cxx.lat")))
    assert [k for k, _ in lats] == ["utt0", "utt1", "utt3", "utt4"]        # utt2 / utt5 are spkC's: no i-vector; input order over sets
    for text in (r.stderr, p.stderr):
        assert "Zero-length utterance: empty" in text and "No iVector available for utterance utt2" in text
        assert "Decoded 7 utterances, 3 with errors." in text
    assert sorted(l for l in r.stderr.splitlines() if l.startswith("utt")) == sorted(l for l in p.stderr.splitlines() if l.startswith("utt"))
    assert sorted(l for l in r.stderr.splitlines() if "Log-like per frame" in l) == sorted(l for l in p.stderr.splitlines() if "Log-like per frame" in l)
    words = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "words.ark"), "int32"))
    ali = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(tmp_path / "ali.txt")}
    for k, lat in lats:
        bp_words, bp_ali = latbin.best_path(lat)[:2]
        assert list(words[k]) == [int(x) for x in bp_words] and ali[k] == [int(x) for x in bp_ali]
    # waveforms in (features on the device), raw lattices out, one set
    wv = ["--wav", "--determinize-lattice=false", str(tmp_path / "final.mdl"), fst_, "scp:%s" % (tmp_path / "wav.scp")]
    r = subprocess.run([exe] + common + wv + ["ark:%s" % (tmp_path / "cxx_raw.lat")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    p = subprocess.run([sys.executable, ROOT + "/tools/nnet3_latgen_faster_batch.py"] + common + wv + ["ark:%s" % (tmp_path / "py_raw.lat")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    assert open(tmp_path / "cxx_raw.lat", "rb").read() == open(tmp_path / "py_raw.lat", "rb").read()
    assert "Decoded 6 utterances, 0 with errors." in r.stderr
    # the recipe's online i-vectors, estimated on the device (--ivector-extraction-config: final.ie / final.dubm / final.mat /
    # global_cmvn.stats read by the library itself, kamd_ivector_info_read): features in and waveforms in
    from kaldi_amd import ivector
    allf = np.concatenate([mf.ComputeFeatures(wv) for wv in waves])
    info = ivector.make_synthetic(num_gauss=32, ivector_dim=10, seed=6, feat_mean=allf.mean(0), feat_std=allf.std(0))
    conf = ivector.write_config_dir(tmp_path / "extractor", info)
    # (both chunkings of the reference: DecodableNnetSimple's -- nnet3-latgen-faster -- and NnetBatchComputer's tasks -- nnet3-latgen-faster-batch)
    for tag, inp, rule in (("feats", ["scp:%s" % (tmp_path / "feats.scp")], "simple"), ("wav", ["--wav", "scp:%s" % (tmp_path / "wav.scp")], "simple"),
                           ("feats_bc", ["scp:%s" % (tmp_path / "feats.scp")], "batch_computer")):
        args = common + ["--ivector-extraction-config=%s" % conf, "--frames-per-chunk=50", "--chunk-rule=%s" % rule, "--set-frames=300",
                         str(tmp_path / "final_iv.mdl"), fst_]
        args = args[:-2] + inp[:-1] + args[-2:] + inp[-1:]
        r = subprocess.run([exe] + args + ["ark:%s" % (tmp_path / ("cxx_oiv_%s.lat" % tag))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        p = subprocess.run([sys.executable, ROOT + "/tools/nnet3_latgen_faster_batch.py"] + args + ["ark:%s" % (tmp_path / ("py_oiv_%s.lat" % tag))],
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        assert open(tmp_path / ("cxx_oiv_%s.lat" % tag), "rb").read() == open(tmp_path / ("py_oiv_%s.lat" % tag), "rb").read()
        assert len(list(latbin.read_lattices("ark:%s" % (tmp_path / ("cxx_oiv_%s.lat" % tag))))) == 6
    # (the same utterances with one constant i-vector per speaker decode to other lattices: the online estimates matter)
    assert open(tmp_path / "cxx_oiv_feats.lat", "rb").read() != open(tmp_path / "cxx.lat", "rb").read()
    assert open(tmp_path / "cxx_oiv_feats.lat", "rb").read() != open(tmp_path / "cxx_oiv_feats_bc.lat", "rb").read()      # 17 against 16 frames per chunk
    # the reference's exits: usage without arguments, 255 + a message for a model that is not there
    assert subprocess.run([exe], capture_output=True).returncode == 1
    bad = subprocess.run([exe] + common + [str(tmp_path / "absent.mdl"), fst_, "scp:%s" % (tmp_path / "feats.scp"), "ark:/dev/null"],
                         capture_output=True, text=True)
    assert bad.returncode == 255 and "cannot open" in bad.stderr


def test_online2_wav_nnet3_latgen_faster_example_writes_the_python_tools_archive(tmp_path):
    """examples/online2_wav_nnet3_latgen_faster.cc (BASELINE configs[4] as a C++ host program: online.conf, spk2utt, wav.scp,
    a model with an i-vector input, the extraction config read by the library) against tools/online2_wav_nnet3_latgen_faster.py
    on the same files: the CompactLattice archive byte for byte -- plain, with --do-endpointing and with
    --ivector-silence-weighting.* -- whatever the --batch; a speaker's second utterance starts from the first one's
    adaptation state in both."""
    import sys
    import wave
    from kaldi_amd import feat, ivector, latbin, nnet
    from kaldi_amd import io as kio
    from tests.mdl_writer import write_mdl
    lib = os.path.join(ROOT, "kaldi_amd", "lib")
    exe = str(tmp_path / "online2-wav-nnet3-latgen-faster-amd")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "online2_wav_nnet3_latgen_faster.cc"),
                           "-o", exe, "-L", lib, "-lkaldi_amd", "-lpthread", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"])
    g = synth.make_hclg(num_units=25, vocab=60, n_hist=12, seed=6)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, ivector_dim=16, seed=12, output_scale=3.0)
    write_mdl(tmp_path / "final.mdl", m, num_units=25)
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    waves = {"a1": 2.1, "a2": 1.4, "b1": 1.8, "c1": 0.9}
    waves = {k: np.round(synth.make_wave(d, seed=90 + i)).astype(np.float32) for i, (k, d) in enumerate(waves.items())}
    op = abi.mfcc_opts_hires()
    allf = np.concatenate([feat.Mfcc(op).ComputeFeatures(w) for w in waves.values()])
    info = ivector.make_synthetic(num_gauss=64, ivector_dim=16, seed=9, feat_mean=allf.mean(0), feat_std=allf.std(0), max_count=10.0)
    iconf = ivector.write_config_dir(tmp_path / "ivector_extractor", info)
    (tmp_path / "mfcc.conf").write_text("--use-energy=false   # hires\n--num-mel-bins=40\n--num-ceps=40\n--low-freq=20\n--high-freq=-400\n")
    (tmp_path / "online.conf").write_text("--feature-type=mfcc\n--mfcc-config=%s\n--ivector-extraction-config=%s\n--endpoint.silence-phones=1:2\n" %
                                          (tmp_path / "mfcc.conf", iconf))
    with open(tmp_path / "wav.scp", "w") as scp:
        for k, w in waves.items():
            if k == "c1":
                continue                                     # spkC's audio is missing: the reference's warning
            with wave.open(str(tmp_path / (k + ".wav")), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(w.astype("<i2").tobytes())
            scp.write("%s %s\n" % (k, tmp_path / (k + ".wav")))
    (tmp_path / "spk2utt").write_text("spkA a1 a2\nspkB b1\nspkC c1\n")
    common = ["--config=%s" % (tmp_path / "online.conf"), "--beam=15", "--max-active=7000", "--lattice-beam=8", "--acoustic-scale=1.0",
              "--frame-subsampling-factor=3", "--max-seconds=3", "--chunk-length=0.18"]
    files = [str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst"), "ark:%s" % (tmp_path / "spk2utt"), "scp:%s" % (tmp_path / "wav.scp")]
    variants = {
        "plain": [],
        "endpoint": ["--do-endpointing=true", "--endpoint.silence-phones=" + ":".join(str(p) for p in range(1, 21)),
                     "--endpoint.rule3.min-trailing-silence=0.06", "--endpoint.rule3.max-relative-cost=inf"],
        "weighted": ["--ivector-silence-weighting.silence-phones=" + ":".join(str(p) for p in range(1, 26, 2)),
                     "--ivector-silence-weighting.silence-weight=0.001", "--ivector-silence-weighting.max-state-duration=5"],
    }
    got = {}
    for tag, extra in variants.items():
        p = subprocess.run([sys.executable, ROOT + "/tools/online2_wav_nnet3_latgen_faster.py"] + common + extra + ["--batch=3"] + files +
                           ["ark:%s" % (tmp_path / ("py_%s.lat" % tag))], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        want = open(tmp_path / ("py_%s.lat" % tag), "rb").read()
        for batch in (1, 3):
            r = subprocess.run([exe] + common + extra + ["--batch=%d" % batch] + files + ["ark:%s" % (tmp_path / ("cxx_%s_%d.lat" % (tag, batch)))],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            assert open(tmp_path / ("cxx_%s_%d.lat" % (tag, batch)), "rb").read() == want, (tag, batch)
            for text in (r.stderr, p.stderr):
                assert "Did not find audio for utterance c1" in text and "Decoded 3 utterances, 1 with errors." in text
            assert sorted(l for l in r.stderr.splitlines() if "log-like per frame" in l) == sorted(l for l in p.stderr.splitlines() if "log-like per frame" in l)
        got[tag] = want
    assert got["plain"] != got["endpoint"] and got["plain"] != got["weighted"]
    assert [k for k, _ in latbin.read_lattices("ark:%s" % (tmp_path / "cxx_plain_1.lat"))] == ["a1", "a2", "b1"]
    # --feature-type=fbank (OnlineFbank as the base feature): a model without an i-vector input over 40 log-mel bins
    m_fb = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, seed=13, output_scale=3.0)
    write_mdl(tmp_path / "final_fbank.mdl", m_fb, num_units=25)
    (tmp_path / "fbank.conf").write_text("--num-mel-bins=40\n--dither=0\n--window-type=hamming\n--use-energy=false\n")
    (tmp_path / "fbank_online.conf").write_text("--feature-type=fbank\n--fbank-config=%s\n" % (tmp_path / "fbank.conf"))
    fb_common = ["--config=%s" % (tmp_path / "fbank_online.conf")] + common[1:]
    fb_files = [str(tmp_path / "final_fbank.mdl")] + files[1:]
    p = subprocess.run([sys.executable, ROOT + "/tools/online2_wav_nnet3_latgen_faster.py"] + fb_common + ["--batch=2"] + fb_files +
                       ["ark:%s" % (tmp_path / "py_fbank.lat")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    r = subprocess.run([exe] + fb_common + ["--batch=3"] + fb_files + ["ark:%s" % (tmp_path / "cxx_fbank.lat")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(tmp_path / "cxx_fbank.lat", "rb").read() == open(tmp_path / "py_fbank.lat", "rb").read()
    assert [k for k, _ in latbin.read_lattices("ark:%s" % (tmp_path / "cxx_fbank.lat"))] == ["a1", "a2", "b1"]
    # the reference's exits: usage without arguments; 255 + a message for a model without the i-vector config
    assert subprocess.run([exe], capture_output=True).returncode == 1
    (tmp_path / "plain.conf").write_text("--feature-type=mfcc\n--mfcc-config=%s\n" % (tmp_path / "mfcc.conf"))
    bad = subprocess.run([exe, "--config=%s" % (tmp_path / "plain.conf")] + files + ["ark:/dev/null"], capture_output=True, text=True)
    assert bad.returncode == 255 and "ivector" in bad.stderr
