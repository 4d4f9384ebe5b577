"""GPU parity proper: the HIP decoder (through the C-ABI) must reproduce the canonical
CPU oracle BIT-EXACTLY: same surviving tokens (frame, HCLG state), same forward costs,
same links and weights, same per-frame cutoffs and cost offsets, same work counters."""
import ctypes as C

import numpy as np
import pytest

from kaldi_amd import abi, decoder, synth
from kaldi_amd._lib import lib
from oracle import orc
from tests.util import lattice_diff, lattices_equal

pytestmark = pytest.mark.gpu


def sizes(lanes=1, hash_cap=1 << 14, toks=1 << 18, links=1 << 19, frames=512):
    return abi.DecoderSizes(lanes, hash_cap, toks, links, frames)


def run_both(g, ll, cfg, sz=None):
    G = decoder.Graph(g)
    d = decoder.LatticeFasterDecoder(G, cfg, sz or sizes())
    d.Decode(ll)
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    return d, o


def assert_same(d, o):
    ln, lo = d.GetRawLattice(), o.GetRawLattice()
    assert lattices_equal(ln, lo), lattice_diff(ln, lo)
    tn, to = d.trace(), o.trace()
    np.testing.assert_array_equal(tn[0], to[0])
    np.testing.assert_array_equal(tn[1].view(np.uint32), to[1].view(np.uint32))
    np.testing.assert_array_equal(tn[2].view(np.uint32), to[2].view(np.uint32))
    np.testing.assert_array_equal(d.counters()[:7], o.counters()[:7])
    assert d.FinalRelativeCost() == o.FinalRelativeCost()
    if lo is None:                      # every token died (random graphs with dead ends): no lattice on either side
        assert ln is None and d.GetBestPath() is None
        return
    bn, bo = d.GetBestPath(), lo.best_path()
    assert bn["words"].tolist() == bo["words"].tolist()
    assert bn["alignment"].tolist() == bo["alignment"].tolist()
    assert bn["graph_cost"] == bo["graph_cost"] and bn["acoustic_cost"] == bo["acoustic_cost"]
    assert d.FinalRelativeCost() == o.FinalRelativeCost()


@pytest.mark.parametrize("seed", range(4))
def test_hclg_peaked_recipe_config(seed):
    g = synth.make_hclg(num_units=40, vocab=150, n_hist=25, seed=seed)
    ll, words, _ = synth.sample_utterance(g, n_words=7, seed=seed, peak=7.0)
    d, o = run_both(g, ll, abi.decoder_config_recipe())
    assert_same(d, o)
    assert d.GetBestPath()["words"].tolist() == words


@pytest.mark.parametrize("seed", range(4))
def test_random_graph_noise(seed):
    g = synth.make_random_graph(num_states=500, num_labels=40, mean_arcs=3.5, seed=seed, final_frac=0.2)
    ll = synth.random_loglikes(30, g.num_pdfs, seed=seed, scale=2.0)
    cfg = abi.decoder_config_recipe()
    cfg.beam, cfg.lattice_beam = 6.0, 4.0
    d, o = run_both(g, ll, cfg)
    assert_same(d, o)


@pytest.mark.parametrize("max_active,min_active", [(150, 20), (60, 0), (2147483647, 0), (40, 40), (5000, 300)])
def test_max_min_active_select(max_active, min_active):
    """Exact radix select == nth_element on every GetCutoff branch."""
    g = synth.make_hclg(num_units=16, vocab=80, n_hist=10, seed=21)
    ll = synth.random_loglikes(25, g.num_pdfs, seed=3, scale=0.7)
    cfg = abi.decoder_config_recipe()
    cfg.max_active, cfg.min_active = max_active, min_active
    d, o = run_both(g, ll, cfg)
    assert_same(d, o)
    if max_active < 1000:
        assert (o.trace()[0] > max_active).any()


def test_min_active_binding_with_tight_beam():
    g = synth.make_random_graph(num_states=400, num_labels=30, mean_arcs=4, seed=7, final_frac=0.5)
    ll = synth.random_loglikes(20, g.num_pdfs, seed=8, scale=3.0)
    cfg = abi.decoder_config_default()
    cfg.beam, cfg.max_active, cfg.min_active, cfg.lattice_beam = 2.0, 60, 25, 3.0
    d, o = run_both(g, ll, cfg)
    assert_same(d, o)


def test_hub_states_wave_expansion():
    """Unigram hub with hundreds of arcs exercises the wavefront-cooperative path."""
    g = synth.make_hclg(num_units=30, vocab=700, n_hist=6, fanout=(40, 90), seed=4)
    ll, words, _ = synth.sample_utterance(g, n_words=5, seed=1, peak=6.0)
    d, o = run_both(g, ll, abi.decoder_config_recipe())
    assert_same(d, o)


def test_advance_in_chunks_and_no_final_state():
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=3)
    ll, _, _ = synth.sample_utterance(g, n_words=6, seed=4)
    ll = ll[:-2]                      # stop mid-word: no final state reached
    cfg = abi.decoder_config_recipe()
    G = decoder.Graph(g)
    d = decoder.LatticeFasterDecoder(G, cfg, sizes())
    d.InitDecoding()
    for i in range(0, ll.shape[0], 5):
        d.AdvanceDecoding(ll[i:i + 5])
    d.FinalizeDecoding()
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    assert_same(d, o)
    assert d.NumFramesDecoded() == ll.shape[0]


def test_single_frame_and_reuse_of_decoder_object():
    g = synth.make_hclg(num_units=8, vocab=10, n_hist=3, seed=1)
    cfg = abi.decoder_config_recipe()
    G = decoder.Graph(g)
    d = decoder.LatticeFasterDecoder(G, cfg, sizes())
    for T, seed in ((1, 2), (9, 3), (2, 4)):      # one decoder object, several utterances
        ll = synth.random_loglikes(T, g.num_pdfs, seed=seed)
        d.Decode(ll)
        o = orc.Decoder(g, cfg, 1)
        o.Decode(ll)
        assert_same(d, o)


def test_batch_of_lanes_matches_per_utterance():
    g = synth.make_hclg(num_units=32, vocab=100, n_hist=16, seed=8)
    cfg = abi.decoder_config_recipe()
    utts = [synth.sample_utterance(g, n_words=3 + (i % 4), seed=50 + i, peak=6.5)[0] for i in range(9)]
    bd = decoder.BatchDecoder(decoder.Graph(g), cfg, sizes(lanes=9))
    lats = bd.decode(utts)
    for i, ll in enumerate(utts):
        o = orc.Decoder(g, cfg, 1)
        o.Decode(ll)
        assert lattices_equal(lats[i], o.GetRawLattice()), (i, lattice_diff(lats[i], o.GetRawLattice()))


def test_capacity_overflow_is_reported_not_silent():
    g = synth.make_hclg(num_units=16, vocab=80, n_hist=10, seed=21)
    ll = synth.random_loglikes(25, g.num_pdfs, seed=3, scale=0.7)
    G = decoder.Graph(g)
    d = decoder.LatticeFasterDecoder(G, abi.decoder_config_recipe(), sizes(hash_cap=1 << 14, toks=2000, links=3000))
    from kaldi_amd._lib import KamdError
    with pytest.raises(KamdError, match="capacity"):
        d.Decode(ll)


def test_epsilon_cycle_rejected():
    g = synth.make_random_graph(num_states=20, seed=1)
    g.arcs["ilabel"][g.arc_off[5]] = 0
    g.arcs["nextstate"][g.arc_off[5]] = 5
    from kaldi_amd._lib import KamdError
    with pytest.raises(KamdError, match="psilon"):
        decoder.Graph(g)


@pytest.mark.parametrize("seed", [0, 1])
def test_midsize_graph_saturated(seed):
    """~60k states, max-active binding on most frames, thousands of tokens per frame."""
    g = synth.make_hclg(num_units=200, vocab=3000, n_hist=300, fanout=(10, 60), seed=seed)
    ll = synth.random_loglikes(40, g.num_pdfs, seed=seed, scale=1.0)
    cfg = abi.decoder_config_recipe()
    cfg.max_active = 2000
    d, o = run_both(g, ll, cfg, sizes(hash_cap=1 << 16, toks=1 << 21, links=1 << 22))
    assert_same(d, o)
    assert (o.trace()[0] > 2000).any()


@pytest.mark.parametrize("words", [0, 64, 1024, 8192])
def test_small_level1_tables_take_the_level2_and_large_frame_paths(words):
    """The level-1 (LDS) table region shrunk (kamd_decoder_set_level1_table): frames of thousands of tokens then overflow into
    the level-2 table in HBM, switch between the half-region and the whole-region table from frame to frame and keep the
    commit's worklists in HBM -- the paths a full-size region only takes on frames of > 5 k / > 10 k tokens.  Same results."""
    g = synth.make_hclg(num_units=200, vocab=3000, n_hist=300, fanout=(10, 60), seed=0)
    ll = _mixed_load(g)[:40]
    cfg = abi.decoder_config_recipe()
    cfg.max_active = 3000
    G = decoder.Graph(g)
    d = decoder.LatticeFasterDecoder(G, cfg, sizes(hash_cap=1 << 16, toks=1 << 21, links=1 << 22))
    d.SetLevel1Table(words)
    d.Decode(ll)
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    assert_same(d, o)
    assert (o.trace()[0] > 6000).any() and (o.trace()[0] < 200).any()
    if words < 8192:
        assert int(d.counters()[7]) > 0          # tokens did go to the level-2 table


def _mixed_load(g):
    """frames of tens of thousands of tokens (flat scores) alternating with frames of a few
    tokens (sharply peaked scores): both finalize modes and both transitions between them"""
    a = synth.random_loglikes(12, g.num_pdfs, seed=1, scale=0.6)
    u, _, _ = synth.sample_utterance(g, n_words=3, seed=2, peak=9.0, noise=0.5)
    b = synth.random_loglikes(12, g.num_pdfs, seed=3, scale=0.6)
    return np.concatenate([a, u[:14], b, u[14:24], a[:6]]).astype(np.float32)


def test_finalize_frames_larger_than_lds_working_set():
    g = synth.make_hclg(num_units=200, vocab=3000, n_hist=300, fanout=(10, 60), seed=0)
    ll = _mixed_load(g)
    cfg = abi.decoder_config_recipe()
    cfg.max_active = 30000
    d, o = run_both(g, ll, cfg, sizes(hash_cap=1 << 16, toks=1 << 21, links=1 << 23))
    ntok = o.trace()[0]
    assert ntok.max() > 20000 and ntok[15:25].max() < 1000
    assert_same(d, o)


@pytest.mark.parametrize("big", [False, True])
def test_finalize_with_no_slack_in_the_arenas(big):
    """Finalize stages the surviving lattice from the top of the arenas downward; it must
    stay exact when the arenas are only just large enough for the decode itself."""
    from kaldi_amd._lib import KamdError
    if big:
        g = synth.make_hclg(num_units=200, vocab=3000, n_hist=300, fanout=(10, 60), seed=0)
        ll = _mixed_load(g)[:30]
        cfg = abi.decoder_config_recipe()
        cfg.max_active = 30000
        hc = 1 << 16
    else:
        g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=4)
        ll = synth.random_loglikes(30, g.num_pdfs, seed=5, scale=0.4)   # nearly everything survives
        cfg = abi.decoder_config_recipe()
        hc = 1 << 14
    G = decoder.Graph(g)

    def fits(toks, links):
        d = decoder.LatticeFasterDecoder(G, cfg, sizes(hash_cap=hc, toks=toks, links=links))
        try:
            d.Decode(ll)
        except KamdError:
            return None
        return d

    def smallest(f, lo, hi):        # smallest size in (lo, hi] for which f(size) succeeds
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if f(mid) is not None:
                hi = mid
            else:
                lo = mid
        return hi

    T, L = 1 << 21, 1 << 23
    assert fits(T, L) is not None
    t_min = smallest(lambda t: fits(t, L), 0, T)
    l_min = smallest(lambda l: fits(t_min, l), 0, L)
    # the number of candidate links a frame records depends on the order in which the running
    # cutoff tightens (not on the result), so the link threshold moves a little run to run
    d = None
    for extra in range(0, 2048, 32):
        d = fits(t_min, l_min + extra)
        if d is not None:
            break
    assert d is not None and fits(t_min - 1, L) is None
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    assert_same(d, o)


def test_graph_from_openfst_file_and_lattice_archive(tmp_path):
    """HCLG.fst -> kamd_graph_read_openfst -> decode -> lattice archive, the file-level drop-in
    of nnet3-latgen-faster --determinize-lattice=false."""
    from kaldi_amd import io as kio
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=3)
    ll, words, _ = synth.sample_utterance(g, n_words=5, seed=4, peak=6.0)
    cfg = abi.decoder_config_recipe()
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    G = decoder.Graph.from_file(tmp_path / "HCLG.fst")
    assert G.num_states() == g.num_states and G.num_arcs() == g.num_arcs
    d = decoder.LatticeFasterDecoder(G, cfg, sizes(), tid2pdf=g.tid2pdf)
    d.Decode(ll)
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    assert_same(d, o)
    lat = d.GetRawLattice()
    kio.write_lattice(tmp_path / "lat.1", "utt1", lat, binary=True, append=False)
    (key, st, fin, arcs), = list(kio.read_lattices(tmp_path / "lat.1"))
    assert key == "utt1" and st == lat.start and np.array_equal(arcs, lat.arcs)


def test_python_decode_utterance_lattice_faster(tmp_path):
    """kaldi_amd.wrappers: the DecodeUtteranceLatticeFaster tail in Python."""
    import io as pyio
    from kaldi_amd import io as kio
    from kaldi_amd import wrappers
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=3)
    ll, words, _ = synth.sample_utterance(g, n_words=5, seed=4, peak=3.0)
    cfg = abi.decoder_config_recipe()
    d = decoder.LatticeFasterDecoder(decoder.Graph(g), cfg, sizes())
    tid_phone = np.zeros(g.tid2pdf.size, np.int32)
    tid_phone[1::2] = np.arange(1, (g.tid2pdf.size - 1) // 2 + 1)
    log = pyio.StringIO()
    out = {}
    ok, like, w = wrappers.decode_utterance_lattice_faster(d, ll, "u1", acoustic_scale=0.5, determinize=True,
                                                           tid_phone=tid_phone, lattice_path=tmp_path / "clat.ark",
                                                           words_out=out, log=log)
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    bp = o.GetRawLattice().best_path()
    assert ok and w == bp["words"].tolist() == out["u1"]
    assert abs(like + bp["graph_cost"] + bp["acoustic_cost"]) < 1e-3
    raw = open(tmp_path / "clat.ark", "rb").read()
    assert raw.startswith(b"u1 ") and raw[3] == 214 and b"compactlattice44" in raw[:64]
    assert "Log-like per frame for utterance u1" in log.getvalue()
    ok, _, _ = wrappers.decode_utterance_lattice_faster(d, ll, "u2", determinize=False, lattice_path=tmp_path / "lat.ark",
                                                        log=log)
    (key, st, fin, arcs), = list(kio.read_lattices(tmp_path / "lat.ark"))
    assert ok and key == "u2" and arcs.size == o.GetRawLattice().arcs.size


def test_pipeline_grows_on_capacity_overflow():
    from kaldi_amd import nnet, pipeline
    from kaldi_amd._lib import KamdError
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=0.5)
    cfg = abi.decoder_config_recipe()
    waves = [synth.make_wave(d, seed=40 + i) for i, d in enumerate((1.5, 0.9))]
    small = abi.DecoderSizes(2, 1 << 12, 3000, 4000, 128)
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, cfg, max_utts=2, max_seconds=2.0, sizes=small)
    pipe.load(waves)
    with pytest.raises(KamdError, match="capacity"):
        pipe.run()
    pipe.run(auto_grow=8)
    assert pipe.sizes.arena_tokens > small.arena_tokens
    res = pipe.results()
    ref = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, cfg, max_utts=2, max_seconds=2.0).decode(waves)
    for a, b in zip(res, ref):
        assert lattices_equal(a["lattice"], b["lattice"])


def test_latgen_faster_mapped_tool(tmp_path):
    """files in / files out: id2pdf + HCLG.fst + loglikes.ark -> lat.ark + words, against a direct decode."""
    import subprocess
    import sys
    from kaldi_amd import io as kio
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=6)
    utts = {}
    for i in range(3):
        ll, words, _ = synth.sample_utterance(g, n_words=3 + i, seed=30 + i, peak=5.0)
        utts["utt%d" % i] = (ll, words)
        kio.write_matrix_ark(tmp_path / "ll.ark", "utt%d" % i, ll, binary=bool(i % 2), append=i > 0)
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    (tmp_path / "id2pdf.int").write_text("id2pdf " + " ".join(str(int(x)) for x in g.tid2pdf) + " \n")
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    cmd = [sys.executable, root + "/tools/latgen_faster_mapped.py", "--acoustic-scale=1.0", "--beam=15", "--lattice-beam=8",
           "--max-active=7000", "--determinize-lattice=0", str(tmp_path / "id2pdf.int"), str(tmp_path / "HCLG.fst"),
           str(tmp_path / "ll.ark"), str(tmp_path / "lat.ark"), str(tmp_path / "words.txt")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Done 3 utterances, failed for 0" in r.stderr
    got_words = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(tmp_path / "words.txt")}
    lats = {k: (st, fin, arcs) for k, st, fin, arcs in kio.read_lattices(tmp_path / "lat.ark")}
    cfg = abi.decoder_config_recipe()
    for key, (ll, words) in utts.items():
        assert got_words[key] == words
        o = orc.Decoder(g, cfg, 1)
        o.Decode(ll)
        assert lats[key][2].size == o.GetRawLattice().arcs.size
    # the same command line as a C++ host program over kaldi_amd.hpp (examples/latgen_faster_mapped.cc):
    # identical words, alignments and raw lattice archive
    exe = str(tmp_path / "latgen-faster-mapped-amd")
    libdir = root + "/kaldi_amd/lib"
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", root + "/include", root + "/examples/latgen_faster_mapped.cc", "-o", exe,
                           "-L", libdir, "-lkaldi_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    r = subprocess.run([exe, "--acoustic-scale=1.0", "--beam=15", "--lattice-beam=8", "--max-active=7000", "--determinize-lattice=false",
                        str(tmp_path / "id2pdf.int"), str(tmp_path / "HCLG.fst"), "ark:%s" % (tmp_path / "ll.ark"),
                        "ark:%s" % (tmp_path / "lat_cxx.ark"), "ark,t:%s" % (tmp_path / "words_cxx.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Done 3 utterances, failed for 0" in r.stderr
    assert open(tmp_path / "words_cxx.txt").read() == open(tmp_path / "words.txt").read()
    assert open(tmp_path / "lat_cxx.ark", "rb").read() == open(tmp_path / "lat.ark", "rb").read()
    r = subprocess.run([exe, "--no-such=1", "a", "b", "c", "d"], capture_output=True, text=True)
    assert r.returncode == 255
    # --config file, scp input with file:offset entries and a pipe entry
    from kaldi_amd import table
    with table.TableWriter("ark,scp:%s,%s" % (tmp_path / "ll2.ark", tmp_path / "ll2.scp"), "matrix") as w2:
        for k, (ll_, _) in utts.items():
            w2.write(k, ll_)
    lines = open(tmp_path / "ll2.scp").read().splitlines()
    kio.write_matrix_ark(tmp_path / "one.ark", "zzz", utts["utt2"][0], binary=True, append=False)
    lines[2] = "utt2 dd if=%s bs=1 skip=4 2>/dev/null |" % (tmp_path / "one.ark")         # the object without its "zzz " key
    open(tmp_path / "ll2.scp", "w").write("\n".join(lines) + "\n")
    (tmp_path / "dec.conf").write_text("--beam=15   # as above\n--lattice_beam=8\n--max-active=7000\n--Determinize-Lattice=false\n")
    r = subprocess.run([exe, "--config=%s" % (tmp_path / "dec.conf"), "--acoustic-scale=1.0", str(tmp_path / "id2pdf.int"), str(tmp_path / "HCLG.fst"),
                        "scp:%s" % (tmp_path / "ll2.scp"), "ark:%s" % (tmp_path / "lat_cxx2.ark"), "ark,t:%s" % (tmp_path / "words_cxx2.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(tmp_path / "words_cxx2.txt").read() == open(tmp_path / "words.txt").read()
    assert open(tmp_path / "lat_cxx2.ark", "rb").read() == open(tmp_path / "lat.ark", "rb").read()


def test_nnet3_latgen_faster_tool(tmp_path):
    """final.mdl + HCLG.fst + wav.scp -> CompactLattice archive + words, against the in-memory pipeline."""
    import subprocess
    import sys
    import wave
    from kaldi_amd import io as kio
    from kaldi_amd import nnet, pipeline
    from tests.mdl_writer import write_mdl
    g = synth.make_hclg(num_units=25, vocab=60, n_hist=12, seed=6)
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, g.num_pdfs, input_dim=40, seed=12, output_scale=3.0)
    write_mdl(tmp_path / "final.mdl", m, num_units=25)
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    waves = [np.round(synth.make_wave(d, seed=50 + i)).astype(np.float32) for i, d in enumerate((1.2, 2.0, 0.8))]
    with open(tmp_path / "wav.scp", "w") as scp:
        for i, w in enumerate(waves):
            with wave.open(str(tmp_path / ("u%d.wav" % i)), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
                f.writeframes(w.astype("<i2").tobytes())
            scp.write("utt%d %s\n" % (i, tmp_path / ("u%d.wav" % i)))
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    (tmp_path / "decode.config").write_text("--beam=15.0   # steps/nnet3/decode.sh\n--max-active=7000\n--lattice_beam=8.0\n")
    tool = [sys.executable, root + "/tools/nnet3_latgen_faster.py", "--config=%s" % (tmp_path / "decode.config"),
            "--acoustic-scale=1.0", "--frame-subsampling-factor=3", "--batch=2"]
    gz = tmp_path / "lat.1.gz"
    r = subprocess.run(tool + ["--wav", str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst"), "scp:%s" % (tmp_path / "wav.scp"),
                               "ark:| gzip -c > %s" % gz, "ark,t:%s" % (tmp_path / "words.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Done 3 utterances, failed for 0" in r.stderr
    # the same through the in-memory pipeline with the ORIGINAL model and a table built like the writer's
    g.tid2pdf = np.concatenate([[-1], np.stack([2 * np.arange(25) + 1, 2 * np.arange(25)], 1).reshape(-1)]).astype(np.int32)
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, abi.decoder_config_recipe(), max_utts=3, max_seconds=2.5)
    ref = pipe.decode(waves)
    got = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(tmp_path / "words.txt")}
    for i in range(3):
        assert got["utt%d" % i] == ref[i]["words"].tolist()
    import gzip
    raw = gzip.open(gz).read()
    assert raw.startswith(b"utt0 ") and raw.count(b"compactlattice44") == 3
    # features-rspecifier form: features dumped to an ark,scp pair (as compute-mfcc-feats would), raw lattices out
    from kaldi_amd import table
    with table.TableWriter("ark,scp:%s,%s" % (tmp_path / "feats.ark", tmp_path / "feats.scp"), "matrix") as w:
        for i in range(3):
            w.write("utt%d" % i, pipe.features(i))
    r = subprocess.run(tool + ["--determinize-lattice=false", str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst"),
                               "scp:%s" % (tmp_path / "feats.scp"), "ark:%s" % (tmp_path / "raw.lat"),
                               "ark,t:%s" % (tmp_path / "words2.txt"), "ark:%s" % (tmp_path / "ali.ark")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(tmp_path / "words2.txt").read() == open(tmp_path / "words.txt").read()
    alis = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "ali.ark"), "int32"))
    lats = dict(table.SequentialTableReader("ark:%s" % (tmp_path / "raw.lat"), "lattice"))
    for i in range(3):
        assert alis["utt%d" % i].tolist() == ref[i]["best"]["alignment"].tolist()
        start, final, arcs = lats["utt%d" % i]
        assert arcs.size == ref[i]["lattice"].arcs.shape[0]
    # bad usage is an error exit, like KALDI_ERR
    r = subprocess.run(tool + ["--no-such-option=1", "a", "b", "c", "d"], capture_output=True, text=True)
    assert r.returncode == 255 and "Invalid option" in r.stderr


def test_more_pdfs_than_the_lds_row_holds():
    """P = 9000: only part of the log-likelihood row is staged in LDS, the rest is read from HBM."""
    g = synth.make_hclg(num_units=4500, vocab=300, n_hist=20, seed=8)
    assert g.num_pdfs == 9000
    ll = synth.random_loglikes(20, g.num_pdfs, seed=9, scale=1.0)
    d, o = run_both(g, ll, abi.decoder_config_recipe(), sizes(hash_cap=1 << 16, toks=1 << 21, links=1 << 22))
    assert_same(d, o)


def test_pipeline_ragged_batch_with_empty_and_tiny_utterances():
    """Ragged batch: empty waveform, one shorter than a frame, a one-frame utterance, and
    normal ones; the short ones are skipped (None), the rest decode as if alone."""
    from kaldi_amd import nnet, pipeline
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    cfg = abi.decoder_config_recipe()
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, cfg, max_utts=6, max_seconds=3.0)
    waves = [np.zeros(0, np.float32), synth.make_wave(1.1, seed=1), np.zeros(399, np.float32),
             synth.make_wave(0.025, seed=2), synth.make_wave(2.0, seed=3)]
    res = pipe.decode(waves)
    assert res[0] is None and res[2] is None
    assert res[3]["lattice"].num_frames == 1
    solo = pipe.decode([waves[4]])
    assert lattices_equal(solo[0]["lattice"], res[4]["lattice"])      # batch independence
    res2 = pipe.decode(waves)
    assert lattices_equal(res2[4]["lattice"], res[4]["lattice"])
    for u in (1, 3, 4):
        o = orc.Decoder(g, cfg, 1)
        lane = [i for i in (1, 3, 4)].index(u)
        o.Decode(pipe.loglikes(lane))
        assert lattices_equal(res2[u]["lattice"], o.GetRawLattice())
    assert pipe.decode([np.zeros(10, np.float32)]) == [None]


@pytest.mark.parametrize("bounds", [[5], [7, 30], [3, 4, 60, 1000]])
def test_pipeline_overlapped_nnet_slices_give_the_same_lattices(bounds):
    """kamd_pipeline_set_overlap: the nnet stage cut in time and overlapped with the search.  Same
    log-likelihood rows (bit-equal), same lattices, whatever the boundaries (inside the model's
    context, past the end of short utterances, past the end of all)."""
    from kaldi_amd import nnet, pipeline
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=3.0)
    cfg = abi.decoder_config_recipe()
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, cfg, max_utts=6, max_seconds=4.0)
    waves = [synth.make_wave(d, seed=10 + i) for i, d in enumerate((3.1, 0.2, 1.0, 0.025, 2.4, 0.6))]
    want = pipe.decode(waves)
    want_ll = [pipe.loglikes(u).copy() for u in range(len(waves))]
    pipe.set_overlap(bounds)
    got = pipe.decode(waves)
    for u in range(len(waves)):
        np.testing.assert_array_equal(pipe.loglikes(u), want_ll[u])
        assert lattices_equal(got[u]["lattice"], want[u]["lattice"]), (u, lattice_diff(got[u]["lattice"], want[u]["lattice"]))
        assert got[u]["words"].tolist() == want[u]["words"].tolist()
    assert lib().kamd_decoder_last_advance_launches(pipe.dec._dec) == min(len(bounds) + 1, 1 + sum(b < want_ll[0].shape[0] for b in bounds))
    assert all(x >= 0 for x in pipe.last_stage_ms)
    pipe.set_overlap([])
    again = pipe.decode(waves)
    assert lattices_equal(again[0]["lattice"], want[0]["lattice"])
    assert lib().kamd_decoder_last_advance_launches(pipe.dec._dec) == 1


def test_no_device_memory_leak_over_create_destroy_cycles():
    """A server creates and drops graphs, models, decoders, pipelines, extractors and stream batches for its whole
    life: after warm-up, repeated cycles leave the free device memory where it was."""
    import gc
    from kaldi_amd import ivector, nnet, online, pipeline
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, ivector_dim=16)
    info = ivector.make_synthetic(num_gauss=32, ivector_dim=16, seed=1)
    waves = [synth.make_wave(1.0, seed=i) for i in range(3)]

    def cycle():
        ie = ivector.IvectorExtractor(info)
        pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, abi.decoder_config_recipe(), max_utts=3, max_seconds=2.0)
        pipe.set_ivector_extractor(ie, 50)
        pipe.decode(waves)
        sb = online.StreamBatch(abi.mfcc_opts_hires(), pipe.nnet, pipe.graph, abi.decoder_config_recipe(), 2, max_seconds=2.0,
                                sizes=abi.DecoderSizes(2, 1 << 12, 1 << 16, 1 << 17, 256))
        sb.set_ivector_extractor(ie, 20)
        sb.start([0])
        sb.accept(0, waves[0], input_finished=True)
        sb.advance([0])
        sb.finalize([0])
        del sb, pipe, ie
        gc.collect()

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert lib().kamd_device_mem_info(C.byref(f), C.byref(t)) == 0
        return f.value
    for _ in range(3):
        cycle()
    base = free_bytes()
    for _ in range(10):
        cycle()
    assert abs(free_bytes() - base) < (8 << 20), (base, free_bytes())


def test_loglike_rows_narrower_than_the_graphs_pdfs_are_refused():
    """A log-likelihood matrix with fewer columns than the pdfs the graph's arcs map to would be read out of bounds (the
    frame's row is DMA'd into LDS): the launch is refused by name instead.  (Found through the C++ mirror passing one
    entry too many of the transition-id table, whose garbage tail entry raised the pdf count.)"""
    from kaldi_amd._lib import KamdError
    g = synth.make_hclg(num_units=16, vocab=50, n_hist=8, seed=4)
    ll = synth.random_loglikes(12, g.num_pdfs, seed=1)
    d = decoder.LatticeFasterDecoder(decoder.Graph(g), abi.decoder_config_recipe(), sizes())
    with pytest.raises(KamdError, match="log-likelihood rows of %d columns" % (g.num_pdfs - 3)):
        d.Decode(np.ascontiguousarray(ll[:, :g.num_pdfs - 3]))
    d2 = decoder.LatticeFasterDecoder(decoder.Graph(g), abi.decoder_config_recipe(), sizes())
    d2.Decode(ll)                      # the right width decodes
    assert d2.GetRawLattice() is not None
