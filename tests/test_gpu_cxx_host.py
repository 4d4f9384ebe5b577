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
    exe = build_cxx(str(tmp_path))
    out = subprocess.check_output([exe, str(fx)], text=True).strip().splitlines()
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
    assert out[3] == "badconfig threw"
    assert bp["words"].tolist() == words
