"""kamd_lattice_prune (PruneLattice on a raw lattice held on the host; the exact form of GetRawLatticePruned,
decoder/lattice-faster-online-decoder.cc:168-265) against path enumeration on small random acyclic lattices."""
import numpy as np

from kaldi_amd import abi, decoder


def _random_lattice(rng, n):
    arcs = []
    for s in range(n - 1):
        for _ in range(int(rng.integers(1, 4))):
            arcs.append((s, int(rng.integers(s + 1, n)), int(rng.integers(1, 9)), int(rng.integers(0, 4)), float(rng.uniform(0, 3)), float(rng.uniform(0, 3))))
    A = np.zeros(len(arcs), abi.LAT_ARC_DTYPE)
    for k, a in enumerate(arcs):
        A[k] = a
    final = np.full(n, np.inf, np.float32)
    final[n - 1] = 0.5
    for s in range(n - 1):
        if rng.random() < 0.2:
            final[s] = float(rng.uniform(0, 2))
    return decoder.Lattice(0, np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32), np.zeros(n, np.float32), final, A, n)


def test_prune_keeps_exactly_the_paths_within_the_beam():
    rng = np.random.default_rng(4)
    for trial in range(60):
        lat = _random_lattice(rng, int(rng.integers(3, 10)))
        paths = []

        def walk(s, cost, states, arcs):
            if np.isfinite(lat.final[s]):
                paths.append((cost + float(lat.final[s]), list(states), list(arcs)))
            for k in np.nonzero(lat.arcs["src"] == s)[0]:
                a = lat.arcs[k]
                walk(int(a["dst"]), cost + float(a["graph_cost"]) + float(a["acoustic_cost"]), states + [int(a["dst"])], arcs + [int(k)])
        walk(0, 0.0, [0], [])
        best = min(p[0] for p in paths)
        for beam in (0.0, 0.7, 2.5, 100.0):
            want_states, want_arcs = set(), set()
            for c, st, ar in paths:
                if c <= best + beam + 1e-9:
                    want_states.update(st); want_arcs.update(ar)
            got = decoder.prune_lattice(lat, beam)
            assert got is not None and set(got.hclg.tolist()) == want_states, (trial, beam)
            assert got.arcs.size == len(want_arcs)
            assert got.start == 0 and got.hclg[got.start] == 0
            # endpoints renumbered consistently: an arc's endpoints name the same HCLG states as before
            for a in got.arcs:
                assert any(int(lat.arcs[k]["src"]) == int(got.hclg[a["src"]]) and int(lat.arcs[k]["dst"]) == int(got.hclg[a["dst"]]) and
                           lat.arcs[k]["graph_cost"] == a["graph_cost"] for k in want_arcs)
