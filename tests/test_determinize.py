"""Lattice determinization (host side).  PARITY UNPINNED (no OpenFst, no
lattice fixture in the reference); what is checked are the defining properties of
DeterminizeLatticePruned (lat/determinize-lattice-pruned.h:35-120):
  * the output is deterministic and epsilon-free on word labels;
  * for every word sequence the output has ONE path; its weight is the weight of the best
    raw path with that word sequence and its transition-id string is that path's alignment;
  * with an infinite beam every word sequence of the raw lattice is kept, with a finite beam
    every kept path is within the beam and every raw path well inside the beam is kept."""
import numpy as np
import pytest

from kaldi_amd import abi, synth
from kaldi_amd import io as kio
from oracle import orc


def raw_lattice(seed, n_words=3, peak=2.0, lattice_beam=6.0):
    g = synth.make_hclg(num_units=24, vocab=40, n_hist=10, seed=seed)
    ll, _, _ = synth.sample_utterance(g, n_words=n_words, seed=seed + 1, peak=peak)
    cfg = abi.decoder_config_recipe()
    cfg.lattice_beam = lattice_beam
    d = orc.Decoder(g, cfg, 1)
    d.Decode(ll)
    tid_phone = np.zeros(g.tid2pdf.size, np.int32)
    tid_phone[1::2] = np.arange(1, (g.tid2pdf.size - 1) // 2 + 1)     # forward tids enter unit u
    return d.GetRawLattice(), tid_phone


def raw_adj(lat):
    adj = [[] for _ in range(lat.frame.size)]
    for a in lat.arcs:
        adj[a["src"]].append(a)
    return adj


def best_raw_path_for_words(lat, adj, words):
    """Viterbi over (raw state, #words consumed): (cost, g, a, tids) of the best raw path that
    outputs exactly `words`; states are topologically ordered by (frame, epsilon links)."""
    S, n = lat.frame.size, len(words)
    INF = float("inf")
    best = {}
    best[(lat.start, 0)] = (0.0, 0.0, 0.0, ())
    # relax in frame order; epsilon links stay inside a frame, so iterate to a fixpoint per frame
    order = np.argsort(lat.frame, kind="stable")
    by_frame = {}
    for s in order:
        by_frame.setdefault(int(lat.frame[s]), []).append(int(s))
    for f in sorted(by_frame):
        changed = True
        while changed:
            changed = False
            for s in by_frame[f]:
                for k in range(n + 1):
                    cur = best.get((s, k))
                    if cur is None:
                        continue
                    for a in adj[s]:
                        k2 = k
                        if a["olabel"] != 0:
                            if k < n and words[k] == a["olabel"]:
                                k2 = k + 1
                            else:
                                continue
                        g2, a2 = cur[1] + float(a["graph_cost"]), cur[2] + float(a["acoustic_cost"])
                        t2 = cur[3] + ((int(a["ilabel"]),) if a["ilabel"] != 0 else ())
                        key = (int(a["dst"]), k2)
                        old = best.get(key)
                        if old is None or g2 + a2 < old[0] - 1e-7:
                            best[key] = (g2 + a2, g2, a2, t2)
                            if lat.frame[a["dst"]] == f:
                                changed = True
    out = None
    for s in range(S):
        if np.isfinite(lat.final[s]) and (s, n) in best:
            c = best[(s, n)]
            tot = (c[0] + float(lat.final[s]), c[1] + float(lat.final[s]), c[2], c[3])
            if out is None or tot[0] < out[0]:
                out = tot
    return out


def clat_paths(cl, limit=400):
    """all (words, g, a, tids) paths of a small compact lattice (DFS, capped)."""
    adj = [[] for _ in range(cl.num_states)]
    for i, a in enumerate(cl.arcs):
        adj[a["src"]].append(i)
    out = []

    def dfs(s, words, g, a, tids):
        if len(out) >= limit:
            return
        if np.isfinite(cl.final[2 * s]):
            out.append((tuple(words), g + float(cl.final[2 * s]), a + float(cl.final[2 * s + 1]),
                        tuple(tids) + tuple(int(x) for x in cl.final_string(s))))
        for i in adj[s]:
            x = cl.arcs[i]
            dfs(int(x["dst"]), words + [int(x["label"])], g + float(x["graph_cost"]), a + float(x["acoustic_cost"]),
                tids + [int(t) for t in cl.arc_string(i)])

    dfs(cl.start, [], 0.0, 0.0, [])
    return out


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("phone", [False, True])
def test_determinized_lattice_properties(seed, phone):
    lat, tid_phone = raw_lattice(seed)
    cl = kio.determinize_lattice(lat, 1e30, tid_phone if phone else None)
    assert cl.reached_beam and cl.start == 0 and cl.num_states > 0
    # deterministic, epsilon free
    for s in range(cl.num_states):
        labels = cl.arcs["label"][cl.arcs["src"] == s]
        assert (labels != 0).all() and len(set(labels.tolist())) == labels.size
    adj = raw_adj(lat)
    paths = clat_paths(cl, 5000)
    assert len(paths) >= 5 and len({p[0] for p in paths}) == len(paths)   # one path per word sequence
    bp = lat.best_path()
    best = min(paths, key=lambda p: p[1] + p[2])
    assert list(best[0]) == bp["words"].tolist()
    assert list(best[3]) == bp["alignment"].tolist()
    assert abs(best[1] - bp["graph_cost"]) < 1e-3 and abs(best[2] - bp["acoustic_cost"]) < 1e-3
    for words, g, a, tids in paths[:40]:
        ref = best_raw_path_for_words(lat, adj, list(words))
        assert ref is not None, words
        assert abs((g + a) - ref[0]) < 2e-3, (words, g + a, ref[0])
        # the alignment is that of a best path (ties may pick another equally good one)
        assert len(tids) == lat.frame.max() and abs(g - ref[1]) < 2e-3


@pytest.mark.parametrize("seed", range(3))
def test_every_raw_word_sequence_is_kept_without_pruning(seed):
    lat, tid_phone = raw_lattice(seed)
    cl = kio.determinize_lattice(lat, 1e30, tid_phone)
    have = {p[0]: p[1] + p[2] for p in clat_paths(cl, 100000)}
    adj = raw_adj(lat)
    rng = np.random.default_rng(seed)
    n_ok = 0
    for _ in range(300):                       # random walks through the raw lattice
        s, words, cost = lat.start, [], 0.0
        while adj[s]:
            a = adj[s][rng.integers(len(adj[s]))]
            if a["olabel"]:
                words.append(int(a["olabel"]))
            cost += float(a["graph_cost"]) + float(a["acoustic_cost"])
            s = int(a["dst"])
        if not np.isfinite(lat.final[s]):
            continue
        cost += float(lat.final[s])
        assert tuple(words) in have
        assert have[tuple(words)] <= cost + 2e-3
        n_ok += 1
    assert n_ok > 50


def test_pruning_beam():
    lat, tid_phone = raw_lattice(1, n_words=3, peak=1.5, lattice_beam=6.0)
    full = kio.determinize_lattice(lat, 1e30, tid_phone)
    pruned = kio.determinize_lattice(lat, 2.0, tid_phone)
    pf, pp = clat_paths(full, 5000), clat_paths(pruned, 5000)
    best = min(c[1] + c[2] for c in pf)
    costs = {p[0]: p[1] + p[2] for p in pf}
    kept = {p[0] for p in pp}
    assert kept and kept <= set(costs)
    for p in pp:
        assert p[1] + p[2] <= best + 2.0 + 1e-3
        assert abs(costs[p[0]] - (p[1] + p[2])) < 2e-3
    if len(pf) < 5000:
        for w, c in costs.items():
            if c <= best + 2.0 - 1e-2:
                assert w in kept
    assert pruned.arcs.size <= full.arcs.size


@pytest.mark.parametrize("binary", [True, False])
def test_compact_lattice_archive(tmp_path, binary):
    lat, tid_phone = raw_lattice(2)
    cl = kio.determinize_lattice(lat, 6.0, tid_phone)
    p = tmp_path / "lat.1"
    cl.write(p, "utt1", binary=binary, append=False, acoustic_scale=0.5)
    raw = p.read_bytes()
    assert raw.startswith(b"utt1 ")
    if binary:
        assert raw[5] == 214 and b"compactlattice44" in raw[:64] and b"vector" in raw[:32]
    else:
        lines = raw.decode().split("\n")
        assert lines[0] == "utt1 " and lines[-1] == "" and lines[-2] == ""
        first = lines[1].split("\t")
        assert first[0] == "0" and len(first) in (3, 4)
        if len(first) == 4:
            g, a, s = first[3].split(",")
            float(g), float(a)
            assert all(t.isdigit() for t in s.split("_") if t)


def test_word_only_and_no_determinization_modes():
    lat, tid_phone = raw_lattice(3)
    o = kio.determinize_opts_default()
    o.phone_determinize, o.word_determinize = 0, 0          # "copying lattice without determinization"
    cl = kio.determinize_lattice(lat, 5.0, tid_phone, o)
    assert cl.arcs.size == lat.arcs.size and (cl.arcs["str_len"] <= 1).all()
    o = kio.determinize_opts_default()
    o.word_determinize = 0                                   # phone pass only: not deterministic on words
    cl2 = kio.determinize_lattice(lat, 5.0, tid_phone, o)
    assert cl2.num_states > 0
