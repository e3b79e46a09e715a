"""Lattice determinisation + wire format on the host (no GPU): rs_lattice_entry_from_raw runs the same DeterminizeLattice /
CompactLatticeArkEntry code rs_result_lattice uses, on raw lattices made here.

Checked against a brute-force enumeration written in this file (every path of the raw lattice, grouped by word sequence,
best alignment per sequence under LatticeWeight's order: total cost, then graph cost -- fstext/lattice-weight.h:294-307),
and, when oracle/_ref is built, by pushing the bytes through the REFERENCE's lattice-to-nbest | nbest-to-linear.
"""
import math

import numpy as np
import pytest

from tests.lattice_io import BIN, read_compact_lattice, reference_paths

INF = float("inf")


def random_raw_lattice(rng, n_states, fan, n_words, eps_rate=0.3, quantum=0.0):
    """Acyclic state-level lattice in time order: arcs go forward, a share of them carry no word (label 0).  quantum > 0
    rounds the costs to multiples of it, which makes equal-cost alignments common."""
    def cost(scale):
        c = rng.random() * scale
        return float(np.float32(round(c / quantum) * quantum if quantum else c))
    arcs = []
    tid = 1
    for s in range(n_states - 1):
        for _ in range(int(rng.integers(1, fan + 1))):
            d = int(rng.integers(s + 1, min(n_states, s + 4)))
            w = 0 if rng.random() < eps_rate else int(rng.integers(1, n_words + 1))
            arcs.append((s, d, w, tid, cost(3), cost(5)))
            tid += 1
    final = [INF] * n_states
    final[n_states - 1] = cost(1)
    if n_states > 3:
        final[n_states - 2] = cost(2)
    return arcs, final


def brute_force(n_states, start, final, arcs):
    """{word sequence: (graph, acoustic, transition-ids)} of the best alignment per sequence."""
    out_arcs = [[] for _ in range(n_states)]
    for a in arcs:
        out_arcs[a[0]].append(a)
    best = {}

    def walk(s, words, g, ac, tids):
        if math.isfinite(final[s]):
            key, cand = tuple(words), (g + final[s] + ac, g + final[s], ac, list(tids))
            if key not in best or cand[:2] < best[key][:2]:
                best[key] = cand
        for (_, d, w, t, ag, aa) in out_arcs[s]:
            walk(d, words + [w] if w else words, g + ag, ac + aa, tids + [t])
    walk(start, [], 0.0, 0.0, [])
    return {k: v[1:] for k, v in best.items()}


def paths_of_compact(start, finals, arcs):
    out = []

    def walk(s, words, g, ac, tids):
        if s in finals:
            fg, fa, ft = finals[s]
            out.append((tuple(words), g + fg, ac + fa, tids + ft))
        for (label, ag, aa, at, d) in arcs[s]:
            walk(d, words + [label] if label else words, g + ag, ac + aa, tids + at)
    walk(start, [], 0.0, 0.0, [])
    return out


@pytest.mark.parametrize("seed", range(12))
def test_determinised_lattice_against_brute_force(seed):
    from rhasspy_speech_amd import _lib
    rng = np.random.default_rng(seed)
    n_states = int(rng.integers(4, 11))
    arcs, final = random_raw_lattice(rng, n_states, fan=3, n_words=3)
    beam = [1e9, 6.0, 2.5][seed % 3]
    want = brute_force(n_states, 0, final, arcs)
    entry = _lib.lattice_entry_from_raw(n_states, 0, final, arcs, beam, key="k1")
    start, finals, carcs = read_compact_lattice(entry, "k1")
    for row in carcs:
        labels = [a[0] for a in row]
        assert 0 not in labels, "epsilon arc in the determinised lattice"
        assert len(labels) == len(set(labels)), "not deterministic on word labels"
    got = paths_of_compact(start, finals, carcs)
    assert len({p[0] for p in got}) == len(got), "a word sequence appears twice"
    best_total = min(g + a for g, a, _ in want.values())
    by_words = {p[0]: p for p in got}
    for words, (g, a, tids) in want.items():
        p = by_words.get(words)
        if p is None:
            assert g + a > best_total + beam - 1e-3, f"{words} (total {g + a}, best {best_total}) is missing"
            continue
        assert p[1] == pytest.approx(g, abs=1e-4) and p[2] == pytest.approx(a, abs=1e-4), words
        assert p[3] == tids, words
    assert set(by_words) <= set(want)
    assert min(want, key=lambda k: (want[k][0] + want[k][1], want[k][0])) in by_words


@pytest.mark.parametrize("seed", range(200, 208))
def test_equal_cost_alignments_follow_the_weight_order(seed):
    """Costs on a 0.5 grid: many alignments of a word sequence tie on total cost; the one kept has the smallest graph cost
    (which alignment wins among exact (total, graph) ties is not defined, so the transition-ids are not compared)."""
    from rhasspy_speech_amd import _lib
    rng = np.random.default_rng(seed)
    n_states = int(rng.integers(5, 10))
    arcs, final = random_raw_lattice(rng, n_states, fan=3, n_words=2, quantum=0.5)
    want = brute_force(n_states, 0, final, arcs)
    start, finals, carcs = read_compact_lattice(_lib.lattice_entry_from_raw(n_states, 0, final, arcs, 1e9), "utt")
    got = {p[0]: p for p in paths_of_compact(start, finals, carcs)}
    assert set(got) == set(want)
    for words, (g, a, _) in want.items():
        assert (got[words][1], got[words][2]) == (g, a), words


def test_single_state_and_empty_word_sequences():
    from rhasspy_speech_amd import _lib
    # a lattice that is its start state alone, final: one path with no words
    entry = _lib.lattice_entry_from_raw(1, 0, [0.75], [], 10.0)
    start, finals, arcs = read_compact_lattice(entry, "utt")
    assert paths_of_compact(start, finals, arcs) == [((), 0.75, 0.0, [])]
    # only epsilon arcs: still one path, carrying the cheaper alignment's transition-ids
    entry = _lib.lattice_entry_from_raw(3, 0, [INF, INF, 0.0], [(0, 1, 0, 7, 1.0, 1.0), (0, 1, 0, 8, 0.5, 1.0), (1, 2, 0, 9, 0.0, 2.0)], 10.0)
    start, finals, arcs = read_compact_lattice(entry, "utt")
    (words, g, a, tids), = paths_of_compact(start, finals, arcs)
    assert words == () and (g, a) == (0.5, 3.0) and tids == [8, 9]


def test_bad_arguments_are_errors():
    from rhasspy_speech_amd import _lib
    with pytest.raises(_lib.RsError):
        _lib.lattice_entry_from_raw(2, 5, [INF, 0.0], [(0, 1, 1, 1, 0.0, 0.0)], 10.0)
    with pytest.raises(_lib.RsError):
        _lib.lattice_entry_from_raw(2, 0, [INF, 0.0], [(0, 9, 1, 1, 0.0, 0.0)], 10.0)


@pytest.mark.skipif(not (BIN / "lattice-to-nbest").exists(), reason="oracle/_ref is not built (bash oracle/build_ref.sh)")
@pytest.mark.parametrize("seed", [100, 101, 102, 103])
def test_reference_tools_list_the_same_paths(seed, tmp_path):
    from rhasspy_speech_amd import _lib
    rng = np.random.default_rng(seed)
    n_states = int(rng.integers(5, 10))
    arcs, final = random_raw_lattice(rng, n_states, fan=3, n_words=4)
    want = brute_force(n_states, 0, final, arcs)
    ark = tmp_path / "lat.ark"
    ark.write_bytes(_lib.lattice_entry_from_raw(n_states, 0, final, arcs, 1e9, key="utt"))
    listed = reference_paths(ark, 10000, tmp_path)
    assert len(listed) == len(want)
    totals = [p["graph"] + p["acoustic"] for p in listed]
    assert totals == sorted(totals)
    for p in listed:
        g, a, tids = want[tuple(p["words"])]
        assert p["ali"] == tids
        assert p["graph"] == pytest.approx(g, abs=1e-3) and p["acoustic"] == pytest.approx(a, abs=1e-3)


def test_subset_reached_again_with_a_better_cost_is_expanded_again():
    """A subset first reached through an expensive arc and expanded under that forward cost, then reached again more cheaply:
    the arcs pruned the first time must come back (the reference's pruned determinisation works its queue best-first,
    determinize-lattice-pruned.cc).  S -1(3)-> X, S -2(0)-> Y -3(0)-> X, X -4(0)-> F, X -5(3)-> F, beam 4: the path 2 3 5
    costs 3 and is inside the beam although 1 5 (cost 6) is not."""
    from rhasspy_speech_amd import _lib
    S, Y, X, F = 0, 1, 2, 3
    arcs = [(S, X, 1, 1, 3.0, 0.0), (S, Y, 2, 2, 0.0, 0.0), (Y, X, 3, 3, 0.0, 0.0), (X, F, 4, 4, 0.0, 0.0), (X, F, 5, 5, 3.0, 0.0)]
    final = [INF, INF, INF, 0.0]
    entry = _lib.lattice_entry_from_raw(4, 0, final, arcs, 4.0, key="k")
    start, finals, carcs = read_compact_lattice(entry, "k")
    got = {p[0]: p[1] + p[2] for p in paths_of_compact(start, finals, carcs)}
    # (1 5, cost 6, rides along: the determinised state after `1` and after `2 3` is one shared state, as in the reference's lattice)
    assert got == {(1, 4): 3.0, (1, 5): 6.0, (2, 3, 4): 0.0, (2, 3, 5): 3.0}, got
