"""TEST INFRASTRUCTURE ONLY -- CPU oracle, part 4: n-best word sequences from a raw state-level lattice.

Restates the *result* of the reference's lattice post-processing:
  DeterminizeLatticePhonePrunedWrapper(raw, lattice_beam)   lat/determinize-lattice-pruned.cc:1488-1513
      -> one path per word sequence, weighted by its best alignment, pruned to best + lattice_beam
  lattice-to-nbest --n=N --acoustic-scale=s                  latbin/lattice-to-nbest.cc:80-110
      -> ShortestPath(N) on (graph + s * acoustic)
  nbest-to-linear                                           latbin/nbest-to-linear.cc:67-87
      -> word ids, and (graph cost, acoustic cost) per path
without building the determinised lattice: a best-first search over word-prefixes whose search nodes are the
weighted state subsets lattice determinisation would create (epsilon-closed, best (graph, acoustic) pair per
lattice state, compared on the float sum then on the graph part like LatticeWeight's Compare,
fstext/lattice-weight.h:294-307), with the exact backward cost as heuristic.  Paths therefore pop in increasing
total cost, one per distinct word sequence.
"""
from __future__ import annotations

import heapq
import itertools
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class Lattice:
    start: int
    state_frame: np.ndarray
    final: np.ndarray          # per state graph-side final cost, inf = non-final
    src: np.ndarray
    dst: np.ndarray
    ilabel: np.ndarray
    olabel: np.ndarray
    graph: np.ndarray
    acoustic: np.ndarray

    @property
    def num_states(self) -> int:
        return len(self.final)

    @property
    def num_arcs(self) -> int:
        return len(self.src)


@dataclass
class Path:
    words: List[int]
    graph_cost: float
    acoustic_cost: float

    @property
    def total(self) -> float:
        return self.graph_cost + self.acoustic_cost


def _better(a: Tuple[float, float], b: Tuple[float, float]) -> bool:
    """LatticeWeight Compare: smaller total wins; on equal totals the smaller graph cost."""
    ta, tb = a[0] + a[1], b[0] + b[1]
    return ta < tb or (ta == tb and a[0] < b[0])


def _topo_order(l: Lattice, out_arcs: List[List[int]]) -> np.ndarray:
    n = l.num_states
    indeg = np.zeros(n, np.int64)
    np.add.at(indeg, l.dst, 1)
    order = np.full(n, -1, np.int64)
    stack = [s for s in range(n) if indeg[s] == 0]
    k = 0
    while stack:
        s = stack.pop()
        order[s] = k
        k += 1
        for a in out_arcs[s]:
            d = int(l.dst[a])
            indeg[d] -= 1
            if indeg[d] == 0:
                stack.append(d)
    assert k == n, "lattice has a cycle"
    return order


def nbest(l: Lattice, n: int, lattice_beam: float, acoustic_scale: float = 1.0, max_sequences: int = 20000) -> List[Path]:
    if l.num_states == 0 or l.start < 0:
        return []
    out_arcs: List[List[int]] = [[] for _ in range(l.num_states)]
    for a in range(l.num_arcs):
        out_arcs[int(l.src[a])].append(a)
    topo = _topo_order(l, out_arcs)
    rank = np.argsort(topo)
    g = l.graph.astype(np.float64)
    ac = l.acoustic.astype(np.float64)
    fin = l.final.astype(np.float64)
    # exact backward cost (unscaled: the determinisation beam is on the unscaled lattice)
    beta = np.full(l.num_states, np.inf)
    for s in rank[::-1]:
        b = fin[s]
        for a in out_arcs[s]:
            c = g[a] + ac[a] + beta[l.dst[a]]
            if c < b:
                b = c
        beta[s] = b
    best_total = beta[l.start]
    if not np.isfinite(best_total):
        return []
    cutoff = best_total + lattice_beam

    def closure(seed: Dict[int, Tuple[float, float]]) -> Dict[int, Tuple[float, float]]:
        cur = dict(seed)
        heap = [(int(topo[q]), q) for q in cur]
        heapq.heapify(heap)
        done = set()
        while heap:
            _, q = heapq.heappop(heap)
            if q in done:
                continue
            done.add(q)
            pq = cur[q]
            for a in out_arcs[q]:
                if l.olabel[a] != 0:
                    continue
                d = int(l.dst[a])
                cand = (pq[0] + g[a], pq[1] + ac[a])
                if cand[0] + cand[1] + beta[d] > cutoff:
                    continue
                if d not in cur or _better(cand, cur[d]):
                    cur[d] = cand
                    heapq.heappush(heap, (int(topo[d]), d))
        return cur

    counter = itertools.count()
    # heap items: (f, tiebreak, kind, payload) ; kind 0 = prefix node (seed subset), 1 = complete path
    heap: List[Tuple[float, int, int, object]] = [(best_total, next(counter), 0, ((), {int(l.start): (0.0, 0.0)}))]
    found: List[Path] = []
    want = n if acoustic_scale == 1.0 else max_sequences
    while heap and len(found) < want:
        f, _, kind, payload = heapq.heappop(heap)
        if f > cutoff:
            break
        if kind == 1:
            found.append(payload)
            continue
        words, seed = payload
        sub = closure(seed)
        # completing here
        bestf = None
        for q, p in sub.items():
            if np.isfinite(fin[q]):
                cand = (p[0] + fin[q], p[1])
                if bestf is None or _better(cand, bestf):
                    bestf = cand
        if bestf is not None:
            heapq.heappush(heap, (bestf[0] + bestf[1], next(counter), 1, Path(list(words), bestf[0], bestf[1])))
        # extending by one word
        nxt: Dict[int, Dict[int, Tuple[float, float]]] = {}
        for q, p in sub.items():
            for a in out_arcs[q]:
                w = int(l.olabel[a])
                if w == 0:
                    continue
                d = int(l.dst[a])
                cand = (p[0] + g[a], p[1] + ac[a])
                if cand[0] + cand[1] + beta[d] > cutoff:
                    continue
                dd = nxt.setdefault(w, {})
                if d not in dd or _better(cand, dd[d]):
                    dd[d] = cand
        for w, seed2 in nxt.items():
            f2 = min(p[0] + p[1] + beta[q] for q, p in seed2.items())
            heapq.heappush(heap, (f2, next(counter), 0, (words + (w,), seed2)))
    if acoustic_scale != 1.0:
        found.sort(key=lambda p: p.graph_cost + acoustic_scale * p.acoustic_cost)
    return found[:n]
