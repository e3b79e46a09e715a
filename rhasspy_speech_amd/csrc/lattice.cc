#include "lattice.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <deque>
#include <map>
#include <memory>
#include <queue>
#include <unordered_map>

#include "kaldi_io.h"

namespace rs {
namespace {

struct Pair { double g, a; };
inline bool Better(const Pair &x, const Pair &y) {
  double tx = x.g + x.a, ty = y.g + y.a;
  return tx < ty || (tx == ty && x.g < y.g);
}

// A search node = one word prefix; `seed` = best (graph, acoustic) per lattice state right after the prefix's
// last word arc (before epsilon closure).
struct Node {
  std::vector<int32_t> words;
  std::vector<std::pair<int, Pair>> seed;
  bool complete = false;
  Pair total{0, 0};
};
struct HeapItem {
  double f;
  long id;
  std::shared_ptr<Node> node;
  bool operator<(const HeapItem &o) const { return f > o.f || (f == o.f && id > o.id); }   // min-heap
};

}  // namespace

std::vector<NbestPath> LatticeNbest(const RawLattice &lat, int n, double lattice_beam, double acoustic_scale) {
  std::vector<NbestPath> out;
  const int N = lat.num_states;
  if (N == 0 || lat.start < 0 || n <= 0) return out;
  const double INF = std::numeric_limits<double>::infinity();
  // adjacency (CSR)
  std::vector<int> begin(N + 1, 0);
  for (auto &a : lat.arcs) begin[a.src + 1]++;
  for (int s = 0; s < N; s++) begin[s + 1] += begin[s];
  std::vector<int> order(lat.arcs.size()), fill(begin.begin(), begin.end() - 1);
  for (size_t i = 0; i < lat.arcs.size(); i++) order[fill[lat.arcs[i].src]++] = (int)i;
  // topological index (Kahn)
  std::vector<int> indeg(N, 0), topo(N, -1), by_rank;
  for (auto &a : lat.arcs) indeg[a.dst]++;
  std::vector<int> stack;
  for (int s = 0; s < N; s++) if (indeg[s] == 0) stack.push_back(s);
  by_rank.reserve(N);
  while (!stack.empty()) {
    int s = stack.back();
    stack.pop_back();
    topo[s] = (int)by_rank.size();
    by_rank.push_back(s);
    for (int k = begin[s]; k < begin[s + 1]; k++) {
      int d = lat.arcs[order[k]].dst;
      if (--indeg[d] == 0) stack.push_back(d);
    }
  }
  if ((int)by_rank.size() != N) Fail("lattice has a cycle");
  // exact backward cost on the unscaled lattice
  std::vector<double> beta(N, INF);
  for (int r = N - 1; r >= 0; r--) {
    int s = by_rank[r];
    double b = lat.final_cost[s];
    for (int k = begin[s]; k < begin[s + 1]; k++) {
      const RawLattice::Arc &a = lat.arcs[order[k]];
      double c = a.graph + a.acoustic + beta[a.dst];
      if (c < b) b = c;
    }
    beta[s] = b;
  }
  const double best_total = beta[lat.start];
  if (!(best_total < INF)) return out;
  const double cutoff = best_total + lattice_beam;

  std::priority_queue<HeapItem> heap;
  long ids = 0;
  {
    auto root = std::make_shared<Node>();
    root->seed.push_back({lat.start, Pair{0, 0}});
    heap.push({best_total, ids++, root});
  }
  const size_t want = acoustic_scale == 1.0 ? (size_t)n : (size_t)20000;
  std::vector<NbestPath> found;
  // per-state scratch with stamps (round 5: an unordered_map + a map + a priority queue built and torn down per expanded prefix)
  std::vector<Pair> cur(N);
  std::vector<int> stamp(N, 0), done(N, 0), members;
  int epoch = 0;
  struct Step { int olabel, dst; Pair w; };
  std::vector<Step> steps;
  while (!heap.empty() && found.size() < want) {
    HeapItem it = heap.top();
    heap.pop();
    if (it.f > cutoff) break;
    Node &nd = *it.node;
    if (nd.complete) {
      NbestPath p;
      p.words = nd.words;
      p.graph_cost = nd.total.g;
      p.acoustic_cost = nd.total.a;
      found.push_back(std::move(p));
      continue;
    }
    // epsilon closure in topological order
    epoch++;
    members.clear();
    std::priority_queue<std::pair<int, int>, std::vector<std::pair<int, int>>, std::greater<std::pair<int, int>>> q;
    for (auto &sp : nd.seed) {
      const int st = sp.first;
      if (stamp[st] != epoch) { stamp[st] = epoch; members.push_back(st); cur[st] = sp.second; q.push({topo[st], st}); }
      else if (Better(sp.second, cur[st])) { cur[st] = sp.second; q.push({topo[st], st}); }
    }
    while (!q.empty()) {
      int s = q.top().second;
      q.pop();
      if (done[s] == epoch) continue;
      done[s] = epoch;
      const Pair ps = cur[s];
      for (int k = begin[s]; k < begin[s + 1]; k++) {
        const RawLattice::Arc &a = lat.arcs[order[k]];
        if (a.olabel != 0) continue;
        Pair cand{ps.g + a.graph, ps.a + a.acoustic};
        if (cand.g + cand.a + beta[a.dst] > cutoff) continue;
        if (stamp[a.dst] != epoch) { stamp[a.dst] = epoch; members.push_back(a.dst); cur[a.dst] = cand; q.push({topo[a.dst], a.dst}); }
        else if (Better(cand, cur[a.dst])) { cur[a.dst] = cand; q.push({topo[a.dst], a.dst}); }
      }
    }
    // (members in the order the closure processed them was the order of round 5's `members`: popped states, i.e. topological;
    // the results below do not depend on it -- minima under a strict order with ties resolved by the sort further down)
    std::sort(members.begin(), members.end(), [&](int x, int y) { return topo[x] < topo[y]; });
    // complete here?
    bool have = false;
    Pair bestf{0, 0};
    for (int s : members) {
      if (lat.final_cost[s] < INF) {
        Pair cand{cur[s].g + lat.final_cost[s], cur[s].a};
        if (!have || Better(cand, bestf)) { bestf = cand; have = true; }
      }
    }
    if (have) {
      auto c = std::make_shared<Node>();
      c->words = nd.words;
      c->complete = true;
      c->total = bestf;
      heap.push({bestf.g + bestf.a, ids++, c});
    }
    // extend by one word: (word, destination) pairs, the best weight per pair
    steps.clear();
    for (int s : members) {
      const Pair ps = cur[s];
      for (int k = begin[s]; k < begin[s + 1]; k++) {
        const RawLattice::Arc &a = lat.arcs[order[k]];
        if (a.olabel == 0) continue;
        Pair cand{ps.g + a.graph, ps.a + a.acoustic};
        if (cand.g + cand.a + beta[a.dst] > cutoff) continue;
        steps.push_back({a.olabel, a.dst, cand});
      }
    }
    std::stable_sort(steps.begin(), steps.end(), [](const Step &x, const Step &y) { return x.olabel != y.olabel ? x.olabel < y.olabel : x.dst < y.dst; });
    for (size_t i = 0; i < steps.size();) {
      auto c = std::make_shared<Node>();
      c->words = nd.words;
      c->words.push_back(steps[i].olabel);
      double f2 = INF;
      size_t j = i;
      while (j < steps.size() && steps[j].olabel == steps[i].olabel) {
        Pair best = steps[j].w;
        size_t k2 = j + 1;
        while (k2 < steps.size() && steps[k2].olabel == steps[j].olabel && steps[k2].dst == steps[j].dst) { if (Better(steps[k2].w, best)) best = steps[k2].w; k2++; }
        c->seed.push_back({steps[j].dst, best});
        f2 = std::min(f2, best.g + best.a + beta[steps[j].dst]);
        j = k2;
      }
      heap.push({f2, ids++, c});
      i = j;
    }
  }
  if (acoustic_scale != 1.0)
    std::stable_sort(found.begin(), found.end(), [&](const NbestPath &x, const NbestPath &y) {
      return x.graph_cost + acoustic_scale * x.acoustic_cost < y.graph_cost + acoustic_scale * y.acoustic_cost;
    });
  if ((int)found.size() > n) found.resize(n);
  return found;
}

// ------------------------------------------------------------------------------------------ determinisation
namespace {

struct Elem {           // subset element: lattice state + residual weight / alignment relative to the subset's incoming arc
  int state;
  Pair w;
  std::vector<int32_t> tids;
};

struct SubsetKey {
  std::vector<Elem> elems;       // sorted by state
  bool operator<(const SubsetKey &o) const {
    if (elems.size() != o.elems.size()) return elems.size() < o.elems.size();
    for (size_t i = 0; i < elems.size(); i++) {
      const Elem &a = elems[i], &b = o.elems[i];
      if (a.state != b.state) return a.state < b.state;
      if (a.w.g != b.w.g) return a.w.g < b.w.g;
      if (a.w.a != b.w.a) return a.w.a < b.w.a;
      if (a.tids != b.tids) return a.tids < b.tids;
    }
    return false;
  }
};

}  // namespace

CompactLat DeterminizeLattice(const RawLattice &lat, double beam) {
  CompactLat out;
  const int N = lat.num_states;
  if (N == 0 || lat.start < 0) return out;
  const double INF = std::numeric_limits<double>::infinity();
  std::vector<int> begin(N + 1, 0);
  for (auto &a : lat.arcs) begin[a.src + 1]++;
  for (int s = 0; s < N; s++) begin[s + 1] += begin[s];
  std::vector<int> order(lat.arcs.size()), fill(begin.begin(), begin.end() - 1);
  for (size_t i = 0; i < lat.arcs.size(); i++) order[fill[lat.arcs[i].src]++] = (int)i;
  std::vector<int> indeg(N, 0), topo(N, -1), by_rank;
  for (auto &a : lat.arcs) indeg[a.dst]++;
  std::vector<int> stack;
  for (int s = 0; s < N; s++) if (indeg[s] == 0) stack.push_back(s);
  while (!stack.empty()) {
    int s = stack.back();
    stack.pop_back();
    topo[s] = (int)by_rank.size();
    by_rank.push_back(s);
    for (int k = begin[s]; k < begin[s + 1]; k++) { int d = lat.arcs[order[k]].dst; if (--indeg[d] == 0) stack.push_back(d); }
  }
  if ((int)by_rank.size() != N) Fail("lattice has a cycle");
  std::vector<double> beta(N, INF);
  for (int r = N - 1; r >= 0; r--) {
    int s = by_rank[r];
    double b = lat.final_cost[s];
    for (int k = begin[s]; k < begin[s + 1]; k++) {
      const RawLattice::Arc &a = lat.arcs[order[k]];
      b = std::min(b, a.graph + a.acoustic + beta[a.dst]);
    }
    beta[s] = b;
  }
  const double best_total = beta[lat.start];
  if (!(best_total < INF)) return out;
  const double cutoff = best_total + beam;

  // epsilon closure (word label 0) of a seed set, pruned with the forward cost `alpha` of the subset; returns the elements
  // sorted by state, best (weight, alignment) per state
  // Round 6.  A state-level lattice of this decoder is chains of hundreds of arcs without a word label (one transition-id each) between
  // a few dozen word arcs, so an epsilon closure walks most of an utterance and an element's alignment is hundreds of ids long.
  // Round 5 kept a (state, weight, alignment) element for EVERY state of the closure and copied the alignment for every arc it looked
  // at: O(states x length) per closure, twice over.  Now
  //   * the closure keeps, per state, the best weight and a back pointer (predecessor state + transition-id, or the seed it came in
  //     by) in flat arrays with stamps -- nothing is copied while it runs;
  //   * only the states that matter outside the closure become elements of the subset: those with a word arc out and the final ones
  //     (Kaldi's "minimal representation" of a subset, determinize-lattice-pruned.cc: ConvertToMinimal drops the states all of whose
  //     arcs are epsilon); the others can only be left by arcs the closure has already followed.  Their alignments are read off the
  //     back pointers, once.
  // Subsets that differed in interior states only are now one output state: an equivalent lattice (same word sequences, same best
  // alignment and costs per sequence: tests/test_lattice_cpu.py against brute-force enumeration and the reference's tools), fewer states.
  std::vector<char> is_exit(N, 0);
  for (int s2 = 0; s2 < N; s2++) {
    if (lat.final_cost[s2] < INF) is_exit[s2] = 1;
    for (int k = begin[s2]; k < begin[s2 + 1] && !is_exit[s2]; k++) if (lat.arcs[order[k]].olabel != 0) is_exit[s2] = 1;
  }
  struct ClState { Pair w; int parent, tid, seed; };      // parent -1: a seed (index `seed` of the seed vector)
  std::vector<ClState> cl(N);
  std::vector<int> cl_stamp(N, 0), cl_done(N, 0), cl_members, cl_rev;
  int cl_epoch = 0;
  auto closure = [&](std::vector<Elem> seed, double alpha) {
    cl_epoch++;
    cl_members.clear();
    std::priority_queue<std::pair<int, int>, std::vector<std::pair<int, int>>, std::greater<std::pair<int, int>>> q;
    for (size_t i = 0; i < seed.size(); i++) {
      const int st = seed[i].state;
      if (cl_stamp[st] != cl_epoch) { cl_stamp[st] = cl_epoch; cl_members.push_back(st); cl[st] = ClState{seed[i].w, -1, 0, (int)i}; q.push({topo[st], st}); }
      else if (Better(seed[i].w, cl[st].w)) { cl[st] = ClState{seed[i].w, -1, 0, (int)i}; q.push({topo[st], st}); }
    }
    while (!q.empty()) {
      const int s = q.top().second;
      q.pop();
      if (cl_done[s] == cl_epoch) continue;
      cl_done[s] = cl_epoch;
      const Pair ws = cl[s].w;
      for (int k = begin[s]; k < begin[s + 1]; k++) {
        const RawLattice::Arc &a = lat.arcs[order[k]];
        if (a.olabel != 0) continue;
        const Pair w{ws.g + a.graph, ws.a + a.acoustic};
        if (alpha + w.g + w.a + beta[a.dst] > cutoff) continue;
        const bool seen = cl_stamp[a.dst] == cl_epoch;
        if (seen && !Better(w, cl[a.dst].w)) continue;
        if (!seen) { cl_stamp[a.dst] = cl_epoch; cl_members.push_back(a.dst); }
        cl[a.dst] = ClState{w, s, a.ilabel, -1};
        q.push({topo[a.dst], a.dst});
      }
    }
    std::sort(cl_members.begin(), cl_members.end());
    std::vector<Elem> v;
    for (int st : cl_members) {
      if (!is_exit[st]) continue;
      Elem e;
      e.state = st;
      e.w = cl[st].w;
      cl_rev.clear();
      int p = st;
      while (cl[p].parent >= 0) { if (cl[p].tid != 0) cl_rev.push_back(cl[p].tid); p = cl[p].parent; }
      const std::vector<int32_t> &head = seed[cl[p].seed].tids;
      e.tids.reserve(head.size() + cl_rev.size());
      e.tids.assign(head.begin(), head.end());
      e.tids.insert(e.tids.end(), cl_rev.rbegin(), cl_rev.rend());
      v.push_back(std::move(e));
    }
    return v;
  };
  // common weight (the best element's, LatticeWeight order) and common alignment prefix are moved out of the subset
  auto normalise = [&](std::vector<Elem> *v, CompactLat::Weight *common) {
    Pair best = (*v)[0].w;
    for (auto &e : *v) if (Better(e.w, best)) best = e.w;
    size_t pre = (*v)[0].tids.size();
    for (auto &e : *v) {
      size_t k = 0;
      while (k < pre && k < e.tids.size() && e.tids[k] == (*v)[0].tids[k]) k++;
      pre = k;
    }
    common->graph = best.g;
    common->acoustic = best.a;
    common->tids.assign((*v)[0].tids.begin(), (*v)[0].tids.begin() + pre);
    for (auto &e : *v) { e.w.g -= best.g; e.w.a -= best.a; e.tids.erase(e.tids.begin(), e.tids.begin() + pre); }
  };
  std::map<SubsetKey, int> ids;
  std::deque<SubsetKey> subsets;      // (a deque: references stay valid while it grows, the subset under expansion is not copied)
  std::vector<double> alpha;          // best forward total of every output state
  // Subsets are expanded best-first on alpha (as Kaldi's pruned determinisation processes its queue,
  // determinize-lattice-pruned.cc); a subset that is reached again with a better forward cost after it was expanded is
  // expanded again, because arcs pruned under the worse alpha may now be inside the beam.
  std::vector<double> expanded_at;    // alpha the subset was last expanded with (+inf = never)
  std::priority_queue<std::pair<double, int>, std::vector<std::pair<double, int>>, std::greater<std::pair<double, int>>> todo;
  auto state_of = [&](std::vector<Elem> v, double a) {
    SubsetKey key{std::move(v)};
    auto it = ids.find(key);
    if (it != ids.end()) {
      if (a < alpha[it->second]) { alpha[it->second] = a; todo.push({a, it->second}); }      // reached more cheaply: expand (again) under the better alpha
      return it->second;
    }
    const int id = (int)subsets.size();
    ids.emplace(key, id);
    subsets.push_back(std::move(key));
    alpha.push_back(a);
    expanded_at.push_back(INF);
    todo.push({a, id});
    out.arcs.emplace_back();
    out.final_w.emplace_back();
    out.is_final.push_back(0);
    return id;
  };
  // start: a state whose single epsilon-word arc carries the common part of the initial subset
  {
    std::vector<Elem> init = closure({Elem{lat.start, Pair{0, 0}, {}}}, 0.0);
    if (init.empty()) return out;          // (nothing with a word arc or a final weight within the beam: cannot happen when best_total is finite)
    CompactLat::Weight common;
    normalise(&init, &common);
    const bool trivial = common.graph == 0 && common.acoustic == 0 && common.tids.empty();
    if (trivial) {
      out.start = state_of(init, 0.0);
    } else {
      out.arcs.emplace_back();
      out.final_w.emplace_back();
      out.is_final.push_back(0);
      subsets.push_back(SubsetKey{});          // placeholder for the extra start state (never expanded)
      alpha.push_back(0.0);
      expanded_at.push_back(0.0);
      out.start = 0;
      const int s1 = state_of(init, common.graph + common.acoustic);
      out.arcs[0].push_back({s1, 0, common});
    }
  }
  while (!todo.empty()) {
    const size_t si = (size_t)todo.top().second;
    const double queued = todo.top().first;
    todo.pop();
    if (subsets[si].elems.empty() || queued > alpha[si] || expanded_at[si] <= alpha[si]) continue;      // placeholder / stale entry
    if (subsets.size() > 2000000) Fail("lattice determinisation: too many states");
    const std::vector<Elem> &elems = subsets[si].elems;
    const double a0 = alpha[si];
    expanded_at[si] = a0;
    out.arcs[si].clear();
    out.is_final[si] = 0;
    // final weight
    {
      bool have = false;
      Elem bestf{};
      for (auto &e : elems) {
        if (!(lat.final_cost[e.state] < INF)) continue;
        const Pair w{e.w.g + lat.final_cost[e.state], e.w.a};
        if (a0 + w.g + w.a > cutoff) continue;
        if (!have || Better(w, bestf.w)) { bestf.state = e.state; bestf.w = w; bestf.tids = e.tids; have = true; }
      }
      if (have) {
        out.is_final[si] = 1;
        out.final_w[si].graph = bestf.w.g;
        out.final_w[si].acoustic = bestf.w.a;
        out.final_w[si].tids = bestf.tids;
      }
    }
    // word arcs
    std::map<int, std::vector<Elem>> nxt;
    for (auto &e : elems) {
      for (int k = begin[e.state]; k < begin[e.state + 1]; k++) {
        const RawLattice::Arc &a = lat.arcs[order[k]];
        if (a.olabel == 0) continue;
        const Pair w{e.w.g + a.graph, e.w.a + a.acoustic};
        if (a0 + w.g + w.a + beta[a.dst] > cutoff) continue;
        Elem c{a.dst, w, e.tids};
        if (a.ilabel != 0) c.tids.push_back(a.ilabel);
        nxt[a.olabel].push_back(std::move(c));
      }
    }
    for (auto &kv : nxt) {
      std::vector<Elem> v = closure(std::move(kv.second), a0);
      if (v.empty()) continue;
      CompactLat::Weight common;
      normalise(&v, &common);
      const int d = state_of(std::move(v), a0 + common.graph + common.acoustic);
      out.arcs[si].push_back({d, kv.first, common});
    }
  }
  return out;
}

std::string CompactLatticeArkEntry(const std::string &key, const CompactLat &clat) {
  std::string o = key + " ";       // no "\0B": this holder's binary form starts with the FST magic (kaldi-lattice.cc:478-500)
  auto put = [&](const void *p, size_t n) { o.append(reinterpret_cast<const char *>(p), n); };
  auto put_i32 = [&](int32_t v) { put(&v, 4); };
  auto put_i64 = [&](int64_t v) { put(&v, 8); };
  auto put_str = [&](const std::string &s) { put_i32((int32_t)s.size()); o.append(s); };
  auto put_w = [&](const CompactLat::Weight &w) {
    const float g = (float)w.graph, a = (float)w.acoustic;
    put(&g, 4);
    put(&a, 4);
    put_i32((int32_t)w.tids.size());
    for (int32_t t : w.tids) put_i32(t);
  };
  int64_t narcs = 0;
  for (auto &v : clat.arcs) narcs += (int64_t)v.size();
  // FstHeader (fst.cc:58-82): magic, fst type, arc type, version, flags, properties, start, #states, #arcs
  put_i32(2125659606);
  put_str("vector");
  put_str("compactlattice44");
  put_i32(2);
  put_i32(0);
  const uint64_t props = 0x1ull | 0x2ull;            // kExpanded | kMutable: nothing else is claimed
  put(&props, 8);
  put_i64(clat.arcs.empty() ? -1 : clat.start);
  put_i64((int64_t)clat.arcs.size());
  put_i64(narcs);
  const float inf = std::numeric_limits<float>::infinity();
  for (size_t s = 0; s < clat.arcs.size(); s++) {
    if (clat.is_final[s]) {
      put_w(clat.final_w[s]);
    } else {                                          // CompactLatticeWeight::Zero() = (inf, inf, empty string)
      put(&inf, 4);
      put(&inf, 4);
      put_i32(0);
    }
    put_i64((int64_t)clat.arcs[s].size());
    for (auto &a : clat.arcs[s]) {
      put_i32(a.label);
      put_i32(a.label);
      put_w(a.w);
      put_i32(a.dst);
    }
  }
  return o;
}

}  // namespace rs
