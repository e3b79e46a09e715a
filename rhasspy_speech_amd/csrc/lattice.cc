#include "lattice.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
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
  std::unordered_map<int, Pair> cur;
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
    cur.clear();
    std::priority_queue<std::pair<int, int>, std::vector<std::pair<int, int>>, std::greater<std::pair<int, int>>> q;
    for (auto &sp : nd.seed) { cur[sp.first] = sp.second; q.push({topo[sp.first], sp.first}); }
    std::vector<char> done_flag;
    std::unordered_map<int, char> done;
    std::vector<int> members;
    while (!q.empty()) {
      int s = q.top().second;
      q.pop();
      if (done.count(s)) continue;
      done[s] = 1;
      members.push_back(s);
      const Pair ps = cur[s];
      for (int k = begin[s]; k < begin[s + 1]; k++) {
        const RawLattice::Arc &a = lat.arcs[order[k]];
        if (a.olabel != 0) continue;
        Pair cand{ps.g + a.graph, ps.a + a.acoustic};
        if (cand.g + cand.a + beta[a.dst] > cutoff) continue;
        auto f = cur.find(a.dst);
        if (f == cur.end() || Better(cand, f->second)) {
          cur[a.dst] = cand;
          q.push({topo[a.dst], a.dst});
        }
      }
    }
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
    // extend by one word
    std::map<int, std::unordered_map<int, Pair>> nxt;
    for (int s : members) {
      const Pair ps = cur[s];
      for (int k = begin[s]; k < begin[s + 1]; k++) {
        const RawLattice::Arc &a = lat.arcs[order[k]];
        if (a.olabel == 0) continue;
        Pair cand{ps.g + a.graph, ps.a + a.acoustic};
        if (cand.g + cand.a + beta[a.dst] > cutoff) continue;
        auto &dd = nxt[a.olabel];
        auto f = dd.find(a.dst);
        if (f == dd.end() || Better(cand, f->second)) dd[a.dst] = cand;
      }
    }
    for (auto &kv : nxt) {
      auto c = std::make_shared<Node>();
      c->words = nd.words;
      c->words.push_back(kv.first);
      double f2 = INF;
      for (auto &sp : kv.second) {
        c->seed.push_back({sp.first, sp.second});
        f2 = std::min(f2, sp.second.g + sp.second.a + beta[sp.first]);
      }
      heap.push({f2, ids++, c});
    }
  }
  if (acoustic_scale != 1.0)
    std::stable_sort(found.begin(), found.end(), [&](const NbestPath &x, const NbestPath &y) {
      return x.graph_cost + acoustic_scale * x.acoustic_cost < y.graph_cost + acoustic_scale * y.acoustic_cost;
    });
  if ((int)found.size() > n) found.resize(n);
  return found;
}

}  // namespace rs
