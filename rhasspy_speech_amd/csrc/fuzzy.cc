#include "fuzzy.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <sstream>
#include <unordered_map>

namespace rs {

namespace {

// Python's repr of the running penalty (penalty += 0.1 in double) is what fstcompile parses into a float32 weight
float PenaltyOfRank(int k) {
  double p = 0.0;
  for (int i = 0; i < k; i++) p += 0.1;
  char buf[64];
  std::snprintf(buf, sizeof(buf), "%.17g", p);
  return std::strtof(buf, nullptr);
}

struct FanArc { int label; int next; float w; };

}  // namespace

FuzzyResult FuzzyMatcher::Match(const std::string &nbest_text) const {
  FuzzyResult res;
  // ---- the fan of hypotheses (transcribe_util.py:22-41): state 0 = start, a fresh state per word
  std::vector<std::vector<FanArc>> fan(1);
  std::vector<char> fan_final(1, 0);
  {
    std::istringstream is(nbest_text);
    std::string line;
    int rank = 0;
    while (std::getline(is, line)) {
      std::istringstream ls(line);
      std::string tok;
      if (!(ls >> tok)) continue;                  // blank line
      const float pen = PenaltyOfRank(rank);
      int state = 0;
      while (ls >> tok) {
        char *end = nullptr;
        const long id = std::strtol(tok.c_str(), &end, 10);
        if (end == tok.c_str() || *end != 0 || id < 0) Fail("fuzzy match: n-best symbol is not a word id: " + tok);
        fan.emplace_back();
        fan_final.push_back(0);
        const int nxt = (int)fan.size() - 1;
        fan[state].push_back({(int)id, nxt, pen});
        state = nxt;
      }
      fan_final[state] = 1;
      rank++;
    }
  }
  // ---- the composition, materialised the way `fstcompose` writes it (compose.h:336-376): composed state = (fan state,
  // grammar state), numbered in order of discovery while the states are expanded in increasing number; the arcs of a state
  // are the grammar's input-epsilon arcs first, then for every fan arc the grammar arcs with that input label (the fan has
  // no epsilons, so the sequence filter never blocks anything).  Ties between equally cheap paths are broken by this
  // numbering and arc order further down, so they are reproduced, not chosen.
  const float INF = std::numeric_limits<float>::infinity();
  struct Node { int a, g; float dist; int parent; int il, ol; float w; };
  struct CArc { int next, il, ol; float w; };
  std::vector<Node> nodes;
  std::vector<std::vector<CArc>> carcs;
  std::unordered_map<unsigned long long, int> index;
  auto node_of = [&](int a, int g) {
    const unsigned long long key = ((unsigned long long)(unsigned)a << 32) | (unsigned)g;
    auto it = index.find(key);
    if (it != index.end()) return it->second;
    nodes.push_back({a, g, INF, -1, 0, 0, 0.f});
    index.emplace(key, (int)nodes.size() - 1);
    return (int)nodes.size() - 1;
  };
  node_of(0, g_.start);
  for (size_t n = 0; n < nodes.size(); n++) {
    const int a = nodes[n].a, g = nodes[n].g;
    carcs.emplace_back();
    const uint32_t b = g_.arc_begin[g], e = g_.arc_begin[g + 1];
    for (uint32_t i = b; i < e && g_.arcs[i].ilabel == 0; i++) {
      const int m = node_of(a, g_.arcs[i].nextstate);
      carcs[n].push_back({m, 0, g_.arcs[i].olabel, g_.arcs[i].weight});
    }
    for (const FanArc &fa : fan[a]) {
      uint32_t lo = b, hi = e;
      while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (g_.arcs[mid].ilabel < fa.label) lo = mid + 1; else hi = mid; }
      for (uint32_t i = lo; i < e && g_.arcs[i].ilabel == fa.label; i++) {
        const int m = node_of(fa.next, g_.arcs[i].nextstate);
        carcs[n].push_back({m, fa.label, g_.arcs[i].olabel, fa.w + g_.arcs[i].weight});
      }
    }
    if (nodes.size() > 20000000) Fail("fuzzy match: the composition is too large");
  }
  // ---- `fstshortestpath` (shortest-path.h:177-250) with its automatic queue (queue.h:586-670): every composed state
  // carries the epsilon self loop of G.fuzzy.fst, so each state is its own strongly connected component with a LIFO
  // queue and the states are visited in the topological order of the component numbering = reverse post-order of a
  // depth-first search that follows the arcs in order.  A state's distance / parent only change on a strict improvement.
  const int N = (int)nodes.size();
  std::vector<int> order;             // post-order
  {
    std::vector<char> seen(N, 0);     // 1 = on the stack, 2 = finished
    std::vector<std::pair<int, size_t>> stack;
    stack.push_back({0, 0});
    seen[0] = 1;
    while (!stack.empty()) {
      const int s = stack.back().first;
      size_t &ai = stack.back().second;
      if (ai < carcs[s].size()) {
        const int t = carcs[s][ai++].next;
        if (!seen[t]) { seen[t] = 1; stack.push_back({t, 0}); }
        else if (seen[t] == 1 && t != s) Fail("fuzzy match: G.fuzzy.fst has a cycle of input-epsilon arcs (only the self loops are expected)");
      } else {
        seen[s] = 2;
        order.push_back(s);
        stack.pop_back();
      }
    }
  }
  nodes[0].dist = 0.f;
  float best_final = INF;
  int best_node = -1;
  for (size_t oi = order.size(); oi-- > 0;) {
    const int n = order[oi];
    const float sd = nodes[n].dist;
    if (!(sd < INF)) continue;
    if (fan_final[nodes[n].a] && g_.final_cost[nodes[n].g] < INF) {
      const float f = sd + g_.final_cost[nodes[n].g];
      if (f < best_final) { best_final = f; best_node = n; }
    }
    for (const CArc &ca : carcs[n]) {
      const float nd = sd + ca.w;
      if (nd < nodes[ca.next].dist) {
        nodes[ca.next].dist = nd;
        nodes[ca.next].parent = n; nodes[ca.next].il = ca.il; nodes[ca.next].ol = ca.ol; nodes[ca.next].w = ca.w;
      }
    }
  }
  if (best_node < 0) return res;
  // ---- the path, first arc first
  std::vector<int> path;
  for (int n = best_node; nodes[n].parent >= 0; n = nodes[n].parent) path.push_back(n);
  // fstrmepsilon | fsttopsort | fstproject --project_type=output | fstprint: epsilon:epsilon arcs disappear, their
  // weights go (Times, left to right) into the next remaining arc; weights print with 9 significant digits unless zero
  float pending = 0.f;
  bool have_pending = false;
  for (size_t i = path.size(); i-- > 0;) {
    const Node &nd = nodes[path[i]];
    if (nd.il == 0 && nd.ol == 0) {
      pending = have_pending ? pending + nd.w : nd.w;
      have_pending = true;
      continue;
    }
    const float w = have_pending ? pending + nd.w : nd.w;
    have_pending = false;
    pending = 0.f;
    if (w != 0.f) {
      char buf[64];
      std::snprintf(buf, sizeof(buf), "%.9g", (double)w);
      res.cost += std::strtod(buf, nullptr);
    }
    if (nd.ol != 0) res.olabels.push_back(nd.ol);
  }
  res.matched = !res.olabels.empty();
  if (!res.matched) res.cost = 0.0;
  return res;
}

}  // namespace rs
