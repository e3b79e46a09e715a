// The rescoring path of rhasspy_speech/transcribe_wav.py:107-232 (`async_transcribe_rescore`; the streaming twin is
// transcribe_stream.py:131-274) on the lattices this library produces -- host side, no processes.
//
// The reference decodes with the OLD graph and pipes the lattice through seven Kaldi tools:
//   lattice-scale --lm-scale=0.0                  graph costs := 0 (the old graph's LM / pronunciation / transition costs go)
//   lattice-to-phone-lattice final.mdl            labels := phones read off the transition-ids (lattice-functions.cc:423-441:
//                                                  a phone is emitted on the non-self-loop transition out of HMM state 0)
//   lattice-compose - Ldet.fst                    phones -> words of the NEW lexicon; Ldet = L_disambig.fst without its
//                                                  #0 arcs, determinised (fstdeterminizestar), disambiguation symbols removed
//   lattice-determinize                           one (best) alignment per word sequence (--prune=false: no beam)
//   lattice-compose --phi-label=#0 - G.fst        adds the NEW LM; #0 arcs are failure (back-off) transitions
//   lattice-add-trans-probs --transition-scale=1.0 --self-loop-scale=0.1 final.mdl
//   lattice-to-nbest --n=N --acoustic-scale=A | nbest-to-linear
// A word sequence the new lexicon / grammar cannot produce drops out on the way: the README's out-of-vocabulary rejection.
//
// Here: the determinised lattice of rs_decode_opts.emit_lattice (lattice.cc) is expanded to transition-id level with phone
// labels, composed with the lexicon transducer, determinised again (lattice.cc: DeterminizeLattice), composed with G under
// failure-arc semantics, given its transition costs, and ranked.  Composing with L_disambig itself (minus the #0 arcs,
// disambiguation symbols read as epsilon) instead of its determinised form gives the same weighted phone->word relation
// in the tropical semiring -- determinisation only merges paths with equal labels keeping the cheapest -- and the next
// step keeps the cheapest alignment per word sequence anyway, so Ldet.fst never has to be built (the reference rebuilds it on
// every call).  Checked against the reference's own tools on its own lattices: tests/test_rescore_cpu.py.
#include "rescore.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <queue>
#include <set>
#include <unordered_map>

#include "kaldi_io.h"

namespace rs {

namespace {
const double kInf = std::numeric_limits<double>::infinity();

struct PairHash {
  size_t operator()(const std::pair<int, int> &p) const { return ((size_t)(uint32_t)p.first << 32) ^ (uint32_t)p.second; }
};
}  // namespace

Rescorer::Rescorer(const std::string &lang_dir) {
  // #0 of the word table (transcribe_wav.py:119-127)
  {
    std::ifstream in(lang_dir + "/words.txt");
    if (!in) Fail("cannot open " + lang_dir + "/words.txt");
    std::string w;
    long id;
    while (in >> w >> id) if (w == "#0") phi_ = (int)id;
    if (phi_ < 0) Fail("No value for disambiguation state (#0)");      // the reference's ValueError
  }
  std::set<int> disambig;
  {
    std::ifstream in(lang_dir + "/phones/disambig.int");
    if (!in) Fail("cannot open " + lang_dir + "/phones/disambig.int");
    long id;
    while (in >> id) disambig.insert((int)id);
  }
  // L: drop the arcs whose OUTPUT label is #0 (the awk filter, transcribe_wav.py:133), disambiguation symbols -> epsilon
  {
    Hclg L;
    L.Read(lang_dir + "/L_disambig.fst");
    l_start_ = L.start;
    l_final_.assign(L.final_cost.begin(), L.final_cost.end());
    l_arcs_.resize(L.num_states());
    for (int s = 0; s < L.num_states(); s++)
      for (uint32_t a = L.arc_begin[s]; a < L.arc_begin[s + 1]; a++) {
        FstArc arc = L.arcs[a];
        if (arc.olabel == phi_) continue;
        if (disambig.count(arc.ilabel)) arc.ilabel = 0;
        l_arcs_[s].push_back(arc);
      }
  }
  // G with PropagateFinal (fstext-utils-inl.h:1094-1136): a state that is not final inherits the final weight reachable
  // through its back-off arc
  {
    Hclg G;
    G.Read(lang_dir + "/G.fst");
    g_start_ = G.start;
    g_final_.assign(G.final_cost.begin(), G.final_cost.end());
    g_arcs_.resize(G.num_states());
    g_phi_.assign(G.num_states(), -1);
    for (int s = 0; s < G.num_states(); s++)
      for (uint32_t a = G.arc_begin[s]; a < G.arc_begin[s + 1]; a++) {
        const FstArc &arc = G.arcs[a];
        if (arc.ilabel == phi_) {
          if (g_phi_[s] >= 0) Fail("Phi nondeterminism found");
          g_phi_[s] = (int)g_arcs_[s].size();
        }
        g_arcs_[s].push_back(arc);
      }
    std::vector<char> done(G.num_states(), 0);
    std::function<void(int, int)> propagate = [&](int s, int depth) {
      if (done[s]) return;
      done[s] = 1;
      if (depth > G.num_states()) Fail("G.fst: loop of back-off arcs");
      if (std::isfinite(g_final_[s]) || g_phi_[s] < 0) return;
      const FstArc &pa = g_arcs_[s][g_phi_[s]];
      if (pa.nextstate == s) return;
      propagate(pa.nextstate, depth + 1);
      if (std::isfinite(g_final_[pa.nextstate])) g_final_[s] = g_final_[pa.nextstate] + pa.weight;
    };
    for (int s = 0; s < G.num_states(); s++) propagate(s, 0);
  }
}

std::vector<NbestPath> Rescorer::Rescore(const CompactLat &clat, const TransitionModel &tm, int nbest, double acoustic_scale) const {
  std::vector<NbestPath> none;
  if (clat.start < 0 || clat.arcs.empty()) return none;
  const int num_tids = (int)tm.id2pdf.size() - 1;
  auto phone_of = [&](int tid) {     // ConvertLatticeToPhones
    if (tid <= 0 || tid > num_tids) Fail("lattice carries an invalid transition-id " + std::to_string(tid));
    return (tm.id2hmm_state[tid] == 0 && !tm.id2self_loop[tid]) ? tm.id2phone[tid] : 0;
  };
  // ---- 1 + 2. graph costs to zero, transition-id level, phones as labels
  RawLattice ph;
  ph.num_states = (int)clat.arcs.size();
  ph.start = clat.start;
  auto new_state = [&]() { return ph.num_states++; };
  auto chain = [&](int src, int dst, const CompactLat::Weight &w) {
    const size_t n = w.tids.size();
    if (n == 0) { ph.arcs.push_back({src, dst, 0, 0.0, w.acoustic, 0}); return; }
    int cur = src;
    for (size_t k = 0; k < n; k++) {
      const int nxt = k + 1 == n ? dst : new_state();
      ph.arcs.push_back({cur, nxt, phone_of(w.tids[k]), 0.0, k == 0 ? w.acoustic : 0.0, w.tids[k]});
      cur = nxt;
    }
  };
  std::vector<int> final_states;
  for (size_t s = 0; s < clat.arcs.size(); s++) {
    for (auto &a : clat.arcs[s]) chain((int)s, a.dst, a.w);
    if (clat.is_final[s]) {
      const int f = new_state();
      chain((int)s, f, clat.final_w[s]);
      final_states.push_back(f);
    }
  }
  ph.final_cost.assign(ph.num_states, kInf);
  for (int f : final_states) ph.final_cost[f] = 0.0;
  std::vector<std::vector<int>> ph_out(ph.num_states);
  for (size_t i = 0; i < ph.arcs.size(); i++) ph_out[ph.arcs[i].src].push_back((int)i);
  // ---- 3. compose with the lexicon transducer (phones -> words); epsilon moves on either side are free
  RawLattice wl;
  {
    std::unordered_map<std::pair<int, int>, int, PairHash> id;
    std::vector<std::pair<int, int>> todo;
    auto sid = [&](int a, int b) {
      auto it = id.find({a, b});
      if (it != id.end()) return it->second;
      const int k = (int)id.size();
      id.emplace(std::make_pair(a, b), k);
      todo.push_back({a, b});
      return k;
    };
    wl.start = sid(ph.start, l_start_);
    for (size_t i = 0; i < todo.size(); i++) {
      if (todo.size() > 4000000) Fail("rescoring: the phone lattice composed with the lexicon grew beyond 4 M states");
      const int ps = todo[i].first, ls = todo[i].second, src = (int)i;
      for (int ai : ph_out[ps]) {
        const RawLattice::Arc &a = ph.arcs[ai];
        if (a.olabel == 0) { wl.arcs.push_back({src, sid(a.dst, ls), 0, a.graph, a.acoustic, a.ilabel}); continue; }
        for (const FstArc &la : l_arcs_[ls])
          if (la.ilabel == a.olabel) wl.arcs.push_back({src, sid(a.dst, la.nextstate), la.olabel, a.graph + la.weight, a.acoustic, a.ilabel});
      }
      for (const FstArc &la : l_arcs_[ls])
        if (la.ilabel == 0) wl.arcs.push_back({src, sid(ps, la.nextstate), la.olabel, (double)la.weight, 0.0, 0});
    }
    wl.num_states = (int)todo.size();
    wl.final_cost.assign(wl.num_states, kInf);
    for (size_t i = 0; i < todo.size(); i++) {
      const double pf = ph.final_cost[todo[i].first], lf = l_final_[todo[i].second];
      if (std::isfinite(pf) && std::isfinite(lf)) wl.final_cost[i] = pf + lf;
    }
  }
  // ---- 4. best alignment per word sequence, no beam
  CompactLat det = DeterminizeLattice(wl, kInf);
  if (det.start < 0 || det.arcs.empty()) return none;
  // ---- 5 + 6. compose with G (failure arcs), add the scaled transition log-probabilities of every alignment
  auto trans_cost = [&](const std::vector<int32_t> &tids) {      // AddTransitionProbs (hmm-utils.cc:1065-1084,1121-1149), scales 1.0 / 0.1
    const float tscale = 1.0f, sscale = 0.1f;
    double c = 0.0;
    for (int32_t t : tids) {
      float lp;
      if (tm.id2self_loop[t]) lp = sscale * tm.log_prob[t];
      else lp = sscale * tm.non_self_loop_log_prob[t] + tscale * (tm.log_prob[t] - tm.non_self_loop_log_prob[t]);
      c -= (double)lp;
    }
    return c;
  };
  RawLattice fin;
  {
    // States are (lattice state, G state, filter state): ComposeFst's sequence filter (compose-filter.h, SequenceComposeFilter)
    // orders moves along G's input-epsilon arcs -- hassil's grammars keep <eps> arcs: kaldi.py:321-341 never removes them --
    // before epsilon moves of the lattice, so every pair of paths is composed exactly once.  Filter state 1 = "a G epsilon was
    // taken at a lattice state that has epsilon arcs of its own"; the lattice's epsilon arcs may be taken in state 0 only.
    struct Key { int ds, gs, f; };
    std::unordered_map<uint64_t, int> id;
    std::vector<Key> todo;
    const uint64_t ng = (uint64_t)g_arcs_.size();
    auto sid = [&](int a, int b, int f) {
      const uint64_t k = ((uint64_t)a * ng + (uint64_t)b) * 2 + (uint64_t)f;
      auto it = id.find(k);
      if (it != id.end()) return it->second;
      const int n = (int)id.size();
      id.emplace(k, n);
      todo.push_back({a, b, f});
      return n;
    };
    fin.start = sid(det.start, g_start_, 0);
    for (size_t i = 0; i < todo.size(); i++) {
      if (todo.size() > 4000000) Fail("rescoring: the word lattice composed with G grew beyond 4 M states");
      const int ds = todo[i].ds, gs0 = todo[i].gs, fs = todo[i].f, src = (int)i;
      size_t n_eps = 0;
      for (auto &a : det.arcs[ds]) n_eps += a.label == 0;
      const bool alleps1 = n_eps == det.arcs[ds].size() && !det.is_final[ds], noeps1 = n_eps == 0;
      // G moves alone along an input-epsilon arc (the matcher's implicit loop on the lattice side comes first in ComposeFst)
      if (!alleps1)
        for (const FstArc &ga : g_arcs_[gs0])
          if (ga.ilabel == 0) fin.arcs.push_back({src, sid(ds, ga.nextstate, noeps1 ? 0 : 1), ga.olabel, (double)ga.weight, 0.0, 0});
      for (auto &a : det.arcs[ds]) {
        const double tc = trans_cost(a.w.tids);
        if (a.label == 0) {
          if (fs == 0) fin.arcs.push_back({src, sid(a.dst, gs0, 0), 0, a.w.graph + tc, a.w.acoustic, 0});
          continue;
        }
        // PhiMatcher: the word's own arcs, else follow the back-off arc (paying it) and look again
        int gs = gs0;
        double backoff = 0.0;
        for (int hops = 0; hops <= (int)g_arcs_.size(); hops++) {
          bool hit = false;
          for (const FstArc &ga : g_arcs_[gs])
            if (ga.ilabel == a.label && ga.ilabel != phi_) {
              hit = true;
              fin.arcs.push_back({src, sid(a.dst, ga.nextstate, 0), ga.olabel, a.w.graph + tc + backoff + ga.weight, a.w.acoustic, 0});
            }
          if (hit || g_phi_[gs] < 0) break;
          const FstArc &pa = g_arcs_[gs][g_phi_[gs]];
          backoff += pa.weight;
          gs = pa.nextstate;
        }
      }
    }
    fin.num_states = (int)todo.size();
    fin.final_cost.assign(fin.num_states, kInf);
    for (size_t i = 0; i < todo.size(); i++) {      // final weights carry an alignment: through one more arc
      const int ds = todo[i].ds, gs = todo[i].gs;
      if (!det.is_final[ds] || !std::isfinite(g_final_[gs])) continue;
      const CompactLat::Weight &fw = det.final_w[ds];
      const int f = fin.num_states++;
      fin.final_cost.push_back(0.0);
      fin.arcs.push_back({(int)i, f, 0, fw.graph + trans_cost(fw.tids) + g_final_[gs], fw.acoustic, 0});
    }
  }
  // ---- 7. lattice-to-nbest --acoustic-scale=A | nbest-to-linear
  return LatticeNbest(fin, nbest, kInf, acoustic_scale);
}

// ------------------------------------------------------------------------------------------------ CompactLattice reader
// One binary table entry "<key> " + VectorFst<CompactLatticeArc> (the inverse of lattice.cc: CompactLatticeArkEntry; also what
// the reference's online2-wav-nnet3-latgen-faster writes), for the host-only parity tests of the rescoring path.
CompactLat ParseCompactLatticeEntry(const char *p, size_t n, std::string *key) {
  size_t pos = 0;
  auto need = [&](size_t k) { if (pos + k > n) Fail("CompactLattice entry: unexpected end of data"); };
  auto i32 = [&]() { need(4); int32_t v; std::memcpy(&v, p + pos, 4); pos += 4; return v; };
  auto i64 = [&]() { need(8); int64_t v; std::memcpy(&v, p + pos, 8); pos += 8; return v; };
  auto f32 = [&]() { need(4); float v; std::memcpy(&v, p + pos, 4); pos += 4; return v; };
  auto str = [&]() { const int32_t k = i32(); need((size_t)k); std::string s(p + pos, (size_t)k); pos += (size_t)k; return s; };
  std::string k;
  while (pos < n && p[pos] != ' ') k.push_back(p[pos++]);
  if (pos >= n) Fail("CompactLattice entry: no key");
  pos++;
  if (key) *key = k;
  if (pos + 2 <= n && p[pos] == '\0' && p[pos + 1] == 'B') pos += 2;
  if (i32() != 2125659606) Fail("CompactLattice entry: bad FST magic");
  const std::string fst_type = str(), arc_type = str();
  if (fst_type != "vector" || arc_type.rfind("compactlattice", 0) != 0) Fail("CompactLattice entry: unexpected FST type " + fst_type + " / " + arc_type);
  (void)i32();                       // version
  const int32_t flags = i32();
  if (flags & 3) Fail("CompactLattice entry: symbol tables are not supported");
  (void)i64();                       // properties
  const int64_t start = i64(), ns = i64();
  (void)i64();                       // number of arcs
  CompactLat lat;
  lat.start = (int)start;
  lat.arcs.resize((size_t)ns);
  lat.final_w.resize((size_t)ns);
  lat.is_final.assign((size_t)ns, 0);
  auto weight = [&](CompactLat::Weight *w) {
    w->graph = f32();
    w->acoustic = f32();
    const int32_t len = i32();
    w->tids.resize((size_t)len);
    for (int32_t i = 0; i < len; i++) w->tids[i] = i32();
  };
  for (int64_t s = 0; s < ns; s++) {
    CompactLat::Weight fw;
    weight(&fw);
    if (std::isfinite(fw.graph) || std::isfinite(fw.acoustic)) { lat.is_final[s] = 1; lat.final_w[s] = fw; }
    const int64_t na = i64();
    for (int64_t a = 0; a < na; a++) {
      CompactLat::Arc arc;
      arc.label = i32();
      (void)i32();
      weight(&arc.w);
      arc.dst = i32();
      lat.arcs[s].push_back(std::move(arc));
    }
  }
  return lat;
}

}  // namespace rs
