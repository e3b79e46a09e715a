// Weighted-transducer operations of the graph-construction chain; see graph_build.h.
//
// Reference semantics followed (file:line under /root/reference/kaldi):
//   Compose            openfst/src/include/fst/compose-filter.h:192-266 (sequence filter), src/fstext/table-matcher.h:300-325
//   DeterminizeStar    src/fstext/determinize-star-inl.h:195-260 (worklist), :623-830 (epsilon closure), :832-1040 (output)
//   MinimizeEncoded    src/fstext/fstext-utils.h:113-121 (quantise, encode labels + weights, acceptor minimise, decode)
//   PushSpecial        src/fstext/push-special.cc:90-250
//   RemoveEpsLocal     src/fstext/remove-eps-local-inl.h:46-310
//   ComposeContext     src/fstext/context-fst.cc:27-330, src/fstext/deterministic-fst-inl.h:408-505
#include "graph_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <set>
#include <unordered_map>

#include "kaldi_io.h"

namespace rs {
namespace gb {

// ------------------------------------------------------------------------------------------------------------ I/O
Fst ReadFst(const std::string &path) {
  Hclg h;
  h.Read(path);
  Fst f;
  const int ns = h.num_states();
  f.start = h.start;
  f.arcs.resize(ns);
  f.fin = h.final_cost;
  for (int s = 0; s < ns; s++) {
    f.arcs[s].reserve(h.arc_begin[s + 1] - h.arc_begin[s]);
    for (uint32_t a = h.arc_begin[s]; a < h.arc_begin[s + 1]; a++)
      f.arcs[s].push_back({h.arcs[a].ilabel, h.arcs[a].olabel, h.arcs[a].weight, h.arcs[a].nextstate});
  }
  return f;
}

namespace {
template <typename T> void Put(std::string *o, T v) { o->append(reinterpret_cast<const char *>(&v), sizeof(T)); }
void PutStr(std::string *o, const char *s) { Put<int32_t>(o, (int32_t)std::strlen(s)); o->append(s); }
}  // namespace

void WriteFst(const Fst &f, const std::string &path, bool const_type) {
  // openfst/src/lib/fst.cc:84-96 (header), include/fst/const-fst.h:102-110 / vector-fst.h (bodies).  Properties: kExpanded only;
  // every other bit is left "unknown", which readers recompute when they need it.
  std::string o;
  Put<int32_t>(&o, 2125659606);
  PutStr(&o, const_type ? "const" : "vector");
  PutStr(&o, "standard");
  Put<int32_t>(&o, 2);
  Put<int32_t>(&o, 0);
  Put<uint64_t>(&o, 0x1ull);
  Put<int64_t>(&o, f.start);
  Put<int64_t>(&o, f.NumStates());
  Put<int64_t>(&o, (int64_t)f.NumArcs());
  if (const_type) {
    uint32_t pos = 0;
    for (int s = 0; s < f.NumStates(); s++) {
      uint32_t ni = 0, no = 0;
      for (const Arc &a : f.arcs[s]) { ni += a.il == 0; no += a.ol == 0; }
      Put<float>(&o, f.fin[s]);
      Put<uint32_t>(&o, pos);
      Put<uint32_t>(&o, (uint32_t)f.arcs[s].size());
      Put<uint32_t>(&o, ni);
      Put<uint32_t>(&o, no);
      pos += (uint32_t)f.arcs[s].size();
    }
    for (int s = 0; s < f.NumStates(); s++)
      for (const Arc &a : f.arcs[s]) { Put<int32_t>(&o, a.il); Put<int32_t>(&o, a.ol); Put<float>(&o, a.w); Put<int32_t>(&o, a.next); }
  } else {
    for (int s = 0; s < f.NumStates(); s++) {
      Put<float>(&o, f.fin[s]);
      Put<int64_t>(&o, (int64_t)f.arcs[s].size());
      for (const Arc &a : f.arcs[s]) { Put<int32_t>(&o, a.il); Put<int32_t>(&o, a.ol); Put<float>(&o, a.w); Put<int32_t>(&o, a.next); }
    }
  }
  std::ofstream os(path, std::ios::binary);
  os.write(o.data(), (std::streamsize)o.size());
  os.close();                                   // a full disk shows at the flush
  if (!os.good()) Fail("Error writing FST to " + path);
}

std::vector<int32_t> ReadIntList(const std::string &path) {
  std::ifstream is(path);
  if (!is.good()) Fail("Could not read integer list from " + path);
  std::vector<int32_t> v;
  long x;
  while (is >> x) v.push_back((int32_t)x);
  return v;
}

void WriteILabelInfo(const std::vector<std::vector<int32_t>> &info, const std::string &path) {
  // context-fst.cc:325-332 in binary mode: "\0B", then int32 size and one integer vector per label (base/io-funcs-inl.h:29-53,
  // 180-200: each basic value is preceded by its byte size)
  std::string o("\0B", 2);
  auto put_i32 = [&](int32_t v) { o.push_back(4); Put<int32_t>(&o, v); };
  put_i32((int32_t)info.size());
  for (const auto &v : info) {
    o.push_back(4);
    Put<int32_t>(&o, (int32_t)v.size());
    for (int32_t x : v) Put<int32_t>(&o, x);
  }
  std::ofstream os(path, std::ios::binary);
  os.write(o.data(), (std::streamsize)o.size());
  if (!os.good()) Fail("Error writing ilabel info to " + path);
}

// ------------------------------------------------------------------------------------------------------------ basics
void Connect(Fst *f) {
  const int ns = f->NumStates();
  if (ns == 0 || f->start < 0) { *f = Fst(); return; }
  std::vector<char> acc(ns, 0), coacc(ns, 0);
  std::vector<int> stack{f->start};
  acc[f->start] = 1;
  while (!stack.empty()) {
    const int s = stack.back();
    stack.pop_back();
    for (const Arc &a : f->arcs[s]) if (!acc[a.next]) { acc[a.next] = 1; stack.push_back(a.next); }
  }
  std::vector<std::vector<int>> rev(ns);
  for (int s = 0; s < ns; s++) for (const Arc &a : f->arcs[s]) rev[a.next].push_back(s);
  for (int s = 0; s < ns; s++) if (f->fin[s] != kInf) { coacc[s] = 1; stack.push_back(s); }
  while (!stack.empty()) {
    const int s = stack.back();
    stack.pop_back();
    for (int p : rev[s]) if (!coacc[p]) { coacc[p] = 1; stack.push_back(p); }
  }
  std::vector<int> map(ns, -1);
  int n = 0;
  for (int s = 0; s < ns; s++) if (acc[s] && coacc[s]) map[s] = n++;
  if (map[f->start] < 0) { *f = Fst(); return; }
  Fst g;
  g.arcs.resize(n);
  g.fin.resize(n);
  g.start = map[f->start];
  for (int s = 0; s < ns; s++) {
    if (map[s] < 0) continue;
    g.fin[map[s]] = f->fin[s];
    auto &out = g.arcs[map[s]];
    for (const Arc &a : f->arcs[s]) if (map[a.next] >= 0) out.push_back({a.il, a.ol, a.w, map[a.next]});
  }
  *f = std::move(g);
}

void ArcSort(Fst *f, bool by_ilabel) {
  for (auto &v : f->arcs)
    std::stable_sort(v.begin(), v.end(), [by_ilabel](const Arc &x, const Arc &y) {
      if (by_ilabel) return x.il < y.il || (x.il == y.il && x.ol < y.ol);     // ILabelCompare / OLabelCompare (arcsort.h): ties on the other label
      return x.ol < y.ol || (x.ol == y.ol && x.il < y.il);
    });
}

namespace {
struct TripleHash {
  size_t operator()(const std::pair<std::pair<int, int>, int> &k) const {
    return ((size_t)(unsigned)k.first.first * 0x9E3779B97F4A7C15ull) ^ ((size_t)(unsigned)k.first.second * 0xC2B2AE3D27D4EB4Full) ^ (size_t)k.second;
  }
};
}  // namespace

Fst Compose(const Fst &a_in, const Fst &b_in, bool connect) {
  Fst a = a_in, b = b_in;
  ArcSort(&a, false);        // output epsilons of a state first, then by output label
  ArcSort(&b, true);
  Fst out;
  if (a.start < 0 || b.start < 0) return out;
  // per state of a: does it have output-epsilon arcs / only those (and no final weight)?
  const int na = a.NumStates();
  std::vector<char> noeps1(na), alleps1(na);
  for (int s = 0; s < na; s++) {
    size_t ne = 0;
    for (const Arc &x : a.arcs[s]) ne += x.ol == 0;
    noeps1[s] = ne == 0;
    alleps1[s] = ne == a.arcs[s].size() && a.fin[s] == kInf;
  }
  typedef std::pair<std::pair<int, int>, int> Key;
  std::unordered_map<Key, int, TripleHash> index;
  std::vector<Key> states;
  auto state_of = [&](int s1, int s2, int fs) {
    const Key k{{s1, s2}, fs};
    auto it = index.find(k);
    if (it != index.end()) return it->second;
    const int id = out.AddState();
    index.emplace(k, id);
    states.push_back(k);
    return id;
  };
  out.start = state_of(a.start, b.start, 0);
  for (size_t n = 0; n < states.size(); n++) {
    const int s1 = states[n].first.first, s2 = states[n].first.second, fs = states[n].second;
    if (a.fin[s1] != kInf && b.fin[s2] != kInf) out.fin[n] = a.fin[s1] + b.fin[s2];
    const auto &a1 = a.arcs[s1];
    const auto &b2 = b.arcs[s2];
    // the right side moves alone on an input epsilon (the left one stays): not from a state of `a` that can only leave by
    // output epsilons; afterwards the left side may not take an output epsilon of its own until a label has been matched
    size_t b_first = 0;
    for (; b_first < b2.size() && b2[b_first].il == 0; b_first++)
      if (!alleps1[s1]) {
        const Arc &y = b2[b_first];
        const int d = state_of(s1, y.next, noeps1[s1] ? 0 : 1);
        out.arcs[n].push_back({0, y.ol, y.w, d});
      }
    // the left side moves alone on an output epsilon
    size_t a_first = 0;
    for (; a_first < a1.size() && a1[a_first].ol == 0; a_first++)
      if (fs == 0) {
        const Arc &x = a1[a_first];
        const int d = state_of(x.next, s2, 0);
        out.arcs[n].push_back({x.il, 0, x.w, d});
      }
    // matching labels: walk the shorter arc list, binary-search the other (the table matcher's job in fsttablecompose: the
    // loop state of H has an arc per phone-in-context)
    if (a1.size() - a_first <= b2.size() - b_first) {
      for (size_t i = a_first; i < a1.size(); i++) {
        const Arc &x = a1[i];
        auto lo = std::lower_bound(b2.begin() + b_first, b2.end(), x.ol, [](const Arc &y, int l) { return y.il < l; });
        for (; lo != b2.end() && lo->il == x.ol; ++lo) {
          const int d = state_of(x.next, lo->next, 0);
          out.arcs[n].push_back({x.il, lo->ol, x.w + lo->w, d});
        }
      }
    } else {
      for (size_t j = b_first; j < b2.size(); j++) {
        const Arc &y = b2[j];
        auto lo = std::lower_bound(a1.begin() + a_first, a1.end(), y.il, [](const Arc &x, int l) { return x.ol < l; });
        for (; lo != a1.end() && lo->ol == y.il; ++lo) {
          const int d = state_of(lo->next, y.next, 0);
          out.arcs[n].push_back({lo->il, y.ol, lo->w + y.w, d});
        }
      }
    }
  }
  if (connect) Connect(&out);
  return out;
}

// ------------------------------------------------------------------------------------------------------------ DeterminizeStar
namespace {

inline float LogPlus(float f1, float f2) {      // fst::LogWeight Plus (float-weight.h): -log(e^-f1 + e^-f2), the correction in double
  if (f1 == kInf) return f2;
  if (f2 == kInf) return f1;
  if (f1 > f2) return (float)(f2 - std::log1p(std::exp(-(double)(f1 - f2))));
  return (float)(f1 - std::log1p(std::exp(-(double)(f2 - f1))));
}
inline bool ApproxEq(float a, float b, float delta) { return a <= b + delta && b <= a + delta; }

class StringTable {
 public:
  StringTable() { seqs_.emplace_back(); }                 // id 0 = the empty string
  int Append(int id, int label) {
    std::vector<int32_t> v = seqs_[id];
    v.push_back(label);
    return IdOf(v);
  }
  int IdOf(const std::vector<int32_t> &v) {
    if (v.empty()) return 0;
    auto it = ids_.find(v);
    if (it != ids_.end()) return it->second;
    seqs_.push_back(v);
    ids_.emplace(v, (int)seqs_.size() - 1);
    return (int)seqs_.size() - 1;
  }
  const std::vector<int32_t> &Seq(int id) const { return seqs_[id]; }
  int RemovePrefix(int id, size_t n) {
    if (n == 0) return id;
    const std::vector<int32_t> &v = seqs_[id];
    return IdOf(std::vector<int32_t>(v.begin() + n, v.end()));
  }

 private:
  std::vector<std::vector<int32_t>> seqs_;
  std::map<std::vector<int32_t>, int> ids_;
};

struct Element { int state, str; float w; };

struct TempArc { int il, ostr, next; float w; };      // next = -1: a final weight

class Determinizer {
 public:
  Determinizer(const Fst &in, bool use_log, float delta) : in_(in), log_(use_log), delta_(delta) {}

  Fst Run() {
    Fst out;
    if (in_.start < 0) return out;
    SubsetId({{in_.start, 0, 0.0f}});
    while (!queue_.empty()) {
      const int id = queue_.front();         // newest first, as the reference's deque is used (push_front / pop_front)
      queue_.pop_front();
      Process(id);
    }
    // ---- to an ordinary transducer: an output string longer than one label becomes a chain of extra states; the input label
    // and the weight sit on the first arc (determinize-star-inl.h:915-985)
    const int n = (int)temp_.size();
    for (int s = 0; s < n; s++) out.AddState();
    out.start = 0;
    for (int s = 0; s < n; s++)
      for (const TempArc &t : temp_[s]) {
        const std::vector<int32_t> &seq = strings_.Seq(t.ostr);
        int cur = s;
        if (t.next < 0) {
          for (size_t i = 0; i < seq.size(); i++) {
            const int nx = out.AddState();
            out.arcs[cur].push_back({0, seq[i], i == 0 ? t.w : 0.0f, nx});
            cur = nx;
          }
          out.fin[cur] = seq.empty() ? t.w : 0.0f;
        } else {
          for (size_t i = 0; i + 1 < seq.size(); i++) {
            const int nx = out.AddState();
            out.arcs[cur].push_back({i == 0 ? t.il : 0, seq[i], i == 0 ? t.w : 0.0f, nx});
            cur = nx;
          }
          out.arcs[cur].push_back({seq.size() <= 1 ? t.il : 0, seq.empty() ? 0 : seq.back(), seq.size() <= 1 ? t.w : 0.0f, t.next});
        }
      }
    return out;
  }

 private:
  float Plus(float a, float b) const { return log_ ? LogPlus(a, b) : std::min(a, b); }

  // a subset is looked up by its (state, string) pairs; the weights must agree within delta (determinize-star-inl.h:285-325)
  int SubsetId(const std::vector<Element> &sub) {
    size_t h = 0, factor = 1;
    for (const Element &e : sub) { h *= factor; h += (size_t)e.state + (size_t)103333 * (size_t)e.str; factor *= 23531; }
    auto range = table_.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
      const std::vector<Element> &o = subsets_[it->second];
      if (o.size() != sub.size()) continue;
      bool same = true;
      for (size_t i = 0; i < o.size() && same; i++)
        same = o[i].state == sub[i].state && o[i].str == sub[i].str && ApproxEq(o[i].w, sub[i].w, delta_);
      if (same) return it->second;
    }
    const int id = (int)subsets_.size();
    // (fstdeterminizestar has --max-states for inputs that are not determinizable -- a lexicon without disambiguation symbols,
    // say -- and otherwise runs until memory is gone; here the construction gives up with a message)
    if (id > 100000000) Fail("Determinization aborted since passed 100000000 states (is the input determinizable?)");
    subsets_.push_back(sub);
    temp_.emplace_back();
    table_.emplace(h, id);
    queue_.push_front(id);
    return id;
  }

  // ---- epsilon closure of a subset (determinize-star-inl.h:623-830): weights reaching a state over several epsilon paths are
  // added; a state is re-expanded only while the not yet propagated part still changes its weight by more than delta
  struct Info { Element e; float pending; bool queued; };
  void Closure(const std::vector<Element> &in, std::vector<Element> *out) {
    info_.clear();
    pos_.clear();
    std::deque<int> q;
    auto add = [&](int state, int str, float w) {
      auto it = pos_.find(state);
      if (it == pos_.end()) {
        pos_.emplace(state, (int)info_.size());
        info_.push_back({{state, str, kInf}, w, true});
        q.push_back(state);
        return;
      }
      Info &f = info_[it->second];
      if (f.e.str != str) Fail("FST was not functional -> not determinizable");
      f.pending = Plus(f.pending, w);
      if (!f.queued) {
        const float tot = Plus(f.e.w, f.pending);
        if (!ApproxEq(tot, f.e.w, delta_)) { f.queued = true; q.push_back(state); }
      }
    };
    auto expand = [&](int state, int str, float w, std::vector<Element> *first_level) {
      for (const Arc &a : in_.arcs[state]) {
        if (a.il != 0) continue;
        const int nstr = a.ol == 0 ? str : strings_.Append(str, a.ol);
        if (first_level) first_level->push_back({a.next, nstr, w + a.w});
        else add(a.next, nstr, w + a.w);
      }
    };
    std::vector<Element> first;
    for (const Element &e : in) expand(e.state, e.str, e.w, &first);
    if (first.empty()) { *out = in; return; }
    for (const Element &e : in) {                      // the members themselves: weight not yet "processed"
      pos_.emplace(e.state, (int)info_.size());
      info_.push_back({{e.state, e.str, kInf}, e.w, false});
    }
    for (const Element &e : first) add(e.state, e.str, e.w);
    int guard = 0;
    while (!q.empty()) {
      const int s = q.front();
      q.pop_front();
      Info &f = info_[pos_[s]];
      const float w = f.pending;
      f.e.w = Plus(f.e.w, w);
      f.pending = kInf;
      f.queued = false;
      const int str = f.e.str;
      if (++guard > 100000000) Fail("Determinization aborted: epsilon closure does not terminate");
      expand(s, str, w, nullptr);
    }
    out->clear();
    for (Info &f : info_) {
      if (f.pending != kInf) f.e.w = Plus(f.e.w, f.pending);
      out->push_back(f.e);
    }
    std::sort(out->begin(), out->end(), [](const Element &x, const Element &y) { return x.state < y.state; });
  }

  void Process(int id) {
    std::vector<Element> closed;
    {
      const std::vector<Element> sub = subsets_[id];       // (copy: subsets_ grows below)
      Closure(sub, &closed);
    }
    // final weight
    bool is_final = false;
    int fstr = 0;
    float fw = 0.0f;
    for (const Element &e : closed) {
      const float f = in_.fin[e.state];
      if (f == kInf) continue;
      if (!is_final) { is_final = true; fstr = e.str; fw = e.w + f; }
      else {
        if (fstr != e.str) Fail("FST was not functional -> not determinizable");
        fw = Plus(fw, e.w + f);
      }
    }
    if (is_final) temp_[id].push_back({0, fstr, -1, fw});
    // transitions, grouped by input label
    std::vector<std::pair<int, Element>> all;
    for (const Element &e : closed)
      for (const Arc &a : in_.arcs[e.state]) {
        if (a.il == 0) continue;
        all.push_back({a.il, {a.next, a.ol == 0 ? e.str : strings_.Append(e.str, a.ol), e.w + a.w}});
      }
    std::stable_sort(all.begin(), all.end(), [](const std::pair<int, Element> &x, const std::pair<int, Element> &y) {
      return x.first < y.first || (x.first == y.first && x.second.state < y.second.state);
    });
    size_t i = 0;
    std::vector<Element> sub;
    while (i < all.size()) {
      const int il = all[i].first;
      sub.clear();
      for (; i < all.size() && all[i].first == il; i++) {
        const Element &e = all[i].second;
        if (!sub.empty() && sub.back().state == e.state) {
          if (sub.back().str != e.str) Fail("FST was not functional -> not determinizable");
          sub.back().w = Plus(sub.back().w, e.w);
        } else {
          sub.push_back(e);
        }
      }
      // common output prefix and total weight go onto the arc
      std::vector<int32_t> pre = strings_.Seq(sub[0].str);
      for (size_t k = 1; k < sub.size() && !pre.empty(); k++) {
        const std::vector<int32_t> &o = strings_.Seq(sub[k].str);
        if (o.size() < pre.size()) pre.resize(o.size());
        for (size_t j = 0; j < pre.size(); j++) if (o[j] != pre[j]) { pre.resize(j); break; }
      }
      float tot = sub[0].w;
      for (size_t k = 1; k < sub.size(); k++) tot = Plus(tot, sub[k].w);
      const int common = strings_.IdOf(pre);
      for (Element &e : sub) { e.w = e.w - tot; e.str = strings_.RemovePrefix(e.str, pre.size()); }
      const int dst = SubsetId(sub);
      temp_[id].push_back({il, common, dst, tot});
    }
  }

  const Fst &in_;
  const bool log_;
  const float delta_;
  StringTable strings_;
  std::vector<std::vector<Element>> subsets_;
  std::vector<std::vector<TempArc>> temp_;
  std::unordered_multimap<size_t, int> table_;
  std::deque<int> queue_;
  std::vector<Info> info_;
  std::unordered_map<int, int> pos_;
};

}  // namespace

Fst DeterminizeStar(const Fst &f, bool use_log, float delta) {
  Fst sorted = f;
  ArcSort(&sorted, true);
  return Determinizer(sorted, use_log, delta).Run();
}

// ------------------------------------------------------------------------------------------------------------ MinimizeEncoded
void MinimizeEncoded(Fst *f, float delta) {
  Connect(f);
  const int ns = f->NumStates();
  if (ns == 0) return;
  auto quant = [delta](float w) { return (w == kInf || w == -kInf || w != w) ? w : std::floor(w / delta + 0.5f) * delta; };   // float-weight.h Quantize
  // one symbol per distinct (ilabel, olabel, quantised weight); a final weight is such a symbol on an arc into one extra
  // "superfinal" state (encode.h with kEncodeLabels | kEncodeWeights; arc-map.h MAP_REQUIRE_SUPERFINAL)
  std::map<std::tuple<int, int, float>, int> sym;
  auto sym_of = [&](int il, int ol, float w) {
    auto k = std::make_tuple(il, ol, w);
    auto it = sym.find(k);
    if (it != sym.end()) return it->second;
    const int id = (int)sym.size() + 1;
    sym.emplace(k, id);
    return id;
  };
  const int super = ns;
  std::vector<std::vector<std::pair<int, int>>> enc(ns + 1);       // (symbol, next)
  for (int s = 0; s < ns; s++) {
    for (Arc &a : f->arcs[s]) { a.w = quant(a.w); enc[s].push_back({sym_of(a.il, a.ol, a.w), a.next}); }
    if (f->fin[s] != kInf) { f->fin[s] = quant(f->fin[s]); enc[s].push_back({sym_of(0, 0, f->fin[s]), super}); }
  }
  // coarsest partition such that states of a block have the same set of (symbol, block of the destination): refined until stable
  // (for a deterministic input this is the minimal automaton minimize.h computes, whichever of its algorithms runs)
  std::vector<int> block(ns + 1, 0);
  block[super] = 1;
  int nblocks = 2;
  for (;;) {
    std::map<std::pair<int, std::vector<std::pair<int, int>>>, int> sig_ids;
    std::vector<int> nb(ns + 1);
    for (int s = 0; s <= ns; s++) {
      std::vector<std::pair<int, int>> sig;
      sig.reserve(enc[s].size());
      for (auto &e : enc[s]) sig.push_back({e.first, block[e.second]});
      std::sort(sig.begin(), sig.end());
      sig.erase(std::unique(sig.begin(), sig.end()), sig.end());
      auto key = std::make_pair(block[s], std::move(sig));
      auto it = sig_ids.find(key);
      if (it == sig_ids.end()) it = sig_ids.emplace(std::move(key), (int)sig_ids.size()).first;
      nb[s] = it->second;
    }
    const int n2 = (int)sig_ids.size();
    block.swap(nb);
    if (n2 == nblocks) break;
    nblocks = n2;
  }
  // quotient, blocks numbered by their first member (the superfinal block disappears into final weights)
  std::vector<int> id_of_block(nblocks, -1);
  Fst g;
  for (int s = 0; s < ns; s++)
    if (id_of_block[block[s]] < 0) id_of_block[block[s]] = g.AddState();
  std::vector<char> done(g.NumStates(), 0);
  for (int s = 0; s < ns; s++) {
    const int q = id_of_block[block[s]];
    if (done[q]) continue;
    done[q] = 1;
    std::vector<std::tuple<int, int, int, float>> seen;       // (il, ol, next, w) unique, sorted (ArcUniqueMapper)
    for (const Arc &a : f->arcs[s]) seen.emplace_back(a.il, a.ol, id_of_block[block[a.next]], a.w);
    std::sort(seen.begin(), seen.end());
    seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
    for (auto &t : seen) g.arcs[q].push_back({std::get<0>(t), std::get<1>(t), std::get<3>(t), std::get<2>(t)});
    g.fin[q] = f->fin[s];
  }
  g.start = id_of_block[block[f->start]];
  *f = std::move(g);
}

// ------------------------------------------------------------------------------------------------------------ PushSpecial
void PushSpecial(Fst *f, float delta) {
  const int n = f->NumStates();
  if (n == 0) return;
  const int init = f->start;
  std::vector<double> occ(n, 1.0 / std::sqrt((double)n));
  std::vector<std::vector<std::pair<int, double>>> pred(n);
  for (int s = 0; s < n; s++) {
    for (const Arc &a : f->arcs[s]) pred[a.next].push_back({s, (double)expf(-a.w)});
    const double fin = (double)expf(-f->fin[s]);
    if (fin != 0.0) pred[init].push_back({s, fin});
  }
  auto accuracy = [&]() {
    double mn = 0, mx = 0;
    for (int s = 0; s < n; s++) {
      double sum = 0.0;
      for (const Arc &a : f->arcs[s]) sum += (double)expf(-a.w) * occ[a.next] / occ[s];
      sum += (double)expf(-f->fin[s]) * occ[init] / occ[s];
      if (s == 0) { mn = mx = sum; } else { mn = std::min(mn, sum); mx = std::max(mx, sum); }
    }
    return std::log(mx / mn);
  };
  for (int iter = 0; iter < 200; iter++) {
    std::vector<double> nw(n);
    for (int i = 0; i < n; i++) nw[i] = 0.1 * occ[i];
    for (int i = 0; i < n; i++) for (auto &p : pred[i]) nw[p.first] += occ[i] * p.second;
    double sumsq = 0.0;
    for (int i = 0; i < n; i++) sumsq += nw[i] * nw[i];
    const double inv = 1.0 / std::sqrt(sumsq);
    for (int i = 0; i < n; i++) occ[i] = nw[i] * inv;
    if (iter % 5 == 0 && iter > 0 && accuracy() <= (double)delta) break;
  }
  for (int s = 0; s < n; s++) occ[s] = -std::log(occ[s]);
  for (int s = 0; s < n; s++) {
    for (Arc &a : f->arcs[s]) a.w = (float)((double)a.w + occ[a.next] - occ[s]);
    if (f->fin[s] != kInf) f->fin[s] = f->fin[s] + (float)(occ[init] - occ[s]);
  }
}

// ------------------------------------------------------------------------------------------------------------ symbols, local epsilon removal
void RemoveInputSymbols(Fst *f, const std::vector<int32_t> &syms) {
  std::set<int32_t> set(syms.begin(), syms.end());
  for (auto &v : f->arcs) for (Arc &a : v) if (set.count(a.il)) a.il = 0;
}

namespace {
// remove-eps-local-inl.h:46-310: an arc followed by a state with a single way in (or a single way out) is fused with what
// follows whenever the pair carries at most one input and one output label, in one pass over the states in number order.
class EpsLocal {
 public:
  EpsLocal(Fst *f, bool stochastic_in_log) : f_(f), log_(stochastic_in_log) {
    if (f_->start < 0) return;
    dead_ = f_->AddState();
    const int n = f_->NumStates();
    in_.assign(n, 0);
    out_.assign(n, 0);
    in_[f_->start]++;
    for (int s = 0; s < n; s++) {
      if (f_->fin[s] != kInf) out_[s]++;
      for (const Arc &a : f_->arcs[s]) { in_[a.next]++; out_[s]++; }
    }
    for (int s = 0; s < n; s++)
      for (size_t pos = 0; pos < f_->arcs[s].size(); pos++) RemoveEps(s, pos);
    Connect(f_);
  }

 private:
  static bool Combine(const Arc &a, const Arc &b, Arc *c) {
    if (a.il != 0 && b.il != 0) return false;
    if (a.ol != 0 && b.ol != 0) return false;
    c->w = a.w + b.w;
    c->il = a.il != 0 ? a.il : b.il;
    c->ol = a.ol != 0 ? a.ol : b.ol;
    c->next = b.next;
    return true;
  }
  // the totals that decide how the remaining weights are rescaled: summed as probabilities by the command-line tool
  // (RemoveEpsLocalSpecial, fstrmepslocal's default --stochastic-in-log=true), tropical inside GetHmmAsFsa
  float Plus(float a, float b) const { return log_ ? LogPlus(a, b) : std::min(a, b); }
  static float PlusFinal(float a, float b) { return std::min(a, b); }

  void Pattern1(int s, size_t pos, Arc arc) {       // the next state has this arc as its only way in, several ways out
    const int nx = arc.next;
    float removed = kInf, kept = kInf;
    std::vector<Arc> to_add;
    for (Arc &na : f_->arcs[nx]) {
      if (na.next == dead_) continue;
      Arc c;
      if (Combine(arc, na, &c)) {
        removed = Plus(removed, na.w);
        out_[nx]--;
        in_[na.next]--;
        na.next = dead_;
        to_add.push_back(c);
      } else {
        kept = Plus(kept, na.w);
      }
    }
    if (f_->fin[nx] != kInf) {
      if (arc.il == 0 && arc.ol == 0) {
        removed = Plus(removed, f_->fin[nx]);
        if (f_->fin[s] == kInf) out_[s]++;
        f_->fin[s] = PlusFinal(f_->fin[s], arc.w + f_->fin[nx]);
        out_[nx]--;
        f_->fin[nx] = kInf;
      } else {
        kept = Plus(kept, f_->fin[nx]);
      }
    }
    if (removed != kInf) {
      if (kept == kInf) {
        out_[s]--;
        in_[nx]--;
        f_->arcs[s][pos].next = dead_;
      } else {
        // keep the weights out of the next state "stochastic": what stays is scaled up, the arc into it down (Reweight)
        const float total = Plus(removed, kept);
        const float rw = kept - total;
        f_->arcs[s][pos].w += rw;
        for (Arc &na : f_->arcs[nx]) if (na.next != dead_) na.w -= rw;
        if (f_->fin[nx] != kInf) f_->fin[nx] -= rw;
      }
    }
    for (const Arc &c : to_add) {
      out_[s]++;
      in_[c.next]++;
      f_->arcs[s].push_back(c);
    }
  }

  void Pattern2(int s, size_t pos, Arc arc) {       // the next state has a single way out (an arc or its final weight)
    const int nx = arc.next;
    const bool can_delete_next = in_[nx] == 1;
    bool delete_arc = false;
    if (f_->fin[nx] != kInf) {
      if (arc.il == 0 && arc.ol == 0) {
        if (f_->fin[s] == kInf) out_[s]++;
        f_->fin[s] = PlusFinal(f_->fin[s], arc.w + f_->fin[nx]);
        delete_arc = true;
        if (can_delete_next) { out_[nx]--; f_->fin[nx] = kInf; }
      }
    } else {
      size_t k = 0;
      while (f_->arcs[nx][k].next == dead_) k++;
      Arc na = f_->arcs[nx][k];
      Arc c;
      if (Combine(arc, na, &c)) {
        delete_arc = true;
        if (can_delete_next) {
          out_[nx]--;
          in_[na.next]--;
          f_->arcs[nx][k].next = dead_;
        }
        out_[s]++;
        in_[c.next]++;
        f_->arcs[s].push_back(c);
      }
    }
    if (delete_arc) {
      out_[s]--;
      in_[nx]--;
      f_->arcs[s][pos].next = dead_;
    }
  }

  void RemoveEps(int s, size_t pos) {
    const Arc arc = f_->arcs[s][pos];
    const int nx = arc.next;
    if (nx == dead_ || nx == s) return;
    if (in_[nx] == 1 && out_[nx] > 1) Pattern1(s, pos, arc);
    else if (out_[nx] == 1) Pattern2(s, pos, arc);
  }

  Fst *f_;
  const bool log_;
  int dead_ = -1;
  std::vector<int> in_, out_;
};
}  // namespace

void RemoveEpsLocal(Fst *f, bool stochastic_in_log) { EpsLocal run(f, stochastic_in_log); }

// ------------------------------------------------------------------------------------------------------------ ComposeContext
Fst ComposeContext(const std::vector<int32_t> &disambig_in, int width, int central, Fst lg, std::vector<std::vector<int32_t>> *ilabels) {
  if (width <= 0 || central < 0 || central >= width) Fail("ComposeContext: bad context width / central position");
  std::vector<int32_t> disambig(disambig_in);
  std::sort(disambig.begin(), disambig.end());
  std::set<int32_t> all;
  for (auto &v : lg.arcs) for (const Arc &a : v) if (a.il != 0) all.insert(a.il);
  std::set<int32_t> phones;
  for (int32_t s : all) if (!std::binary_search(disambig.begin(), disambig.end(), s)) phones.insert(s);
  int32_t subseq = 1;
  if (!all.empty()) subseq = std::max(subseq, *all.rbegin() + 1);
  if (!disambig.empty()) subseq = std::max(subseq, disambig.back() + 1);
  if (central != width - 1) {
    // right context: every final state can emit the subsequential symbol into a new final state that loops on it
    // (context-fst.cc:293-322; the original final weights stay)
    std::vector<int> finals;
    for (int s = 0; s < lg.NumStates(); s++) if (lg.fin[s] != kInf) finals.push_back(s);
    const int super = lg.AddState();
    lg.arcs[super].push_back({subseq, 0, 0.0f, super});
    lg.fin[super] = 0.0f;
    for (int s : finals) lg.arcs[s].push_back({subseq, 0, lg.fin[s], super});
  }
  // ---- the inverse context transducer, built on demand: a state is the last width-1 phones; reading a phone emits the label
  // of the window whose central position just became known (context-fst.cc:27-260)
  std::vector<std::vector<int32_t>> &info = *ilabels;
  info.clear();
  struct VecHash {
    size_t operator()(const std::vector<int32_t> &v) const {
      size_t h = 1469598103934665603ull;
      for (int32_t x : v) h = (h ^ (size_t)(uint32_t)x) * 1099511628211ull;
      return h;
    }
  };
  std::unordered_map<std::vector<int32_t>, int, VecHash> label_of, cstate_of;
  std::vector<std::vector<int32_t>> cseq;
  auto find_label = [&](const std::vector<int32_t> &v) {
    auto it = label_of.find(v);
    if (it != label_of.end()) return it->second;
    info.push_back(v);
    label_of.emplace(v, (int)info.size() - 1);
    return (int)info.size() - 1;
  };
  auto find_cstate = [&](const std::vector<int32_t> &v) {
    auto it = cstate_of.find(v);
    if (it != cstate_of.end()) return it->second;
    cseq.push_back(v);
    cstate_of.emplace(v, (int)cseq.size() - 1);
    return (int)cseq.size() - 1;
  };
  find_label({});                                                    // 0 = epsilon
  find_cstate(std::vector<int32_t>(width - 1, 0));                   // 0 = nothing seen yet
  int pseudo_eps = 0;
  if (width > central + 1 && !disambig.empty()) pseudo_eps = find_label({0});     // "#-1" (context-fst.cc:62-78)
  std::vector<char> phone_flag(phones.empty() ? 1 : *phones.rbegin() + 1, 0);
  for (int32_t p : phones) phone_flag[p] = 1;
  auto is_phone = [&](int32_t l) { return l > 0 && l < (int32_t)phone_flag.size() && phone_flag[l]; };
  auto c_final = [&](int cs) {
    if (central < width - 1) return cseq[cs][central] == subseq;
    return true;
  };
  // GetArc(state, ilabel) -> (olabel of C^-1 = label of the result, next state); false = no such arc
  auto c_arc = [&](int cs, int32_t il, int *olabel, int *next) {
    if (std::binary_search(disambig.begin(), disambig.end(), il)) {
      *olabel = find_label({-il});
      *next = cs;
      return true;
    }
    const std::vector<int32_t> seq = cseq[cs];          // (copy: cseq grows below)
    if (is_phone(il)) {
      if (!seq.empty() && seq.back() == subseq) return false;
    } else if (il == subseq) {
      if (central + 1 == width || seq[central] == subseq) return false;
    } else {
      Fail("ComposeContext: invalid input label " + std::to_string(il) + " (confusion about phone list or disambig symbols?)");
    }
    std::vector<int32_t> full(seq);
    full.push_back(il);
    for (int i = central + 1; i < width; i++) if (full[i] == subseq) full[i] = 0;
    std::vector<int32_t> nseq(seq);
    if (!nseq.empty()) { nseq.erase(nseq.begin()); nseq.push_back(il); }
    *next = find_cstate(nseq);
    *olabel = full[central] == 0 ? pseudo_eps : find_label(full);
    return true;
  };
  // ---- composition, breadth first over (context state, lg state) (deterministic-fst-inl.h:408-505; not trimmed)
  Fst out;
  if (lg.start < 0) return out;
  struct PairHash2 { size_t operator()(const std::pair<int, int> &k) const { return ((size_t)(uint32_t)k.first << 32) ^ (size_t)(uint32_t)k.second; } };
  std::unordered_map<std::pair<int, int>, int, PairHash2> index;
  std::deque<std::pair<int, int>> q;
  index[{0, lg.start}] = out.AddState();
  out.start = 0;
  q.push_back({0, lg.start});
  while (!q.empty()) {
    const std::pair<int, int> cur = q.front();
    q.pop_front();
    const int id = index[cur];
    if (c_final(cur.first) && lg.fin[cur.second] != kInf) out.fin[id] = lg.fin[cur.second];
    for (const Arc &a : lg.arcs[cur.second]) {
      int ol = 0, ncs = cur.first;
      if (a.il != 0 && !c_arc(cur.first, a.il, &ol, &ncs)) continue;
      const std::pair<int, int> np{ncs, a.next};
      auto it = index.find(np);
      int nid;
      if (it == index.end()) {
        nid = out.AddState();
        index.emplace(np, nid);
        q.push_back(np);
      } else {
        nid = it->second;
      }
      out.arcs[id].push_back({ol, a.ol, a.w, nid});
    }
  }
  return out;
}

}  // namespace gb
}  // namespace rs

// ------------------------------------------------------------------------------------------------------------ comparison (fstequivalent's role)
namespace rs {
namespace gb {

std::vector<std::vector<int32_t>> ReadILabelInfo(const std::string &path) {
  KaldiReader r(path);
  const int32_t n = r.ReadInt32();
  if (n < 0) Fail(path + ": bad ilabel info");
  std::vector<std::vector<int32_t>> info(n);
  for (int32_t i = 0; i < n; i++) r.ReadIntVector(&info[i]);
  return info;
}

// Same transducer up to state numbering, arc order and `delta` on the weights?  Pairs the states breadth first from the start
// states, matching the arcs of a pair after sorting them by (ilabel, olabel, weight); exact for the outputs of the deterministic
// stages (at most one arc per label pair and state).  Returns "" or a description of the first difference.
std::string Isomorphic(const Fst &a_in, const Fst &b_in, float delta) {
  Fst a = a_in, b = b_in;
  if ((a.start < 0) != (b.start < 0)) return "one FST is empty";
  if (a.start < 0) return "";
  if (a.NumStates() != b.NumStates()) return "state counts differ: " + std::to_string(a.NumStates()) + " vs " + std::to_string(b.NumStates());
  if (a.NumArcs() != b.NumArcs()) return "arc counts differ: " + std::to_string(a.NumArcs()) + " vs " + std::to_string(b.NumArcs());
  auto sort_arcs = [](Fst *f) {
    for (auto &v : f->arcs)
      std::sort(v.begin(), v.end(), [](const Arc &x, const Arc &y) { return std::tie(x.il, x.ol, x.w) < std::tie(y.il, y.ol, y.w); });
  };
  sort_arcs(&a);
  sort_arcs(&b);
  std::vector<int> map(a.NumStates(), -1), back(b.NumStates(), -1);
  std::deque<int> q;
  map[a.start] = b.start;
  back[b.start] = a.start;
  q.push_back(a.start);
  auto close = [delta](float x, float y) { return (x == kInf && y == kInf) || std::fabs(x - y) <= delta; };
  while (!q.empty()) {
    const int s = q.front();
    q.pop_front();
    const int t = map[s];
    if (!close(a.fin[s], b.fin[t])) return "final weights differ at state " + std::to_string(s) + "/" + std::to_string(t);
    if (a.arcs[s].size() != b.arcs[t].size()) return "out-degrees differ at state " + std::to_string(s) + "/" + std::to_string(t);
    for (size_t k = 0; k < a.arcs[s].size(); k++) {
      const Arc &x = a.arcs[s][k], &y = b.arcs[t][k];
      if (x.il != y.il || x.ol != y.ol || !close(x.w, y.w))
        return "arc " + std::to_string(k) + " differs at state " + std::to_string(s) + "/" + std::to_string(t) + ": " + std::to_string(x.il) + ":" +
               std::to_string(x.ol) + "/" + std::to_string(x.w) + " vs " + std::to_string(y.il) + ":" + std::to_string(y.ol) + "/" + std::to_string(y.w);
      if (map[x.next] == -1 && back[y.next] == -1) { map[x.next] = y.next; back[y.next] = x.next; q.push_back(x.next); }
      else if (map[x.next] != y.next) return "destinations are paired inconsistently at state " + std::to_string(s) + "/" + std::to_string(t);
    }
  }
  return "";
}

namespace {
// tropical weight of the (input string, output string) pair in f: cheapest accepting path with exactly these labels
float PairWeight(const Fst &f, const std::vector<int32_t> &is, const std::vector<int32_t> &os) {
  const int ni = (int)is.size() + 1, no = (int)os.size() + 1;
  auto idx = [&](int s, int p, int q) { return ((size_t)s * ni + p) * no + q; };
  std::vector<float> dist((size_t)f.NumStates() * ni * no, kInf);
  std::vector<char> inq(dist.size(), 0);
  std::deque<size_t> queue;
  dist[idx(f.start, 0, 0)] = 0.0f;
  queue.push_back(idx(f.start, 0, 0));
  size_t guard = 0;
  while (!queue.empty()) {
    const size_t u = queue.front();
    queue.pop_front();
    inq[u] = 0;
    if (++guard > 200000000) Fail("PairWeight: no convergence (negative epsilon cycle?)");
    const int q = (int)(u % no), p = (int)((u / no) % ni), s = (int)(u / no / ni);
    const float d = dist[u];
    for (const Arc &a : f.arcs[s]) {
      int p2 = p, q2 = q;
      if (a.il != 0) { if (p >= ni - 1 || is[p] != a.il) continue; p2++; }
      if (a.ol != 0) { if (q >= no - 1 || os[q] != a.ol) continue; q2++; }
      const size_t v = idx(a.next, p2, q2);
      if (d + a.w < dist[v] - 1e-7f) {
        dist[v] = d + a.w;
        if (!inq[v]) { inq[v] = 1; queue.push_back(v); }
      }
    }
  }
  float best = kInf;
  for (int s = 0; s < f.NumStates(); s++)
    if (f.fin[s] != kInf) best = std::min(best, dist[idx(s, ni - 1, no - 1)] + f.fin[s]);
  return best;
}
}  // namespace

// Randomised equivalence in the tropical semiring (what `fstequivalent --random=true` does): label pairs of random accepting
// paths of either transducer must weigh the same (within delta) in both.  Returns "" or the first counter-example.
std::string RandEquivalent(const Fst &a, const Fst &b, float delta, int npaths, int max_len, uint64_t seed) {
  if ((a.start < 0) != (b.start < 0)) return "one FST is empty";
  if (a.start < 0) return "";
  auto rnd = [&seed]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(seed >> 33); };
  const Fst *fs[2] = {&a, &b};
  for (int side = 0; side < 2; side++) {
    const Fst &f = *fs[side];
    // arcs to the nearest final state, to finish a walk that got long
    std::vector<int> togo(f.NumStates(), 1 << 30);
    {
      std::vector<std::vector<int>> rev(f.NumStates());
      for (int s = 0; s < f.NumStates(); s++) for (const Arc &x : f.arcs[s]) rev[x.next].push_back(s);
      std::deque<int> q;
      for (int s = 0; s < f.NumStates(); s++) if (f.fin[s] != kInf) { togo[s] = 0; q.push_back(s); }
      while (!q.empty()) { const int s = q.front(); q.pop_front(); for (int p : rev[s]) if (togo[p] > togo[s] + 1) { togo[p] = togo[s] + 1; q.push_back(p); } }
    }
    for (int n = 0; n < npaths; n++) {
      std::vector<int32_t> is, os;
      int s = f.start, len = 0;
      for (;;) {
        const bool fin = f.fin[s] != kInf;
        const auto &arcs = f.arcs[s];
        if (fin && (arcs.empty() || rnd() % (arcs.size() + 1) == 0 || len > max_len)) break;
        if (arcs.empty()) break;
        const Arc *pick = &arcs[rnd() % arcs.size()];
        if (len > max_len) for (const Arc &x : arcs) if (togo[x.next] < togo[pick->next]) pick = &x;
        if (pick->il) is.push_back(pick->il);
        if (pick->ol) os.push_back(pick->ol);
        s = pick->next;
        len++;
        if (len > 4 * max_len + 64) break;
      }
      if (f.fin[s] == kInf) continue;
      const float wa = PairWeight(a, is, os), wb = PairWeight(b, is, os);
      if (!((wa == kInf && wb == kInf) || std::fabs(wa - wb) <= delta * (1.0f + 0.02f * (float)len))) {
        std::string d = "path " + std::to_string(n) + " of " + (side ? "B" : "A") + " weighs " + std::to_string(wa) + " in A and " + std::to_string(wb) + " in B; input";
        for (int32_t x : is) d += " " + std::to_string(x);
        d += " output";
        for (int32_t x : os) d += " " + std::to_string(x);
        return d;
      }
    }
  }
  return "";
}

}  // namespace gb
}  // namespace rs
