// See nnet3_setup.h.  Citations are to /root/reference/kaldi/src.
#include "nnet3_setup.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <sstream>
#include <unordered_map>

namespace rs {

// =============================================================================== glibc rand() / rand_r()

GlibcRand::GlibcRand() {
  // srandom_r(1): stdlib/random_r.c -- the Lehmer sequence 16807 x mod (2^31 - 1) fills the 31-word state of the TYPE_3
  // generator, then 310 outputs are discarded
  int word = 1;
  r_[0] = 1u;
  for (int i = 1; i < 31; i++) {
    const long hi = word / 127773, lo = word % 127773;
    long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = (int)w;
    r_[i] = (unsigned)word;
  }
  for (int i = 0; i < 310; i++) (void)Next();
}

int GlibcRand::Next() {
  r_[f_] += r_[b_];
  const int out = (int)(r_[f_] >> 1);
  f_ = (f_ + 1) % 31;
  b_ = (b_ + 1) % 31;
  return out;
}

int GlibcRand::RandR(unsigned *seed) {
  unsigned next = *seed;
  next = next * 1103515245u + 12345u;
  int result = (int)((next / 65536u) % 2048u);
  next = next * 1103515245u + 12345u;
  result <<= 10;
  result ^= (int)((next / 65536u) % 1024u);
  next = next * 1103515245u + 12345u;
  result <<= 10;
  result ^= (int)((next / 65536u) % 1024u);
  *seed = next;
  return result;
}

void DitherNoise(long offset, int t0, int t1, int win, float *out) {
  GlibcRand g;
  for (long i = 0; i < offset + t0; i++) (void)g.Next();
  for (int t = t0; t < t1; t++) {
    unsigned seed = (unsigned)g.Next() + 27437u;                     // RandomState::RandomState, kaldi-math.cc:69
    float *row = out + (size_t)(t - t0) * win;
    for (int i = 0; i < win; i++) {
      // RandGauss (kaldi-math.h:155-158); g++ draws the logarithm's operand first
      const float u1 = (float)((GlibcRand::RandR(&seed) + 1.0) / (2147483647 + 2.0));
      const float u2 = (float)((GlibcRand::RandR(&seed) + 1.0) / (2147483647 + 2.0));
      row[i] = (float)(sqrtf(-2 * logf(u1)) * cosf(2 * M_PI * u2));
    }
  }
}

namespace {

struct Rng {
  GlibcRand g;
  long calls = 0;
  int Rand() { calls++; return g.Next(); }
  int RandInt(int lo, int hi) {                // kaldi-math.cc:95-119: an empty range draws nothing
    if (hi == lo) return lo;
    return lo + Rand() % (hi + 1 - lo);
  }
};

// =============================================================================== descriptors (nnet-descriptor.cc)

enum Kind { kNode, kConst, kAppend, kSum, kFailover, kIfDefined, kOffset, kSwitch, kScale, kRound, kReplaceIndex };

struct G {
  Kind kind = kNode;
  std::vector<G> ch;
  int v1 = -1, v2 = -1;
  float alpha = 0.0f;
};

std::vector<std::string> Tokenize(const std::string &s) {
  std::vector<std::string> t;
  size_t i = 0;
  while (i < s.size()) {
    const char c = s[i];
    if (std::isspace((unsigned char)c)) { i++; continue; }
    if (c == '(' || c == ')' || c == ',') { t.emplace_back(1, c); i++; continue; }
    size_t b = i;
    while (i < s.size() && !std::isspace((unsigned char)s[i]) && s[i] != '(' && s[i] != ')' && s[i] != ',') i++;
    t.push_back(s.substr(b, i - b));
  }
  return t;
}

struct Parser {
  const std::vector<std::string> &tok;
  const std::vector<std::string> &names;
  const std::string &text;
  size_t p = 0;
  const std::string &Cur() const {
    static const std::string end = "end of input";
    return p < tok.size() ? tok[p] : end;
  }
  void Expect(const char *t) {
    if (Cur() != t) Fail("nnet3 descriptor '" + text + "': expected '" + t + "', got '" + Cur() + "'");
    p++;
  }
  int Int() {
    char *e;
    const long v = std::strtol(Cur().c_str(), &e, 10);
    if (*e != 0 || Cur().empty()) Fail("nnet3 descriptor '" + text + "': expected an integer, got '" + Cur() + "'");
    p++;
    return (int)v;
  }
  float Real() {
    char *e;
    const float v = std::strtof(Cur().c_str(), &e);
    if (*e != 0 || Cur().empty()) Fail("nnet3 descriptor '" + text + "': expected a number, got '" + Cur() + "'");
    p++;
    return v;
  }
  G Parse() {                                                        // GeneralDescriptor::Parse (:597-650)
    static const std::pair<const char *, Kind> reserved[] = {
        {"Append", kAppend}, {"Sum", kSum}, {"Failover", kFailover}, {"IfDefined", kIfDefined}, {"Offset", kOffset},
        {"Switch", kSwitch}, {"Scale", kScale}, {"Const", kConst}, {"Round", kRound}, {"ReplaceIndex", kReplaceIndex}};
    G g;
    bool is_reserved = false;
    for (auto &r : reserved)
      if (Cur() == r.first) { g.kind = r.second; is_reserved = true; }
    if (!is_reserved) {
      for (size_t i = 0; i < names.size(); i++)
        if (names[i] == Cur()) { g.kind = kNode; g.v1 = (int)i; p++; return g; }
      Fail("nnet3 descriptor '" + text + "': expected a descriptor, got '" + Cur() + "'");
    }
    p++;
    Expect("(");
    switch (g.kind) {
      case kAppend: case kSum: case kSwitch:
        g.ch.push_back(Parse());
        while (Cur() == ",") { p++; g.ch.push_back(Parse()); }
        Expect(")");
        break;
      case kFailover:
        g.ch.push_back(Parse());
        Expect(",");
        g.ch.push_back(Parse());
        Expect(")");
        break;
      case kIfDefined:
        g.ch.push_back(Parse());
        Expect(")");
        break;
      case kScale:
        g.alpha = Real();
        Expect(",");
        g.ch.push_back(Parse());
        Expect(")");
        break;
      case kConst:
        g.alpha = Real();
        Expect(",");
        g.v1 = Int();
        Expect(")");
        break;
      case kOffset:
        g.ch.push_back(Parse());
        Expect(",");
        g.v1 = Int();
        g.v2 = 0;
        if (Cur() == ",") { p++; g.v2 = Int(); }
        Expect(")");
        break;
      case kRound:
        g.ch.push_back(Parse());
        Expect(",");
        g.v1 = Int();
        Expect(")");
        break;
      case kReplaceIndex:
        g.ch.push_back(Parse());
        Expect(",");
        if (Cur() == "t") g.v1 = 0; else if (Cur() == "x") g.v1 = 1; else Fail("nnet3 descriptor '" + text + "': expected 't' or 'x'");
        p++;
        Expect(",");
        g.v2 = Int();
        Expect(")");
        break;
      default: break;
    }
    return g;
  }
};

int NumAppendTerms(const G &g) {                                     // (:763-782)
  if (g.kind == kNode || g.kind == kConst) return 1;
  if (g.kind == kAppend) {
    int n = 0;
    for (auto &c : g.ch) n += NumAppendTerms(c);
    return n;
  }
  const int n = NumAppendTerms(g.ch[0]);
  for (size_t i = 1; i < g.ch.size(); i++)
    if (NumAppendTerms(g.ch[i]) != n) Fail("nnet3 descriptor: operands with different numbers of Append terms");
  return n;
}

G AppendTerm(const G &g, int term) {                                 // (:784-814)
  if (g.kind == kNode || g.kind == kConst) { G r = g; r.ch.clear(); return r; }
  if (g.kind == kAppend) {
    for (auto &c : g.ch) {
      const int n = NumAppendTerms(c);
      if (term < n) return AppendTerm(c, term);
      term -= n;
    }
    Fail("nnet3 descriptor: internal error (append term)");
  }
  G r;
  r.kind = g.kind; r.v1 = g.v1; r.v2 = g.v2; r.alpha = g.alpha;
  for (auto &c : g.ch) r.ch.push_back(AppendTerm(c, term));
  return r;
}

void Take(G &d, size_t child) {      // d becomes its child
  G tmp = std::move(d.ch[child]);
  d = std::move(tmp);
}

bool Normalize(G &d) {                                               // GeneralDescriptor::Normalize (:836-975), one pass
  bool changed = false, fall = false;
  const Kind k = d.kind;
  if (k == kOffset) {
    G &child = d.ch[0];
    if (child.kind == kOffset) {
      d.v1 += child.v1;
      d.v2 += child.v2;
      G grand = std::move(child.ch[0]);
      d.ch[0] = std::move(grand);
      changed = true;
      fall = true;
    } else if (d.v1 == 0 && d.v2 == 0) {
      Take(d, 0);
      changed = true;
    } else {
      fall = true;
    }
  }
  if (fall || k == kSwitch || k == kRound || k == kReplaceIndex) {
    G &child = d.ch[0];
    if (child.kind == kSum || child.kind == kFailover || child.kind == kIfDefined) {
      if (d.ch.size() > 1) Fail("nnet3 descriptor: Sum(), Failover() or IfDefined() inside Switch() cannot be normalised");
      G c = std::move(d.ch[0]);
      for (auto &gc : c.ch) {
        G w;
        w.kind = d.kind; w.v1 = d.v1; w.v2 = d.v2; w.alpha = d.alpha;
        w.ch.push_back(std::move(gc));
        gc = std::move(w);
      }
      d.kind = c.kind; d.v1 = c.v1; d.v2 = c.v2;
      d.ch = std::move(c.ch);
      changed = true;
    }
  } else if (k == kSum) {
    if (d.ch.size() == 1) {
      Take(d, 0);
      changed = true;
    } else if (d.ch.size() > 2) {
      G rest;
      rest.kind = kSum;
      for (size_t i = 1; i < d.ch.size(); i++) rest.ch.push_back(std::move(d.ch[i]));
      d.ch.resize(1);
      d.ch.push_back(std::move(rest));
      changed = true;
    }
  } else if (k == kScale) {
    G &child = d.ch[0];
    if (child.kind == kOffset || child.kind == kReplaceIndex || child.kind == kRound) {
      std::swap(d.kind, child.kind);
      std::swap(d.alpha, child.alpha);
      std::swap(d.v1, child.v1);
      std::swap(d.v2, child.v2);
      changed = true;
    } else if (child.kind == kSum) {
      G c = std::move(d.ch[0]);
      d.ch.clear();
      for (auto &gc : c.ch) {
        G w;
        w.kind = kScale; w.alpha = d.alpha;
        w.ch.push_back(std::move(gc));
        d.ch.push_back(std::move(w));
      }
      d.kind = kSum;
      d.alpha = 0.0f;
      changed = true;
    } else if (child.kind == kScale) {
      d.alpha *= child.alpha;
      G grand = std::move(child.ch[0]);
      d.ch[0] = std::move(grand);
      changed = true;
    } else if (child.kind != kNode) {
      Fail("nnet3 descriptor: unhandled Scale() nesting (push Scale() inside the other expressions)");
    }
  }
  for (auto &c : d.ch) changed = changed || Normalize(c);            // the reference short-circuits the same way
  return changed;
}

std::string Print(const G &g, const std::vector<std::string> &names) {
  char buf[64];
  switch (g.kind) {
    case kNode: return names[g.v1];
    case kConst: std::snprintf(buf, sizeof buf, "Const(%g, %d)", g.alpha, g.v1); return buf;
    case kScale: std::snprintf(buf, sizeof buf, "Scale(%g, ", g.alpha); return buf + Print(g.ch[0], names) + ")";
    case kOffset: return "Offset(" + Print(g.ch[0], names) + ", " + std::to_string(g.v1) + (g.v2 != 0 ? ", " + std::to_string(g.v2) : "") + ")";
    case kRound: return "Round(" + Print(g.ch[0], names) + ", " + std::to_string(g.v1) + ")";
    case kReplaceIndex: return "ReplaceIndex(" + Print(g.ch[0], names) + ", " + (g.v1 == 0 ? "t" : "x") + ", " + std::to_string(g.v2) + ")";
    default: break;
  }
  std::string s = g.kind == kAppend ? "Append(" : g.kind == kSum ? "Sum(" : g.kind == kFailover ? "Failover(" : g.kind == kIfDefined ? "IfDefined(" : "Switch(";
  for (size_t i = 0; i < g.ch.size(); i++) s += (i ? ", " : "") + Print(g.ch[i], names);
  return s + ")";
}

G ParseDescriptor(const std::string &text, const std::vector<std::string> &names) {   // Descriptor::Parse (:490-507)
  const std::vector<std::string> tok = Tokenize(text);
  Parser ps{tok, names, text};
  G g = ps.Parse();
  if (ps.p != tok.size()) Fail("nnet3 descriptor '" + text + "': expected end of input, got '" + ps.Cur() + "'");
  const int n = NumAppendTerms(g);
  G out;
  if (n == 1) out = AppendTerm(g, 0);
  else {
    out.kind = kAppend;
    for (int i = 0; i < n; i++) out.ch.push_back(AppendTerm(g, i));
  }
  while (Normalize(out)) {}
  return out;
}

struct Index { int n = 0, t = 0, x = 0; };
struct Cindex {
  int node = 0;
  Index i;
  bool operator==(const Cindex &o) const { return node == o.node && i.n == o.i.n && i.t == o.i.t && i.x == o.i.x; }
};
struct CindexHash {
  size_t operator()(const Cindex &c) const {
    uint64_t h = (uint64_t)(uint32_t)c.node * 0x9E3779B97F4A7C15ull;
    h ^= (uint64_t)(uint32_t)c.i.t * 0xC2B2AE3D27D4EB4Full + (uint64_t)(uint32_t)c.i.n * 1000003ull + (uint64_t)(uint32_t)c.i.x * 7919ull;
    return (size_t)(h ^ (h >> 29));
  }
};

int MathMod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

Cindex MapToInput(const G *g, Index ind) {                           // ForwardingDescriptor::MapToInput (:88-234)
  for (;;) {
    switch (g->kind) {
      case kNode: return Cindex{g->v1, ind};
      case kScale: g = &g->ch[0]; break;
      case kOffset: ind.t += g->v1; ind.x += g->v2; g = &g->ch[0]; break;
      case kRound: ind.t -= MathMod(ind.t, g->v1); g = &g->ch[0]; break;
      case kReplaceIndex: if (g->v1 == 0) ind.t = g->v2; else ind.x = g->v2; g = &g->ch[0]; break;
      case kSwitch: g = &g->ch[MathMod(ind.t, (int)g->ch.size())]; break;
      default: Fail("nnet3 descriptor: internal error (forwarding descriptor)");
    }
  }
}

int Lcm(int a, int b) { return a / std::gcd(a, b) * b; }

int FwdModulus(const G &g) {
  if (g.kind == kNode) return 1;
  if (g.kind == kRound) return g.v1;
  if (g.kind == kSwitch) {
    int m = (int)g.ch.size();
    for (auto &c : g.ch) m = Lcm(m, FwdModulus(c));
    return m;
  }
  return FwdModulus(g.ch[0]);
}
int SumModulus(const G &g) {
  if (g.kind == kConst) return 1;
  if (g.kind == kIfDefined) return SumModulus(g.ch[0]);
  if (g.kind == kSum || g.kind == kFailover) return Lcm(SumModulus(g.ch[0]), SumModulus(g.ch[1]));
  return FwdModulus(g);
}

void SumDeps(const G &g, const Index &ind, std::vector<Cindex> *out) {          // SumDescriptor::GetDependencies
  if (g.kind == kConst) return;
  if (g.kind == kIfDefined) return SumDeps(g.ch[0], ind, out);
  if (g.kind == kSum || g.kind == kFailover) { SumDeps(g.ch[0], ind, out); SumDeps(g.ch[1], ind, out); return; }
  out->push_back(MapToInput(&g, ind));
}

using CSet = std::function<bool(const Cindex &)>;

bool SumComputable(const G &g, const Index &ind, const CSet &cs, std::vector<Cindex> *used) {   // (:303-401)
  if (g.kind == kConst) return true;
  if (g.kind == kIfDefined) {
    std::vector<Cindex> tmp;
    if (SumComputable(g.ch[0], ind, cs, used ? &tmp : nullptr) && used) used->insert(used->end(), tmp.begin(), tmp.end());
    return true;
  }
  if (g.kind == kSum || g.kind == kFailover) {
    std::vector<Cindex> u1, u2;
    const bool c1 = SumComputable(g.ch[0], ind, cs, used ? &u1 : nullptr), c2 = SumComputable(g.ch[1], ind, cs, used ? &u2 : nullptr);
    if (g.kind == kSum) {
      if (!(c1 && c2)) return false;
      if (used) { used->insert(used->end(), u1.begin(), u1.end()); used->insert(used->end(), u2.begin(), u2.end()); }
      return true;
    }
    if (c1) { if (used) used->insert(used->end(), u1.begin(), u1.end()); return true; }
    if (c2) { if (used) used->insert(used->end(), u2.begin(), u2.end()); return true; }
    return false;
  }
  const Cindex c = MapToInput(&g, ind);
  const bool ok = cs(c);
  if (ok && used) used->push_back(c);
  return ok;
}

struct Desc {
  G g;
  size_t NumParts() const { return g.kind == kAppend ? g.ch.size() : 1; }
  const G &Part(size_t i) const { return g.kind == kAppend ? g.ch[i] : g; }
  void Deps(const Index &ind, std::vector<Cindex> *out) const {
    out->clear();
    for (size_t i = 0; i < NumParts(); i++) SumDeps(Part(i), ind, out);
  }
  bool Computable(const Index &ind, const CSet &cs, std::vector<Cindex> *used) const {   // (:556-571)
    if (used) used->clear();
    for (size_t i = 0; i < NumParts(); i++)
      if (!SumComputable(Part(i), ind, cs, used)) { if (used) used->clear(); return false; }
    return true;
  }
  int Modulus() const {
    int m = 1;
    for (size_t i = 0; i < NumParts(); i++) m = Lcm(m, SumModulus(Part(i)));
    return m;
  }
  // ModelCollapser::DescriptorIsCollapsible (nnet-utils.cc:1551-1565)
  int CollapsibleNode() const {
    int ans = -2;
    for (size_t i = 0; i < NumParts(); i++) {
      const G *p = &Part(i);
      int n = -1;
      if (p->kind != kIfDefined && p->kind != kSum && p->kind != kFailover && p->kind != kConst) {
        if (p->kind == kOffset) p = &p->ch[0];
        if (p->kind == kScale && p->ch[0].kind == kNode) p = &p->ch[0];     // a scaled node is still a SimpleForwardingDescriptor
        if (p->kind == kNode) n = p->v1;
      }
      if (ans == -2) ans = n;
      else if (ans != -1 && n != ans) ans = -1;
    }
    return ans == -2 ? -1 : ans;
  }
  void MaxAbsOffset(int *m) const { Walk(g, m); }
  static void Walk(const G &g, int *m) {
    if (g.kind == kOffset) *m = std::max(*m, std::abs(g.v1));
    for (auto &c : g.ch) Walk(c, m);
  }
};

// =============================================================================== node list (nnet-nnet.cc:189-460)

std::map<std::string, std::string> Fields(const std::string &line, std::string *first) {
  std::map<std::string, std::string> kv;
  std::istringstream is(line);
  is >> *first;
  std::string rest;
  std::getline(is, rest);
  std::vector<std::pair<size_t, size_t>> keys;
  size_t i = 0;
  while (i < rest.size()) {
    if (std::isspace((unsigned char)rest[i])) { i++; continue; }
    const size_t b = i;
    while (i < rest.size() && (std::isalnum((unsigned char)rest[i]) || rest[i] == '-' || rest[i] == '_')) i++;
    if (i < rest.size() && rest[i] == '=' && i > b && (b == 0 || std::isspace((unsigned char)rest[b - 1]))) keys.emplace_back(b, i);
    while (i < rest.size() && !std::isspace((unsigned char)rest[i])) i++;
  }
  for (size_t k = 0; k < keys.size(); k++) {
    const size_t vb = keys[k].second + 1, ve = k + 1 < keys.size() ? keys[k + 1].first : rest.size();
    std::string v = rest.substr(vb, ve - vb);
    while (!v.empty() && std::isspace((unsigned char)v.back())) v.pop_back();
    kv[rest.substr(keys[k].first, keys[k].second - keys[k].first)] = v;
  }
  return kv;
}

struct Node {
  enum Type { kInput, kDescriptor, kComponent, kDimRange } type = kInput;
  std::string name, line;       // line: the original config line (input / output / dim-range nodes are written back verbatim)
  Desc desc;
  int comp = -1, src = -1;
  bool component_input = false; // the hidden "<name>_input" descriptor of a component node
};

struct Net {
  std::vector<Node> nodes;
  std::vector<std::string> *comp_names;
  std::vector<Component> *comps;

  std::vector<std::string> Names() const {
    std::vector<std::string> n;
    for (auto &x : nodes) n.push_back(x.name);
    return n;
  }
  int Find(const std::string &name) const {
    for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].name == name) return (int)i;
    return -1;
  }
  bool IsOutput(int i) const { return nodes[i].type == Node::kDescriptor && (i + 1 == (int)nodes.size() || nodes[i + 1].type != Node::kComponent); }
  int Modulus() const {
    int m = 1;
    for (auto &n : nodes) if (n.type == Node::kDescriptor) m = Lcm(m, n.desc.Modulus());
    return m;
  }
  const std::string &CType(int c) const { return (*comps)[c].type; }
  const std::vector<int32_t> *TimeOffsets(int c) const {
    const Component &k = (*comps)[c];
    if (k.type != "TdnnComponent") return nullptr;
    auto it = k.iv.find("<TimeOffsets>");
    if (it == k.iv.end()) Fail("nnet3: TdnnComponent without <TimeOffsets>");
    return &it->second;
  }
  int CompIndex(const std::string &name) const {
    auto it = std::find(comp_names->begin(), comp_names->end(), name);
    return it == comp_names->end() ? -1 : (int)(it - comp_names->begin());
  }
};

Net BuildNet(const std::vector<std::string> &cfg, std::vector<std::string> *comp_names, std::vector<Component> *comps) {
  Net net;
  net.comp_names = comp_names;
  net.comps = comps;
  std::vector<std::map<std::string, std::string>> kvs;
  for (auto &l : cfg) {
    if (l.find_first_not_of(" \t") == std::string::npos || l[l.find_first_not_of(" \t")] == '#') continue;
    std::string first;
    auto kv = Fields(l, &first);
    Node n;
    n.line = l;
    if (first == "input-node") { n.type = Node::kInput; n.name = kv["name"]; }
    else if (first == "component-node") {
      Node d;
      d.type = Node::kDescriptor; d.name = kv["name"] + "_input"; d.component_input = true; d.line = l;
      net.nodes.push_back(d);
      kvs.push_back(kv);
      n.type = Node::kComponent; n.name = kv["name"];
    } else if (first == "output-node") { n.type = Node::kDescriptor; n.name = kv["name"]; }
    else if (first == "dim-range-node") { n.type = Node::kDimRange; n.name = kv["name"]; }
    else if (first == "component") continue;
    else Fail("nnet3: unsupported config line: " + l);
    net.nodes.push_back(n);
    kvs.push_back(kv);
  }
  const std::vector<std::string> names = net.Names();
  for (size_t i = 0; i < net.nodes.size(); i++) {
    Node &n = net.nodes[i];
    if (n.type == Node::kDescriptor) n.desc.g = ParseDescriptor(kvs[i]["input"], names);
    else if (n.type == Node::kComponent) {
      n.comp = net.CompIndex(kvs[i]["component"]);
      if (n.comp < 0) Fail("nnet3: component-node " + n.name + " refers to unknown component " + kvs[i]["component"]);
    } else if (n.type == Node::kDimRange) {
      n.src = net.Find(kvs[i]["input-node"]);
      if (n.src < 0) Fail("nnet3: dim-range-node " + n.name + " has unknown input-node");
    }
  }
  return net;
}

// =============================================================================== CollapseModel (nnet-utils.cc:1459-2116)

bool IsAffine(const std::string &t) { return t == "AffineComponent" || t == "NaturalGradientAffineComponent"; }   // dynamic_cast<AffineComponent*>

const MatF &Linear(const Component &c, const std::string &name) {
  auto it = c.m.find("<LinearParams>");
  if (it == c.m.end()) Fail("nnet3: component " + name + " has no <LinearParams>");
  return it->second;
}
std::vector<float> Bias(const Component &c, int rows) {
  auto it = c.v.find("<BiasParams>");
  if (it == c.v.end() || it->second.empty()) return std::vector<float>(rows, 0.0f);
  return it->second;
}

struct Collapser {
  Net &net;
  Rng &rng;

  int Add(const std::string &name, Component c) {
    net.comp_names->push_back(name);
    net.comps->push_back(std::move(c));
    return (int)net.comps->size() - 1;
  }

  // CollapseComponentsAffine (:1777-1856): y = W2 (blockdiag(W1) x + b1 repeated) + b2
  int Affine(int c1, int c2) {
    const std::string &t1 = net.CType(c1), &t2 = net.CType(c2);
    if (!IsAffine(t2) || (t1 != "FixedAffineComponent" && !IsAffine(t1))) return -1;
    const std::string name = (*net.comp_names)[c1] + "." + (*net.comp_names)[c2];
    const int have = net.CompIndex(name);
    if (have >= 0) return have;
    const MatF &W1 = Linear((*net.comps)[c1], (*net.comp_names)[c1]);
    const int in1 = W1.cols, out1 = W1.rows;
    if (in1 > out1) return -1;                     // dimension-reducing first transform: left alone
    const MatF W2 = Linear((*net.comps)[c2], (*net.comp_names)[c2]);
    const int in2 = W2.cols, out2 = W2.rows;
    if (in2 % out1 != 0) Fail("nnet3: cannot collapse " + name + ": input dimension is not a multiple of the first transform's output");
    const int mult = in2 / out1;
    const std::vector<float> b1 = Bias((*net.comps)[c1], out1), b2 = Bias((*net.comps)[c2], out2);
    const MatF W1c = W1;
    rng.Rand();    // AffineComponent::Init(new_input_dim, new_output_dim, 0.0, 0.0): linear_params_.SetRandn() -> RandomState()
    rng.Rand();    //                                                               bias_params_.SetRandn()   -> RandomState()
    Component nc;
    nc.type = "AffineComponent";
    MatF &W = nc.m["<LinearParams>"];
    W.Resize(out2, mult * in1);
    std::vector<float> &b = nc.v["<BiasParams>"];
    b.resize(out2);
    std::vector<double> acc(in1);
    for (int o = 0; o < out2; o++) {
      double bo = b2[o];
      for (int i = 0; i < mult; i++) {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int j = 0; j < out1; j++) {
          const double w2 = W2(o, i * out1 + j);
          bo += w2 * b1[j];
          const float *w1 = &W1c.d[(size_t)j * in1];
          for (int k = 0; k < in1; k++) acc[k] += w2 * w1[k];
        }
        for (int k = 0; k < in1; k++) W(o, i * in1 + k) = (float)acc[k];
      }
      b[o] = (float)bo;
    }
    return Add(name, std::move(nc));
  }

  // CollapseComponentsScale (:1872-1906)
  int Scale(int c1, int c2) {
    if (!IsAffine(net.CType(c1)) || net.CType(c2) != "FixedScaleComponent") return -1;
    auto sit = (*net.comps)[c2].v.find("<Scales>");
    if (sit == (*net.comps)[c2].v.end()) return -1;
    const MatF &W1 = Linear((*net.comps)[c1], (*net.comp_names)[c1]);
    if (W1.rows != (int)sit->second.size()) return -1;
    const std::string name = (*net.comp_names)[c1] + "." + (*net.comp_names)[c2];
    const int have = net.CompIndex(name);
    if (have >= 0) return have;
    Component nc = (*net.comps)[c1];
    const std::vector<float> scales = sit->second;
    MatF &W = nc.m["<LinearParams>"];
    std::vector<float> b = Bias(nc, W.rows);
    for (int r = 0; r < W.rows; r++) {
      b[r] *= scales[r];
      for (int c = 0; c < W.cols; c++) W(r, c) *= scales[r];
    }
    nc.v["<BiasParams>"] = b;
    return Add(name, std::move(nc));
  }

  // OptimizeNode (:1640-1685) with CollapseComponents (:1505-1527) under CollapseModelConfig(): affine, then scale
  bool OptimizeNode(int i) {
    auto &nodes = net.nodes;
    if (nodes[i].type != Node::kDescriptor || i + 1 >= (int)nodes.size() || nodes[i + 1].type != Node::kComponent) return false;
    const int src = nodes[i].desc.CollapsibleNode();
    if (src < 0 || nodes[src].type != Node::kComponent) return false;
    int combined = Affine(nodes[src].comp, nodes[i + 1].comp);
    if (combined == -1) combined = Scale(nodes[src].comp, nodes[i + 1].comp);
    if (combined == -1) return false;
    nodes[i + 1].comp = combined;
    // ReplaceNodeInDescriptor (:1571-1601): the bypassed node's name is replaced by the text of its own input, re-parsed
    const std::vector<std::string> names = net.Names();
    std::vector<std::string> fake = names;
    fake[src] = Print(nodes[src - 1].desc.g, names);
    nodes[i].desc.g = ParseDescriptor(Print(nodes[i].desc.g, fake), names);
    return true;
  }

  void Run() {                                                       // Collapse (:1464-1489)
    bool changed = true;
    for (int iter = 0; changed; iter++) {
      changed = false;
      for (int i = 0; i < (int)net.nodes.size(); i++)
        if (OptimizeNode(i)) changed = true;
      if (iter >= 10) Fail("KALDI_ERR: Something went wrong collapsing model.");
    }
  }
};

// nnet-graph / nnet-nnet.cc:932-949: nodes no output node depends on
std::vector<bool> ReachableFromOutputs(const Net &net) {
  std::vector<bool> seen(net.nodes.size(), false);
  std::vector<int> stack;
  for (int i = 0; i < (int)net.nodes.size(); i++)
    if (net.IsOutput(i)) { seen[i] = true; stack.push_back(i); }
  while (!stack.empty()) {
    const int i = stack.back();
    stack.pop_back();
    std::vector<int> deps;
    const Node &n = net.nodes[i];
    if (n.type == Node::kDescriptor) {
      std::function<void(const G &)> walk = [&](const G &g) {
        if (g.kind == kNode) deps.push_back(g.v1);
        for (auto &c : g.ch) walk(c);
      };
      walk(n.desc.g);
    } else if (n.type == Node::kComponent) deps.push_back(i - 1);
    else if (n.type == Node::kDimRange) deps.push_back(n.src);
    for (int d : deps)
      if (!seen[d]) { seen[d] = true; stack.push_back(d); }
  }
  return seen;
}

// =============================================================================== computation graph (nnet-computation-graph.cc)

enum { kUnknown = 0, kComputable = 1, kNotComputable = 2 };

struct GraphBuilder {
  const Net &net;
  Rng &rng;
  std::vector<Cindex> cindexes;
  std::unordered_map<Cindex, int, CindexHash> ids;
  std::vector<char> is_input, computable, queued, deps_done;
  std::vector<int> usable;
  std::vector<std::vector<int>> deps, depend_on_this;
  std::vector<int> segment_ends, cur, nxt;
  int distance = -1;

  GraphBuilder(const Net &n, Rng &r) : net(n), rng(r) {}

  int Get(const Cindex &c, bool input, bool *is_new) {               // ComputationGraph::GetCindexId (:29-47)
    auto it = ids.find(c);
    if (it != ids.end()) { *is_new = false; return it->second; }
    const int id = (int)cindexes.size();
    ids.emplace(c, id);
    cindexes.push_back(c);
    is_input.push_back(input);
    deps.emplace_back();
    *is_new = true;
    return id;
  }
  int Lookup(const Cindex &c) const { auto it = ids.find(c); return it == ids.end() ? -1 : it->second; }
  void AddInfo() {
    depend_on_this.emplace_back();
    computable.push_back(kUnknown);
    usable.push_back(0);
    queued.push_back(0);
    deps_done.push_back(0);
  }

  using Io = std::vector<std::pair<std::string, std::vector<Index>>>;

  void Compute(const Io &inputs, const Io &outputs) {                // (:462-493) with AddInputs / AddOutputs (:262-314)
    const int start = (int)cindexes.size();
    bool is_new;
    for (auto &io : inputs) {
      const int n = net.Find(io.first);
      if (n < 0) Fail("nnet3: network has no input named " + io.first);
      for (auto &ind : io.second) {
        Get(Cindex{n, ind}, true, &is_new);
        AddInfo();
        computable.back() = kComputable;
      }
    }
    for (auto &io : outputs) {
      const int n = net.Find(io.first);
      if (n < 0) Fail("nnet3: network has no output named " + io.first);
      for (auto &ind : io.second) {
        const int id = Get(Cindex{n, ind}, false, &is_new);
        AddInfo();
        usable.back() = 1;
        queued.back() = 1;
        nxt.push_back(id);
      }
    }
    distance = 0;
    cur.swap(nxt);
    while (distance < 10000) {
      OneIter();
      if (rng.RandInt(1, distance + 1) == 1) Check(start);
      if (cur.empty()) break;
    }
    if (distance >= 10000) Fail("KALDI_ERR: Loop detected while building computation graph (bad network topology?)");
    if (rng.RandInt(1, 2 * ((int)segment_ends.size() + 1)) == 1) Check(start);
  }

  void Check(int start) {                                            // (:496-570): the draws of the self check
    const int num = (int)cindexes.size();
    for (int i = start; i < num; i += 1 + rng.RandInt(0, num / 100)) (void)rng.RandInt(0, i);
  }

  void OneIter() {                                                   // BuildGraphOneIter (:893-913)
    while (!cur.empty()) {
      const int i = cur.back();
      cur.pop_back();
      queued[i] = 0;
      if (!deps_done[i] && usable[i] != 0) {
        deps_done[i] = 1;
        AddDeps(i);
        if (!queued[i]) { queued[i] = 1; nxt.push_back(i); }
      } else if (computable[i] == kUnknown) {
        UpdateComputable(i);
      }
    }
    cur.swap(nxt);
    distance++;
  }

  void InputCindexes(int i, std::vector<Cindex> *out) const {
    const Cindex c = cindexes[i];
    const Node &node = net.nodes[c.node];
    out->clear();
    if (node.type == Node::kDescriptor) node.desc.Deps(c.i, out);
    else if (node.type == Node::kComponent) {
      const std::vector<int32_t> *offs = net.TimeOffsets(node.comp);
      if (!offs) out->push_back(Cindex{c.node - 1, c.i});
      else for (int o : *offs) out->push_back(Cindex{c.node - 1, Index{c.i.n, c.i.t + o, c.i.x}});
    } else if (node.type == Node::kDimRange) out->push_back(Cindex{node.src, c.i});
  }

  void AddDeps(int i) {                                              // AddDependencies (:624-718)
    std::vector<Cindex> in;
    InputCindexes(i, &in);
    std::vector<int> this_dep;
    bool is_new;
    for (auto &c : in) {
      const int d = Get(c, false, &is_new);
      this_dep.push_back(d);
      if (is_new) {
        AddInfo();
        queued.back() = 1;
        nxt.push_back(d);
      }
    }
    std::sort(this_dep.begin(), this_dep.end());
    this_dep.erase(std::unique(this_dep.begin(), this_dep.end()), this_dep.end());
    deps[i] = this_dep;
    for (int d : this_dep) {
      depend_on_this[d].push_back(i);
      IncUsable(d);
    }
  }

  void IncUsable(int i) {                                            // (:856-874)
    if (usable[i]++ == 0 && computable[i] != kNotComputable) {
      for (int d : deps[i]) IncUsable(d);
      if (computable[i] == kUnknown && !queued[i]) { queued[i] = 1; nxt.push_back(i); }
    }
  }
  void DecUsable(int i) {                                            // (:877-890)
    if (--usable[i] == 0 && computable[i] != kNotComputable)
      for (int d : deps[i]) DecUsable(d);
  }

  CSet Set(bool treat_unknown) const {
    return [this, treat_unknown](const Cindex &c) {
      const int i = Lookup(c);
      if (i < 0) return false;
      return computable[i] == kComputable || (treat_unknown && computable[i] == kUnknown);
    };
  }

  bool ComponentComputable(int node_i, const Index &ind, const CSet &cs, std::vector<Cindex> *used) const {
    const std::vector<int32_t> *offs = net.TimeOffsets(net.nodes[node_i].comp);
    if (used) used->clear();
    auto one = [&](const Cindex &c) {
      if (!cs(c)) return false;
      if (used) used->push_back(c);
      return true;
    };
    if (!offs) return one(Cindex{node_i - 1, ind});
    for (int o : *offs)
      if (!one(Cindex{node_i - 1, Index{ind.n, ind.t + o, ind.x}})) return false;
    return true;
  }

  int ComputeComputable(int i) const {                               // ComputeComputableInfo (:721-785)
    const Cindex c = cindexes[i];
    const Node &node = net.nodes[c.node];
    switch (node.type) {
      case Node::kDescriptor:
        if (node.desc.Computable(c.i, Set(false), nullptr)) return kComputable;
        if (!node.desc.Computable(c.i, Set(true), nullptr)) return kNotComputable;
        return kUnknown;
      case Node::kComponent:
        if (ComponentComputable(c.node, c.i, Set(false), nullptr)) return kComputable;
        if (!ComponentComputable(c.node, c.i, Set(true), nullptr)) return kNotComputable;
        return kUnknown;
      case Node::kDimRange: {
        const int j = Lookup(Cindex{node.src, c.i});
        return j >= 0 ? computable[j] : kUnknown;
      }
      default:
        return is_input[i] ? kComputable : kNotComputable;
    }
  }

  void UpdateComputable(int i) {                                     // UpdateComputableInfo (:813-853)
    if (usable[i] == 0) return;
    const int out = ComputeComputable(i);
    computable[i] = (char)out;
    if (out != kUnknown) {
      for (int o : depend_on_this[i])
        if (computable[o] == kUnknown && !queued[o]) { queued[o] = 1; nxt.push_back(o); }
      if (out == kNotComputable && usable[i] != 0)
        for (int d : deps[i]) DecUsable(d);
    }
  }

  std::vector<bool> OutputComputable(const std::string &name, const std::vector<Index> &idx) const {   // GetComputableInfo (:787-810)
    const int n = net.Find(name);
    std::vector<bool> r;
    for (auto &ind : idx) r.push_back(computable[Lookup(Cindex{n, ind})] == kComputable);
    return r;
  }

  void PruneDeps(int i) {                                            // PruneDependencies (:352-447)
    if (computable[i] == kNotComputable || usable[i] == 0) { deps[i].clear(); return; }
    if (computable[i] != kComputable) Fail("nnet3: internal error (pruning a cindex of unknown computability)");
    const Cindex c = cindexes[i];
    const Node &node = net.nodes[c.node];
    if (node.type == Node::kDimRange || node.type == Node::kInput) return;
    std::vector<Cindex> used;
    const bool ok = node.type == Node::kDescriptor ? node.desc.Computable(c.i, Set(false), &used) : ComponentComputable(c.node, c.i, Set(false), &used);
    if (!ok) Fail("nnet3: internal error (computable cindex is not computable)");
    std::vector<int> u;
    for (auto &x : used) u.push_back(Lookup(x));
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    deps[i] = u;
  }

  void Prune() {                                                     // (:572-622) with ComputeRequiredArray (:916-958), Renumber (:53-122)
    const int start = segment_ends.empty() ? 0 : segment_ends.back(), num = (int)cindexes.size();
    for (int i = start; i < num; i++) PruneDeps(i);
    std::vector<char> required(num - start, 0);
    std::vector<int> queue;
    for (int c = start; c < num; c++)
      if (net.IsOutput(cindexes[c].node)) { required[c - start] = 1; queue.push_back(c); }
    while (!queue.empty()) {
      const int c = queue.back();
      queue.pop_back();
      for (int d : deps[c])
        if (d >= start && !required[d - start]) { required[d - start] = 1; queue.push_back(d); }
    }
    std::vector<int> old2new(num - start, -1), new2old;
    for (int c = start; c < num; c++)
      if (required[c - start] || is_input[c]) {
        if (computable[c] != kComputable) Fail("KALDI_ASSERT: You are calling Prune when not everything is computable.");
        old2new[c - start] = start + (int)new2old.size();
        new2old.push_back(c);
      }
    if ((int)new2old.size() != num - start) {
      for (int c = start; c < num; c++)
        if (old2new[c - start] < 0) ids.erase(cindexes[c]);
      for (size_t k = 0; k < new2old.size(); k++) {
        const int o = new2old[k], n = start + (int)k;
        ids[cindexes[o]] = n;
        std::vector<int> d;
        for (int x : deps[o]) d.push_back(x < start ? x : old2new[x - start]);
        cindexes[n] = cindexes[o];
        is_input[n] = is_input[o];
        deps[n] = std::move(d);
      }
      const size_t n2 = start + new2old.size();
      cindexes.resize(n2);
      is_input.resize(n2);
      deps.resize(n2);
    }
    const size_t n2 = cindexes.size();
    computable.assign(computable.begin(), computable.begin() + start);
    computable.resize(n2, kComputable);
    usable.resize(start);
    usable.resize(n2, 1);
    queued.resize(start);
    queued.resize(n2, 0);
    deps_done.resize(start);
    deps_done.resize(n2, 0);
    depend_on_this.resize(start);
    depend_on_this.resize(n2);
    segment_ends.push_back((int)n2);
  }
};

// ComputeSimpleNnetContext (nnet-utils.cc:92-197)
void SimpleNnetContext(const Net &net, Rng &rng, int *left, int *right) {
  const int modulus = net.Modulus();
  const bool has_ivector = net.Find("ivector") != -1;
  for (int window = 40; window < 800; window *= 2) {
    std::vector<int> lefts, rights;
    bool ok = true;
    for (int start = 0; start <= modulus && ok; start++) {
      const int n = rng.Rand() % 10;
      std::vector<Index> idx, iv;
      for (int t = start; t < start + window; t++) idx.push_back(Index{n, t, 0});
      GraphBuilder::Io in{{"input", idx}}, out{{"output", idx}};
      if (has_ivector) {
        for (int t = start - modulus; t < start + window; t++) iv.push_back(Index{n, t, 0});
        in.push_back({"ivector", iv});
      }
      GraphBuilder b(net, rng);
      b.Compute(in, out);
      const std::vector<bool> okv = b.OutputComputable("output", idx);
      const int first_ok = (int)(std::find(okv.begin(), okv.end(), true) - okv.begin());
      const int first_not = (int)(std::find(okv.begin() + first_ok, okv.end(), false) - okv.begin());
      if (first_ok == window || first_not <= first_ok) { ok = false; break; }
      lefts.push_back(first_ok);
      rights.push_back(window - first_not);
    }
    if (!ok) continue;
    *left = *std::max_element(lefts.begin(), lefts.end());
    *right = *std::max_element(rights.begin(), rights.end());
    return;
  }
  Fail("KALDI_ERR: Failure in ComputeSimpleNnetContext (perhaps not a simple nnet?)");
}

// ModifyNnetIvectorPeriod (nnet-compile-looped.cc:28-79)
void ModifyIvectorPeriod(Net &net, int period) {
  const std::vector<std::string> names = net.Names();
  for (auto &n : net.nodes) {
    if (!n.component_input) continue;
    std::string text = Print(n.desc.g, names);
    const size_t pos = text.find("ReplaceIndex(");
    if (pos == std::string::npos) continue;
    const size_t comma = text.find(", t, 0)", pos);
    if (comma == std::string::npos) Fail("KALDI_ERR: Could not process the ReplaceIndex expression in: " + text);
    const std::string inner = text.substr(pos + 13, comma - (pos + 13));
    text = text.substr(0, pos) + "Round(" + inner + ", " + std::to_string(period) + ")" + text.substr(comma + 7);
    n.desc.g = ParseDescriptor(text, names);
  }
}

}  // namespace

Nnet3SetupResult Nnet3Setup(const std::vector<std::string> &config_lines, std::vector<std::string> *component_names,
                            std::vector<Component> *components, int frames_per_chunk, int extra_left_context_initial,
                            int frame_subsampling_factor) {
  Nnet3SetupResult res;
  Rng rng;
  Net net = BuildNet(config_lines, component_names, components);
  int l = 0, r = 0;
  SimpleNnetContext(net, rng, &l, &r);                               // AmNnetSimple::Read -> SetContext (am-nnet-simple.cc:48,80-87)
  Collapser{net, rng}.Run();                                         // CollapseModel(CollapseModelConfig(), &nnet)
  // the collapsed network, as config lines (orphans dropped: Nnet::RemoveOrphanNodes, nnet-nnet.cc:932-949)
  {
    const std::vector<bool> keep = ReachableFromOutputs(net);
    const std::vector<std::string> names = net.Names();
    for (size_t i = 0; i < net.nodes.size(); i++) {
      const Node &n = net.nodes[i];
      if (n.component_input) continue;
      if (n.type == Node::kInput) { res.config.push_back(n.line); continue; }
      if (!keep[i]) continue;
      if (n.type == Node::kComponent)
        res.config.push_back("component-node name=" + n.name + " component=" + (*component_names)[n.comp] + " input=" + Print(net.nodes[i - 1].desc.g, names));
      else
        res.config.push_back(n.line);
    }
  }
  SimpleNnetContext(net, rng, &l, &r);                               // DecodableNnetSimpleLoopedInfo::Init (decodable-simple-looped.cc:55-62)
  res.left_context = l;
  res.right_context = r;
  const int left = l + extra_left_context_initial, right = r;
  int chunk = frames_per_chunk;
  const int modulus = net.Modulus();
  const int fsf = std::max(frame_subsampling_factor, 1);
  while (chunk % modulus != 0 || chunk % fsf != 0) chunk++;          // GetChunkSize (nnet-compile-looped.cc:82-96)
  const bool has_ivector = net.Find("ivector") != -1;
  if (has_ivector) ModifyIvectorPeriod(net, chunk);
  // CompileLooped (:326-345) -> CompileLoopedInternal with 5 requests (:131-300) -> Compiler::CreateComputation (nnet-compile.cc:50-62)
  GraphBuilder b(net, rng);
  std::vector<int> prev_times, seen;
  for (int k = 0; k < 5; k++) {
    const int in0 = k == 0 ? -left : chunk + right + (k - 1) * chunk, in1 = k == 0 ? chunk + right : in0 + chunk;
    std::vector<Index> in, out, iv;
    for (int t = in0; t < in1; t++) in.push_back(Index{0, t, 0});
    for (int t = k * chunk; t < (k + 1) * chunk; t += fsf) out.push_back(Index{0, t, 0});      // (CreateComputationRequestInternal, :111-128)
    GraphBuilder::Io ins{{"input", in}}, outs{{"output", out}};
    if (has_ivector) {
      std::vector<int> times;
      if (k < 3) {
        for (int t = in0; t < in1; t++) {
          const int it = t - MathMod(t, chunk);
          if (std::find(seen.begin(), seen.end(), it) == seen.end() && std::find(times.begin(), times.end(), it) == times.end()) times.push_back(it);
        }
        std::sort(times.begin(), times.end());
        seen.insert(seen.end(), times.begin(), times.end());
      } else {
        for (int t : prev_times) times.push_back(t + chunk);         // ExtrapolateComputationRequest (:246-270)
      }
      prev_times = times;
      for (int t : times) iv.push_back(Index{0, t, 0});
      if (!iv.empty()) ins.push_back({"ivector", iv});
    }
    b.Compute(ins, outs);
    b.Prune();
  }
  // Compiler::SetUpPrecomputedIndexes (nnet-compile.cc:1239-1291): one step per (node, segment) of a feed-forward network;
  // TdnnComponent::PrecomputeIndexes draws once per step (nnet-tdnn-component.cc:553)
  int begin = 0, max_offset = 0;
  for (int end : b.segment_ends) {
    std::vector<int> tdnn;
    for (int c = begin; c < end; c++) {
      const Node &n = net.nodes[b.cindexes[c].node];
      if (n.type == Node::kComponent && net.CType(n.comp) == "TdnnComponent") tdnn.push_back(b.cindexes[c].node);
    }
    std::sort(tdnn.begin(), tdnn.end());
    tdnn.erase(std::unique(tdnn.begin(), tdnn.end()), tdnn.end());
    for (size_t i = 0; i < tdnn.size(); i++) rng.Rand();
    begin = end;
  }
  // Optimize -> SplitRowOps -> InsertCommands (nnet-optimize-utils.cc:2883-2892,4654): the rows a chunk reads from the
  // chunk before it make two-piece multi-row commands, which are split, which draws once
  if (left + right > 0) rng.Rand();
  res.rand_calls = rng.calls;
  // The count above is that of a looped compilation that succeeds with 5 requests.  With a time offset larger than the
  // chunk the optimiser finds no repeating structure in five segments and the reference starts over with 10 (observed on
  // the reference: +-30 at chunk 24; every network with offsets <= chunk compiled at the first attempt).
  {
    const std::vector<bool> keep = ReachableFromOutputs(net);
    for (size_t i = 0; i < net.nodes.size(); i++) {
      if (!keep[i]) continue;
      const Node &n = net.nodes[i];
      if (n.type == Node::kDescriptor) n.desc.MaxAbsOffset(&max_offset);
      if (n.type == Node::kComponent)
        if (const std::vector<int32_t> *offs = net.TimeOffsets(n.comp))
          for (int o : *offs) max_offset = std::max(max_offset, std::abs(o));
    }
    if (max_offset > chunk) {
      res.rand_calls_certain = false;
      res.uncertain_why = "a time offset of " + std::to_string(max_offset) + " frames exceeds --frames-per-chunk=" + std::to_string(chunk);
    }
  }
  return res;
}

}  // namespace rs
