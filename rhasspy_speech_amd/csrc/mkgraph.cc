// The model-dependent steps of graph construction and the mkgraph.sh chain itself; see graph_build.h.
//
// Reference semantics followed (file:line under /root/reference/kaldi/src):
//   ContextDependency   tree/context-dep.cc:25-45 (Compute), :143-180 (I/O); tree/event-map.cc:36-225 (CE / TE / SE maps)
//   MakeHTransducer     hmm/hmm-utils.cc:30-150 (GetHmmAsFsa), :262-333 (GetHTransducer); fstext/fstext-utils-inl.h:712-781 (MakeLoopFst)
//   AddSelfLoops        hmm/hmm-utils.cc:425-560 (reorder = true); fstext/fstext-utils-inl.h:577-650 (MakePrecedingInputSymbolsSameClass)
//   Mkgraph             egs/wsj/s5/utils/mkgraph.sh:72-170
#include <sys/stat.h>
#include <unistd.h>
#include "env.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <set>

#include "graph_build.h"
#include "kaldi_io.h"

namespace rs {
namespace gb {

// ------------------------------------------------------------------------------------------------------------ tree
int32_t ContextDependency::ReadNode(KaldiReader &r) {
  const std::string tok = r.PeekToken();
  if (tok == "NULL") { r.ReadToken(); return -1; }
  Node n;
  if (tok == "CE") {
    r.ReadToken();
    n.kind = 'C';
    n.answer = r.ReadInt32();
  } else if (tok == "TE") {
    r.ReadToken();
    n.kind = 'T';
    n.key = r.ReadInt32();
    const int size = r.ReadInt32();
    r.ExpectToken("(");
    n.children.resize(size);
    for (int i = 0; i < size; i++) n.children[i] = ReadNode(r);
    r.ExpectToken(")");
  } else if (tok == "SE") {
    r.ReadToken();
    n.kind = 'S';
    n.key = r.ReadInt32();
    r.ReadIntVector(&n.yes_set);
    std::sort(n.yes_set.begin(), n.yes_set.end());
    r.ExpectToken("{");
    const int32_t yes = ReadNode(r), no = ReadNode(r);
    r.ExpectToken("}");
    if (yes < 0 || no < 0) Fail("SplitEventMap::Read, NULL pointers.");
    n.children = {yes, no};
  } else {
    Fail("EventMap::read, was not expecting token " + tok.substr(0, 20) + " in " + r.name());
  }
  nodes_.push_back(n);
  return (int32_t)nodes_.size() - 1;
}

void ContextDependency::Read(const std::string &path) {
  KaldiReader r(path);
  r.ExpectToken("ContextDependency");
  n_ = r.ReadInt32();
  p_ = r.ReadInt32();
  std::string tok = r.ReadToken();
  if (tok == "ToLength") {           // old format: a pdf-class-count map nobody uses any more comes first
    ReadNode(r);
    tok = r.ReadToken();
  }
  if (tok != "ToPdf") Fail("Got unexpected token " + tok + " reading context-dependency object.");
  nodes_.clear();
  root_ = ReadNode(r);
  r.ExpectToken("EndContextDependency");
  if (root_ < 0 || n_ <= 0 || p_ < 0 || p_ >= n_) Fail(path + ": bad context-dependency object");
}

bool ContextDependency::Compute(const std::vector<int32_t> &win, int pdf_class, int32_t *pdf) const {
  if ((int)win.size() != n_) Fail("ContextDependencyNew::Compute, the phone sequence has the wrong size");
  auto value_of = [&](int32_t key, int32_t *v) {       // event = {(-1, pdf class), (0, win[0]), ..., (N-1, win[N-1])}
    if (key == -1) { *v = pdf_class; return true; }
    if (key >= 0 && key < n_) { *v = win[key]; return true; }
    return false;
  };
  int32_t cur = root_;
  while (cur >= 0) {
    const Node &n = nodes_[cur];
    int32_t v;
    if (n.kind == 'C') { *pdf = n.answer; return true; }
    if (!value_of(n.key, &v)) return false;
    if (n.kind == 'T') {
      if (v < 0 || v >= (int32_t)n.children.size()) return false;
      cur = n.children[v];
    } else {
      cur = std::binary_search(n.yes_set.begin(), n.yes_set.end(), v) ? n.children[0] : n.children[1];
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------------------------ H
namespace {
// one phone-in-context as an acceptor over transition-ids, without self loops (they are added to the finished graph)
Fst HmmAsFsa(const std::vector<int32_t> &win, int phone, const std::vector<int32_t> &pdfs, const TransitionModel &tm, float tscale) {
  const auto &entry = tm.topo_entries[tm.phone2entry[phone]];
  Fst f;
  for (size_t i = 0; i < entry.size(); i++) f.AddState();
  f.start = 0;
  f.fin[entry.size() - 1] = 0.0f;
  for (int hs = 0; hs < (int)entry.size(); hs++) {
    const int fc = entry[hs].fwd, sc = entry[hs].self;
    for (int k = 0; k < (int)entry[hs].trans.size(); k++) {
      const int dst = entry[hs].trans[k].first;
      if (dst == hs) continue;
      float log_prob;
      int label;
      if (fc < 0) {            // non-emitting state: the topology's own probability, no transition-id
        log_prob = logf(entry[hs].trans[k].second);
        label = 0;
      } else {
        const int ts = tm.TupleToTransitionState(phone, hs, pdfs[fc], pdfs[sc]);
        if (ts == 0) {
          std::string w;
          for (int32_t p : win) w += std::to_string(p) + " ";
          Fail("TransitionModel::TupleToTransitionState, tuple not found (phone window " + w + "): mismatch of tree and model");
        }
        const int tid = tm.tstate_first_tid[ts] + k;
        log_prob = tm.log_prob[tid] - tm.non_self_loop_log_prob[tid];       // GetTransitionLogProbIgnoringSelfLoops
        label = tid;
      }
      f.arcs[hs].push_back({label, label, -log_prob, dst});
    }
  }
  RemoveEpsLocal(&f, false);
  if (tscale != 1.0f)       // ApplyProbabilityScale
    for (int s = 0; s < f.NumStates(); s++) {
      for (Arc &a : f.arcs[s]) a.w *= tscale;
      if (f.fin[s] != kInf) f.fin[s] *= tscale;
    }
  return f;
}
}  // namespace

Fst MakeHTransducer(const std::vector<std::vector<int32_t>> &ilabels, const ContextDependency &tree, const TransitionModel &tm,
                    float tscale, std::vector<int32_t> *disambig_out) {
  if (ilabels.empty() || !ilabels[0].empty()) Fail("make-h-transducer: ilabel info must start with the empty window (epsilon)");
  disambig_out->clear();
  int next_disambig = tm.NumTransitionIds() + 1;
  Fst h;
  const int loop = h.AddState();
  h.start = loop;
  h.fin[loop] = 0.0f;
  // identical HMMs (same phone, same pdfs) are instantiated once; every further label gets a copy of the first arc only
  std::map<std::pair<int, std::vector<int32_t>>, Arc> shared;
  for (int j = 1; j < (int)ilabels.size(); j++) {
    const std::vector<int32_t> &info = ilabels[j];
    if (info.empty()) Fail("make-h-transducer: empty ilabel-info entry");
    if (info[0] < 0 || (info[0] == 0 && info.size() == 1)) {
      if (info.size() != 1) Fail("make-h-transducer: ilabel info of a grammar FST (--nonterm-phones-offset) is not supported");
      const int sym = next_disambig++;
      disambig_out->push_back(sym);
      const int end = h.AddState();
      h.arcs[loop].push_back({sym, j, 0.0f, end});       // the single arc starts in the shared loop state
      h.arcs[end].push_back({0, 0, 0.0f, loop});
      continue;
    }
    if ((int)info.size() != tree.width())
      Fail("Context size mismatch, ilabel-info [from context FST is " + std::to_string(info.size()) + ", context-dependency object expects " +
           std::to_string(tree.width()));
    const int phone = info[tree.central()];
    if (phone == 0) Fail("phone == 0.  Some mismatch happened, or there is a code error.");
    std::vector<int32_t> pdfs(tm.NumPdfClasses(phone));
    for (int c = 0; c < (int)pdfs.size(); c++)
      if (!tree.Compute(info, c, &pdfs[c])) {
        std::string w;
        for (int32_t p : info) w += std::to_string(p) + " ";
        Fail("GetHmmAsFsa: context-dependency object could not produce an answer: pdf-class = " + std::to_string(c) + " ctx-window = " + w);
      }
    const auto key = std::make_pair(phone, pdfs);
    auto it = shared.find(key);
    if (it != shared.end()) {
      Arc a = it->second;
      a.ol = j;
      h.arcs[loop].push_back(a);
      continue;
    }
    const Fst f = HmmAsFsa(info, phone, pdfs, tm, tscale);
    if (f.start < 0) continue;
    // start state shared with the loop state when nothing returns to it and it has exactly one arc and no final weight
    bool returns = false;
    for (auto &v : f.arcs) for (const Arc &a : v) returns |= a.next == f.start;
    const bool share = !returns && f.arcs[f.start].size() == 1 && f.fin[f.start] == kInf;
    std::vector<int> map(f.NumStates());
    for (int s = 0; s < f.NumStates(); s++) map[s] = (s == f.start && share) ? loop : h.AddState();
    if (!share) {
      const Arc a{0, j, 0.0f, map[f.start]};
      shared[key] = a;
      h.arcs[loop].push_back(a);
    }
    for (int s = 0; s < f.NumStates(); s++) {
      for (const Arc &a : f.arcs[s]) {
        const Arc na{a.il, (s == f.start && share) ? j : 0, a.w, map[a.next]};
        h.arcs[map[s]].push_back(na);
        if (s == f.start && share) shared[key] = na;
      }
      if (f.fin[s] != kInf) h.arcs[map[s]].push_back({0, 0, f.fin[s], loop});
    }
  }
  return h;
}

// ------------------------------------------------------------------------------------------------------------ self loops
void AddSelfLoops(const TransitionModel &tm, float self_loop_scale, Fst *f) {
  if (f->start < 0) Fail("add-self-loops: empty FST");
  const int ntid = tm.NumTransitionIds();
  auto cls = [&](int label) {         // transition-id -> transition-state; epsilon and anything else -> 0
    if (label >= 1 && label <= ntid) {
      if (tm.id2self_loop[label]) Fail("AddSelfLoops: graph already has self-loops.");
      return tm.id2tstate[label];
    }
    return 0;
  };
  // ---- every state must be entered by labels of one transition-state only (the start state counts as entered by epsilon):
  // arcs into a state with mixed classes go through a fresh state per (state, class) that continues with an epsilon
  {
    const int ns = f->NumStates();
    std::vector<int> c(ns, -1);
    c[f->start] = 0;
    std::set<int> bad;
    for (int s = 0; s < ns; s++)
      for (const Arc &a : f->arcs[s]) {
        if (c[a.next] == -1) c[a.next] = cls(a.il);
        else if (c[a.next] != cls(a.il)) bad.insert(a.next);
      }
    if (!bad.empty()) {
      std::map<std::pair<int, int>, int> made;
      for (int s = 0; s < ns; s++)
        for (size_t k = 0; k < f->arcs[s].size(); k++) {
          const Arc a = f->arcs[s][k];
          if (a.il == 0 || !bad.count(a.next)) continue;
          const std::pair<int, int> key{a.next, cls(a.il)};
          auto it = made.find(key);
          if (it == made.end()) {
            const int ns2 = f->AddState();
            f->arcs[ns2].push_back({0, 0, 0.0f, a.next});
            it = made.emplace(key, ns2).first;
          }
          f->arcs[s][k].next = it->second;
        }
    }
  }
  const int ns = f->NumStates();
  std::vector<int> state_in(ns, -1);
  for (int s = 0; s < ns; s++)
    for (const Arc &a : f->arcs[s]) {
      const int c = cls(a.il);
      if (state_in[a.next] == -1) state_in[a.next] = c;
      else if (state_in[a.next] != c) Fail("AddSelfLoops: a state is entered by labels of different transition-states");
    }
  if (!(state_in[f->start] == -1 || state_in[f->start] == 0)) Fail("AddSelfLoops: the start state is entered by a transition-id");
  // ---- reorder = true: the self loop sits on the state AFTER the forward transition; everything leaving that state (and its
  // final weight) is multiplied by the probability of not looping
  for (int s = 0; s < ns; s++) {
    if (state_in[s] <= 0) continue;
    const int ts = state_in[s];
    const int first = tm.tstate_first_tid[ts];
    const float fwd = tm.non_self_loop_log_prob[first];
    if (f->fin[s] != kInf) f->fin[s] += -fwd * self_loop_scale;
    for (Arc &a : f->arcs[s]) a.w += -fwd * self_loop_scale;
    const int sl = tm.self_loop_of_id[first];
    if (sl != 0) f->arcs[s].push_back({sl, 0, -tm.log_prob[sl] * self_loop_scale, s});
  }
}

// ------------------------------------------------------------------------------------------------------------ the chain
namespace {
bool Exists(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
void MakeDirs(const std::string &p) {
  for (size_t i = 1; i <= p.size(); i++)
    if (i == p.size() || p[i] == '/') ::mkdir(p.substr(0, i).c_str(), 0755);
}
void CopyFile(const std::string &from, const std::string &to) {
  std::ifstream is(from, std::ios::binary);
  if (!is.good()) Fail("mkgraph: cannot read " + from);
  std::ofstream os(to, std::ios::binary);
  os << is.rdbuf();
  if (!os.good()) Fail("mkgraph: cannot write " + to);
}
}  // namespace

void Mkgraph(const std::string &lang, const std::string &model_dir, const std::string &dir, const MkgraphOptions &opts) {
  const std::string tree_path = model_dir + "/tree", model_path = model_dir + "/final.mdl";
  for (const std::string &req : {lang + "/L_disambig.fst", lang + "/G.fst", lang + "/words.txt", lang + "/phones/disambig.int", model_path, tree_path})
    if (!Exists(req)) Fail("mkgraph.sh: expected " + req + " to exist");
  ContextDependency tree;
  tree.Read(tree_path);
  TransitionModel tm;
  {
    KaldiReader r(model_path);
    tm.Read(r);
  }
  MakeDirs(dir);
  const bool timing = TuneEnv("RS_MKGRAPH_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what, const Fst &f) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "mkgraph: %-28s %8.1f ms  (%d states, %zu arcs)\n", what, std::chrono::duration<double, std::milli>(now - t_last).count(), f.NumStates(), f.NumArcs());
    t_last = now;
  };
  auto dump = [&](const Fst &f, const char *name) { if (!opts.dump_dir.empty()) { MakeDirs(opts.dump_dir); WriteFst(f, opts.dump_dir + "/" + name, false); } };
  // LG
  Fst lg;
  {
    Fst l = ReadFst(lang + "/L_disambig.fst"), g = ReadFst(lang + "/G.fst");
    lg = Compose(l, g);
  }
  lap("L o G", lg);
  lg = DeterminizeStar(lg, true);
  lap("determinizestar", lg);
  MinimizeEncoded(&lg);
  lap("minimizeencoded", lg);
  PushSpecial(&lg);
  lap("pushspecial", lg);
  dump(lg, "LG.fst");
  // CLG
  std::vector<std::vector<int32_t>> ilabels;
  Fst clg = ComposeContext(ReadIntList(lang + "/phones/disambig.int"), tree.width(), tree.central(), std::move(lg), &ilabels);
  ArcSort(&clg, true);
  lap("composecontext", clg);
  dump(clg, "CLG.fst");
  if (!opts.dump_dir.empty()) WriteILabelInfo(ilabels, opts.dump_dir + "/ilabels");
  // Ha
  std::vector<int32_t> disambig_tid;
  Fst ha = MakeHTransducer(ilabels, tree, tm, opts.transition_scale, &disambig_tid);
  lap("make-h-transducer", ha);
  dump(ha, "Ha.fst");
  {
    std::ofstream os(dir + "/disambig_tid.int");
    for (int32_t d : disambig_tid) os << d << "\n";
  }
  // HCLGa
  ArcSort(&ha, false);
  Fst hclg = Compose(ha, clg);
  clg = Fst();
  ha = Fst();
  lap("H o CLG", hclg);
  hclg = DeterminizeStar(hclg, true);
  lap("determinizestar", hclg);
  RemoveInputSymbols(&hclg, disambig_tid);
  RemoveEpsLocal(&hclg, true);
  lap("rmsymbols + rmepslocal", hclg);
  MinimizeEncoded(&hclg);
  lap("minimizeencoded", hclg);
  dump(hclg, "HCLGa.fst");
  // HCLG
  AddSelfLoops(tm, opts.self_loop_scale, &hclg);
  ArcSort(&hclg, true);        // (the decoders want the input epsilons of a state first; fstconvert keeps whatever order it is given)
  // mkgraph.sh:151-164: the graph appears under its final name only when it is complete (a run killed mid-write must not leave a
  // file the up-to-date test of the next run accepts), and an empty result is an error
  if (hclg.start < 0 || hclg.NumStates() == 0) Fail("it looks like the result in " + dir + "/HCLG.fst is empty");
  {
    const std::string tmp = dir + "/HCLG.fst." + std::to_string((long)getpid());
    WriteFst(hclg, tmp, true);
    if (std::rename(tmp.c_str(), (dir + "/HCLG.fst").c_str()) != 0) { std::remove(tmp.c_str()); Fail("cannot move " + tmp + " to " + dir + "/HCLG.fst"); }
  }
  lap("add-self-loops + write", hclg);
  CopyFile(lang + "/words.txt", dir + "/words.txt");
  MakeDirs(dir + "/phones");
  for (const char *opt : {"word_boundary.int", "word_boundary.txt", "align_lexicon.int", "align_lexicon.txt", "disambig.int", "disambig.txt", "silence.csl"})
    if (Exists(lang + "/phones/" + opt)) CopyFile(lang + "/phones/" + opt, dir + "/phones/" + opt);
  if (Exists(lang + "/phones.txt")) CopyFile(lang + "/phones.txt", dir + "/phones.txt");
  {
    std::ofstream os(dir + "/num_pdfs");       // mkgraph.sh:185 (am-info | grep pdfs)
    os << tm.num_pdfs << "\n";
  }
}

}  // namespace gb
}  // namespace rs
