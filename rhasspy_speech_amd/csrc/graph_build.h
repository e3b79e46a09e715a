// Decoding-graph construction: the weighted-transducer operations the reference's
// utils/mkgraph.sh chains as processes (kaldi/egs/wsj/s5/utils/mkgraph.sh:100-150, called from rhasspy_speech/kaldi.py:409-425)
//   fsttablecompose L_disambig.fst G.fst | fstdeterminizestar --use-log=true | fstminimizeencoded | fstpushspecial      -> LG
//   fstcomposecontext --context-size=N --central-position=P --read-disambig-syms=... ilabels LG | fstarcsort           -> CLG
//   make-h-transducer --disambig-syms-out=... --transition-scale=1.0 ilabels tree final.mdl                             -> Ha
//   fsttablecompose Ha CLG | fstdeterminizestar --use-log=true | fstrmsymbols disambig_tid | fstrmepslocal |
//     fstminimizeencoded                                                                                               -> HCLGa
//   add-self-loops --self-loop-scale=... --reorder=true final.mdl | fstconvert --fst_type=const                        -> HCLG
// as in-process functions on one mutable FST type.  Host code: the reference's is host code too, and every step is a
// worklist algorithm over hash tables of subsets / state pairs.  Results are EQUIVALENT transducers (same weighted relation;
// isomorphic wherever the operation's output is canonical, i.e. after determinisation + encoded minimisation); state numbering
// is this implementation's own.
#pragma once
#include <cstdint>
#include <limits>
#include <string>
#include <vector>

#include "model.h"

namespace rs {
namespace gb {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr float kDelta = 1.0f / 1024.0f;      // fst::kDelta

struct Arc {
  int32_t il, ol;
  float w;
  int32_t next;
};

struct Fst {
  int start = -1;
  std::vector<std::vector<Arc>> arcs;
  std::vector<float> fin;            // +inf = not final
  int AddState() { arcs.emplace_back(); fin.push_back(kInf); return (int)arcs.size() - 1; }
  int NumStates() const { return (int)arcs.size(); }
  size_t NumArcs() const { size_t n = 0; for (auto &a : arcs) n += a.size(); return n; }
};

Fst ReadFst(const std::string &path);                       // ConstFst / VectorFst over the standard arc
void WriteFst(const Fst &f, const std::string &path, bool const_type);
void Connect(Fst *f);                                        // fst::Connect: keep accessible + coaccessible states, order preserved
void ArcSort(Fst *f, bool by_ilabel);                        // fstarcsort --sort_type=ilabel|olabel (stable)
Fst Compose(const Fst &a, const Fst &b, bool connect = true);     // fsttablecompose: sequence filter, then Connect
Fst DeterminizeStar(const Fst &f, bool use_log, float delta = kDelta);    // fstdeterminizestar [--use-log]
void MinimizeEncoded(Fst *f, float delta = kDelta);          // fstminimizeencoded
void PushSpecial(Fst *f, float delta = kDelta);              // fstpushspecial
void RemoveInputSymbols(Fst *f, const std::vector<int32_t> &syms);   // fstrmsymbols
// fstrmepslocal (stochastic_in_log = true, the tool's default: RemoveEpsLocalSpecial) / fst::RemoveEpsLocal (false)
void RemoveEpsLocal(Fst *f, bool stochastic_in_log);
// fstcomposecontext: inv(C) o lg for a context window of `width` phones with the central one at `central`; *ilabels gets the
// window (or [-disambig], or [0] for the start-of-sequence pseudo epsilon) each input label of the result stands for.
Fst ComposeContext(const std::vector<int32_t> &disambig, int width, int central, Fst lg, std::vector<std::vector<int32_t>> *ilabels);

// tree/context-dep.cc, tree/event-map.cc: the phonetic decision tree as the decoder-graph builder uses it
class ContextDependency {
 public:
  void Read(const std::string &path);
  int width() const { return n_; }
  int central() const { return p_; }
  bool Compute(const std::vector<int32_t> &phone_window, int pdf_class, int32_t *pdf) const;

 private:
  struct Node {
    char kind = 'C';                    // C = constant, T = table, S = split
    int32_t key = 0, answer = -1;
    std::vector<int32_t> children;      // T: per value (-1 = NULL); S: {yes, no}
    std::vector<int32_t> yes_set;       // S, sorted
  };
  int32_t ReadNode(KaldiReader &r);
  int n_ = 0, p_ = 0, root_ = -1;
  std::vector<Node> nodes_;
};

// make-h-transducer: *disambig_out = the input labels standing for the disambiguation symbols of `ilabels`
Fst MakeHTransducer(const std::vector<std::vector<int32_t>> &ilabels, const ContextDependency &tree, const TransitionModel &tm,
                    float transition_scale, std::vector<int32_t> *disambig_out);
// add-self-loops --reorder=true
void AddSelfLoops(const TransitionModel &tm, float self_loop_scale, Fst *f);

struct MkgraphOptions {
  float transition_scale = 1.0f;     // mkgraph.sh tscale
  float self_loop_scale = 0.1f;      // mkgraph.sh loopscale (rhasspy passes 1.0)
  std::string dump_dir;              // when set: LG.fst, CLG.fst, ilabels, Ha.fst, HCLGa.fst are written there too
};
// utils/mkgraph.sh <lang_dir> <model_dir> <graph_dir>: reads lang_dir/{L_disambig.fst,G.fst,words.txt,phones/disambig.int} and
// model_dir/{tree,final.mdl}; writes graph_dir/{HCLG.fst,words.txt,disambig_tid.int,phones/...}.
void Mkgraph(const std::string &lang_dir, const std::string &model_dir, const std::string &graph_dir, const MkgraphOptions &opts);

void WriteILabelInfo(const std::vector<std::vector<int32_t>> &info, const std::string &path);
std::vector<std::vector<int32_t>> ReadILabelInfo(const std::string &path);
// comparison of two transducers ("" = same, else the first difference): up to state numbering / arc order, and as weighted
// relations on the label pairs of random accepting paths (fstequivalent --random=true)
std::string Isomorphic(const Fst &a, const Fst &b, float delta);
std::string RandEquivalent(const Fst &a, const Fst &b, float delta, int npaths, int max_len, uint64_t seed);
std::vector<int32_t> ReadIntList(const std::string &path);    // one integer per line

}  // namespace gb
}  // namespace rs
