// Rescoring path: see rescore.cc.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "lattice.h"
#include "model.h"

namespace rs {

class Rescorer {
 public:
  // new_lang_dir: L_disambig.fst, G.fst, words.txt (with #0), phones/disambig.int -- the files transcribe_wav.py:117-142 reads
  explicit Rescorer(const std::string &new_lang_dir);
  // The reference's tool chain on one determinised lattice: up to `nbest` word sequences (ids of the NEW words.txt), best
  // first, with nbest-to-linear's graph / acoustic costs.  Empty when nothing survives ("Empty lattice ... incompatible LM?").
  std::vector<NbestPath> Rescore(const CompactLat &clat, const TransitionModel &tm, int nbest, double acoustic_scale) const;
  int phi() const { return phi_; }

 private:
  int phi_ = -1;
  int l_start_ = -1, g_start_ = -1;
  std::vector<std::vector<FstArc>> l_arcs_, g_arcs_;
  std::vector<float> l_final_, g_final_;
  std::vector<int> g_phi_;          // per G state: index of its back-off arc in g_arcs_[s], -1 = none
};

CompactLat ParseCompactLatticeEntry(const char *bytes, size_t n, std::string *key);

}  // namespace rs
