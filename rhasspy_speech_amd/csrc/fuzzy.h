// Host-side fuzzy matcher: the reference's `get_fuzzy_text` (rhasspy_speech/transcribe_util.py:11-88) without the seven
// OpenFst processes it spawns per utterance.  SURVEY.md section 8(f) item 3.
//
// The n-best hypotheses (word ids, best first) become a fan of linear paths whose arcs cost 0.1 * rank; the fan is
// composed with G.fuzzy.fst (the grammar plus `word : <eps> / 1` self loops that let recognised words be dropped,
// kaldi.py:343-408); the cheapest path gives the output labels (the grammar's words and `__output:` meta labels) and the
// cost the caller compares with max_fuzzy_cost.  Costs reproduce what the reference sums from `fstprint`'s text: float32
// tropical arithmetic along the path, epsilon:epsilon arcs folded into the next arc as fstrmepsilon does, every printed
// weight rounded to 9 significant digits, the final weight not counted.
#pragma once
#include <string>
#include <vector>

#include "model.h"

namespace rs {

struct FuzzyResult {
  bool matched = false;           // false: no path, or a path without output words (the reference returns None)
  std::vector<int> olabels;       // non-epsilon output labels along the path
  double cost = 0.0;
};

class FuzzyMatcher {
 public:
  explicit FuzzyMatcher(const std::string &fuzzy_fst_path) { g_.Read(fuzzy_fst_path); }
  // nbest: the text `nbest-to-linear ... ark,t:-` prints ("utt-k id id ...\n" lines)
  FuzzyResult Match(const std::string &nbest_text) const;

 private:
  Hclg g_;      // any VectorFst / ConstFst over the standard arc; arcs ilabel-sorted per state
};

}  // namespace rs
