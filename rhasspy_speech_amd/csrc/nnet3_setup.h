// What the reference's decoder binaries do to the acoustic model between reading it and the first frame, as far as it
// shows in the results:
//
//  (1) CollapseModel(CollapseModelConfig(), &nnet) (online2-wav-nnet3-latgen-faster.cc:171,
//      online2-cli-nnet3-decode-faster.cc:106; nnet3/nnet-utils.cc:1459-2116): with the default configuration
//      (nnet-utils.h:240-249: affine and scale collapsing on, dropout and batch-norm collapsing off) a FixedAffine / Affine
//      component feeding an Affine one is folded into it -- also when several time offsets of the first are appended at the
//      input of the second -- and an Affine followed by a FixedScale is folded.  The network evaluated afterwards has fewer
//      layers and different (pre-multiplied) parameters.
//
//  (2) They call glibc's rand().  Dither() seeds the noise of every frame from rand() (feat/feature-window.cc:90-98,
//      base/kaldi-math.cc:59-70), each utterance is decoded by a fresh process (rhasspy_speech/tools.py:117-147), and
//      before the first frame nnet3 has drawn thousands of values: ComputeSimpleNnetContext (nnet-utils.cc:107), the
//      computation-graph builder's self checks (nnet-computation-graph.cc:481-555), AffineComponent::Init inside
//      CollapseModel (nnet-utils.cc:1850), TdnnComponent::PrecomputeIndexes (nnet-tdnn-component.cc:553) and the row-op
//      splitter (nnet-optimize-utils.cc:4654).  How many is a deterministic function of the network's structure and of
//      --frames-per-chunk; the dither of frame t is seeded by value number (that count + t) of the default-seed sequence.
//
// Nnet3Setup replays both on the parsed model: it returns the collapsed network (config lines + components) and the count.
#pragma once
#include <string>
#include <vector>

#include "model.h"

namespace rs {

struct Nnet3SetupResult {
  std::vector<std::string> config;   // config lines of the collapsed network (orphan nodes removed)
  long rand_calls = 0;               // rand() calls before the first frame
  // false when the 5-request looped compilation the count assumes may fail in the reference (it then retries with 10,
  // 20, ... requests and draws more; nnet-compile-looped.cc:326-345): some time offset exceeds the chunk size
  bool rand_calls_certain = true;
  std::string uncertain_why;
  int left_context = 0, right_context = 0;   // of the collapsed network (ComputeSimpleNnetContext)
};

// `components` / `component_names` gain the combined components CollapseModel creates.
Nnet3SetupResult Nnet3Setup(const std::vector<std::string> &config_lines, std::vector<std::string> *component_names,
                            std::vector<Component> *components, int frames_per_chunk, int extra_left_context_initial,
                            int frame_subsampling_factor = 1);

// glibc rand() of a fresh process (stdlib/random_r.c, TYPE_3 additive feedback generator, seed 1) and rand_r()
// (stdlib/rand_r.c), restated so that the host process's own rand() state is never touched.
class GlibcRand {
 public:
  GlibcRand();
  int Next();
  static int RandR(unsigned *seed);
 private:
  unsigned r_[34];
  int f_ = 3, b_ = 0;
};

// out[(t - t0) * win + i] = RandGauss() number i of frame t (kaldi-math.h:150-158 on rand_r, seed = rand() value number
// offset + t, + 27437): what Dither() adds to sample i of frame t for --dither=1.  Host libm, like the reference.
void DitherNoise(long offset, int t0, int t1, int win, float *out);

}  // namespace rs
