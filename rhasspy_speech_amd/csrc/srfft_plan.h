// Host-side plan for the GPU restatement of Kaldi's split-radix real FFT (matrix/srfft.cc:135-432).
//
// Why a plan: the reference computes the 512-point real FFT of every frame with a recursive, in-place, single
// precision split-radix algorithm followed by a post-processing loop whose twiddle factor is advanced by a float
// recurrence.  Its rounding error (about 1e-3 on the highest cepstra) is what dominates the feature mismatch of any
// "more accurate" FFT.  The recursion is a fixed dataflow graph, so it is flattened here into levels of independent
// butterfly tasks that a wavefront executes lane-parallel with exactly the same float operations on the same
// operands -- bit-identical spectra, no sequential recursion.
#pragma once
#include <vector>

namespace rs {

struct SrfftTask {       // 16 bytes, read by one lane
  int kind_logm;         // kind | logm << 8;  kind 0: L-shaped butterfly n of a block of 2^logm points (logm >= 3),
                         //                    kind 1: whole 4-point block, kind 2: whole 2-point block
  int off;               // first complex index of the block
  int n;                 // butterfly index inside the block (kind 0)
  int tw;                // kind 0: -1 no twiddle (n == 0), -2 the sqrt(1/2) case (n == m/8), else index into tw[]
};

struct SrfftPlan {
  int logn = 0;                       // complex size N = 2^logn = padded / 2
  std::vector<SrfftTask> tasks;       // level-major
  std::vector<int> level_begin;       // num_levels + 1 offsets into tasks
  std::vector<float> tw;              // per kind-0 twiddle entry: 6 floats {cn, spcn, smcn, c3n, spc3n, smc3n}
  std::vector<int> perm;              // N: the bit-reversal pass as a gather, out[i] = in[perm[i]]
  std::vector<float> kn;              // 2 * (N/2 + 1): (re, im) of the post-processing factor for k = 0 .. N/2
};

SrfftPlan BuildSrfftPlan(int padded_window);

}  // namespace rs
