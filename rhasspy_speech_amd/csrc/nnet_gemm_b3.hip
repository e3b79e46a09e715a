// The TDNN layer GEMM on the fp16 matrix cores with FP32 results: every FP32 operand is split into two fp16 parts
//        x = x1 + x2 (+ at most 2^-22 |x|),   x1 = fp16(x), x2 = fp16(x - x1)        (round to nearest even)
// which together carry 22 bits of the significand, and a product is accumulated as
//        a b ~= a2 b1 + a1 b2 + a1 b1
// by three v_mfma_f32_32x32x16_f16 with FP32 accumulation (smallest terms first).  Each partial product is exact in FP32
// (11 x 11 significand bits); what is dropped (a2 b2 and the two representation errors) is below 2^-21 |a b|, and measured
// against the reference's goldens the log-likelihoods are as close as those of an FP32 BLAS (the order of the FP32 sums is
// what the error is made of: 1.9e-5 against 1.6e-5 on the zamia-size model, profiles/r04/split_numerics.txt).  Rounds 1-3 used
// three bf16 parts and six products per element: twice the matrix-core work and 6 instead of 4 bytes per operand element.
// Range: fp16 ends at 65504.  Weights are static: every output column is scaled by a power of two so that its largest
// weight lies in [2^14, 2^15) (low parts normal; the epilogue multiplies the accumulators by the inverse, exactly).
// Activations are split as they are: anything below 65520 in magnitude is carried with an absolute error below 2^-25 (low
// parts may be subnormal -- the matrix cores keep fp16 subnormal inputs, profiles/micro/mfma_f16_denorm.hip); a kernel that
// meets a larger one raises GemmDev::ovf and the host repeats the call on the exact-FP32 kernels (engine.cc), which stay
// selectable (RS_GEMM_B3=0) and are what small layers use.  The fp16 matrix cores run 16x the rate of the FP32-input MFMA
// (MI355X_MICROARCH.md), so three of them per k-step cost 3/16 of the exact-FP32 path.
// tests/test_gpu_parity.py holds the result to the same 1e-4 log-likelihood bound against the reference as the FP32 path.
//
// Same segmented-K contract as nnet_kernels.hip (GemmDev: the splice never exists in memory).  Block tile (32 MR) x 256:
// four waves side by side, each (32 MR) x 64 = MR x 2 accumulator tiles of 32x32.  K advances 16 per step:
//   weights: split ONCE on the host into the fragment order of the MFMA B operand -- for every (k-step, 32-column tile,
//            part) one 1 KiB block [k-group 2][column 32][8 fp16].  A wave is the only consumer of its 64 columns, so its
//            four B fragments never touch LDS: each is one global_load_dwordx4 (lane l takes bytes 16 l .. 16 l + 15 of
//            the block: 1 KiB of consecutive memory per instruction);
//   activations (shared by the four waves): FP32 rows from the producer's buffer -> registers (16 bytes per lane) ->
//            split -> two 8-byte LDS writes into fragment order [k-group][row][8 fp16] (two LDS stages, one barrier per
//            step), read back with one conflict-free ds_read_b128 at 16 x lane per fragment, used and dropped;
//   pipeline: three register sets rotate -- during step t the weights of step t + 2 and the activations of step t + 3 are
//            requested and the activations of step t + 1 are split into LDS.
// (History of the three-bf16 form of this kernel: profiles/r01)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include "env.h"

#include "nnet_b3_common.h"

namespace rs {
namespace {
using namespace b3;

// WM = 1: 256 threads, two workgroups per CU.  WM = 2: 512 threads = two such wave rows stacked (64 MR rows), launched with
// (almost) all of the CU's LDS so that no other workgroup shares the CU -- used when several decode pipelines are in flight
// (see engine.cc: ProcessTurn for why this kernel must then keep other kernels' workgroups off its CU).
// Measurement only (profiles/micro/pk_perturber2.sh; results WRONG with a bit set): 1 = no MFMAs, 2 = no split / LDS stores of the
// activations, 4 = no epilogue, 8 = no fragment reads + MFMAs (the whole step), 16 = the split without its |x| maximum
#ifndef RS_B3_ABLATE
#define RS_B3_ABLATE 0
#endif
template <int MR, bool MIXED, int WM>
__global__ __launch_bounds__(256 * WM, WM == 1 ? 2 : 1) void GemmKernelB3(GemmDev d, int rows, int nbig, const int *__restrict__ row_ivec, int epi_mode) {
  static_assert(!(MIXED && WM > 1), "two tile heights only with one wave row");
  constexpr int RT = MR * WM, BM = 32 * RT, BN = kB3BN, NT = 256 * WM;
  constexpr int STAGE = RT * kB3Parts * kB3FragBytes;      // activations only: the weights go straight to registers
  constexpr int UNITS = BM * 4, NA = (UNITS + NT - 1) / NT;      // 16-byte activation loads per k-step and thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;          // wave row (0 .. WM-1), 64-column slice
  // XCD-aware tile order (see GemmKernel): all column tiles of a row tile go to one XCD back to back
  // Two tile heights in one launch: the first nbig row tiles are BM rows tall, the rows after them are cut into tiles of
  // half that height (MR even), launched last.  The launcher sizes nbig to whole rounds of the chip, so that the rest --
  // which would otherwise be one mostly empty round of tall tiles -- spreads over all CUs as short ones.
  const int ncol = (d.n + BN - 1) / BN;
  const int big_blocks = (nbig + 7) / 8 * 8 * ncol;
  const bool small = MIXED && (int)blockIdx.x >= big_blocks;
  const int mr_eff = small ? MR / 2 : MR;                         // row tiles (of 32) this workgroup computes
  const int bid = small ? blockIdx.x - big_blocks : blockIdx.x, xcd = bid & 7, local = bid >> 3;
  const int rt = (local / ncol) * 8 + xcd, ct = local % ncol;
  const int row0 = small ? nbig * BM + rt * (BM / 2) : rt * BM, n0 = ct * BN;
  if (small ? row0 >= rows : rt >= nbig) return;
  f32x16 acc[MR][2];
#pragma unroll
  for (int i = 0; i < MR; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // activation staging: unit u = tid + 256 h -> tile row u / 4, floats 4 (u % 4) .. + 3 of the k-step
  int grow[NA], a_lds[NA];
  bool a_on[NA];
  const int kq = tid & 3;
#pragma unroll
  for (int h = 0; h < NA; h++) {
    const int u = tid + NT * h, r = u >> 2;
    a_on[h] = u < (MIXED ? mr_eff * 128 : UNITS);
    grow[h] = row0 + (a_on[h] ? r : 0);
    if (grow[h] >= rows) grow[h] = 0;          // clamped rows are dropped in the epilogue
    if (d.row_map) grow[h] = d.row_map[grow[h]];
    // fragment image of row tile r / 32, part p at + p * RT KiB: [k-group][row][8 fp16]
    a_lds[h] = (r >> 5) * kB3FragBytes + (kq >> 1) * 512 + (r & 31) * 16 + (kq & 1) * 8;
  }
  const float *aptr[NA];
  f32x4 av0[NA], av1[NA], av2[NA];             // three k-steps of activations in flight
  int lim0 = 0, lim1 = 0, lim2 = 0;
  int seg = 0, k0 = 0, nt = 0;              // (segment, k0) cursor of the next k-step to load; a segment spans its padded width
  for (int sgi = 0; sgi < d.nsegs; sgi++) nt += ((d.segs[sgi].ncols + kGemmBK - 1) / kGemmBK) * (kGemmBK / kB3KS);
  // per-segment scalars live in the lanes of two VGPRs (lane s = segment s) and are fetched with v_readlane: reading them
  // from the kernel arguments inside the loop cost two or three s_load + s_waitcnt lgkmcnt(0) round trips per k-step, and
  // that wait also drains the LDS queue
  const int seg_ncols_v = lane < d.nsegs ? d.segs[lane < kMaxSegs ? lane : 0].ncols : 0;
  const int seg_rowoff_v = lane < d.nsegs ? d.segs[lane < kMaxSegs ? lane : 0].row_off : 0;
  const int nsegs = d.nsegs, ld0 = d.segs[0].ld, rowoff0 = d.segs[0].row_off;
  auto enter_segment = [&]() __attribute__((always_inline)) {
    const GemmSegDev &sg = d.segs[seg];
#pragma unroll
    for (int h = 0; h < NA; h++) {
      const long arow = sg.per_utt ? (long)row_ivec[grow[h]] : (long)grow[h] + sg.row_off;
      aptr[h] = sg.src + arow * sg.ld + sg.col0 + kq * 4;
    }
  };
  // K order.  Default: segment after segment.  d.interleave (all segments are row-shifted views of one buffer, the TDNN
  // Append(Offset(x, -k), x, Offset(x, k))): k-step t belongs to segment t % nsegs at columns 16 (t / nsegs), so the
  // shifted reads of the same columns follow each other and hit in L2 instead of streaming the activations from HBM once
  // per segment (FETCH_SIZE was 5x the algorithmic bytes); W3 is laid out in the same order by the host.
  const bool inter = d.interleave != 0;
  auto issue_a = [&](f32x4 (&av)[NA], int &staged_lim) __attribute__((always_inline)) {
    const long delta = inter ? (long)(__builtin_amdgcn_readlane(seg_rowoff_v, seg) - rowoff0) * ld0 : 0;
#pragma unroll
    for (int h = 0; h < NA; h++) av[h] = *reinterpret_cast<const f32x4 *>(aptr[h] + delta);      // 16-byte aligned rows (launcher)
    const int ncols = __builtin_amdgcn_readlane(seg_ncols_v, seg), padded = (ncols + kGemmBK - 1) / kGemmBK * kGemmBK;
    staged_lim = ncols - k0 - kq * 4;
    if (inter) {
      if (++seg == nsegs) {
        seg = 0;
        if (k0 + kB3KS < padded) {
          k0 += kB3KS;
#pragma unroll
          for (int h = 0; h < NA; h++) aptr[h] += kB3KS;
        }                    // else: past the last k-step, stay in place (those requests are never used)
      }
      return;
    }
#pragma unroll
    for (int h = 0; h < NA; h++) aptr[h] += kB3KS;
    k0 += kB3KS;
    if (k0 >= padded) {
      if (seg + 1 < nsegs) {
        seg++; k0 = 0; enter_segment();
      } else {                 // past the last k-step (the pipeline requests up to three steps beyond it): stay in place
        k0 -= kB3KS;
#pragma unroll
        for (int h = 0; h < NA; h++) aptr[h] -= kB3KS;
      }
    }
  };
  bool over = false;                           // an activation this thread split is beyond the fp16 split's range (or not a number)
  float amax[NA];                              // max |x| over what this thread splits of its row(s): every fourth float4 of the layer's K
#pragma unroll
  for (int h = 0; h < NA; h++) amax[h] = 0.f;
  auto store_a = [&](int stage, const f32x4 (&av)[NA], int staged_lim) __attribute__((always_inline)) {
    unsigned char *As = smem + stage * STAGE;
    if (RS_B3_ABLATE & 2) return;
#pragma unroll
    for (int h = 0; h < NA; h++) {
      if (!a_on[h]) continue;
      const f32x4 x0 = av[h];
      const f32x4 x = f32x4{staged_lim > 0 ? x0[0] : 0.f, staged_lim > 1 ? x0[1] : 0.f, staged_lim > 2 ? x0[2] : 0.f, staged_lim > 3 ? x0[3] : 0.f};
      f16x4 p1, p2;
      if (RS_B3_ABLATE & 16) (void)Split2(x, &p1, &p2); else { over |= B3Over(Split2(x, &p1, &p2)); amax[h] = B3AbsMax(amax[h], x); }
      *reinterpret_cast<f16x4 *>(As + a_lds[h]) = p1;
      *reinterpret_cast<f16x4 *>(As + a_lds[h] + RT * kB3FragBytes) = p2;
    }
  };
  // weights: k-step t, this wave's 2 column tiles x 2 parts = 4 consecutive KiB of W3
  constexpr int P = kB3Parts;
  const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(d.W3) + (size_t)(n0 / 32 + wn * 2) * P * kB3FragBytes + lane * 16;
  const size_t wstep = (size_t)(d.n3 / 32) * P * kB3FragBytes;
  auto load_b = [&](f16x8 (&bf)[2][P]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int p = 0; p < P; p++) bf[j][p] = *reinterpret_cast<const f16x8 *>(wsrc + (j * P + p) * kB3FragBytes);
    wsrc += wstep;
  };
  // One k-step of MFMAs.  Activation fragments are transient (read, used, dropped): the low part meets the weights' high
  // part, the high part meets both -- smallest terms first for every accumulator.
  auto step = [&](int t, const f16x8 (&bf)[2][P]) __attribute__((always_inline)) {
    const unsigned char *As = smem + (t & 1) * STAGE + lane * 16;
    if (RS_B3_ABLATE & 8) return;
    // fragment reads run one fragment ahead of the MFMAs that use them
    f16x8 cur = *reinterpret_cast<const f16x8 *>(As + ((P - 1) * RT + wm * MR) * kB3FragBytes), nxt = cur;
#pragma unroll
    for (int idx = 0; idx < P * MR; idx++) {
      const int pa = P - 1 - idx / MR, i = idx % MR;
      if (idx + 1 < P * MR) {
        const int pa2 = P - 1 - (idx + 1) / MR, i2 = (idx + 1) % MR;
        nxt = *reinterpret_cast<const f16x8 *>(As + (pa2 * RT + wm * MR + i2) * kB3FragBytes);
      }
      if (!MIXED || i < mr_eff) {
#pragma unroll
        for (int pb = P - 1; pb >= 0; pb--) {
          if (pb > P - 1 - pa) continue;
#pragma unroll
          for (int j = 0; j < 2; j++) {
            if (RS_B3_ABLATE & 1) acc[i][j][0] += (float)cur[0] * (float)bf[j][pb][0];
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][pb], cur, acc[i][j], 0, 0, 0);
          }
        }
      }
      cur = nxt;
    }
  };
  // Software pipeline, everything rotating over three register sets: during k-step t the weights of step t + 2 and the
  // activations of step t + 3 are requested, the activations of step t + 1 (requested two steps ago) are split into LDS.
  // A lone workgroup on a CU (the last, partly filled round of a launch) then still has two full steps of latency cover.
  // The requests run unconditionally: past the end they re-read the last activations (never stored) and two padding
  // k-steps of W3 (never multiplied).
  f16x8 b0[2][P], b1[2][P], b2[2][P];
  if (nt == 0) return;
  enter_segment();
  load_b(b0);
  issue_a(av0, lim0);
  load_b(b1);
  issue_a(av1, lim1);
  issue_a(av2, lim2);
  store_a(0, av0, lim0);
  dd::LdsBarrier();
#define RS_B3_SUBSTEP(T, BCUR, BNEXT2, ANEXT, LNEXT, AFREE, LFREE)                       \
  {                                                                                      \
    load_b(BNEXT2);                                                                      \
    store_a(((T) + 1) & 1, ANEXT, LNEXT);                                                \
    issue_a(AFREE, LFREE);                                                               \
    step((T), BCUR);                                                                     \
    dd::LdsBarrier();                                                                    \
  }
  const int nfull = nt / 3 * 3;              // whole rotations in the loop (no exits from inside it: they made the
#pragma nounroll                             // register allocator spill the weight sets), the remainder after it
  for (int t = 0; t < nfull; t += 3) {
    RS_B3_SUBSTEP(t, b0, b2, av1, lim1, av0, lim0)
    RS_B3_SUBSTEP(t + 1, b1, b0, av2, lim2, av1, lim1)
    RS_B3_SUBSTEP(t + 2, b2, b1, av0, lim0, av2, lim2)
  }
  if (nt - nfull >= 1) RS_B3_SUBSTEP(nfull, b0, b2, av1, lim1, av0, lim0)
  if (nt - nfull == 2) RS_B3_SUBSTEP(nfull + 1, b1, b0, av2, lim2, av1, lim1)
#undef RS_B3_SUBSTEP
  if (over) d.ovf[0] = 1;
  {
    // a row's K is dealt out to four neighbouring lanes (kq = lane & 3): their maxima together are the row's
    bool under = false;
#pragma unroll
    for (int h = 0; h < NA; h++) {
      float m = amax[h];
      m = fmaxf(m, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(m), 0xB1, 0xF, 0xF, true)));      // quad_perm [1,0,3,2]
      m = fmaxf(m, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(m), 0x4E, 0xF, 0xF, true)));      // quad_perm [2,3,0,1]
      under |= a_on[h] && B3Under(m);
    }
    if (under) d.ovf[1] = 1;
  }
  if (RS_B3_ABLATE & 4) { float fs = 0.f; for (int i = 0; i < MR; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) fs += acc[i][j][r]; if (fs == 12345.f) d.out[0] = fs; return; }
#include "nnet_b3_epilogue.inc"
}

template <int MR, bool MIXED, int WM>
void LaunchB3(const GemmDev &d, int rows, int nbig, const int *row_ivec, hipStream_t s) {
  constexpr int BM = 32 * MR * WM;
  constexpr size_t stage = 2 * (size_t)(MR * WM * kB3Parts * kB3FragBytes), ctile = kB3EpiBytes;
  // WM = 2 asks for all but 1 KiB of the CU's LDS: no other workgroup fits beside it
  constexpr size_t smem = WM == 1 ? (stage > ctile ? stage : ctile) : (size_t)159 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&GemmKernelB3<MR, MIXED, WM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int ncol = (d.n + kB3BN - 1) / kB3BN;
  const int rest = std::max(rows - nbig * BM, 0), nsmall = MIXED ? (rest + BM / 2 - 1) / (BM / 2) : 0;
  const int blocks = ((nbig + 7) / 8 * 8 + (nsmall + 7) / 8 * 8) * ncol;
  hipLaunchKernelGGL((GemmKernelB3<MR, MIXED, WM>), dim3(blocks), dim3(256 * WM), smem, s, d, rows, nbig, row_ivec, GemmEpiMode(d, rows));
}

}  // namespace

// Up to 45 % of the 256-column tiles may be padding: per padded column the split-fp16 kernels are about three times as fast as
// the exact-FP32 kernel with its 128-column tiles (hidden layer 105 us for 33 GFLOP against 163 us for the 15 GFLOP of the pruned
// output layer, profiles/r04), so e.g. the headline's 362 output columns (29 % padding in two tiles) belong here: 163 -> 85 us.
bool GemmB3PaddingOk(int n, int n3) {
  static const int pct = [] { const char *e = TuneEnv("RS_GEMM_B3_PAD"); return e ? std::atoi(e) : 45; }();
  // One tile wide, at least 96 columns (round 6: a factorised TDNN's 128-wide bottlenecks, K = 2048): the exact-FP32 kernel ran such a
  // layer at 109 TFLOP/s, 0.7 of ITS peak (403 us for 84 k rows), the split kernels take 62.5 % padding and are still 2.4 times as fast
  // (profiles/r06/tdnnf_notes.txt); RS_GEMM_B3_NARROW=0 (read per model load) keeps the 45 % rule alone.
  if (n3 == kB3BN && n >= 96) { const char *e = std::getenv("RS_GEMM_B3_NARROW"); if (!(e && std::atoi(e) == 0)) return true; }
  return (long)(n3 - n) * 100 <= (long)n3 * pct;
}

bool GemmB3Usable(const GemmDev &d) {
  const char *e = std::getenv("RS_GEMM_B3");          // read per call: the parity test flips it between two decodes
  if ((e && std::atoi(e) == 0) || !d.W3 || d.n3 < kB3BN) return false;
  if (!GemmB3PaddingOk(d.n, d.n3)) return false;
  for (int i = 0; i < d.nsegs; i++)
    if ((d.segs[i].ld & 3) || (d.segs[i].col0 & 3) || (reinterpret_cast<uintptr_t>(d.segs[i].src) & 15)) return false;
  return true;
}

void LaunchGemmB3(const GemmDev &d0, int rows, const int *row_ivec, hipStream_t s) {
  if (d0.res) {      // (a folded residual: this kernel's epilogue does not add it -- nnet_gemm_b3i.hip)
    LaunchGemmB3(GemmWithoutResidual(d0), rows, row_ivec, s);
    LaunchResidualAdd(d0, rows, s);
    return;
  }
  const GemmDev &d = d0;
  static int num_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  static int force_mr = [] { const char *e = TuneEnv("RS_GEMM_B3_MR"); return e ? std::atoi(e) : 0; }();
  static int mixed = [] { const char *e = TuneEnv("RS_GEMM_B3_MIXED"); return e ? std::atoi(e) : 1; }();
  const int ncol = (d.n + kB3BN - 1) / kB3BN;
  if (d.exclusive) {
    // one 512-thread workgroup per CU; tile height 128 or 192 rows, whichever leaves the fuller last round
    auto rounds1 = [&](int bm) { return (double)(((long)((rows + bm - 1) / bm) * ncol + num_cu - 1) / num_cu) * bm; };
    if (rounds1(192) * 0.97 < rounds1(128)) LaunchB3<3, false, 2>(d, rows, (rows + 191) / 192, row_ivec, s);
    else LaunchB3<2, false, 2>(d, rows, (rows + 127) / 128, row_ivec, s);
    return;
  }
  const long slots = std::max(2L * num_cu / std::max(d.share, 1), 8L);      // two workgroups per CU; the device may be shared
  // Tile height: rounds of `slots` tiles, each as long as the tile is tall, weighted by the measured per-row efficiency
  // of the height (taller tiles stream the weights for more rows: 0.78 at 128 rows).
  auto rounds = [&](long row_tiles) { return (double)((row_tiles * ncol + slots - 1) / slots); };
  auto cost = [&](int bm, double eff) { return rounds((rows + bm - 1) / bm) * bm * eff; };
  int mr = 2, nbig = (rows + 63) / 64;
  double best = cost(64, 1.0);
  if (cost(96, 0.97) < best) { best = cost(96, 0.97); mr = 3; nbig = (rows + 95) / 96; }
  if (cost(128, 0.78) < best) { best = cost(128, 0.78); mr = 4; nbig = (rows + 127) / 128; }
  if (mixed) {
    // whole rounds of 128-row tiles, the remaining rows as 64-row tiles of the same launch
    const long full = (long)(rows / 128) * ncol / slots * slots / ncol;        // 128-row tiles in whole rounds
    const long rest = rows - full * 128;
    const double c = rounds(full) * 128 * 0.78 + rounds((rest + 63) / 64) * 64 * 1.0;
    if (full > 0 && c < best) { best = c; mr = 4; nbig = (int)full; }
  }
  if (force_mr >= 2 && force_mr <= 4) { mr = force_mr; nbig = (rows + 32 * mr - 1) / (32 * mr); }
  if (mr == 2) LaunchB3<2, false, 1>(d, rows, nbig, row_ivec, s);
  else if (mr == 3) LaunchB3<3, false, 1>(d, rows, nbig, row_ivec, s);
  else if ((long)nbig * 128 >= rows) LaunchB3<4, false, 1>(d, rows, nbig, row_ivec, s);
  else LaunchB3<4, true, 1>(d, rows, nbig, row_ivec, s);
}

}  // namespace rs
