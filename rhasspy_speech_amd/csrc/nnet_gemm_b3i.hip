// The TDNN layer GEMM of nnet_gemm_b3.hip for sources that already exist as operand images (kernels.h: ActImage): the
// producing layer's epilogue stored its result split into the two fp16 parts, in A-fragment order, so the consumer's
// K-loop has no FP32 loads, no splitting and no LDS writes of its own left -- what nnet_gemm_b3.hip spends about half of
// its loop on (profiles/r01: matrix cores 42 % busy; VALU split + ds_write + a barrier per 16-wide k-step).
//
// Same arithmetic (three v_mfma_f32_32x32x16_f16 per product, smallest terms first, FP32 accumulation) and the same tile:
// (32 MR) x 256 per workgroup, four waves side by side, each (32 MR) x 64.  Per 16-wide k-step:
//   activations: 2 MR fragments of 1 KiB, copied verbatim from the image into LDS by global_load_lds_dwordx4 (the image
//                block IS the fragment: lane l supplies the address of row l & 31, k-group l >> 5, so row-shifted TDNN
//                segments and row maps cost nothing but address arithmetic); wave w stages row tile w;
//   weights:     as before, four fragments per wave straight into registers, three register sets rotating;
//   reads:       conflict-free ds_read_b128 at 16 x lane per fragment.
// Two k-steps form one LDS stage (BK = 32): one barrier per 32 of K instead of per 16; two stages, the DMA of stage t + 1
// runs during the MFMAs of stage t.
//
// Measured (profiles/r02): the kernel runs the hidden layers in the same time as nnet_gemm_b3.hip -- the
// in-loop split and the LDS writes it removes were not what the waves wait for.  Two further experiments on this kernel, both
// reverted: (1) hipcc puts `s_waitcnt vmcnt(0)` in front of the first use of an ordinary load result while an LDS-DMA is in
// flight, draining the weight prefetch at every second k-step; hiding the weight loads in inline asm with hand-counted waits
// is bit-exact only when every wait is vmcnt(0) -- counted waits that leave the DMA in flight behind the weight loads gave
// wrong tiles, i.e. LDS-DMA and register loads do NOT retire in issue order with respect to each other on this part -- and
// the all-vmcnt(0) form is no faster (1.64 ms for the nnet stage either way); (2) three instead of two weight register sets
// beside the asm loads spill.  The tile quantisation (656 row tiles of 128 on 512 slots) and the L2 stream of the weights
// (24 KiB per k-step and workgroup) are what is left to attack.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include "env.h"

#include "nnet_b3_common.h"

namespace rs {
namespace {
using namespace b3;

constexpr int kKPS = 2;                               // k-steps per LDS stage (template parameter KPS of the kernel; 8 for the 32-row tile)
// Timing ablations (profiles/micro/b3i_ablate.sh; results are WRONG with any bit set): 1 = the weight pointer never advances
// (every k-step re-reads the same 24 KiB: no L2 weight stream), 2 = the activation DMA always fetches k-step 0 (no HBM
// activation stream), 4 = no MFMAs, 8 = no DMA and no weight loads (matrix cores + LDS reads + barriers only), 16 = no
// per-stage wait / barrier.
#ifndef RS_B3I_ABLATE
#define RS_B3I_ABLATE 0
#endif

// KPS = k-steps per LDS stage.  A stage is waited for (vmcnt(0): the compiler drains everything in front of the first use of a
// weight register while an LDS-DMA is in flight anyway) and fenced by one barrier; the DMA of stage s + 1 is issued at the start of
// stage s.  With the 32-row tile of a small launch (one workgroup per CU, 12 MFMAs per k-step) a two-k-step stage is over long
// before its successor has arrived: 24 exposed round trips per layer (38 us for 5000 rows); eight k-steps per stage leave six.
template <int MR, bool MIXED, int KPS = kKPS>
__global__ __launch_bounds__(256, 2) void GemmKernelB3I(GemmDev d, int rows, int nbig, int epi_mode) {
  constexpr int BM = 32 * MR, BN = kB3BN;
  constexpr int P = kB3Parts, KSTEP_BYTES = MR * P * kB3FragBytes, STAGE = KPS * KSTEP_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;                               // 64-column slice; also the row tile this wave stages
  const int ncol = (d.n + BN - 1) / BN;
  const int big_blocks = (nbig + 7) / 8 * 8 * ncol;
  const bool small = MIXED && (int)blockIdx.x >= big_blocks;
  const int mr_eff = small ? MR / 2 : MR;
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

  // ---- activation staging: this wave copies row tile `wave` (all three parts) of every k-step
  const bool stager = wave < mr_eff;
  int grow = row0 + wave * 32 + (lane & 31);
  if (grow >= rows) grow = 0;                          // clamped rows are dropped in the epilogue
  if (d.row_map) grow = d.row_map[grow];
  const int kg_off = (lane >> 5) * 512;
  // per-segment scalars in the lanes of VGPRs (lane s = segment s), fetched with v_readlane (see nnet_gemm_b3.hip)
  const int sl_ = lane < d.nsegs ? (lane < kMaxSegs ? lane : 0) : 0;
  const int seg_rowoff_v = lane < d.nsegs ? d.segs[sl_].row_off : 0;
  const int seg_ks0_v = lane < d.nsegs ? d.segs[sl_].col0 / kB3KS : 0;
  const int seg_nks_v = lane < d.nsegs ? (d.segs[sl_].ncols + kB3KS - 1) / kB3KS : 0;
  const int nsegs = d.nsegs;
  const bool inter = d.interleave != 0;
  int nt = 0;                                          // k-steps in total
  for (int sgi = 0; sgi < d.nsegs; sgi++) nt += (d.segs[sgi].ncols + kB3KS - 1) / kB3KS;
  // (segment, k-step inside it) of the next k-step to stage; interleaved order: step t = segment t % nsegs, k-step t / nsegs
  int seg = 0, ks = 0;
  const unsigned char *img_base = d.segs[0].img.base;
  size_t part_bytes = d.segs[0].img.part_bytes;
  int img_nks = d.segs[0].img.nks, img_guard = d.segs[0].img.guard;
  auto stage_kstep = [&](unsigned char *dst) __attribute__((always_inline)) {     // dst: this k-step's 2 MR KiB in LDS
    if (stager) {
      const int phys = grow + __builtin_amdgcn_readlane(seg_rowoff_v, seg) + img_guard;
      const unsigned char *src = img_base + ((size_t)(phys >> 5) * img_nks + (__builtin_amdgcn_readlane(seg_ks0_v, seg) + ((RS_B3I_ABLATE & 2) ? 0 : ks))) * kB3FragBytes +
                                 kg_off + (phys & 31) * 16;
      if (!(RS_B3I_ABLATE & 8))
#pragma unroll
      for (int p = 0; p < P; p++)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + p * part_bytes),
                                         (void __attribute__((address_space(3))) *)(dst + (p * MR + wave) * kB3FragBytes), 16, 0, 0);
    }
    if (inter) {
      if (++seg == nsegs) { seg = 0; ks++; }
    } else if (++ks >= __builtin_amdgcn_readlane(seg_nks_v, seg)) {
      ks = 0;
      if (seg + 1 < nsegs) {
        seg++;
        if (d.segs[seg].img.base != img_base) {
          img_base = d.segs[seg].img.base; part_bytes = d.segs[seg].img.part_bytes; img_nks = d.segs[seg].img.nks; img_guard = d.segs[seg].img.guard;
        }
      }
    }
  };
  // weights: k-step t, this wave's 2 column tiles x 2 parts = 4 consecutive KiB of W3I
  const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(d.W3I) + (size_t)(n0 / 32 + wn * 2) * P * kB3FragBytes + lane * 16;
  const size_t wstep = (size_t)(d.n3 / 32) * P * kB3FragBytes;
  auto load_b = [&](f16x8 (&bf)[2][P]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int p = 0; p < P; p++) {
        if (RS_B3I_ABLATE & 8) { if (wsrc == nullptr) bf[j][p] = *reinterpret_cast<const f16x8 *>(wsrc + (j * P + p) * kB3FragBytes); }
        else bf[j][p] = *reinterpret_cast<const f16x8 *>(wsrc + (j * P + p) * kB3FragBytes);
      }
    if (!(RS_B3I_ABLATE & 1)) wsrc += wstep;
  };
  // one k-step of MFMAs from the fragments at `As` (this k-step's image in LDS)
  auto step = [&](const unsigned char *As, const f16x8 (&bf)[2][P]) __attribute__((always_inline)) {
    const unsigned char *Al = As + lane * 16;
    f16x8 cur = *reinterpret_cast<const f16x8 *>(Al + ((P - 1) * MR) * kB3FragBytes), nxt = cur;
#pragma unroll
    for (int idx = 0; idx < P * MR; idx++) {
      const int pa = P - 1 - idx / MR, i = idx % MR;
      if (idx + 1 < P * MR) {
        const int pa2 = P - 1 - (idx + 1) / MR, i2 = (idx + 1) % MR;
        nxt = *reinterpret_cast<const f16x8 *>(Al + (pa2 * MR + i2) * kB3FragBytes);
      }
      if ((RS_B3I_ABLATE & 4) ? (lane == 99 && cur[0] == (_Float16)12345.f) : (!MIXED || i < mr_eff)) {
#pragma unroll
        for (int pb = P - 1; pb >= 0; pb--) {
          if (pb > P - 1 - pa) continue;
#pragma unroll
          for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][pb], cur, acc[i][j], 0, 0, 0);
        }
      }
      cur = nxt;
    }
  };
  if (nt == 0) return;
  // ---- pipeline: LDS stage s holds k-steps 2 s, 2 s + 1; weights rotate over three register sets two k-steps ahead
  const int nstage = (nt + KPS - 1) / KPS;
  f16x8 b0[2][P], b1[2][P], b2[2][P];
  load_b(b0);
  load_b(b1);
#pragma unroll
  for (int q = 0; q < KPS; q++) if (q < nt) stage_kstep(smem + q * KSTEP_BYTES);
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // The loop body handles one k-step; the stage protocol (DMA of the next stage issued at the first k-step of a stage, waited
  // for and fenced by the barrier after its second) is written out per position in the 6-step rotation (2 stages x 3 weight sets).
  int t = 0;
#define RS_B3I_KSTEP(BCUR, BNEXT2)                                                                   \
  {                                                                                                  \
    const int st = t / KPS, kk = t % KPS;                                                            \
    load_b(BNEXT2);                              /* weights of k-step t + 2 (padding steps past the end) */ \
    if (kk == 0 && st + 1 < nstage) {            /* DMA of the next stage */                        \
      unsigned char *nx = smem + ((st + 1) & 1) * STAGE;                                             \
      _Pragma("unroll") for (int q = 0; q < KPS; q++)                                                \
        if (KPS * (st + 1) + q < nt) stage_kstep(nx + q * KSTEP_BYTES);                              \
    }                                                                                                \
    step(smem + (st & 1) * STAGE + kk * KSTEP_BYTES, BCUR);                                          \
    if (!(RS_B3I_ABLATE & 16) && (kk == KPS - 1 || t + 1 == nt)) { /* stage done: next stage's DMA landed, everyone done reading this one */ \
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
      __builtin_amdgcn_s_barrier();                                                                  \
    }                                                                                                \
    t++;                                                                                             \
  }
#pragma nounroll
  while (t + 6 <= nt) {
    RS_B3I_KSTEP(b0, b2)
    RS_B3I_KSTEP(b1, b0)
    RS_B3I_KSTEP(b2, b1)
    RS_B3I_KSTEP(b0, b2)
    RS_B3I_KSTEP(b1, b0)
    RS_B3I_KSTEP(b2, b1)
  }
  // remainder (< 6 k-steps), same rotation
  if (t < nt) RS_B3I_KSTEP(b0, b2)
  if (t < nt) RS_B3I_KSTEP(b1, b0)
  if (t < nt) RS_B3I_KSTEP(b2, b1)
  if (t < nt) RS_B3I_KSTEP(b0, b2)
  if (t < nt) RS_B3I_KSTEP(b1, b0)
#undef RS_B3I_KSTEP
  constexpr int RT = MR, NT = 256;
  const int wm = 0;
#include "nnet_b3_epilogue.inc"
}

template <int MR, bool MIXED, int KPS = kKPS>
void LaunchB3I(const GemmDev &d, int rows, int nbig, hipStream_t s) {
  constexpr int BM = 32 * MR;
  constexpr size_t stage = 2 * (size_t)KPS * MR * kB3Parts * kB3FragBytes, ctile = kB3EpiBytes;
  constexpr size_t smem = stage > ctile ? stage : ctile;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&GemmKernelB3I<MR, MIXED, KPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int ncol = (d.n + kB3BN - 1) / kB3BN;
  const int rest = std::max(rows - nbig * BM, 0), nsmall = MIXED ? (rest + BM / 2 - 1) / (BM / 2) : 0;
  const int blocks = ((nbig + 7) / 8 * 8 + (nsmall + 7) / 8 * 8) * ncol;
  hipLaunchKernelGGL((GemmKernelB3I<MR, MIXED, KPS>), dim3(blocks), dim3(256), smem, s, d, rows, nbig, GemmEpiMode(d, rows));
}

// f32 rows -> operand image: one wave per (row block, k-step) 1 KiB block, both parts
// one workgroup per 32 rows (of the row list, or of the buffer), wave w converts the k-steps w, w + 4, ...; the four waves' maxima
// of |x| over a row together are the row's (B3Under, nnet_b3_common.h).
// With a residual (GemmDev::res / res_img of a layer that did NOT run on GemmKernelB3J, which adds it in registers): the row is
// src + res_scale * residual first -- the same float operations -- and goes back to `src` when somebody reads it as floats.
struct ResidualDev { const float *res; int res_ld; float scale; ActImage img; int write_back; };
__global__ __launch_bounds__(256) void ToImageKernel(float *__restrict__ src, int ld, int dim, int rows, ActImage img, int *ovf,
                                                     const int *__restrict__ row_map, ResidualDev rd) {
  __shared__ unsigned rmx[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 32) rmx[threadIdx.x] = 0u;
  __syncthreads();
  const int idx = blockIdx.x * 32 + (lane & 31);
  const bool rok = idx < rows;
  const int row = rok ? (row_map ? row_map[idx] : idx) : 0, phys = row + img.guard;
  const int nks = (dim + 15) / 16;
  bool over = false;
  float rm = 0.f;
  for (int ks = wave; ks < nks && rok; ks += 4) {
    const int col = ks * 16 + (lane >> 5) * 8;
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      lo[e] = col + e < dim ? src[(size_t)row * ld + col + e] : 0.f;
      hi[e] = col + 4 + e < dim ? src[(size_t)row * ld + col + 4 + e] : 0.f;
    }
    if (rd.res || rd.img.base) {
      f32x4 rl, rh;
      if (rd.img.base) {
        const int rp = row + rd.img.guard;
        const unsigned char *rs = rd.img.base + ((size_t)(rp >> 5) * rd.img.nks + ks) * kB3FragBytes + (lane >> 5) * 512 + (rp & 31) * 16;
        const f16x8 r1 = *reinterpret_cast<const f16x8 *>(rs), r2 = *reinterpret_cast<const f16x8 *>(rs + rd.img.part_bytes);
#pragma unroll
        for (int e = 0; e < 4; e++) { rl[e] = (float)r1[e] + (float)r2[e]; rh[e] = (float)r1[4 + e] + (float)r2[4 + e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          rl[e] = col + e < dim ? rd.res[(size_t)row * rd.res_ld + col + e] : 0.f;
          rh[e] = col + 4 + e < dim ? rd.res[(size_t)row * rd.res_ld + col + 4 + e] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        lo[e] = __fadd_rn(rd.scale != 1.0f || rd.img.base ? __fmul_rn(rl[e], rd.scale) : rl[e], lo[e]);
        hi[e] = __fadd_rn(rd.scale != 1.0f || rd.img.base ? __fmul_rn(rh[e], rd.scale) : rh[e], hi[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; e++) { if (col + e >= dim) lo[e] = 0.f; if (col + 4 + e >= dim) hi[e] = 0.f; }
      if (rd.write_back) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (col + e < dim) src[(size_t)row * ld + col + e] = lo[e];
          if (col + 4 + e < dim) src[(size_t)row * ld + col + 4 + e] = hi[e];
        }
      }
    }
    if (img.base) {
      f16x8 p1, p2;
      over |= B3Over(Split2(lo, hi, &p1, &p2));
      rm = B3AbsMax(B3AbsMax(rm, lo), hi);
      unsigned char *dst = img.base + ((size_t)(phys >> 5) * img.nks + ks) * kB3FragBytes + (lane >> 5) * 512 + (phys & 31) * 16;
      *reinterpret_cast<f16x8 *>(dst) = p1;
      *reinterpret_cast<f16x8 *>(dst + img.part_bytes) = p2;
    }
  }
  if (over) ovf[0] = 1;
  atomicMax(&rmx[lane & 31], __float_as_uint(rm));
  __syncthreads();
  if (threadIdx.x < 32 && B3Under(__uint_as_float(rmx[threadIdx.x]))) ovf[1] = 1;
}

}  // namespace

size_t ActImagePartBytes(int rows, int guard, int dim) {
  const size_t row_blocks = ((size_t)rows + 2 * (size_t)guard + 31) / 32 + 1;
  return row_blocks * (size_t)((dim + 15) / 16) * b3::kB3FragBytes;
}

void LaunchToImage(const float *src, int ld, int dim, int rows, const ActImage &img, int *ovf, hipStream_t s, const int *row_map) {
  if (rows <= 0 || img.nks <= 0) return;
  hipLaunchKernelGGL(ToImageKernel, dim3((rows + 31) / 32), dim3(256), 0, s, const_cast<float *>(src), ld, dim, rows, img, ovf, row_map, ResidualDev{nullptr, 0, 1.0f, ActImage{nullptr, 0, 0, 0}, 0});
}

// A layer with a folded residual on a kernel that does not add it itself (GemmKernelB3 / GemmKernelB3I: the extra operands pushed their
// 128-row shapes past 256 registers): the GEMM writes plain floats, then this pass adds the residual and cuts the operand image.  Same
// float operations as GemmKernelB3J's in-register form, so a layer's result does not depend on which of the kernels ran it.
GemmDev GemmWithoutResidual(const GemmDev &d) {
  GemmDev g = d;
  g.res = nullptr; g.res_ld = 0; g.res_scale = 1.0f;
  g.res_img = ActImage{nullptr, 0, 0, 0};
  g.out_img = ActImage{nullptr, 0, 0, 0};
  g.write_f32 = 1;
  return g;
}
void LaunchResidualAdd(const GemmDev &d, int rows, hipStream_t s) {
  if (rows <= 0) return;
  ResidualDev rd{d.res_img.base ? nullptr : d.res, d.res_ld, d.res_scale, d.res_img, d.write_f32 || !d.out_img.base ? 1 : 0};
  hipLaunchKernelGGL(ToImageKernel, dim3((rows + 31) / 32), dim3(256), 0, s, d.out, d.ldo, d.n, rows, d.out_img, d.ovf, d.row_map, rd);
}

bool GemmImagesEnabled() {
  const char *e = std::getenv("RS_GEMM_B3I"), *e3 = std::getenv("RS_GEMM_B3");          // read per call (tests flip them)
  return !(e && std::atoi(e) == 0) && !(e3 && std::atoi(e3) == 0);
}

bool GemmB3IUsable(const GemmDev &d) {
  if (!GemmImagesEnabled() || !d.W3I || d.n3 < kB3BN) return false;
  if (!GemmB3PaddingOk(d.n, d.n3)) return false;
  for (int i = 0; i < d.nsegs; i++)
    if (!d.segs[i].img.base || d.segs[i].per_utt || (d.segs[i].col0 % kB3KS) != 0) return false;
  if (d.interleave)
    for (int i = 1; i < d.nsegs; i++) if (d.segs[i].img.base != d.segs[0].img.base) return false;
  return true;
}

void LaunchGemmB3I(const GemmDev &d0, int rows, hipStream_t s) {
  const GemmDev d = d0.res ? GemmWithoutResidual(d0) : d0;
  static int num_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  static int force_mr = [] { const char *e = TuneEnv("RS_GEMM_B3_MR"); return e ? std::atoi(e) : 0; }();
  const int ncol = (d.n + kB3BN - 1) / kB3BN;
  const long slots = std::max(2L * num_cu / std::max(d.share, 1), 8L);      // two workgroups per CU; the device may be shared
  // Tile height: rounds of `slots` tiles, each as long as the tile is tall, weighted by the per-row cost of the height
  // (a 64-row tile streams the weights for half as many rows as a 128-row one)
  auto rounds = [&](long row_tiles) { return (double)((row_tiles * ncol + slots - 1) / slots); };
  static const double eff64 = [] { const char *e = TuneEnv("RS_GEMM_B3I_EFF64"); return e ? std::atof(e) : 1.3; }();
  // whole rounds of 128-row tiles, the remaining rows as 64-row tiles of the same launch
  const long full = (long)(rows / 128) * ncol / slots * slots / ncol;
  const long rest = rows - full * 128;
  const double c_mixed = rounds(full) * 128 + rounds((rest + 63) / 64) * 64 * eff64;
  const double c_128 = rounds((rows + 127) / 128) * 128, c_64 = rounds((rows + 63) / 64) * 64 * eff64;
  // a launch of less than one round (a stream advance: a few thousand rows) is as long as ONE tile is: the 32-row tile spreads it
  // over four times as many CUs as the 128-row one, each streaming the same weights for a quarter of the rows
  static const double eff32 = [] { const char *e = TuneEnv("RS_GEMM_B3I_EFF32"); return e ? std::atof(e) : 1.7; }();
  const double c_32 = rounds((rows + 31) / 32) * 32 * eff32;
  int mr = 4, nbig = (rows + 127) / 128;
  bool mixed = false;
  if (c_32 < c_64 && c_32 < c_128 && c_32 <= c_mixed) { mr = 1; nbig = (rows + 31) / 32; }
  else if (c_64 < c_128 && c_64 <= c_mixed) { mr = 2; nbig = (rows + 63) / 64; }
  else if (full > 0 && c_mixed < c_128) { mixed = true; nbig = (int)full; }
  if (force_mr == 1) { mr = 1; nbig = (rows + 31) / 32; mixed = false; }
  if (force_mr == 2) { mr = 2; nbig = (rows + 63) / 64; mixed = false; }
  if (force_mr == 4) { mr = 4; nbig = (rows + 127) / 128; mixed = false; }
  static const int kps1 = [] { const char *e = TuneEnv("RS_GEMM_B3I_KPS"); return e ? std::atoi(e) : 8; }();
  if (mr == 1 && GemmB3JSmallUsable(d0)) { LaunchGemmB3JSmall(d0, rows, s); return; }      // (adds a folded residual itself)
  if (mr == 1 && kps1 == 8) LaunchB3I<1, false, 8>(d, rows, nbig, s);
  else if (mr == 1 && kps1 == 4) LaunchB3I<1, false, 4>(d, rows, nbig, s);
  else if (mr == 1) LaunchB3I<1, false>(d, rows, nbig, s);
  else if (mr == 2) LaunchB3I<2, false>(d, rows, nbig, s);
  else if (mixed) LaunchB3I<4, true>(d, rows, nbig, s);
  else LaunchB3I<4, false>(d, rows, nbig, s);
  if (d0.res) LaunchResidualAdd(d0, rows, s);
}

}  // namespace rs
