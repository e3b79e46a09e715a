// The image-fed TDNN layer GEMM (nnet_gemm_b3i.hip) with BOTH operands staged through LDS by DMA and every wait placed by
// hand.  Same arithmetic, bit-identical results.
//
// Why: the timing ablations of GemmKernelB3I (profiles/micro/b3i_ablate.sh, profiles/r02/b3i_ablate.txt) show that its
// data movement and its matrix-core work do not overlap at all -- 106 us with the loads removed, 65 us with the MFMAs removed,
// 183 us together -- and the ISA shows why: the compiler's wait-count insertion puts `s_waitcnt vmcnt(0)` (a) in front of every
// LDS-DMA issue that follows ordinary loads (draining the weight prefetch it was just given) and (b) in front of the first
// ds_read after a DMA issue (draining the DMA of the NEXT stage before the MFMAs of this one start): whatever it cannot prove
// independent of an in-flight LDS-DMA waits for everything.  Here the loop contains no memory instruction the compiler models:
//   * weights travel like the activations: global_load_lds_dwordx4 into LDS (the weight image is already in fragment order);
//     all in-flight memory operations are then DMAs of one kind, counted by vmcnt in issue order;
//   * fragments are read with ds_read_b128 in inline asm and released with hand-written lgkmcnt waits that carry the registers
//     they release as operands (the MFMAs that use them cannot be scheduled above the wait);
//   * one `s_waitcnt vmcnt(N)` per k-step leaves the DMAs of the k-steps ahead in flight.
// Two shapes (template parameter WM = wave rows of four waves):
//   WM = 1 (default): 128 x 256 tile, four waves, ring of three k-step stages (8 KiB activations + 16 KiB weights each, 72 KiB),
//     two workgroups per CU -- one's epilogue and pipeline fill hide behind the other's loop; the DMA runs two k-steps ahead
//     with counted vmcnt waits (rounds 2-3, three bf16 parts: 36 KiB per stage, two stages, one k-step ahead).
//   WM = 2 (RS_GEMM_B3J_WM=2): 256 x 256 tile, eight waves, the two wave rows share the weight fragments (half the L2 weight
//     stream per row), ring of three stages (96 KiB), one workgroup per CU, DMA two k-steps ahead with counted vmcnt waits.
//     Measured slower (232 vs 179 us per hidden layer): with one workgroup per CU nothing hides a tile's epilogue -- all CUs
//     write their 129 MB of output at the same moment -- nor its pipeline fill (profiles/r02/b3j_wm2_ablate.txt).
// Rows that do not fill whole rounds of full-height tiles run as half-height tiles of the same launch.
// STRIP (round 5): a TDNN layer's three spliced k-steps read the SAME 16 input columns at three row offsets (nnet-tdnn-component.cc:
// 181-213, sum over offsets of in[t + o] W_o^T).  Instead of one set of activation fragments per offset (3 x 8 DMA instructions per
// workgroup) ONE strip of 192 rows x 16 columns (both parts; the tile's 128 rows + the offsets' span) is staged per 16-column
// group -- 12 instructions, three per wave -- in [part][k-group][row][16 B] order, and the fragment of offset o is the same
// ds_read_b128 pattern started o rows further down: fewer DMA INSTRUCTIONS per MFMA, which is what the loop is bound by (a wave held
// at a global_load_lds issues no MFMA, profiles/r04/b3j_notes.txt).  Same fragments, same products in the same order: bit-identical.
// What the ablations of the default shape say is left (profiles/r02/b3j_wm1_ablate.txt): data movement alone 118 us, matrix
// cores alone 122 us, together 179 us.  Two workgroups per CU move 72 KiB per k-step = 56 B/clk/CU of the 64 the L2 -> CU path
// delivers, so the loads are throughput-bound for as long as the MFMAs run; only a tile that re-uses the weights across more
// rows per CU (WM = 2 with its epilogue overlapped by a persistent loop over tiles) lowers that.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "env.h"

#include "nnet_b3_common.h"

namespace rs {
namespace {
using namespace b3;

constexpr int kJP = kB3Parts;
// WM wave rows of four waves each, MRT 32-row blocks per wave:
//   WM = 2 -> 256-row tile, 512 threads, three stages (96 KiB, one workgroup per CU);
//   WM = 1, MRT = 4 -> 128-row tile, 256 threads, three stages (72 KiB, two workgroups per CU: one's epilogue hides behind the other's loop);
//   WM = 1, MRT = 5 (round 6) -> 160-row tile, 160 accumulator registers per wave.  A tile's k loop is bound by the DMA instructions that
//     stage its WEIGHTS (16 KiB per k-step whatever the height: profiles/r04/b3j_notes.txt), so a launch costs about as many loop times as
//     it has rounds of tiles, full-height or not; the headline's hidden layers have 76-84 k rows = 596-653 tiles of 128 rows = two rounds of
//     the 512 slots, and 477-522 tiles of 160.  The launcher takes this shape where it saves a round (LaunchGemmB3J).
//   WM = 2, WN = 2 (round 6, "narrow"): 256 x 128 tile of four waves, two wave rows x two wave columns, for layers of at most 128 columns
//     (a factorised TDNN's bottlenecks): the same 24 KiB and 24 MFMAs per wave and k-step as the 128 x 256 tile, all of them useful --
//     on the 256-column shapes half of such a layer's weight stream and MFMAs are padding.
template <int WM, int MRT = 4, int WN = 4> struct JShape {
  static constexpr int kThreads = 64 * WM * WN, kWaves = WM * WN, kRowBlocks = MRT * WM;
  static constexpr int kColTiles = 2 * WN;                                  // 32-column tiles per tile
  static constexpr int kBBytes = kColTiles * kJP * kB3FragBytes;            // 16 KiB (8 narrow): [column tile][part] fragments
  static constexpr int kABytes = kRowBlocks * kJP * kB3FragBytes;         // [part][row block] fragments
  static constexpr int kStage = kABytes + kBBytes;
#ifndef RS_B3J_STAGES
#define RS_B3J_STAGES 3
#endif
#ifndef RS_B3J_STAGES_WM2
#define RS_B3J_STAGES_WM2 4
#endif
  static constexpr int kStages = (WM == 2 && WN == 4) ? RS_B3J_STAGES_WM2 : RS_B3J_STAGES, kAhead = kStages - 1;      // (eight waves: one workgroup per CU, four stages)
  static constexpr int kColTilesPerWave = kColTiles / kWaves;              // weight column tiles a wave stages
};

// Timing ablations (results WRONG with a bit set): 512 = weight DMAs of different workgroups ask for different k-steps, 1024 = no weight DMA, 2048 = no activation DMA, 4 = no MFMAs, 8 = no DMA, 16 = no per-k-step barrier, 64 = no epilogue, 128 = epilogue without its image stores, 256 = image stores folded into a 1 MB window (no HBM write stream)
#ifndef RS_B3J_ABLATE
#define RS_B3J_ABLATE 0
#endif
#define RS_DS_READ(dst, addr, off) __asm__ volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// 64 lanes x 16 bytes from per-lane global addresses to LDS at lds_addr + 16 lane (M0 = wave-uniform LDS byte address).  In asm
// because the compiler drains vmcnt in front of an LDS-DMA builtin that follows other LDS-DMAs still in flight.
#define RS_DMA16(lds_addr, gptr) \
  if (!(RS_B3J_ABLATE & 8)) __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(gptr) : "memory")
// the same with a cache policy suffix for the activation stream (profiles/micro/b3j_policy.sh: "", " nt", " sc1", " sc0 sc1" are
// within 1 % of each other -- the vector L1 does not turn the two workgroups' identical weight reads into one)
#ifndef RS_B3J_A_POLICY
#define RS_B3J_A_POLICY ""
#endif
#define RS_DMA16_STREAM(lds_addr, gptr) \
  if (!(RS_B3J_ABLATE & 8)) __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" RS_B3J_A_POLICY : : "s"(lds_addr), "v"(gptr) : "memory")

#ifdef RS_B3J_NT_STORE
#define RS_IMG_STORE(ptr, v) __builtin_nontemporal_store(v, ptr)
#else
#define RS_IMG_STORE(ptr, v) *(ptr) = (v)
#endif

// -DRS_B3J_TRACE: every workgroup leaves {start, end of the k loop, end of the epilogue} (s_memrealtime, 10 ns ticks), the
// hardware id of its CU and the shader clocks its k loop took (s_memtime); RS_B3J_TRACE_FILE=<path> makes the launcher dump the records of one hidden-layer launch
// (profiles/micro/b3j_trace.sh, b3j_trace_read.py)
#ifdef RS_B3J_TRACE
__device__ unsigned long long g_b3j_trace[8192 * 6];
#define RS_TRACE(SLOT) if (threadIdx.x == 0 && blockIdx.x < 8192) { g_b3j_trace[blockIdx.x * 6 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); if ((SLOT) < 2) g_b3j_trace[blockIdx.x * 6 + 4 + (SLOT)] = __builtin_amdgcn_s_memtime(); }
#define RS_TRACE_ID() if (threadIdx.x == 0 && blockIdx.x < 8192) { unsigned hw; __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); g_b3j_trace[blockIdx.x * 6 + 3] = ((unsigned long long)xcc << 32) | hw; }
#else
#define RS_TRACE(SLOT) do { } while (0)
#define RS_TRACE_ID() do { } while (0)
#endif

// SDIV: the small tiles of a MIXED launch are 1 / SDIV of the full height (2: the tail of a batch launch; 4: a launch of 32-row tiles
// only -- a stream advance's few thousand rows, one tile's worth of time on four times the CUs)
template <int WM, bool MIXED, bool STRIP, int SDIV = 2, int MRT = 4, int WN = 4>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) void GemmKernelB3J(GemmDev d, int rows, int nbig, int nfirst, int epi_mode) {
  RS_TRACE(0);
  RS_TRACE_ID();
  typedef JShape<WM, MRT, WN> SH;
  constexpr int kJBBytes = SH::kBBytes;
  static_assert(!STRIP || (WM == 1 && WN == 4 && SH::kStages == 3), "the strip form: four waves side by side, ring of three weight stages");
  static_assert(WN == 4 || (WN == 2 && WM == 2 && MRT == 4), "the narrow shape: two wave rows x two wave columns");
  static_assert(MRT == 4 || (MRT == 5 && WM == 1 && SDIV == 2), "five row blocks per wave: the 160-row tile of four waves");
  constexpr int kJRowBlocks = SH::kRowBlocks, kJABytes = SH::kABytes, kJAhead = SH::kAhead, kJThreads = SH::kThreads, kJWaves = SH::kWaves;
  // STRIP: a stage of the ring holds a k-step's weights only; behind the ring two strips (the 16-column group in use, the next one)
  constexpr int kJStage = STRIP ? kJBBytes : SH::kStage, kBOff = STRIP ? 0 : kJABytes;
  // the strip: the tile's rows + the 64 rows the offsets may span, staged 64 rows per instruction (the last one half masked when the
  // count is not a multiple of 64)
  constexpr int kStripRows = 32 * MRT + 64, kStripBytes = kJP * 2 * kStripRows * 16, kStripInstr = (kStripRows + 63) / 64;
  constexpr unsigned kStripBase = (unsigned)SH::kStages * kJBBytes;
  constexpr int MR = MRT, BM = 32 * kJRowBlocks, BN = 64 * WN;
  constexpr int kSmallBlocks = MR / SDIV, kSmallBM = 32 * WM * kSmallBlocks;      // a MIXED launch's small tiles
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ncol = (d.n + BN - 1) / BN;
  // Block order: `nfirst` half-height tiles, then the full-height tiles, then the remaining half-height ones.  With all tiles of
  // a round the same height every workgroup reaches its epilogue at the same moment and the launch pays for a burst of output
  // stores HBM cannot absorb (40 us of a 178 us hidden layer, profiles/r02/b3j_wm1_ablate.txt); half-height tiles in the first
  // round put the rounds of the two halves of the device out of phase, so most stores drain under somebody else's MFMAs.
  // (nfirst < 0: the first 2 |nfirst| tiles alternate instead, eight half-height, eight full-height, ... -- which of the two orders
  // puts the two workgroups of a CU out of phase depends on how the dispatcher places consecutive workgroups)
  const bool alt = nfirst < 0;
  if (alt) nfirst = -nfirst;
  const int big_blocks = (nbig + 7) / 8 * 8 * ncol, first_blocks = MIXED ? (nfirst + 7) / 8 * 8 * ncol : 0;
  bool small;
  int bid;
  if (!MIXED) { small = false; bid = blockIdx.x; }
  else if (!alt) {
    small = (int)blockIdx.x < first_blocks || (int)blockIdx.x >= first_blocks + big_blocks;
    bid = small ? ((int)blockIdx.x < first_blocks ? blockIdx.x : blockIdx.x - big_blocks) : blockIdx.x - first_blocks;
  } else {
    const int b = blockIdx.x, gsz = 8 * ncol;
    if (b < 2 * first_blocks) {
      const int grp = b / gsz, within = b % gsz;
      small = (grp & 1) == 0;
      bid = (grp >> 1) * gsz + within;
    } else {
      const int b2 = b - 2 * first_blocks, big_left = big_blocks - first_blocks;
      small = b2 >= big_left;
      bid = first_blocks + (small ? b2 - big_left : b2);
    }
  }
  const int mr_eff = small ? kSmallBlocks : MR;                  // 32-row blocks per wave
  const int xcd = bid & 7, local = bid >> 3;
  const int rt = (local / ncol) * 8 + xcd, ct = local % ncol;
  const int row0 = small ? nbig * BM + rt * kSmallBM : rt * BM, n0 = ct * BN;
  if (small ? row0 >= rows : rt >= nbig) return;
  f32x16 acc[MR][2];
#pragma unroll
  for (int i = 0; i < MR; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // ---- staging: wave w copies activation row block w (three parts) and weight column tile w (three parts) of every k-step
  const int nrb = small ? kSmallBlocks * WM : kJRowBlocks;
  // (wave w stages row block w and, where the tile has more blocks than the workgroup waves, block w + waves too)
  const int n_my = __builtin_amdgcn_readfirstlane((wave < nrb ? 1 : 0) + (wave + kJWaves < nrb ? 1 : 0));
  const bool stager = n_my > 0;
  int grow = row0 + wave * 32 + (lane & 31);
  if (grow >= rows) grow = 0;                          // clamped rows are dropped in the epilogue
  if (d.row_map) grow = d.row_map[grow];
  int grow2 = row0 + (wave + kJWaves) * 32 + (lane & 31);
  if (grow2 >= rows || kJRowBlocks <= kJWaves) grow2 = 0;
  if (d.row_map && kJRowBlocks > kJWaves) grow2 = d.row_map[grow2];
  const int kg_off = (lane >> 5) * 512;
  const int sl_ = lane < d.nsegs ? (lane < kMaxSegs ? lane : 0) : 0;
  int seg_rowoff_v = lane < d.nsegs ? d.segs[sl_].row_off : 0;
  int seg_ks0_v = lane < d.nsegs ? d.segs[sl_].col0 / kB3KS : 0;
  int seg_nks_v = lane < d.nsegs ? (d.segs[sl_].ncols + kB3KS - 1) / kB3KS : 0;
  const int nsegs = d.nsegs;
  const bool inter = d.interleave != 0;
  int nt = 0;
  for (int sgi = 0; sgi < d.nsegs; sgi++) nt += (d.segs[sgi].ncols + kB3KS - 1) / kB3KS;
  if (nt == 0) return;
  int seg = 0, ks = 0;
  // per-segment image descriptors in lanes too: the loop must not contain scalar memory loads (they share lgkmcnt with the
  // hand-counted ds_reads and return out of order)
  const unsigned long long seg_base_u = lane < d.nsegs ? (unsigned long long)(uintptr_t)d.segs[sl_].img.base : 0ull;
  const unsigned long long seg_part_u = lane < d.nsegs ? (unsigned long long)d.segs[sl_].img.part_bytes : 0ull;
  int seg_base_lo = (int)(unsigned)seg_base_u, seg_base_hi = (int)(unsigned)(seg_base_u >> 32);
  int seg_part_lo = (int)(unsigned)seg_part_u, seg_part_hi = (int)(unsigned)(seg_part_u >> 32);
  int seg_inks_v = lane < d.nsegs ? d.segs[sl_].img.nks : 0, seg_guard_v = lane < d.nsegs ? d.segs[sl_].img.guard : 0;
  // Everything loaded above is consumed here once, unconditionally: otherwise the compiler's wait-count analysis carries "load
  // pending" around the loop's back edge and drains vmcnt at the first use in EVERY iteration (seen in the ISA).
  __asm__ volatile("" : "+v"(grow2));
  __asm__ volatile("" : "+v"(grow), "+v"(seg_rowoff_v), "+v"(seg_ks0_v), "+v"(seg_nks_v), "+v"(seg_base_lo), "+v"(seg_base_hi), "+v"(seg_part_lo),
                   "+v"(seg_part_hi), "+v"(seg_inks_v), "+v"(seg_guard_v));
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  // STRIP: this wave's share of a strip is (part wave >> 1, k-group wave & 1): three instructions of 64 rows.  Per-lane source
  // addresses for the segment's first k-step; strip row r = image row of (tile row 0 + smallest offset) + r, clamped to the image.
  unsigned long long sp0 = 0, sp1 = 0, sp2 = 0, sp3 = 0;
  int a_row[MR];
#pragma unroll
  for (int i = 0; i < MR; i++) a_row[i] = 0;
  int sh1 = 0, sh2 = 0;                                  // byte shift of the second / third offset's fragments inside the strip
  if constexpr (STRIP) {
    const ActImage im = d.segs[0].img;
    const int r0 = d.segs[0].row_off, max_phys = (int)(im.part_bytes / ((size_t)im.nks * kB3FragBytes)) * 32 - 1;
    const unsigned char *b0 = im.base + (size_t)(wave >> 1) * im.part_bytes + (size_t)(d.segs[0].col0 / kB3KS) * kB3FragBytes + (wave & 1) * 512;
    const int p0 = d.row_map ? d.row_map[row0] : row0;       // physical row of the tile's first row: strip row r = physical row p0 + r0 + r
    auto at = [&](int j) {
      int phys = p0 + r0 + j * 64 + lane + im.guard;
      phys = phys < 0 ? 0 : (phys > max_phys ? max_phys : phys);
      return (unsigned long long)(uintptr_t)(b0 + (size_t)(phys >> 5) * im.nks * kB3FragBytes + (phys & 31) * 16);
    };
    sp0 = at(0); sp1 = at(1); sp2 = at(2);
    if (kStripInstr > 3) sp3 = at(3);
#pragma unroll
    for (int i = 0; i < MR; i++) {      // strip row (x 16 bytes) of this lane's row of row block i: rows of a tile are not consecutive under a row map
      const int gr = row0 + (wm * mr_eff + i) * 32 + (lane & 31);
      const bool ok = gr < rows && i < mr_eff;
      a_row[i] = ok ? ((d.row_map ? d.row_map[gr] : gr) - p0) * 16 : 0;
    }
    sh1 = __builtin_amdgcn_readfirstlane((d.segs[1].row_off - r0) * 16);
    sh2 = __builtin_amdgcn_readfirstlane((d.segs[2].row_off - r0) * 16);
    __asm__ volatile("" : "+v"(sp0), "+v"(sp1), "+v"(sp2), "+s"(sh1), "+s"(sh2), "+v"(a_row[0]), "+v"(a_row[1]), "+v"(a_row[2]), "+v"(a_row[3]));
    if constexpr (MR > 4) __asm__ volatile("" : "+v"(sp3), "+v"(a_row[MR - 1]));
  }
  auto stage_strip = [&](int ks16, unsigned buf) __attribute__((always_inline)) {      // the strip of 16-column group ks16 into strip buffer buf
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + kStripBase + buf * kStripBytes + (unsigned)wave * (kStripRows * 16));
    const unsigned long long ko = (unsigned long long)ks16 * kB3FragBytes;
    RS_DMA16_STREAM(dst, reinterpret_cast<const unsigned char *>((uintptr_t)(sp0 + ko)));
    RS_DMA16_STREAM(dst + 1024u, reinterpret_cast<const unsigned char *>((uintptr_t)(sp1 + ko)));
    RS_DMA16_STREAM(dst + 2048u, reinterpret_cast<const unsigned char *>((uintptr_t)(sp2 + ko)));
    if constexpr (kStripInstr > 3) {      // rows 192 .. of a taller strip; beyond its last row (a half instruction) the lanes are off
      if (kStripRows % 64 == 0 || lane < kStripRows % 64) RS_DMA16_STREAM(dst + 3072u, reinterpret_cast<const unsigned char *>((uintptr_t)(sp3 + ko)));
    }
  };
  constexpr int CTW = SH::kColTilesPerWave;
  const unsigned char *wsrc = reinterpret_cast<const unsigned char *>(d.W3I) + (size_t)(n0 / 32 + wave * CTW) * kJP * kB3FragBytes + lane * 16;
  const size_t wstep = (size_t)(d.n3 / 32) * kJP * kB3FragBytes;
  const bool wtile_ok = n0 / 32 + wave * CTW + CTW <= d.n3 / 32;     // (tiles past the padded width: their columns are dropped in the epilogue)
  auto stage_kstep = [&](unsigned soff) __attribute__((always_inline)) {     // soff: byte offset of this k-step's stage in LDS
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + soff);
    if (!STRIP && stager && !(RS_B3J_ABLATE & 2048)) {
      const unsigned char *img_base = reinterpret_cast<const unsigned char *>(
          (uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(seg_base_hi, seg) << 32) | (unsigned)__builtin_amdgcn_readlane(seg_base_lo, seg)));
      const size_t part_bytes = (size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(seg_part_hi, seg) << 32) | (unsigned)__builtin_amdgcn_readlane(seg_part_lo, seg));
      const int img_nks = __builtin_amdgcn_readlane(seg_inks_v, seg), img_guard = __builtin_amdgcn_readlane(seg_guard_v, seg);
      const int phys = grow + __builtin_amdgcn_readlane(seg_rowoff_v, seg) + img_guard;
      const unsigned char *src = img_base + ((size_t)(phys >> 5) * img_nks + (__builtin_amdgcn_readlane(seg_ks0_v, seg) + ks)) * kB3FragBytes +
                                 kg_off + (phys & 31) * 16;
#pragma unroll
      for (int p = 0; p < kJP; p++) {
        const unsigned char *g = src + p * part_bytes;
        RS_DMA16_STREAM(dst + (unsigned)((p * kJRowBlocks + wave) * kB3FragBytes), g);
      }
      if (kJRowBlocks > kJWaves && n_my == 2) {
        const int phys2 = grow2 + __builtin_amdgcn_readlane(seg_rowoff_v, seg) + img_guard;
        const unsigned char *src2 = img_base + ((size_t)(phys2 >> 5) * img_nks + (__builtin_amdgcn_readlane(seg_ks0_v, seg) + ks)) * kB3FragBytes +
                                    kg_off + (phys2 & 31) * 16;
#pragma unroll
        for (int p = 0; p < kJP; p++) {
          const unsigned char *g = src2 + p * part_bytes;
          RS_DMA16_STREAM(dst + (unsigned)((p * kJRowBlocks + wave + kJWaves) * kB3FragBytes), g);
        }
      }
    }
    if (!(RS_B3J_ABLATE & 1024)) {
      const unsigned char *ws = wtile_ok ? wsrc : reinterpret_cast<const unsigned char *>(d.W3I) + lane * 16;
      if (RS_B3J_ABLATE & 512) {      // every workgroup asks for a different k-step's weights at any moment (is it the same L2 lines for all?)
        const size_t off = (size_t)(ws - reinterpret_cast<const unsigned char *>(d.W3I)) + (size_t)(bid % 16) * 2 * wstep;
        ws = reinterpret_cast<const unsigned char *>(d.W3I) + off % ((size_t)nt * wstep);
      }
#pragma unroll
      for (int p = 0; p < kJP * CTW; p++) {
        const unsigned char *g = ws + p * kB3FragBytes;
        RS_DMA16(dst + (unsigned)(kBOff + (wave * kJP * CTW + p) * kB3FragBytes), g);
      }
      wsrc += wstep;
    }
    if (STRIP) {
    } else if (inter) {
      if (++seg == nsegs) { seg = 0; ks++; }
    } else if (++ks >= __builtin_amdgcn_readlane(seg_nks_v, seg)) {
      ks = 0;
      if (seg + 1 < nsegs) seg++;
    }
  };
  // ---- one k-step of MFMAs from stage `sbase` (LDS byte address of the stage)
  // (STRIP: a fragment row is a strip row: lane l reads strip row a_row[i] / 16 + the offset's shift, of k-group l >> 5; + part * 2 * 192 rows)
  const unsigned a_lane = STRIP ? lds0 + kStripBase + (unsigned)((lane >> 5) * kStripRows * 16)
                                : lds0 + (unsigned)(wm * mr_eff) * kB3FragBytes + lane * 16;          // + stage; + (part * row blocks + i) KiB
  const unsigned b_lane = lds0 + kBOff + (unsigned)(wn * 2) * kJP * kB3FragBytes + lane * 16;    // + stage; + (j * parts + part) KiB
  auto step = [&](unsigned soff, unsigned a_off) __attribute__((always_inline)) {      // a_off (STRIP): strip buffer + the k-step's offset shift
    const unsigned aa = a_lane + (STRIP ? a_off : soff), ba = b_lane + soff;
    unsigned aa_i[MR];      // (STRIP)
#pragma unroll
    for (int i = 0; i < MR; i++) aa_i[i] = aa + (unsigned)a_row[i];
    f16x8 bf[2][kJP];
    RS_DS_READ(bf[0][0], ba, 0 * 1024); RS_DS_READ(bf[1][0], ba, 2 * 1024);
    f16x8 af[kJP][MR];
    // activation fragments in the order they are used: low part first (pa = 1, 0), row blocks inside
#define RS_A_READ(PA, I) RS_DS_READ(af[PA][I], (STRIP ? aa_i[I] : aa), (STRIP ? (PA) * 2 * kStripRows * 16 : ((PA) * kJRowBlocks + (I)) * 1024))
    RS_A_READ(1, 0);
    RS_A_READ(1, 1);
    RS_DS_READ(bf[0][1], ba, 1 * 1024); RS_DS_READ(bf[1][1], ba, 3 * 1024);
    __asm__ volatile("s_waitcnt lgkmcnt(3)" : "+v"(bf[0][0]), "+v"(bf[1][0]), "+v"(af[1][0]));
#define RS_MFMAS(PA, I)                                                                                       \
    if ((RS_B3J_ABLATE & 4) ? false : (!MIXED || (I) < mr_eff)) {                                             \
      _Pragma("unroll") for (int pb = kJP - 1; pb >= 0; pb--) {                                               \
        if (pb > kJP - 1 - (PA)) continue;                                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                                                         \
          acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][pb], af[PA][I], acc[I][j], 0, 0, 0);       \
      }                                                                                                       \
    }
    // software pipeline over the 8 fragments: the read of fragment n + 2 is issued before the MFMAs of fragment n; the wait
    // in front of fragment n + 1 leaves one read in flight (the weights' low parts, first needed by fragment (0, 0), arrive behind
    // the first two activation fragments)
#define RS_STEP(PA, I, PA1, I1, PA2, I2, N)                                                                   \
    RS_A_READ(PA2, I2);                                                                                       \
    RS_MFMAS(PA, I)                                                                                           \
    __asm__ volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(af[PA1][I1]));
    RS_STEP(1, 0, 1, 1, 1, 2, 3)          // still in flight behind (1, 1): the weights' two low parts and (1, 2)
    RS_STEP(1, 1, 1, 2, 1, 3, 1)
    __asm__ volatile("" : "+v"(bf[0][1]), "+v"(bf[1][1]));          // (that wait covered them too)
    if constexpr (MR == 4) {
      RS_STEP(1, 2, 1, 3, 0, 0, 1)
      RS_STEP(1, 3, 0, 0, 0, 1, 1)
      RS_STEP(0, 0, 0, 1, 0, 2, 1)
      RS_STEP(0, 1, 0, 2, 0, 3, 1)
      RS_MFMAS(0, 2)
      __asm__ volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][3]));
      RS_MFMAS(0, 3)
    } else {
      RS_STEP(1, 2, 1, 3, 1, MR - 1, 1)
      RS_STEP(1, 3, 1, MR - 1, 0, 0, 1)
      RS_STEP(1, MR - 1, 0, 0, 0, 1, 1)
      RS_STEP(0, 0, 0, 1, 0, 2, 1)
      RS_STEP(0, 1, 0, 2, 0, 3, 1)
      RS_STEP(0, 2, 0, 3, 0, MR - 1, 1)
      RS_MFMAS(0, 3)
      __asm__ volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][MR - 1]));
      RS_MFMAS(0, MR - 1)
    }
#undef RS_STEP
#undef RS_MFMAS
#undef RS_A_READ
  };
  // ---- pipeline: DMA of k-step t + 2 is issued in k-step t (after the barrier that says everybody is done with k-step t - 1,
  // whose stage it overwrites); before it, this wave waits for its own DMAs of k-step t, leaving those of k-step t + 1 in flight
  // (-DRS_B3J_STAGES=2: the DMA runs one k-step ahead and the wait is for everything this wave has in flight)
  unsigned strip_cur = 0;                               // (STRIP) strip buffer of the 16-column group the loop is in
  if (STRIP) stage_strip(0, 0u);
  stage_kstep(0u);
  if (kJAhead > 1 && nt > 1) stage_kstep((unsigned)kJStage);
  if (kJAhead > 2 && nt > 2) stage_kstep(2u * (unsigned)kJStage);
  int t = 0;
  constexpr int kOther = kJP * CTW, kOwn = kJP + kOther, kOwn2 = 2 * kJP + kOther;      // DMAs per k-step of a wave that stages no / one / two activation row blocks
#define RS_VMWAIT(N) __asm__ volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory")
// STRIP (three stages, DMA two k-steps ahead, the loop below unrolled by three so that stage S = position of the k-step in its
// 16-column group): what a wave issues in k-step t is the weights of k-step t + 2 (kOther instructions) and, at position 1, the strip
// of the next group (three more).  The wait of k-step t leaves what was issued in k-step t - 1 in flight: 4 + 3 at position 2.
#define RS_WAIT_OWN(S)                                                                                         \
  if (STRIP) { if (t + 1 < nt) { if ((S) == 2) RS_VMWAIT(kOther + kStripInstr); else RS_VMWAIT(kOther); } else RS_VMWAIT(0); }   \
  else if (kJAhead > 2 && t + 2 < nt) { if (n_my == 2) RS_VMWAIT(kOwn2 * 2); else if (n_my == 1) RS_VMWAIT(kOwn * 2); else RS_VMWAIT(kOther * 2); }          \
  else if (kJAhead > 1 && t + 1 < nt) { if (n_my == 2) RS_VMWAIT(kOwn2); else if (n_my == 1) RS_VMWAIT(kOwn); else RS_VMWAIT(kOther); }                   \
  else RS_VMWAIT(0);
#define RS_B3J_KSTEP(S, S2)                                                                                    \
  {                                                                                                            \
    RS_WAIT_OWN(S)                                                                                             \
    if (!(RS_B3J_ABLATE & 16)) __builtin_amdgcn_s_barrier();                                                   \
    if (t + kJAhead < nt) {                                                                                    \
      stage_kstep((unsigned)(S2) * kJStage);                                                                   \
      if (STRIP && (S) == 1) stage_strip((t + 2) / 3, strip_cur ^ 1u);                                         \
    }                                                                                                          \
    step((unsigned)(S) * kJStage, strip_cur * kStripBytes + (unsigned)((S) == 0 ? 0 : (S) == 1 ? sh1 : sh2));   \
    if (STRIP && (S) == 2) strip_cur ^= 1u;                                                                    \
    t++;                                                                                                       \
  }
  static_assert(SH::kStages >= 2 && SH::kStages <= 4, "ring of two to four stages");
  if (SH::kStages == 4) {
#pragma nounroll
    while (t + 4 <= nt) {
      RS_B3J_KSTEP(0, 3)
      RS_B3J_KSTEP(1, 0)
      RS_B3J_KSTEP(2, 1)
      RS_B3J_KSTEP(3, 2)
    }
    if (t < nt) RS_B3J_KSTEP(0, 3)
    if (t < nt) RS_B3J_KSTEP(1, 0)
    if (t < nt) RS_B3J_KSTEP(2, 1)
  } else if (SH::kStages == 3) {
#pragma nounroll
    while (t + 3 <= nt) {
      RS_B3J_KSTEP(0, 2)
      RS_B3J_KSTEP(1, 0)
      RS_B3J_KSTEP(2, 1)
    }
    if (t < nt) RS_B3J_KSTEP(0, 2)
    if (t < nt) RS_B3J_KSTEP(1, 0)
  } else {
#pragma nounroll
    while (t + 2 <= nt) {
      RS_B3J_KSTEP(0, 1)
      RS_B3J_KSTEP(1, 0)
    }
    if (t < nt) RS_B3J_KSTEP(0, 1)
  }
#undef RS_B3J_KSTEP
#undef RS_WAIT_OWN
#undef RS_VMWAIT
  __builtin_amdgcn_s_barrier();                        // the stages become the epilogue's transpose buffer
  RS_TRACE(1);

  if (RS_B3J_ABLATE & 64) { float fs = 0.f; for (int i = 0; i < MR; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) fs += acc[i][j][r]; if (fs == 12345.f) d.out[0] = fs; return; }
  // ---- epilogue.  The MFMAs take the WEIGHT fragment as their first operand, so the accumulators hold the tile transposed:
  // a lane owns output row (lane & 31) of a 32-row block and register r is column (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the
  // 32-column tile -- the same products summed in the same order as with the operands the other way round, bit for bit.  Four
  // consecutive registers are then four consecutive columns of one row, and what a lane lacks of a 16-byte image unit (eight
  // consecutive columns of a row) sits in the lane 32 above / below it: one v_permlane32_swap per register pair and the
  // accumulators ARE the image units.  A layer whose result is only read as an image (every hidden layer) goes from registers to
  // HBM with no LDS transposition, no barrier and no second pass over the tile; the FP32 rows of an output layer still go
  // through LDS, now written four columns at a time.
  {
    constexpr int C_LD = BN + 4, NT = kJThreads;
    float *Cs = reinterpret_cast<float *>(smem);
    float *eb = Cs + 32 * C_LD;                   // [4][BN]: bias, scale, offset of the tile's columns; inverse of the weight image's column scale
    unsigned *rmx = reinterpret_cast<unsigned *>(eb + 4 * BN);      // [BM]: max |x| of the tile's rows over what is split (bits of a non-negative float)
    const int half = lane >> 5;
    for (int c = tid; c < BM; c += NT) rmx[c] = 0u;
    for (int c = tid; c < BN; c += NT) {
      const int col = n0 + c;
      const bool cok = col < d.n;
      eb[c] = (d.bias && cok) ? d.bias[col] : 0.f;
      eb[BN + c] = (epi_mode == 2 && cok) ? d.stages[1].scale[col] : 1.f;
      eb[2 * BN + c] = (epi_mode == 2 && cok) ? d.stages[1].offset[col] : 0.f;
      eb[3 * BN + c] = cok ? d.w3_inv_scale[col] : 0.f;
    }
    bool over = false;                             // a value this thread split into fp16 parts is beyond their range (or not a number)
    dd::LdsBarrier();
    // the fused stages on four consecutive columns (tile-local column CL0 .. CL0 + 3) of one row
#define RS_EPI4(V, CL0)                                                                                        \
    {                                                                                                            \
      const f32x4 b4 = *reinterpret_cast<const f32x4 *>(&eb[(CL0)]);                                             \
      const f32x4 w4 = *reinterpret_cast<const f32x4 *>(&eb[3 * BN + (CL0)]);                                    \
      _Pragma("unroll") for (int e = 0; e < 4; e++) V[e] = __fadd_rn(b4[e], __fmul_rn(V[e], w4[e]));           \
      if (epi_mode == 1 || epi_mode == 2) {                                                                      \
        _Pragma("unroll") for (int e = 0; e < 4; e++) V[e] = V[e] > 0.f ? V[e] : 0.f;                          \
      }                                                                                                          \
      if (epi_mode == 2) {                                                                                       \
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(&eb[BN + (CL0)]);                                      \
        const f32x4 o4 = *reinterpret_cast<const f32x4 *>(&eb[2 * BN + (CL0)]);                                  \
        _Pragma("unroll") for (int e = 0; e < 4; e++) V[e] = __fadd_rn(__fmul_rn(V[e], s4[e]), o4[e]);         \
      } else if (epi_mode == 3) {                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                        \
          const int gc = n0 + (CL0) + e < d.n ? n0 + (CL0) + e : 0;                                              \
          for (int st = 0; st < d.nstages; st++) V[e] = ApplyStage(d.stages[st], V[e], gc);                     \
        }                                                                                                        \
      }                                                                                                          \
    }
    const bool direct = d.out_img.base && !d.write_f32 && (!d.res || d.res_img.base) && !(RS_B3J_ABLATE & (128 | 256));
    if (direct) {
      // destination rows first: a row-map load between two stores makes the compiler wait for vmcnt(0), stores included
      int phys[MR];
      bool rok[MR];
#pragma unroll
      for (int i = 0; i < MR; i++) {
        const int row = row0 + (wm * mr_eff + i) * 32 + (lane & 31);
        rok[i] = row < rows && i < mr_eff;
        phys[i] = rok[i] ? (d.row_map ? d.row_map[row] : row) + d.out_img.guard : 0;
      }
      const int rguard = d.res_img.guard - d.out_img.guard;      // (a folded residual read through its image: the same rows of that image)
      // (the block numbers are macro arguments: acc[] must never be indexed by a variable the compiler might not unroll)
      float rm0 = 0.f, rm1 = 0.f, rm2 = 0.f, rm3 = 0.f, rm4 = 0.f;      // max |x| over what this lane splits of its row of row block 0 .. 4
      // A folded residual's image units of a 32 x 32 block are requested one block AHEAD of the block's own stores: the memory counter
      // runs in issue order, so a load issued behind a block's stores is waited for together with those stores -- one trip to memory
      // and back per block, serially (the layer of the factorised model: 268 us, 201 without its residual).  Two register sets, A / B
      // (whole row blocks ahead -- 64 registers -- spilled the 160-row shapes).
      f16x8 resA[2][2], resB[2][2];      // [k-step of the 32-column tile][part]
      const bool res_on = d.res_img.base != nullptr;
      const unsigned char *res_half = res_on ? d.res_img.base + half * 512 : nullptr;
      const int res_nks = res_on ? d.res_img.nks : 1;
#define RS_RES_LOAD(I, J, R)                                                                                   \
      if (res_on && (!MIXED || (I) < mr_eff)) {                                                                  \
        const int rp = rok[I] ? phys[I] + rguard : 0;          /* (a row that is not stored: any row of the image) */ \
        _Pragma("unroll") for (int ksi = 0; ksi < 2; ksi++) {                                                  \
          int ksg = (n0 + wn * 64 + (J) * 32 + 16 * ksi) >> 4;                                                   \
          ksg = ksg < res_nks ? ksg : res_nks - 1;                                                               \
          const unsigned char *rs = res_half + ((size_t)(rp >> 5) * res_nks + ksg) * kB3FragBytes + (rp & 31) * 16; \
          R[ksi][0] = *reinterpret_cast<const f16x8 *>(rs);                                                      \
          R[ksi][1] = *reinterpret_cast<const f16x8 *>(rs + d.res_img.part_bytes);                               \
        }                                                                                                        \
      }
#define RS_DIRECT(I, J, R)                                                                                     \
      if (!MIXED || (I) < mr_eff) {                                                                              \
        const int cb = wn * 64 + (J) * 32;                                                                       \
        f32x4 q[4];                                                                                              \
        _Pragma("unroll") for (int g = 0; g < 4; g++) {                                                        \
          _Pragma("unroll") for (int e = 0; e < 4; e++) q[g][e] = acc[I][J][4 * g + e];                        \
          RS_EPI4(q[g], cb + 8 * g + 4 * half)                                                                   \
        }                                                                                                        \
        /* lanes 32..63 of q[0] <-> lanes 0..31 of q[1] (and q[2] <-> q[3]): q[2 k], q[2 k + 1] = columns 16 k + 8 half .. + 7 */ \
        _Pragma("unroll") for (int g = 0; g < 4; g += 2)                                                       \
          _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                      \
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[g][e]), __float_as_uint(q[g + 1][e]), false, false); \
            q[g][e] = __uint_as_float(sw[0]);                                                                    \
            q[g + 1][e] = __uint_as_float(sw[1]);                                                                \
          }                                                                                                      \
        _Pragma("unroll") for (int ksi = 0; ksi < 2; ksi++) {                                                  \
          const int col = n0 + cb + 16 * ksi + 8 * half;                                                         \
          if (rok[I] && (col >> 4) < d.out_img.nks) {                                                            \
            f32x4 lo = q[2 * ksi], hi = q[2 * ksi + 1];                                                          \
            if (res_on) {                                                                                        \
              const f16x8 r1 = R[ksi][0], r2 = R[ksi][1];                                                        \
              _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                  \
                lo[e] = __fadd_rn(__fmul_rn((float)r1[e] + (float)r2[e], d.res_scale), lo[e]);                   \
                hi[e] = __fadd_rn(__fmul_rn((float)r1[4 + e] + (float)r2[4 + e], d.res_scale), hi[e]);           \
              }                                                                                                  \
            }                                                                                                    \
            _Pragma("unroll") for (int e = 0; e < 4; e++) { if (col + e >= d.n) lo[e] = 0.f; if (col + 4 + e >= d.n) hi[e] = 0.f; } \
            f16x8 p1, p2;                                                                                        \
            over |= B3Over(Split2(lo, hi, &p1, &p2));                                                            \
            rm##I = B3AbsMax(B3AbsMax(rm##I, lo), hi);                                                           \
            unsigned char *dst = d.out_img.base + ((size_t)(phys[I] >> 5) * d.out_img.nks + (col >> 4)) * kB3FragBytes + half * 512 + (phys[I] & 31) * 16; \
            RS_IMG_STORE(reinterpret_cast<f16x8 *>(dst), p1);                                                    \
            RS_IMG_STORE(reinterpret_cast<f16x8 *>(dst + d.out_img.part_bytes), p2);                             \
          }                                                                                                      \
        }                                                                                                        \
      }
      RS_RES_LOAD(0, 0, resA)
      RS_RES_LOAD(0, 1, resB)
      RS_DIRECT(0, 0, resA) RS_RES_LOAD(1, 0, resA)
      RS_DIRECT(0, 1, resB) RS_RES_LOAD(1, 1, resB)
      RS_DIRECT(1, 0, resA) RS_RES_LOAD(2, 0, resA)
      RS_DIRECT(1, 1, resB) RS_RES_LOAD(2, 1, resB)
      RS_DIRECT(2, 0, resA) RS_RES_LOAD(3, 0, resA)
      RS_DIRECT(2, 1, resB) RS_RES_LOAD(3, 1, resB)
      RS_DIRECT(3, 0, resA)
      if constexpr (MR > 4) { RS_RES_LOAD(4, 0, resA) }
      RS_DIRECT(3, 1, resB)
      if constexpr (MR > 4) { RS_RES_LOAD(4, 1, resB) RS_DIRECT(4, 0, resA) RS_DIRECT(4, 1, resB) }
#undef RS_DIRECT
#undef RS_RES_LOAD
      if (over) d.ovf[0] = 1;
      {
        const int rb = (wm * mr_eff) * 32 + (lane & 31);
        atomicMax(&rmx[rb], __float_as_uint(rm0));
        if (mr_eff > 1) atomicMax(&rmx[rb + 32], __float_as_uint(rm1));
        if (mr_eff > 2) { atomicMax(&rmx[rb + 64], __float_as_uint(rm2)); atomicMax(&rmx[rb + 96], __float_as_uint(rm3)); }
        if (MR > 4 && mr_eff > 4) atomicMax(&rmx[rb + 128], __float_as_uint(rm4));
        dd::LdsBarrier();
        if (tid < BM && B3Under(__uint_as_float(rmx[tid]))) d.ovf[1] = 1;
      }
#ifdef RS_B3J_TRACE
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      RS_TRACE(2);
#endif
      return;
    }
    const bool vec_out = ((d.ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.out) & 15) == 0) && (((d.n + 3) & ~3) <= d.ldo);
    const bool vec_res = d.res && ((d.res_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.res) & 15) == 0) && (((d.n + 3) & ~3) <= d.res_ld);
#define RS_PUT_SLAB(A)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                              \
      _Pragma("unroll") for (int g = 0; g < 4; g++) {                                                            \
        const int cl0 = wn * 64 + j * 32 + 8 * g + 4 * half;                                                     \
        f32x4 v;                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 4; e++) v[e] = A[j][4 * g + e];                                  \
        RS_EPI4(v, cl0)                                                                                          \
        *reinterpret_cast<f32x4 *>(&Cs[(lane & 31) * C_LD + cl0]) = v;                                           \
      }                                                                                                          \
    }
    // Destination rows of everything this thread will store, looked up BEFORE the first store: a row-map load between two
    // stores makes the compiler wait for vmcnt(0) -- which on this ISA also counts the stores in flight -- so every store used
    // to wait for the previous one to reach memory (seen in the ISA; profiles/r02/b3j_wm1_ablate.txt).
    int img_phys[WM * MR];                    // image path: a thread's units of a slab all sit on row tid & 31
    int f32_phys[WM * MR][8 * BN / NT];       // FP32 path: unit q of a slab (32 rows x BN / 4 float4s) sits on row (tid + NT q) / (BN / 4)
#pragma unroll
    for (int sl = 0; sl < WM * MR; sl++) {
      const int row = row0 + sl * 32 + (tid & 31);
      img_phys[sl] = (d.out_img.base && row < rows) ? (d.row_map ? d.row_map[row] : row) + d.out_img.guard : 0;
#pragma unroll
      for (int q = 0; q < 8 * BN / NT; q++) {
        const int r2 = row0 + sl * 32 + (tid + NT * q) / (BN / 4);
        f32_phys[sl][q] = ((d.write_f32 || d.res) && r2 < rows) ? (d.row_map ? d.row_map[r2] : r2) : 0;
      }
    }
    // (the slab number is a macro argument: acc[] must never be indexed by a loop variable the compiler might not unroll -- that put
    // the accumulators in scratch -- and a lambda capturing `d` makes the compiler copy the 1.2 KB argument block to scratch)
#define RS_SLAB(SL)                                                                                            \
    if (!(MIXED && small && (SL) >= WM * kSmallBlocks)) {     /* workgroup-uniform: a half-height tile has half the slabs */ \
      if (MIXED && small) { if (wm == (SL) / kSmallBlocks) { RS_PUT_SLAB(acc[(SL) % kSmallBlocks]) } } \
      else if (wm == (SL) / MR) { RS_PUT_SLAB(acc[(SL) % MR]) } \
      dd::LdsBarrier(); \
      if (d.res) {      /* a folded residual sum (LayerOp::res_buf): see nnet_b3_epilogue.inc */ \
_Pragma("unroll") \
        for (int q = 0; q < 8 * BN / NT; q++) { \
          const int unit = tid + NT * q, rl = unit / (BN / 4), c4 = (unit % (BN / 4)) * 4; \
          const int row = row0 + (SL) * 32 + rl, col = n0 + c4; \
          if (row < rows && col < d.n) { \
            f32x4 v = *reinterpret_cast<const f32x4 *>(&Cs[rl * C_LD + c4]); \
            f32x4 r; \
            if (d.res_img.base) { \
              const int ip = f32_phys[(SL)][q] + d.res_img.guard; \
              const unsigned char *rs = d.res_img.base + ((size_t)(ip >> 5) * d.res_img.nks + (col >> 4)) * kB3FragBytes + ((col >> 3) & 1) * 512 + (ip & 31) * 16 + (col & 7) * 2; \
              const f16x4 r1 = *reinterpret_cast<const f16x4 *>(rs), r2 = *reinterpret_cast<const f16x4 *>(rs + d.res_img.part_bytes); \
_Pragma("unroll") \
              for (int e = 0; e < 4; e++) r[e] = (float)r1[e] + (float)r2[e]; \
            } else { \
              const float *rp = d.res + (size_t)f32_phys[(SL)][q] * d.res_ld + col; \
              if (vec_res) r = *reinterpret_cast<const f32x4 *>(rp); \
              else { for (int e = 0; e < 4; e++) r[e] = col + e < d.n ? rp[e] : 0.f; } \
            } \
_Pragma("unroll") \
            for (int e = 0; e < 4; e++) v[e] = __fadd_rn(d.res_scale != 1.0f ? __fmul_rn(r[e], d.res_scale) : r[e], v[e]); \
            *reinterpret_cast<f32x4 *>(&Cs[rl * C_LD + c4]) = v; \
            if (d.write_f32) { \
              float *op = d.out + (size_t)f32_phys[(SL)][q] * d.ldo + col; \
              if (vec_out) *reinterpret_cast<f32x4 *>(op) = v; \
              else { for (int e = 0; e < 4; e++) if (col + e < d.n) op[e] = v[e]; } \
            } \
          } \
        } \
        dd::LdsBarrier(); \
      } else if (vec_out && d.write_f32) { \
_Pragma("unroll") \
        for (int q = 0; q < 8 * BN / NT; q++) { \
          const int unit = tid + NT * q, rl = unit / (BN / 4), c4 = (unit % (BN / 4)) * 4; \
          const int row = row0 + (SL) * 32 + rl, col = n0 + c4; \
          if (row < rows && col < d.n) \
            *reinterpret_cast<f32x4 *>(d.out + (size_t)f32_phys[(SL)][q] * d.ldo + col) = \
                *reinterpret_cast<const f32x4 *>(&Cs[rl * C_LD + c4]); \
        } \
      } else if (d.write_f32) { \
        for (int idx = tid; idx < 32 * BN; idx += NT) { \
          const int rl = idx / BN, cl = idx % BN; \
          const int row = row0 + (SL) * 32 + rl, col = n0 + cl; \
          if (row < rows && col < d.n) d.out[(size_t)(d.row_map ? d.row_map[row] : row) * d.ldo + col] = Cs[rl * C_LD + cl]; \
        } \
      } \
      if (d.out_img.base) { \
        float rm = 0.f; \
_Pragma("unroll") \
        for (int q = 0; q < 4 * BN / NT; q++) { \
          const int unit = tid + NT * q, rl = unit & 31, kg = (unit >> 5) & 1, ksi = unit >> 6; \
          const int row = row0 + (SL) * 32 + rl, col = n0 + ksi * 16 + kg * 8; \
          if (row < rows && (col >> 4) < d.out_img.nks) { \
            f32x4 lo = *reinterpret_cast<const f32x4 *>(&Cs[rl * C_LD + ksi * 16 + kg * 8]); \
            f32x4 hi = *reinterpret_cast<const f32x4 *>(&Cs[rl * C_LD + ksi * 16 + kg * 8 + 4]); \
_Pragma("unroll") \
            for (int e = 0; e < 4; e++) { if (col + e >= d.n) lo[e] = 0.f; if (col + 4 + e >= d.n) hi[e] = 0.f; } \
            f16x8 p1, p2; \
            over |= B3Over(Split2(lo, hi, &p1, &p2)); \
            rm = B3AbsMax(B3AbsMax(rm, lo), hi); \
            const int phys = img_phys[(SL)]; \
            unsigned char *dst = d.out_img.base + ((size_t)(phys >> 5) * d.out_img.nks + (col >> 4)) * kB3FragBytes + kg * 512 + (phys & 31) * 16; \
            if ((RS_B3J_ABLATE & 128) && p1[0] != (_Float16)12345.f) dst = nullptr; \
            if (RS_B3J_ABLATE & 256) dst = d.out_img.base + ((size_t)(dst - d.out_img.base) & 0xFFFFFu); \
            if (dst) { \
            RS_IMG_STORE(reinterpret_cast<f16x8 *>(dst), p1); \
            RS_IMG_STORE(reinterpret_cast<f16x8 *>(dst + d.out_img.part_bytes), p2); } \
          } \
        } \
        atomicMax(&rmx[(SL) * 32 + (tid & 31)], __float_as_uint(rm)); \
      } \
      dd::LdsBarrier(); \
    }
    RS_SLAB(0) RS_SLAB(1) RS_SLAB(2) RS_SLAB(3)
    if constexpr (WM * MR > 4) { RS_SLAB(4) }
    if constexpr (WM == 2) { RS_SLAB(5) RS_SLAB(6) RS_SLAB(7) }
#undef RS_SLAB
#undef RS_PUT_SLAB
#undef RS_EPI4
    if (over) d.ovf[0] = 1;
    if (d.out_img.base && tid < BM && B3Under(__uint_as_float(rmx[tid]))) d.ovf[1] = 1;
  }
}

template <int WM, bool MIXED, bool STRIP, int SDIV = 2, int MRT = 4, int WN = 4>
void LaunchB3J(const GemmDev &d, int rows, int nbig, int nfirst, hipStream_t s) {
  typedef JShape<WM, MRT, WN> SH;
  constexpr int BM = 32 * SH::kRowBlocks, kSmallBM = 32 * WM * (MRT / SDIV), BN = 64 * WN;
  // (STRIP: the ring holds weights only, two strips of 32 MRT + 64 rows x 16 columns x two parts behind it)
  constexpr size_t ring = STRIP ? (size_t)SH::kStages * SH::kBBytes + 2 * (size_t)(kJP * 2 * (32 * MRT + 64) * 16) : (size_t)SH::kStages * SH::kStage, ctile = kB3EpiBytes;
  constexpr size_t smem0 = ring > ctile ? ring : ctile;
  // RS_GEMM_B3J_ONE_PER_CU=1 (measurement): ask for so much LDS that only one workgroup fits a CU
  static const bool one_per_cu = [] { const char *e = TuneEnv("RS_GEMM_B3J_ONE_PER_CU"); return e && std::atoi(e) != 0; }();
  const size_t smem = (one_per_cu && smem0 < 100 * 1024) ? 100 * 1024 : smem0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&GemmKernelB3J<WM, MIXED, STRIP, SDIV, MRT, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(smem, 100 * 1024));
    attr_set = true;
  }
  const int ncol = (d.n + BN - 1) / BN;
  const int rest = std::max(rows - nbig * BM, 0), nsmall = MIXED ? (rest + kSmallBM - 1) / kSmallBM : 0;
  // the half-height tiles are numbered through both of their block ranges: the first range holds a multiple of 8 of them
  const bool alt = nfirst < 0;
  nfirst = MIXED ? std::min(std::abs(nfirst) / 8 * 8, nsmall / 8 * 8) : 0;
  if (alt && nbig < nfirst) nfirst = 0;
  const int blocks = ((nbig + 7) / 8 * 8 + nfirst + (std::max(nsmall - nfirst, 0) + 7) / 8 * 8) * ncol;
  hipLaunchKernelGGL((GemmKernelB3J<WM, MIXED, STRIP, SDIV, MRT, WN>), dim3(blocks), dim3(SH::kThreads), smem, s, d, rows, nbig, alt ? -nfirst : nfirst, GemmEpiMode(d, rows));
#ifdef RS_B3J_TRACE
  static int traced = 0;
  const char *tf = TuneEnv("RS_B3J_TRACE_FILE");
  static const int trace_at = [] { const char *e = TuneEnv("RS_B3J_TRACE_AT"); return e ? std::atoi(e) : 40; }();
  if (tf && d.out_img.base && !d.write_f32 && ++traced == trace_at) {       // one hidden-layer launch well after warm-up
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(8192 * 6);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_b3j_trace), h.size() * sizeof(unsigned long long));
    if (FILE *f = std::fopen(tf, "w")) {
      std::fprintf(f, "# blocks %d rows %d nbig %d nfirst %d ncol %d\n", blocks, rows, nbig, nfirst, ncol);
      for (int b = 0; b < blocks && b < 8192; b++)
        std::fprintf(f, "%d %llu %llu %llu %llx %llu\n", b, h[b * 6], h[b * 6 + 1], h[b * 6 + 2], h[b * 6 + 3], h[b * 6 + 5] - h[b * 6 + 4]);
      std::fclose(f);
    }
  }
#endif
}

// The strip form applies to a layer whose three segments are the same 16-column groups of ONE image at three ascending row offsets
// no more than 64 rows apart (a TDNN layer's splice); with a row map (layers evaluated on the rows somebody reads) only when the tile's
// rows, skipped halo rows included, still fit the strip.  RS_GEMM_B3J_STRIP=0 (tests, profiles: read per call)
// keeps the one-fragment-set-per-offset form.
bool JStripOk(const GemmDev &d, int tile_rows = 128) {
  const char *e = TuneEnv("RS_GEMM_B3J_STRIP");
  if (e && std::atoi(e) == 0) return false;
  if (!d.interleave || d.nsegs != 3) return false;
  const int span = tile_rows == 128 ? d.row_map_span128 : d.row_map_span160;      // physical rows the tile's list rows reach over
  if (d.row_map && (span <= 0 || span + (d.segs[2].row_off - d.segs[0].row_off) > tile_rows + 64)) return false;
  const GemmSegDev &a = d.segs[0];
  if (!a.img.base || a.per_utt) return false;
  for (int i = 1; i < 3; i++) {
    const GemmSegDev &b = d.segs[i];
    if (b.img.base != a.img.base || b.img.part_bytes != a.img.part_bytes || b.img.nks != a.img.nks || b.img.guard != a.img.guard || b.per_utt ||
        b.col0 != a.col0 || b.ncols != a.ncols || b.row_off <= d.segs[i - 1].row_off)
      return false;
  }
  return a.col0 % kB3KS == 0 && d.segs[2].row_off - a.row_off <= 64;
}

int JWaveRows() {          // RS_GEMM_B3J_WM = 1 | 2 (read per call)
  const char *e = std::getenv("RS_GEMM_B3J_WM");
  return e && std::atoi(e) == 2 ? 2 : 1;
}
long JSlots(const GemmDev &d, int wm) {
  static int num_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const char *es = std::getenv("RS_GEMM_B3J_SLOTS");          // tests: pretend the device runs this many workgroups at a time
  if (es) return std::max(std::atol(es), 1L);
  return std::max((long)(wm == 1 ? 2 : 1) * num_cu / std::max(d.share, 1), 8L);
}

}  // namespace

// RS_GEMM_B3J=0 switches the kernel off, a value > 1 is the smallest launch (rows) it takes (read per call: tests flip it).
bool GemmB3JUsable(const GemmDev &d, int rows) {
  const char *e = std::getenv("RS_GEMM_B3J");
  if (e && std::atoi(e) == 0) return false;
  const int wm = JWaveRows();
  const int ncol = (d.n + kB3BN - 1) / kB3BN;
  // the 256-row tile needs whole rounds to pay off; a launch that GemmKernelB3I's 32-row tiles finish in one round (a stream
  // advance: a few thousand rows) is faster there -- one tile's worth of time on four times as many CUs
  const long min_default = wm == 2 ? JSlots(d, wm) * 256 / ncol : JSlots(d, 1) * 32 / ncol + 1;
  const int min_rows = e && std::atoi(e) > 1 ? std::atoi(e) : (int)std::min<long>(min_default, 1 << 30);
  return rows >= min_rows;
}

// A launch of less than one round of tiles (a stream advance: a few thousand rows) as 32-row tiles of this kernel: GemmKernelB3I's
// ordinary weight loads are waited for with vmcnt(0) in every k-step (the compiler drains the counter in front of the first use of a
// load result while an LDS-DMA is in flight), 0.59 us per k-step for a workgroup alone on its CU; here nothing in the loop is a load
// the compiler sees.  RS_GEMM_B3J_SMALL=0 (read per call: a test compares the two kernels bit for bit) keeps GemmKernelB3I.
bool GemmB3JSmallUsable(const GemmDev &d) {
  const char *e = std::getenv("RS_GEMM_B3J_SMALL");
  if (e && std::atoi(e) == 0) return false;
  const char *e2 = std::getenv("RS_GEMM_B3J");
  return !(e2 && std::atoi(e2) == 0) && JWaveRows() == 1;
}
void LaunchGemmB3JSmall(const GemmDev &d, int rows, hipStream_t s) { LaunchB3J<1, true, false, 4>(d, rows, 0, 0, s); }

void LaunchGemmB3J(const GemmDev &d, int rows, hipStream_t s) {
  const int wm = JWaveRows();
  const int ncol = (d.n + kB3BN - 1) / kB3BN, bm = 128 * wm;
  const long slots = JSlots(d, wm);
  // whole rounds of full-height tiles; the remaining rows as half-height tiles of the same launch
  const long tiles = rows / bm;
  const long full = tiles * ncol / slots * slots / ncol;
  const bool all_big = full * bm >= rows;
  const int nbig = all_big ? (rows + bm - 1) / bm : (int)full;
  static const int stagger = [] { const char *e = TuneEnv("RS_GEMM_B3J_STAGGER"); return e ? std::atoi(e) : 1; }();
  int nfirst = stagger ? (int)(slots / 2) : 0;          // half-height tiles that go first (LaunchB3J clips it to what there is)
  if (stagger == 2 && nbig >= nfirst) nfirst = -nfirst;
  // Layers of at most 128 columns: the 256 x 128 tile (two wave rows x two wave columns).  RS_GEMM_B3J_NARROW=0 (tests: same bits) keeps
  // the 256-column shapes, half of whose weight stream and MFMAs are padding for such a layer.
  if (wm == 1 && d.n <= 128) {
    const char *en = std::getenv("RS_GEMM_B3J_NARROW");
    if (!(en && std::atoi(en) == 0)) {
      const long tiles_n = rows / 256, full_n = tiles_n / slots * slots;
      const bool all_n = full_n * 256 >= rows || (rows + 255) / 256 <= slots;
      if (all_n) LaunchB3J<2, false, false, 2, 4, 2>(d, rows, (rows + 255) / 256, 0, s);
      else LaunchB3J<2, true, false, 2, 4, 2>(d, rows, (int)full_n, 0, s);
      return;
    }
  }
  // The 160-row tile (five row blocks per wave) where it turns a launch of two rounds of tiles into ONE: a tile's k loop is as long as
  // staging its weights takes, whatever its height, so a long-K launch costs about one loop time per round -- the half-height tiles of
  // the last, partly filled round included.  Measured (profiles/r06/notes_experiments.txt): hidden layers (K = 750) 101 -> 90 us; no gain
  // where K is short (the pre-final and output layers, K = 250: the tile's time is its epilogue, which grows with its rows) or where
  // the taller tiles still need several rounds.  RS_GEMM_B3J_MR=4|5 forces a height (tests: same bits either way).
  if (wm == 1) {
    const char *em = std::getenv("RS_GEMM_B3J_MR");
    const int force = em ? std::atoi(em) : 0;
    int ksteps = 0;
    for (int i = 0; i < d.nsegs; i++) ksteps += (d.segs[i].ncols + kB3KS - 1) / kB3KS;
    const long rounds4 = ((long)((rows + 127) / 128) * ncol + slots - 1) / slots, rounds5 = ((long)((rows + 159) / 160) * ncol + slots - 1) / slots;
    if (force == 5 || (force != 4 && rounds5 == 1 && rounds4 > 1 && ksteps >= 32)) {
      const int nbig5 = (rows + 159) / 160;
      if (JStripOk(d, 160)) LaunchB3J<1, false, true, 2, 5>(d, rows, nbig5, 0, s);
      else LaunchB3J<1, false, false, 2, 5>(d, rows, nbig5, 0, s);
      return;
    }
  }
  if (wm == 2) { if (all_big) LaunchB3J<2, false, false>(d, rows, nbig, 0, s); else LaunchB3J<2, true, false>(d, rows, nbig, nfirst, s); }
  else if (JStripOk(d)) { if (all_big) LaunchB3J<1, false, true>(d, rows, nbig, 0, s); else LaunchB3J<1, true, true>(d, rows, nbig, nfirst, s); }
  else { if (all_big) LaunchB3J<1, false, false>(d, rows, nbig, 0, s); else LaunchB3J<1, true, false>(d, rows, nbig, nfirst, s); }
}

}  // namespace rs
