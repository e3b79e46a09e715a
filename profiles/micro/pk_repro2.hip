// Second attempt at an isolated repro of the packed-FP32 interference (profiles/r03/pk_interference.txt): the exact multi-instruction
// sequences of the SLP-vectorized MfccKernel around its packed operations -- a 32-bit v_mov into ONE half of a register pair right in
// front of the v_pk_* that reads the pair (no wait state: the compiler inserts none), a v_pk_* result read by a 32-bit VALU
// instruction / stored to LDS right behind it, LDS data consumed by a v_pk_* right behind the wait -- as single asm blocks on physical
// registers (nothing can be scheduled between the instructions), known answers on small integers, alone and beside an MFMA burner
// that shares every CU.   hipcc --offload-arch=gfx950 -O2 -o pk_repro2 pk_repro2.hip && ./pk_repro2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SEQ>
__global__ __launch_bounds__(256, 2) void victim(unsigned *errors, int rounds) {
  __shared__ float stage[4 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds = (unsigned)(uintptr_t)stage + (unsigned)(wave * 64 + lane) * 16;
  unsigned bad = 0;
  for (int it = 0; it < rounds; it++) {
    const float a0 = (float)((it * 7 + lane) & 255), a1 = (float)((it * 13 + 3 * lane) & 255) + 300.f;
    const float b0 = (float)((it * 5 + 2 * lane) & 255) + 1000.f, b1 = (float)((it * 11 + lane) & 127) + 2000.f, c = (float)((it + lane) & 63) + 5000.f;
    float r0, r1, e0, e1;
    if (SEQ == 0) {
      // v_mov into the high halves of both pairs, then the packed subtract (MfccKernel, length-4 butterfly shuffle)
      __asm__ volatile(
          "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v104, %6\n\ts_nop 4\n\t"
          "v_mov_b32 v103, v101\n\t"
          "v_mov_b32 v101, v104\n\t"
          "v_pk_add_f32 v[100:101], v[102:103], v[100:101] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v100\n\tv_mov_b32 %1, v101"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c) : "v100", "v101", "v102", "v103", "v104");
      e0 = b0 - a0; e1 = a1 - c;
    } else if (SEQ == 1) {
      // v_mov into the high half, packed add of the pair right behind it (power spectrum)
      __asm__ volatile(
          "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v104, %6\n\ts_nop 4\n\t"
          "v_mov_b32 v101, v104\n\t"
          "v_pk_add_f32 v[100:101], v[102:103], v[100:101]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v100\n\tv_mov_b32 %1, v101"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c) : "v100", "v101", "v102", "v103", "v104");
      e0 = b0 + a0; e1 = b1 + c;
    } else if (SEQ == 2) {
      // packed multiply, ONE wait state (the compiler's), 32-bit add of the two halves
      __asm__ volatile(
          "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\ts_nop 4\n\t"
          "v_pk_add_f32 v[100:101], v[102:103], v[100:101]\n\t"
          "s_nop 0\n\t"
          "v_add_f32 %0, v100, v101\n\t"
          "v_sub_f32 %1, v100, v101"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c) : "v100", "v101", "v102", "v103");
      e0 = (b0 + a0) + (b1 + a1); e1 = (b0 + a0) - (b1 + a1);
    } else if (SEQ == 3) {
      // the same with no wait state at all
      __asm__ volatile(
          "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\ts_nop 4\n\t"
          "v_pk_add_f32 v[100:101], v[102:103], v[100:101]\n\t"
          "v_add_f32 %0, v100, v101\n\t"
          "v_sub_f32 %1, v100, v101"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c) : "v100", "v101", "v102", "v103");
      e0 = (b0 + a0) + (b1 + a1); e1 = (b0 + a0) - (b1 + a1);
    } else if (SEQ == 4) {
      // packed result stored to LDS right behind the instruction, read back (the wave's own words)
      __asm__ volatile(
          "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\ts_nop 4\n\t"
          "v_pk_add_f32 v[100:101], v[102:103], v[100:101]\n\t"
          "ds_write2_b32 %7, v100, v101 offset1:1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "ds_read2_b32 v[104:105], %7 offset1:1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c), "v"(lds) : "v100", "v101", "v102", "v103", "v104", "v105", "memory");
      e0 = b0 + a0; e1 = b1 + a1;
    } else if (SEQ == 6 || SEQ == 7) {
      // WRITE AFTER WRITE: a packed add writes a pair, a second packed add, the compiler's s_nop 0, then a 32-bit v_mov OVERWRITES the
      // high half of the first pair (MfccKernel's power spectrum: "v_pk_add v[26:27]; v_pk_add v[16:17]; s_nop 0; v_mov v27, v17;
      // v_pk_add v[26:27], v[10:11], v[26:27]").  SEQ 7: the same with four wait states in front of the v_mov.
      float x0, x1;
#define WAW_SEQ(NOP)                                                                                                          \
      __asm__ volatile(                                                                                                         \
          "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v104, %6\n\tv_mov_b32 v105, %6\n\ts_nop 4\n\t" \
          "v_pk_add_f32 v[106:107], v[100:101], v[102:103] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                       \
          "v_pk_add_f32 v[108:109], v[100:101], v[102:103]\n\t"                                                              \
          NOP                                                                                                                   \
          "v_mov_b32 v107, v109\n\t"                                                                                         \
          "v_pk_add_f32 v[106:107], v[104:105], v[106:107]\n\t"                                                              \
          "s_nop 4\n\tv_mov_b32 %0, v106\n\tv_mov_b32 %1, v107"                                                              \
          : "=v"(x0), "=v"(x1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c)                                                     \
          : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109")
      if (SEQ == 6) WAW_SEQ("s_nop 0\n\t"); else WAW_SEQ("s_nop 4\n\t");
#undef WAW_SEQ
      r0 = x0; r1 = x1;
      e0 = c + (a0 - b0); e1 = c + (a1 + b1);
    } else if (SEQ == 8) {
      // the power spectrum's multiplies with op_sel forms that IGNORE one half of a source pair, that half holding garbage (NaN, inf,
      // denormal, huge -- whatever an earlier kernel left in the register): v_pk_mul_f32 v[12:13], v[12:13], v[14:15] op_sel:[0,1]
      // op_sel_hi:[0,0] and v_pk_mul_f32 v[14:15], v[14:15], v[26:27] op_sel_hi:[1,0]
      const unsigned junk[4] = {0x7fc12345u, 0x7f800000u, 0x00000123u, 0x7f7fffffu};
      const float g = __uint_as_float(junk[(it + lane) & 3]);
      float x0, x1, y0, y1;
      __asm__ volatile(
          "v_mov_b32 v112, %4\n\tv_mov_b32 v113, %8\n\tv_mov_b32 v114, %5\n\tv_mov_b32 v115, %6\n\tv_mov_b32 v126, %7\n\tv_mov_b32 v127, %8\n\ts_nop 4\n\t"
          "v_pk_mul_f32 v[112:113], v[112:113], v[114:115] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
          "v_pk_mul_f32 v[114:115], v[114:115], v[126:127] op_sel_hi:[1,0]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v112\n\tv_mov_b32 %1, v113\n\tv_mov_b32 %2, v114\n\tv_mov_b32 %3, v115"
          : "=v"(x0), "=v"(x1), "=v"(y0), "=v"(y1) : "v"(a0), "v"(b0), "v"(b1), "v"(a1), "v"(g)
          : "v112", "v113", "v114", "v115", "v126", "v127");
      // a0 * b1, a0 * b0 | b0 * a1, b1 * a1: products of small integers, exact
      r0 = x0 + y0; r1 = x1 - y1;
      e0 = a0 * b1 + b0 * a1; e1 = a0 * b0 - b1 * a1;
    } else if (SEQ == 9 || SEQ == 10) {
      // THE instruction the ISA-level bisect of MfccKernel singled out (pk_asm.sh: unpacking this one alone makes the wrong rows go
      // away): a packed multiply IN PLACE whose HIGH result takes the LOW word of source 0 -- the word the LOW result overwrites:
      //   v_pk_mul_f32 v[12:13], v[12:13], v[14:15] op_sel:[0,1] op_sel_hi:[0,0]       lo = s0.lo * s1.hi,  hi = s0.lo * s1.lo
      // SEQ 10: the same with the result in another pair.  `which` tells the two wrong answers apart.
      float x0, x1;
      if (SEQ == 9)
        __asm__ volatile(
            "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %3\n\tv_mov_b32 v115, %4\n\ts_nop 4\n\t"
            "v_pk_mul_f32 v[112:113], v[112:113], v[114:115] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
            "s_nop 4\n\tv_mov_b32 %0, v112\n\tv_mov_b32 %1, v113"
            : "=v"(x0), "=v"(x1) : "v"(a0), "v"(b0), "v"(b1), "v"(c) : "v112", "v113", "v114", "v115");
      else
        __asm__ volatile(
            "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %3\n\tv_mov_b32 v115, %4\n\ts_nop 4\n\t"
            "v_pk_mul_f32 v[116:117], v[112:113], v[114:115] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
            "s_nop 4\n\tv_mov_b32 %0, v116\n\tv_mov_b32 %1, v117"
            : "=v"(x0), "=v"(x1) : "v"(a0), "v"(b0), "v"(b1), "v"(c) : "v112", "v113", "v114", "v115", "v116", "v117");
      r0 = x0; r1 = x1;
      e0 = a0 * b1; e1 = a0 * b0;
      if (r1 != e1 && r1 == (a0 * b1) * b0) atomicAdd(errors + 1, 1u);      // the high half saw the low RESULT instead of the low source
      if ((r0 != e0 || r1 != e1) && SEQ == 9 && atomicAdd(errors + 2, 1u) < 6u) {       // a few samples: inputs and what came out
        const unsigned k = atomicAdd(errors + 3, 1u);
        if (k < 6u) { float *smp = reinterpret_cast<float *>(errors + 4 + 8 * k); smp[0] = a0; smp[1] = b0; smp[2] = b1; smp[3] = c; smp[4] = r0; smp[5] = r1; smp[6] = (float)lane; smp[7] = (float)it; }
      }
    } else if (SEQ == 11 || SEQ == 12) {
      // WRITE AFTER READ: the instruction above (source 1 read with its halves swapped), and right behind it an instruction that
      // OVERWRITES source 1 -- in MfccKernel the next packed multiply does (v_pk_mul_f32 v[14:15], v[14:15], v[26:27]).  SEQ 12: the
      // operands commuted (the swapped pair as source 0), which the bisect found clean.
      float x0, x1;
      if (SEQ == 11)
        __asm__ volatile(
            "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %3\n\tv_mov_b32 v115, %4\n\tv_mov_b32 v126, %5\n\tv_mov_b32 v127, %5\n\ts_nop 4\n\t"
            "v_pk_mul_f32 v[112:113], v[112:113], v[114:115] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
            "v_pk_mul_f32 v[114:115], v[114:115], v[126:127] op_sel_hi:[1,0]\n\t"
            "s_nop 4\n\tv_mov_b32 %0, v112\n\tv_mov_b32 %1, v113"
            : "=v"(x0), "=v"(x1) : "v"(a0), "v"(b0), "v"(b1), "v"(c) : "v112", "v113", "v114", "v115", "v126", "v127");
      else
        __asm__ volatile(
            "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %3\n\tv_mov_b32 v115, %4\n\tv_mov_b32 v126, %5\n\tv_mov_b32 v127, %5\n\ts_nop 4\n\t"
            "v_pk_mul_f32 v[112:113], v[114:115], v[112:113] op_sel:[1,0] op_sel_hi:[0,0]\n\t"
            "v_pk_mul_f32 v[114:115], v[114:115], v[126:127] op_sel_hi:[1,0]\n\t"
            "s_nop 4\n\tv_mov_b32 %0, v112\n\tv_mov_b32 %1, v113"
            : "=v"(x0), "=v"(x1) : "v"(a0), "v"(b0), "v"(b1), "v"(c) : "v112", "v113", "v114", "v115", "v126", "v127");
      r0 = x0; r1 = x1;
      e0 = a0 * b1; e1 = a0 * b0;
      if (r1 != e1 && r1 == a0 * (b0 * c)) atomicAdd(errors + 1, 1u);      // the high half saw source 1 AFTER the next instruction overwrote it
    } else {
      // LDS data consumed by a packed operation right behind the wait; a 32-bit v_mov into one half in between (both shapes occur)
      __asm__ volatile(
          "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\ts_nop 4\n\t"
          "ds_write2_b32 %7, v100, v101 offset1:1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "ds_read2_b32 v[104:105], %7 offset1:1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mov_b32 v103, v106\n\t"
          "v_pk_add_f32 v[100:101], v[104:105], v[102:103] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v100\n\tv_mov_b32 %1, v101"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c), "v"(lds) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "memory");
      e0 = a0 - b0; e1 = a1 - c;
    }
    bad += (unsigned)(r0 != e0) + (unsigned)(r1 != e1);
  }
  if (bad) atomicAdd(errors, bad);
}

__global__ __launch_bounds__(256, 2) void burner(float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  u32x4 ua, ub;
  for (int k = 0; k < 4; k++) { h = h * 1664525u + 1013904223u; ua[k] = (h & 0x807f807fu) | 0x3f003f00u; h = h * 1664525u + 1013904223u; ub[k] = (h & 0x807f807fu) | 0x3f003f00u; }
  const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  float s = 0;
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the matrix-core instructions the product's kernels use, one burner each: which one does the victim need beside it?
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256, 2) void burner2(float *out, int iters) {
  float s = 0;
  if (KIND == 0) {            // v_mfma_f32_32x32x16_f16 (layer GEMMs)
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)(0.5f + threadIdx.x * 0.001f); b[e] = (_Float16)(0.25f + e * 0.01f); }
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  } else if (KIND == 1) {     // v_mfma_f32_16x16x4_f32 (UBM posteriors, exact-FP32 GEMMs)
    f32x4 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
    const float a = 0.5f + threadIdx.x * 0.001f, b = 0.25f;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
  } else {                    // v_mfma_f64_16x16x4_f64 (iVector products)
    f64x4 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.0;
    const double a = 0.5 + threadIdx.x * 0.001, b = 0.25;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; i++) for (int r = 0; r < 4; r++) s += (float)acc[i][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ... and the other unusual instructions of the product's kernels
template <int KIND>
__global__ __launch_bounds__(256, 2) void burner3(float *out, const unsigned char *src, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned a = threadIdx.x * 2654435761u, b = blockIdx.x * 40503u + 7u;
  if (KIND == 0) {            // v_permlane32_swap (the layer GEMMs' epilogue)
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0] + 1u; b = r[1] ^ a; }
    }
  } else if (KIND == 1) {     // DPP row rotations + v_readlane (the search's reductions)
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        a += (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)a, 0x121, 0xF, 0xF, false);
        a ^= (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)a, 0x124, 0xF, 0xF, false);
        b += (unsigned)__builtin_amdgcn_readlane((int)a, 16);
      }
    }
  } else if (KIND == 2) {     // ds_bpermute + 64-bit LDS atomics
    unsigned long long *tab = reinterpret_cast<unsigned long long *>(smem3);
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = ~0ull;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        a = (unsigned)__builtin_amdgcn_ds_bpermute((int)(((lane + 1 + i) & 63) << 2), (int)a) + 1u;
        atomicMin(&tab[(a >> 7) & 1023], ((unsigned long long)a << 32) | b);
      }
    }
    b += (unsigned)tab[lane];
  } else {                    // global -> LDS DMA (global_load_lds_dwordx4) + ds_read_b128 (the layer GEMMs' loop)
    const unsigned lds0 = (unsigned)(uintptr_t)smem3;
    const unsigned char *g = src + ((size_t)blockIdx.x * 4 + wave) * 4096 + lane * 16;
    for (int it = 0; it < iters; it++) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wave * 4096));
#pragma unroll
      for (int p = 0; p < 4; p++)
        __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst + p * 1024), "v"(g + (size_t)((it * 4 + p) & 255) * 1024 * 1024 / 256) : "memory");
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      u32x4 f;
      __asm__ volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(lds0 + (unsigned)(wave * 4096 + lane * 16)));
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      a += f[0];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)(a ^ b);
}

template <int KIND>
int run_beside3(const char *name, unsigned *d_err, float *d_out, const unsigned char *src, hipStream_t sv, hipStream_t sb) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    CHECK(hipStreamSynchronize(sv));
    for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner3<KIND>, dim3(512), dim3(256), 16384, sb, d_out, src, KIND == 3 ? 30000 : 200000);
    hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, 100000);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
    total += e[0];
  }
  printf("v_pk_mul with source 1 half-swapped beside a burner of %-36s %u wrong of 104.9 G values\n", name, total);
  return 0;
}

// ... where the wave's registers sit in the SIMD's 512-entry file.  GemmKernelB3 holds 240 VGPRs per lane: a victim wave that shares a SIMD
// with one starts high up in the file.  victim_high: the instruction of SEQ 9 in a kernel that itself declares 256 registers (v255
// clobbered), so that of two waves on a SIMD one starts at entry 256 -- no other kernel on the device.
template <bool SWAP_SRC1>
__global__ __launch_bounds__(256, 2) void victim_high(unsigned *errors, unsigned *wave_bad, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned bad = 0;
  for (int it = 0; it < rounds; it++) {
    const float a0 = (float)((it * 7 + lane) & 255), b0 = (float)((it * 5 + 2 * lane) & 255) + 1000.f, b1 = (float)((it * 11 + lane) & 127) + 2000.f, c = 77.f;
    float x0, x1;
    if (SWAP_SRC1)
      __asm__ volatile(
          "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %3\n\tv_mov_b32 v115, %4\n\ts_nop 4\n\t"
          "v_pk_mul_f32 v[112:113], v[112:113], v[114:115] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v112\n\tv_mov_b32 %1, v113"
          : "=v"(x0), "=v"(x1) : "v"(a0), "v"(b0), "v"(b1), "v"(c) : "v112", "v113", "v114", "v115", "v255");
    else
      __asm__ volatile(
          "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %5\n\tv_mov_b32 v114, %3\n\tv_mov_b32 v115, %4\n\ts_nop 4\n\t"
          "v_pk_mul_f32 v[112:113], v[114:115], v[112:113] op_sel:[1,0] op_sel_hi:[0,0]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v112\n\tv_mov_b32 %1, v113"
          : "=v"(x0), "=v"(x1) : "v"(a0), "v"(b0), "v"(b1), "v"(c) : "v112", "v113", "v114", "v115", "v255");
    bad += (unsigned)(x0 != a0 * b1) + (unsigned)(x1 != a0 * b0);
  }
  if (bad) { atomicAdd(errors, bad); if (lane == 0) atomicAdd(wave_bad, 1u); }
}

// ... and no instruction at all: waves being LAUNCHED and retired on the victim's CUs (many short kernels, as a decode call issues them)
template <int REGS>
__global__ __launch_bounds__(256) void tiny(float *out, int n) {
  float acc[REGS];
#pragma unroll
  for (int i = 0; i < REGS; i++) acc[i] = (float)(threadIdx.x + i);
  for (int k = 0; k < n; k++)
#pragma unroll
    for (int i = 0; i < REGS; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
  float s = 0;
#pragma unroll
  for (int i = 0; i < REGS; i++) s += acc[i];
  if (s == 12345.f) out[0] = s;
}
template <int REGS>
int run_beside_launches(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb, int blocks, int work) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    CHECK(hipStreamSynchronize(sv));
    hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, 100000);
    for (int l = 0; l < 3000; l++) hipLaunchKernelGGL(tiny<REGS>, dim3(blocks), dim3(256), 0, sb, d_out, work);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
    total += e[0];
  }
  printf("v_pk_mul with source 1 half-swapped beside %-48s %u wrong of 104.9 G values\n", name, total);
  return 0;
}

// ... the conversions of the FP32 -> two-fp16-parts split (GemmKernelB3 is the kernel the victim needs beside it: pk_perturber.sh)
template <int KIND>
__global__ __launch_bounds__(256, 2) void burner4(float *out, int iters) {
  float x = 1.0f + threadIdx.x * 0.001f, y = 0.5f + blockIdx.x * 0.0001f;
  unsigned p = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (KIND == 0) {          // v_cvt_pk_f16_f32 (VOP3): two floats -> packed halves
        __asm__ volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y));
        x += __uint_as_float((p & 0x7fu) | 0x3a000000u);
      } else if (KIND == 1) {   // v_cvt_f32_f16_sdwa: the HIGH half-word of a register -> float (sub-dword source select)
        __asm__ volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(y) : "v"(p));
        p = p * 1664525u + 1013904223u;
        p = (p & 0x03ff03ffu) | 0x3c003c00u;
        x += y * 1e-6f;
      } else {                  // both, as the split does: hi = cvt(x); lo = cvt(x - float(hi)); pack
        unsigned h;
        float hf;
        __asm__ volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x), "v"(y));
        __asm__ volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(hf) : "v"(h));
        x = x * 1.0001f + (y - hf) * 1e-3f;
        p ^= h;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x + y + (float)(p & 0xff);
}
// ... and the split and the MFMAs in ONE wave, as GemmKernelB3's k loop has them: convert, write the parts to LDS, read fragments back,
// multiply.  KIND 0: everything; 1: no LDS round trip (the converted words feed the MFMAs directly); 2: MFMAs + the sub-dword convert only;
// 3: MFMAs + v_cvt_pk only
template <int KIND>
__global__ __launch_bounds__(256, 2) void burner5(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned stage5[256 * 4];
  f32x16 acc[2];
  for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float x[8];
  for (int e = 0; e < 8; e++) x[e] = 0.5f + threadIdx.x * 0.001f + e * 0.01f;
  u32x4 fa = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, fb = fa;
  for (int it = 0; it < iters; it++) {
    unsigned h[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (KIND != 2) __asm__ volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[e]) : "v"(x[2 * e]), "v"(x[2 * e + 1]));
      else h[e] = __float_as_uint(x[2 * e]) & 0x7fff7fffu;
      float h0 = 0.f, h1 = 0.f;
      if (KIND != 3) {
        __asm__ volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(h0) : "v"(h[e]));
        __asm__ volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(h1) : "v"(h[e]));
      }
      const float r0 = x[2 * e] - h0, r1 = x[2 * e + 1] - h1;
      if (KIND != 2) __asm__ volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo[e]) : "v"(r0), "v"(r1));
      else lo[e] = __float_as_uint(r0);
      x[2 * e] = x[2 * e] * 1.0001f + 1e-4f; x[2 * e + 1] = x[2 * e + 1] * 0.9999f + 1e-4f;
    }
    if (KIND == 0) {
      *reinterpret_cast<u32x4 *>(&stage5[threadIdx.x * 4]) = u32x4{h[0], h[1], h[2], h[3]};
      __builtin_amdgcn_wave_barrier();
      fa = *reinterpret_cast<u32x4 *>(&stage5[(threadIdx.x ^ 1) * 4]);
      fb = u32x4{lo[0], lo[1], lo[2], lo[3]};
    } else {
      fa = u32x4{h[0], h[1], h[2], h[3]};
      fb = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
    const f16x8 a = __builtin_bit_cast(f16x8, fa), b = __builtin_bit_cast(f16x8, fb);
#pragma unroll
    for (int i = 0; i < 2; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
int run_beside5(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    CHECK(hipStreamSynchronize(sv));
    for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner5<KIND>, dim3(512), dim3(256), 0, sb, d_out, 60000);
    hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, 100000);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
    total += e[0];
  }
  printf("v_pk_mul with source 1 half-swapped beside a burner of %-52s %u wrong of 104.9 G values\n", name, total);
  return 0;
}

// ... the VOP3 forms with source modifiers GemmKernelB3 carries (its running maximum of |x| for the range check)
template <int KIND>
__global__ __launch_bounds__(256, 2) void burner6(float *out, int iters) {
  float x = 1.0f + threadIdx.x * 0.001f, y = -0.5f - blockIdx.x * 0.0001f, m = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float t;
      if (KIND == 0) __asm__ volatile("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(t) : "v"(x), "v"(y));
      else __asm__ volatile("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(t) : "v"(x), "v"(y), "v"(m));
      m = t;
      x = -x * 1.0001f; y = y * 0.9999f + 1e-5f;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = m + x + y;
}
template <int KIND>
int run_beside6(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    CHECK(hipStreamSynchronize(sv));
    for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner6<KIND>, dim3(512), dim3(256), 0, sb, d_out, 200000);
    hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, 100000);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
    total += e[0];
  }
  printf("v_pk_mul with source 1 half-swapped beside a burner of %-44s %u wrong of 104.9 G values\n", name, total);
  return 0;
}

// ... the 64-bit integer VALU operations of compiler-generated address arithmetic (GemmKernelB3: 3011 v_lshl_add_u64, 1097 v_mov_b64,
// 886 v_lshlrev_b64 in its listing): they run on the same double-width datapath as the packed FP32 operations
template <int KIND>
__global__ __launch_bounds__(256, 2) void burner7(float *out, int iters) {
  unsigned long long p = 0x100000000ull + threadIdx.x * 8ull, q = blockIdx.x * 4096ull + 3ull, r = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (KIND == 0) __asm__ volatile("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(r) : "v"(p), "v"(q));
      else if (KIND == 1) __asm__ volatile("v_mov_b64_e32 %0, %1" : "=v"(r) : "v"(p));
      else if (KIND == 2) __asm__ volatile("v_lshlrev_b64 %0, 2, %1" : "=v"(r) : "v"(p));
      else {
        unsigned long long t;
        __asm__ volatile("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(t) : "v"(p), "v"(q));
        __asm__ volatile("v_mov_b64_e32 %0, %1" : "=v"(r) : "v"(t));
        __asm__ volatile("v_lshlrev_b64 %0, 2, %1" : "=v"(t) : "v"(r));
        r ^= t;
      }
      p = (r & 0xFFFFFFFFFFull) + 8ull; q += 1ull;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)(unsigned)(p ^ q ^ r);
}
template <int KIND>
int run_beside7(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    CHECK(hipStreamSynchronize(sv));
    for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner7<KIND>, dim3(512), dim3(256), 0, sb, d_out, 150000);
    hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, 100000);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
    total += e[0];
  }
  printf("v_pk_mul with source 1 half-swapped beside a burner of %-44s %u wrong of 104.9 G values\n", name, total);
  return 0;
}

// ... f16 MFMAs again (GemmKernelB3 without its MFMAs leaves the victim alone: pk_perturber2.sh), with the operand VALUES the split
// produces: KIND 0 normal numbers, 1 fp16 SUBNORMALS in one operand (the low parts), 2 subnormals in both, 3 zeros, 4 operands and
// accumulators that change every instruction (fresh registers, as fragments read from LDS are)
template <int KIND>
__global__ __launch_bounds__(256, 2) void burner8(float *out, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  u32x4 ua, ub;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int k = 0; k < 4; k++) {
    h = h * 1664525u + 1013904223u;
    const unsigned sub = (h & 0x03ff03ffu) | 0x00010001u, nor = (h & 0x03ff03ffu) | 0x3c003c00u;
    ua[k] = KIND == 3 ? 0u : (KIND == 1 || KIND == 2) ? sub : nor;
    h = h * 1664525u + 1013904223u;
    ub[k] = KIND == 3 ? 0u : KIND == 2 ? ((h & 0x03ff03ffu) | 0x00010001u) : ((h & 0x03ff03ffu) | 0x3c003c00u);
  }
  for (int it = 0; it < iters; it++) {
    f16x8 a = __builtin_bit_cast(f16x8, ua), b = __builtin_bit_cast(f16x8, ub);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
      if (KIND == 5) { h = h * 1664525u + 1013904223u; h = (h & 0x03ff03ffu) | 0x3c003c00u; }      // the same VALU work, feeding nothing
      if (KIND == 6) { a = __builtin_bit_cast(f16x8, (i & 1) ? ub : ua); }                            // two operand sets alternating, no VALU write
      if (KIND == 4) { ua[i & 3] = ua[i & 3] * 1664525u + 1013904223u; ua[i & 3] = (ua[i & 3] & 0x03ff03ffu) | ((i & 1) ? 0x00010001u : 0x3c003c00u); a = __builtin_bit_cast(f16x8, ua); }
    }
  }
  float s = (float)(h & 0xff);
  for (int i = 0; i < 8; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
int run_beside8(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    CHECK(hipStreamSynchronize(sv));
    for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner8<KIND>, dim3(512), dim3(256), 0, sb, d_out, 50000);
    hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, 100000);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
    total += e[0];
  }
  printf("v_pk_mul with source 1 half-swapped beside f16 MFMAs on %-44s %u wrong of 104.9 G values\n", name, total);
  return 0;
}

// ... and the other way round: which OTHER instructions with sub-register operand selection does the perturber of burner8<4> (VALU results
// on their way into MFMA sources) disturb?  The product's kernels use the sub-dword convert (the fp16 split) and DPP rotations (every
// wave reduction); the packed forms are the compiler's.
template <int FORM>
__global__ __launch_bounds__(256, 2) void victim_forms2(unsigned *errors, int rounds) {
  const int lane = threadIdx.x & 63;
  unsigned bad = 0;
  for (int it = 0; it < rounds; it++) {
    const float a0 = (float)((it * 7 + lane) & 255), a1 = (float)((it * 13 + 3 * lane) & 255) + 300.f;
    const float b0 = (float)((it * 5 + 2 * lane) & 255) + 1000.f, b1 = (float)((it * 11 + lane) & 127) + 2000.f;
    float r0 = 0.f, r1 = 0.f, e0 = 0.f, e1 = 0.f;
    if (FORM == 0) {            // v_cvt_f32_f16_sdwa src0_sel:WORD_1 (GemmKernelB3's split)
      const unsigned packed = ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)a1) << 16) | (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a0);
      __asm__ volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r0) : "v"(packed));
      __asm__ volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(r1) : "v"(packed));
      e0 = a1; e1 = a0;
    } else if (FORM == 1) {     // DPP row rotation as a source (the wave reductions of the search, the CG solve, ...)
      __asm__ volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r0) : "v"(a0), "v"(b0));
      const float nb = (float)((it * 7 + ((lane & 48) | ((lane - 1) & 15))) & 255);      // a0 of the lane one to the right inside the row of 16
      e0 = nb + b0; r1 = e1 = 0.f;
    } else if (FORM == 2) {     // v_pk_add_f32, source 1 half-swapped
      __asm__ volatile(
          "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %3\n\tv_mov_b32 v114, %4\n\tv_mov_b32 v115, %5\n\ts_nop 4\n\t"
          "v_pk_add_f32 v[116:117], v[112:113], v[114:115] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v116\n\tv_mov_b32 %1, v117"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v112", "v113", "v114", "v115", "v116", "v117");
      e0 = a0 + b1; e1 = a1 + b0;
    } else if (FORM == 3) {     // v_pk_fma_f32, source 2 half-swapped
      __asm__ volatile(
          "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %3\n\tv_mov_b32 v114, %4\n\tv_mov_b32 v115, %5\n\tv_mov_b32 v118, 1.0\n\tv_mov_b32 v119, 1.0\n\ts_nop 4\n\t"
          "v_pk_fma_f32 v[116:117], v[112:113], v[118:119], v[114:115] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v116\n\tv_mov_b32 %1, v117"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
      e0 = a0 + b1; e1 = a1 + b0;
    } else {                    // v_pk_mul_f32, source 1 NOT swapped but broadcast from its high word (op_sel:[0,1] op_sel_hi:[1,1])
      __asm__ volatile(
          "v_mov_b32 v112, %2\n\tv_mov_b32 v113, %3\n\tv_mov_b32 v114, %4\n\tv_mov_b32 v115, %5\n\ts_nop 4\n\t"
          "v_pk_mul_f32 v[116:117], v[112:113], v[114:115] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
          "s_nop 4\n\tv_mov_b32 %0, v116\n\tv_mov_b32 %1, v117"
          : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v112", "v113", "v114", "v115", "v116", "v117");
      e0 = a0 * b1; e1 = a1 * b1;
    }
    bad += (unsigned)(r0 != e0) + (unsigned)(r1 != e1);
  }
  if (bad) atomicAdd(errors, bad);
}
template <int FORM>
int run_forms2(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb) {
  for (int with_burner = 0; with_burner < 2; with_burner++) {
    unsigned total = 0;
    for (int rep = 0; rep < 4; rep++) {
      CHECK(hipMemsetAsync(d_err, 0, 256, sv));
      CHECK(hipStreamSynchronize(sv));
      if (with_burner) for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner8<4>, dim3(512), dim3(256), 0, sb, d_out, 50000);
      hipLaunchKernelGGL(victim_forms2<FORM>, dim3(512), dim3(256), 0, sv, d_err, 100000);
      unsigned e = 0;
      CHECK(hipMemcpyAsync(&e, d_err, 4, hipMemcpyDeviceToHost, sv));
      CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
      total += e;
    }
    printf("%-72s %-44s %u wrong\n", name, with_burner ? "beside MFMAs with VALU-written sources:" : "alone:", total);
  }
  return 0;
}

template <int KIND>
int run_beside4(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    CHECK(hipStreamSynchronize(sv));
    for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner4<KIND>, dim3(512), dim3(256), 0, sb, d_out, 200000);
    hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, 100000);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
    total += e[0];
  }
  printf("v_pk_mul with source 1 half-swapped beside a burner of %-44s %u wrong of 104.9 G values\n", name, total);
  return 0;
}

template <int KIND>
int run_beside(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    CHECK(hipStreamSynchronize(sv));
    for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner2<KIND>, dim3(512), dim3(256), 0, sb, d_out, KIND == 2 ? 60000 : 100000);
    hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, 100000);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
    total += e[0];
  }
  printf("v_pk_mul with source 1 half-swapped beside a burner of %-28s %u wrong of 104.9 G values\n", name, total);
  return 0;
}

template <int SEQ>
int run(const char *name, unsigned *d_err, float *d_out, hipStream_t sv, hipStream_t sb) {
  const int rounds = 100000;
  for (int with_burner = 0; with_burner < 2; with_burner++) {
    unsigned total = 0, total_kind = 0;
    for (int rep = 0; rep < 4; rep++) {
      CHECK(hipMemsetAsync(d_err, 0, 256, sv));
      CHECK(hipStreamSynchronize(sv));
      if (with_burner) for (int l = 0; l < 4; l++) hipLaunchKernelGGL(burner, dim3(512), dim3(256), 0, sb, d_out, 100000);
      hipLaunchKernelGGL(victim<SEQ>, dim3(512), dim3(256), 0, sv, d_err, rounds);
      unsigned e[2] = {0, 0};
      CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
      CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
      total += e[0]; total_kind += e[1];
    }
    printf("%-96s %-24s %u wrong of %.1f G values", name, with_burner ? "MFMA burner on every CU:" : "alone:", total, 4.0 * 512 * 256 * 2 * rounds / 1e9);
    if (total_kind) printf("  (%u of them: the one predicted wrong value)", total_kind);
    printf("\n");
  }
  return 0;
}


// Entry point for pk_victim_beside_decode.py: one victim sequence on its own stream while the caller's other threads keep decode
// calls in flight on the same device (the real pipelines instead of the MFMA burner).
extern "C" int pk_victim_run(int seq, int rounds, int launches, unsigned *wrong, unsigned *predicted) {
  static unsigned *d_err = nullptr;
  static hipStream_t sv = nullptr;
  if (!d_err) { if (hipMalloc(&d_err, 256) != hipSuccess) return 1; if (hipStreamCreateWithFlags(&sv, hipStreamNonBlocking) != hipSuccess) return 1; }
  if (hipMemsetAsync(d_err, 0, 256, sv) != hipSuccess) return 1;
  for (int l = 0; l < launches; l++) {
    switch (seq) {
      case 9: hipLaunchKernelGGL(victim<9>, dim3(512), dim3(256), 0, sv, d_err, rounds); break;
      case 10: hipLaunchKernelGGL(victim<10>, dim3(512), dim3(256), 0, sv, d_err, rounds); break;
      case 11: hipLaunchKernelGGL(victim<11>, dim3(512), dim3(256), 0, sv, d_err, rounds); break;
      case 12: hipLaunchKernelGGL(victim<12>, dim3(512), dim3(256), 0, sv, d_err, rounds); break;
      case 2: hipLaunchKernelGGL(victim<2>, dim3(512), dim3(256), 0, sv, d_err, rounds); break;
      default: return 2;
    }
  }
  unsigned e[2] = {0, 0};
  if (hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv) != hipSuccess) return 1;
  if (hipStreamSynchronize(sv) != hipSuccess) return 1;
  *wrong = e[0]; *predicted = e[1];
  if (seq == 9 && e[0]) {
    float smp[48];
    unsigned n = 0;
    if (hipMemcpy(&n, d_err + 3, 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(smp, d_err + 4, sizeof(smp), hipMemcpyDeviceToHost) == hipSuccess)
      for (unsigned k = 0; k < n && k < 6; k++)
        printf("    sample: s0.lo %g  s1 = {%g, %g}  (s0.hi %g)  ->  got {%g, %g}, want {%g, %g}   lane %g round %g\n", smp[8 * k], smp[8 * k + 1], smp[8 * k + 2], smp[8 * k + 3],
               smp[8 * k + 4], smp[8 * k + 5], smp[8 * k] * smp[8 * k + 2], smp[8 * k] * smp[8 * k + 1], smp[8 * k + 6], smp[8 * k + 7]);
  }
  return 0;
}

#ifndef PK_LIBRARY
int main() {
  unsigned *d_err; float *d_out;
  CHECK(hipMalloc(&d_err, 256)); CHECK(hipMalloc(&d_out, 4096 * 256 * 4));
  hipStream_t sv, sb;
  CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  run<0>("v_mov into the high halves of both pairs, v_pk_add (neg) right behind", d_err, d_out, sv, sb);
  run<1>("v_mov into one high half, v_pk_add right behind", d_err, d_out, sv, sb);
  run<2>("v_pk_add, s_nop 0, 32-bit add / sub of its halves", d_err, d_out, sv, sb);
  run<3>("v_pk_add, 32-bit add / sub of its halves with NO wait state", d_err, d_out, sv, sb);
  run<4>("v_pk_add, ds_write2_b32 of the result right behind, read back", d_err, d_out, sv, sb);
  run<5>("ds_read2_b32, wait, v_mov into a half, v_pk_add right behind", d_err, d_out, sv, sb);
  run<6>("v_pk_add writes a pair, v_pk_add, s_nop 0, v_mov OVERWRITES its high half (write after write)", d_err, d_out, sv, sb);
  run<7>("the same with s_nop 4 in front of the v_mov", d_err, d_out, sv, sb);
  run<8>("v_pk_mul op_sel forms that ignore a half, NaN / inf / denormal / huge in the ignored half", d_err, d_out, sv, sb);
  run<9>("v_pk_mul IN PLACE, high result from the LOW word of source 0 (op_sel:[0,1] op_sel_hi:[0,0])", d_err, d_out, sv, sb);
  run<10>("the same instruction with its result in another pair", d_err, d_out, sv, sb);
  run<11>("that instruction, and the next one OVERWRITES its half-swapped source 1 (write after read)", d_err, d_out, sv, sb);
  run<12>("the same with the operands commuted (half-swapped pair as source 0)", d_err, d_out, sv, sb);
  run_beside<0>("v_mfma_f32_32x32x16_f16:", d_err, d_out, sv, sb);
  run_beside<1>("v_mfma_f32_16x16x4_f32:", d_err, d_out, sv, sb);
  run_beside<2>("v_mfma_f64_16x16x4_f64:", d_err, d_out, sv, sb);
  unsigned char *d_src;
  CHECK(hipMalloc(&d_src, (size_t)512 * 4 * 4096 + (2u << 20)));
  CHECK(hipMemset(d_src, 1, (size_t)512 * 4 * 4096 + (2u << 20)));
  run_beside_launches<4>("3000 launches of 2048 short workgroups (few registers):", d_err, d_out, sv, sb, 2048, 20);
  run_beside_launches<64>("3000 launches of 2048 short workgroups (64+ registers):", d_err, d_out, sv, sb, 2048, 4);
  run_beside_launches<4>("3000 launches of 256 workgroups:", d_err, d_out, sv, sb, 256, 200);
  run_beside8<0>("normal numbers:", d_err, d_out, sv, sb);
  run_beside8<1>("fp16 subnormals in one operand:", d_err, d_out, sv, sb);
  run_beside8<2>("fp16 subnormals in both operands:", d_err, d_out, sv, sb);
  run_beside8<3>("zeros:", d_err, d_out, sv, sb);
  run_beside8<4>("operands that change every instruction:", d_err, d_out, sv, sb);
  run_beside8<5>("constant operands, the same VALU work feeding nothing:", d_err, d_out, sv, sb);
  run_beside8<6>("two constant operand sets alternating:", d_err, d_out, sv, sb);
  run_forms2<0>("v_cvt_f32_f16_sdwa src0_sel:WORD_1 (the split's sub-dword convert)", d_err, d_out, sv, sb);
  run_forms2<1>("v_add_f32_dpp row_ror:1 (the wave reductions)", d_err, d_out, sv, sb);
  run_forms2<2>("v_pk_add_f32, source 1 half-swapped", d_err, d_out, sv, sb);
  run_forms2<3>("v_pk_fma_f32, source 2 half-swapped", d_err, d_out, sv, sb);
  run_forms2<4>("v_pk_mul_f32, source 1 broadcast from its high word", d_err, d_out, sv, sb);
  run_beside7<0>("v_lshl_add_u64:", d_err, d_out, sv, sb);
  run_beside7<1>("v_mov_b64:", d_err, d_out, sv, sb);
  run_beside7<2>("v_lshlrev_b64:", d_err, d_out, sv, sb);
  run_beside7<3>("all three 64-bit integer operations:", d_err, d_out, sv, sb);
  run_beside6<0>("v_max_f32_e64 v, |v|, |v|:", d_err, d_out, sv, sb);
  run_beside6<1>("v_max3_f32 v, |v|, |v|, v:", d_err, d_out, sv, sb);
  run_beside4<0>("v_cvt_pk_f16_f32:", d_err, d_out, sv, sb);
  run_beside4<1>("v_cvt_f32_f16_sdwa src0_sel:WORD_1:", d_err, d_out, sv, sb);
  run_beside4<2>("both (the FP32 -> two fp16 parts split):", d_err, d_out, sv, sb);
  for (int swap = 1; swap >= 0; swap--) {
    CHECK(hipMemsetAsync(d_err, 0, 256, sv));
    if (swap) hipLaunchKernelGGL(victim_high<true>, dim3(2048), dim3(256), 0, sv, d_err, d_err + 1, 20000);
    else hipLaunchKernelGGL(victim_high<false>, dim3(2048), dim3(256), 0, sv, d_err, d_err + 1, 20000);
    unsigned e[2] = {0, 0};
    CHECK(hipMemcpyAsync(e, d_err, 8, hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv));
    printf("the instruction in a 256-register kernel, two waves per SIMD, ALONE on the device, %s: %u wrong of %.1f G values, in %u of 8192 waves\n",
           swap ? "source 1 half-swapped" : "operands commuted", e[0], 2048.0 * 256 * 2 * 20000 / 1e9, e[1]);
  }
  run_beside5<0>("split + LDS round trip + f16 MFMAs in one wave:", d_err, d_out, sv, sb);
  run_beside5<1>("split + f16 MFMAs in one wave (no LDS):", d_err, d_out, sv, sb);
  run_beside5<2>("sub-dword convert + f16 MFMAs in one wave:", d_err, d_out, sv, sb);
  run_beside5<3>("v_cvt_pk_f16_f32 + f16 MFMAs in one wave:", d_err, d_out, sv, sb);
  run_beside3<0>("v_permlane32_swap:", d_err, d_out, d_src, sv, sb);
  run_beside3<1>("DPP rotations + v_readlane:", d_err, d_out, d_src, sv, sb);
  run_beside3<2>("ds_bpermute + 64-bit LDS atomics:", d_err, d_out, d_src, sv, sb);
  run_beside3<3>("global_load_lds_dwordx4 + ds_read_b128:", d_err, d_out, d_src, sv, sb);
  return 0;
}

#endif
