// Debug aid: a shared library whose only function keeps the matrix cores (or the vector ALUs) of every CU busy from its own
// stream, with small register / no LDS footprint, so that it co-resides with whatever else runs.  Used by
// profiles/micro/stress_burner.py to tell whether results of OTHER kernels depend on what they share a CU with.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void burn_bf16(float *out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  u32x4 ua, ub;
  for (int k = 0; k < 4; k++) { h = h * 1664525u + 1013904223u; ua[k] = (h & 0x807f807fu) | 0x3f003f00u; h = h * 1664525u + 1013904223u; ub[k] = (h & 0x807f807fu) | 0x3f003f00u; }
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  float s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void burn_f32(float *out, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  float s = 0; for (int i = 0; i < 16; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void burn_valu(float *out, int iters) {
  float x[32];
  for (int i = 0; i < 32; i++) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = x[i] * 1.0001f + 0.5f;
  float s = 0; for (int i = 0; i < 32; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int burn(int kind, int blocks, int iters, int launches) {
  static float *out = nullptr;
  static hipStream_t st = nullptr;
  if (!out) { if (hipMalloc(&out, 8192 * 256 * 4) != hipSuccess) return -1; if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -2; }
  for (int l = 0; l < launches; l++) {
    if (kind == 0) hipLaunchKernelGGL(burn_bf16<4>, dim3(blocks), dim3(256), 0, st, out, iters);
    else if (kind == 3) hipLaunchKernelGGL(burn_bf16<10>, dim3(blocks), dim3(256), 0, st, out, iters / 2);
    else if (kind == 4) hipLaunchKernelGGL(burn_bf16<14>, dim3(blocks), dim3(256), 0, st, out, iters / 3);
    else if (kind == 1) hipLaunchKernelGGL(burn_f32, dim3(blocks), dim3(256), 0, st, out, iters);
    else hipLaunchKernelGGL(burn_valu, dim3(blocks), dim3(256), 0, st, out, iters);
  }
  return hipStreamSynchronize(st) == hipSuccess ? 0 : -3;
}

// A victim with a known answer: every wave keeps a private 4 KiB LDS region and registers full of values it can predict,
// re-checks them for a while and counts what changed under it (returns the number of corrupted words seen).
__global__ __launch_bounds__(256) void victim(unsigned *errors, int rounds) {
  __shared__ unsigned lds[4][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned *mine = lds[wave];
  const unsigned seed = blockIdx.x * 4u + wave;
  for (int i = lane; i < 1024; i += 64) mine[i] = seed * 2654435761u + i;
  float r[16];
  for (int k = 0; k < 16; k++) r[k] = (float)(lane + k);
  unsigned bad = 0;
  for (int it = 0; it < rounds; it++) {
    for (int i = lane; i < 1024; i += 64) {
      const unsigned v = mine[i];
      if (v != seed * 2654435761u + i + it) bad++;
      mine[i] = v + 1;
    }
    for (int k = 0; k < 16; k++) r[k] = r[k] * 1.0f + 1.0f;
  }
  for (int k = 0; k < 16; k++) if (r[k] != (float)(lane + k) + (float)rounds) bad++;
  if (bad) atomicAdd(errors, bad);
}
extern "C" long run_victim(int blocks, int rounds, int launches) {
  static unsigned *err = nullptr;
  static hipStream_t st = nullptr;
  if (!err) { if (hipMalloc(&err, 4) != hipSuccess) return -1; if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -2; }
  (void)hipMemsetAsync(err, 0, 4, st);
  for (int l = 0; l < launches; l++) hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 0, st, err, rounds);
  unsigned h = 0;
  if (hipMemcpyAsync(&h, err, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return -3;
  if (hipStreamSynchronize(st) != hipSuccess) return -4;
  return (long)h;
}
