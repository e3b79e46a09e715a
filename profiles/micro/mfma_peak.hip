#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char *name, F launch, double flop_per_block_iter, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(blocks, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0); launch(blocks, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s blocks=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop_per_block_iter * blocks * iters / (ms * 1e-3) / 1e12);
}
int main() {
  float *out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  for (int blocks : {256, 512, 768, 1024}) {
    run("32x32x2 x4acc ", [&](int b, int it) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(256), 0, 0, out, it); }, 4.0 * 4 * 4096, blocks, iters);
    run("16x16x4 x16acc", [&](int b, int it) { hipLaunchKernelGGL(k16<16>, dim3(b), dim3(256), 0, 0, out, it); }, 4.0 * 16 * 2048, blocks, iters);
  }
  return 0;
}
