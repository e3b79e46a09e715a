#include <hip/hip_runtime.h>
__global__ void k(const float* __restrict__ src, float* out) {
  __shared__ __attribute__((aligned(16))) float buf[64 * 4 * 2];
  const int lane = threadIdx.x;
  // each lane fetches 16 bytes; they land at buf + lane*16
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + lane * 4), (void __attribute__((address_space(3)))*)buf, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 256 + (63 - lane) * 4), (void __attribute__((address_space(3)))*)(buf + 256), 16, 0, 0);
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 8; i++) out[lane * 8 + i] = buf[lane * 8 + i];
}
int main() {
  float *s, *o; hipMalloc(&s, 512 * 4); hipMalloc(&o, 512 * 4);
  float h[512]; for (int i = 0; i < 512; i++) h[i] = i;
  hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o);
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; i++) if (h[i] != i) bad++;
  for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) if (h[256 + l * 4 + j] != 256 + (63 - l) * 4 + j) bad++;
  printf("bad=%d first=%g %g %g second=%g %g\n", bad, h[0], h[1], h[4], h[256], h[260]);
  return 0;
}
