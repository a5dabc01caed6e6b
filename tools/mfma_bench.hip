// Measured ceiling of v_mfma_f32_32x32x2_f32 on this chip: register-only MFMA loop, no memory.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_bench.hip -o tools/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float x) {
  v16f acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = x + threadIdx.x, b = x - threadIdx.x;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // shader clock from the two counters (wall clock = 100 MHz)
    out[0] = (float)(clock64() - c0);
    out[1] = (float)(wall_clock64() - w0);
  }
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu) {
  float* d; hipMalloc(&d, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, 100, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)grid * 4 * iters * 4 * NACC * 4096.0;
  float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("NACC=%d blocks/CU=%d: %.2f ms  %.1f TFLOP/s  shader clock %.0f MHz (%.0f cyc / %.0f ticks)\n", NACC, blocks_per_cu, ms,
         flop / ms * 1e-9, h[0] / h[1] * 100.0, h[0], h[1]);
}
int main() { run<4>(1); run<4>(2); run<2>(2); run<1>(4); run<4>(1); return 0; }
