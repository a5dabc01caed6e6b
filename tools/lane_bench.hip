// Does a wave64 VALU instruction issue faster when only some of its 16-lane quarters are active?
// build: hipcc --offload-arch=gfx950 -O3 tools/lane_bench.hip -o tools/lane_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DP>
__global__ __launch_bounds__(64) void k(float* out, int iters, int active, float seed) {
  if ((int)threadIdx.x >= active) return;
  const long long c0 = clock64();
  if (DP) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
      a0 = fma(a0, 0.999, 0.001); a1 = fma(a1, 0.999, 0.001); a2 = fma(a2, 0.999, 0.001); a3 = fma(a3, 0.999, 0.001);
      a4 = fma(a4, 0.999, 0.001); a5 = fma(a5, 0.999, 0.001); a6 = fma(a6, 0.999, 0.001); a7 = fma(a7, 0.999, 0.001);
    }
    out[blockIdx.x * 64 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
  } else {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
      a0 = fmaf(a0, 0.999f, 0.001f); a1 = fmaf(a1, 0.999f, 0.001f); a2 = fmaf(a2, 0.999f, 0.001f); a3 = fmaf(a3, 0.999f, 0.001f);
      a4 = fmaf(a4, 0.999f, 0.001f); a5 = fmaf(a5, 0.999f, 0.001f); a6 = fmaf(a6, 0.999f, 0.001f); a7 = fmaf(a7, 0.999f, 0.001f);
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[4096] = (float)(clock64() - c0);
}
int main() {
  float* d; hipMalloc(&d, 8192 * 4);
  const int iters = 100000;
  for (int dp = 0; dp < 2; ++dp)
    for (int active : {64, 48, 32, 16, 1}) {
      if (dp) hipLaunchKernelGGL(k<1>, dim3(64), dim3(64), 0, 0, d, iters, active, 1.f);
      else hipLaunchKernelGGL(k<0>, dim3(64), dim3(64), 0, 0, d, iters, active, 1.f);
      hipDeviceSynchronize();
      float cyc; hipMemcpy(&cyc, d + 4096, 4, hipMemcpyDeviceToHost);
      printf("%s active lanes %2d: %.2f cycles per FMA instruction (one wave per SIMD, 8 independent chains)\n", dp ? "f64" : "f32", active,
             cyc / (8.0 * iters));
    }
  return 0;
}
