// how fast does the shader clock run while a stream of tiny kernels executes?  s_memtime (shader clock domain on gfx9)
// against s_memrealtime (100 MHz) around a dependent chain of float64 FMAs of known length
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void chain(double* out, unsigned long long* t, int n) {
  double x = out[threadIdx.x];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x = __builtin_fma(x, 1.0000001, 0.5);
  }
  asm volatile("" ::"v"(x));
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
int main() {
  double* out; unsigned long long* t;
  hipMalloc(&out, 8 * 64); hipMemset(out, 0, 8 * 64); hipMallocManaged(&t, 16);
  for (int mode = 0; mode < 3; ++mode) {
    const int reps = mode == 0 ? 1 : 2000;
    const int grid = mode == 2 ? 2048 : 1;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(chain, dim3(grid), dim3(64), 0, 0, out, t, 64);  // 1024 dependent FMAs
    hipDeviceSynchronize();
    printf("mode %d (%d launches, grid %d): s_memtime %llu ticks, s_memrealtime %llu ticks (x10 ns) for 1024 dependent f64 FMAs -> %.1f ns per FMA\n", mode, reps, grid, t[0], t[1],
           t[1] * 10.0 / 1024);
  }
  return 0;
}
