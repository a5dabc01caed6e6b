// v_exp_f32 / v_fma_f32 / v_cos_f32 issue-rate microbenchmark (standalone).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  for (int i = 0; i < iters; ++i) {
#define STEP(x) if (OP == 0) x = __builtin_amdgcn_exp2f(x * -0.5f); else if (OP == 1) x = fmaf(x, 0.999f, 0.001f); else if (OP == 2) x = __builtin_amdgcn_cosf(x); else x = __builtin_amdgcn_exp2f(fmaf(-x, x, 0.3f));
    STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  float* d; hipMalloc(&d, 256 * 8192 * 4);
  const int iters = 4096; const int blocks = 256 * 8;  // 8 blocks/CU = 8 waves/SIMD
  const char* names[] = {"exp2(x*c) [mul+exp]", "fma", "cos", "exp2(fma(-x,x,c)) [fma+exp]"};
  for (int op = 0; op < 4; ++op) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&] { if (op == 0) k<0><<<blocks, 256>>>(d, iters, 0.1f); else if (op == 1) k<1><<<blocks, 256>>>(d, iters, 0.1f); else if (op == 2) k<2><<<blocks, 256>>>(d, iters, 0.1f); else k<3><<<blocks, 256>>>(d, iters, 0.1f); };
    run(); hipDeviceSynchronize();
    hipEventRecord(a); run(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters * 8;
    printf("%-30s %.3f ms  %.2f T iterations/s  (%.2f cycles per wave-iteration per SIMD at 2.4 GHz)\n", names[op], ms, ops / ms / 1e9,
           2.4e9 * (ms * 1e-3) / (ops / 64 / 1024));
  }
  return 0;
}
