// VALU issue-rate microbenchmark for the BVC accumulate stage (standalone): chip-wide throughput of v_exp_f32,
// v_fma_f32, v_pk_fma_f32, v_pk_add_f32 on their own and of the instruction mixes the kernel issues per term, at 8
// waves per SIMD with 8 independent chains per lane.  From the single-instruction rates it DERIVES the issue ceiling of
// a term of riab_bvc.hip's stage B (one v_exp_f32, one v_pk_fma_f32 and half a v_pk_add_f32 per term — two terms per
// packed instruction) and prints the measured mix beside it: bench.py's `valu` roofline uses the derived figure.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const v2f c0 = {0.999f, 0.998f}, c1 = {0.001f, 0.002f};
  for (int i = 0; i < iters; ++i) {
#define STEP(V) if (OP == 0) V = __builtin_amdgcn_exp2f(V * -0.5f); else if (OP == 1) V = fmaf(V, 0.999f, 0.001f); else if (OP == 2) V = __builtin_amdgcn_cosf(V); else if (OP == 3) V = __builtin_amdgcn_exp2f(fmaf(-V, V, 0.3f)); else if (OP == 4) V = __builtin_amdgcn_exp2f(-fabsf(V));
#define PSTEP(V) if (OP == 5) V = __builtin_elementwise_fma(V, c0, c1); else if (OP == 6) V = V + c1; else if (OP == 7) { const v2f t = __builtin_elementwise_fma(V, c0, c1); const v2f e = __builtin_elementwise_fma(-t, t, c1); V += v2f{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)}; }
    if (OP <= 4) { STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7) }
    else { PSTEP(p0) PSTEP(p1) PSTEP(p2) PSTEP(p3) PSTEP(p4) PSTEP(p5) PSTEP(p6) PSTEP(p7) }
  }
  const v2f ps = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + ps.x + ps.y;
}
int main() {
  float* d; hipMalloc(&d, 256 * 8192 * 4);
  const int iters = 4096; const int blocks = 256 * 8;  // 8 blocks/CU = 8 waves/SIMD
  const char* names[] = {"v_mul + v_exp (exp2(x*c))", "v_fma_f32", "v_cos_f32", "v_fma + v_exp (exp2(fma(-x,x,c)))", "v_exp_f32",
                         "v_pk_fma_f32", "v_pk_add_f32", "BVC term pair: 2 pk_fma + 2 exp + pk_add"};
  double cyc[8];
  int clk_khz = 2400000; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double clk = clk_khz * 1e3;
  for (int op = 0; op < 8; ++op) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&] {
      switch (op) {
        case 0: k<0><<<blocks, 256>>>(d, iters, 0.1f); break; case 1: k<1><<<blocks, 256>>>(d, iters, 0.1f); break;
        case 2: k<2><<<blocks, 256>>>(d, iters, 0.1f); break; case 3: k<3><<<blocks, 256>>>(d, iters, 0.1f); break;
        case 4: k<4><<<blocks, 256>>>(d, iters, 0.1f); break; case 5: k<5><<<blocks, 256>>>(d, iters, 0.1f); break;
        case 6: k<6><<<blocks, 256>>>(d, iters, 0.1f); break; default: k<7><<<blocks, 256>>>(d, iters, 0.1f); break;
      }
    };
    run(); hipDeviceSynchronize();
    hipEventRecord(a); run(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters * 8;   // lane-iterations
    cyc[op] = clk * (ms * 1e-3) / (ops / 64 / 1024);  // cycles per wave-iteration per SIMD
    printf("%-44s %.3f ms  %.2f T lane-iterations/s  %.2f cycles per wave-iteration per SIMD at %.2f GHz\n", names[op], ms, ops / ms / 1e9,
           cyc[op], clk / 1e9);
  }
  // a BVC term = 1 v_exp + 1 v_pk_fma (two packed fmas per two terms) + 1/2 v_pk_add
  const double term = cyc[4] + cyc[5] + 0.5 * cyc[6];
  printf("derived from the single-instruction rates: %.2f cycles per wave-term -> %.2f T terms/s chip-wide (1024 SIMDs x 64 lanes x %.2f GHz)\n",
         term, 1024 * 64 * clk / term / 1e12, clk / 1e9);
  printf("measured with the kernel's own mix: %.2f cycles per wave-term -> %.2f T terms/s\n", cyc[7] / 2, 1024 * 64 * clk / (cyc[7] / 2) / 1e12);
  return 0;
}
