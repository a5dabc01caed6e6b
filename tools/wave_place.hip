// How gfx950 places the waves of a 768-thread workgroup (the RESERVING shape of rate_kernel_gated, include/riab_hip.h
// "Residency"): (1) every workgroup's twelve waves must be three per SIMD; (2) a compute unit must hold exactly two such
// workgroups (6 of 8 wave slots per SIMD), i.e. the chip 512; (3) with the chip full of them, a 256-thread workgroup that
// needs one wave slot per SIMD, 224 registers and 80 KB of LDS (a trajectory workgroup's footprint) must still be placed.
//   hipcc --offload-arch=gfx950 -O2 -o tools/wave_place tools/wave_place.hip && ./tools/wave_place
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every wave records its HW_ID; the workgroup then waits until `target` workgroups have arrived (or a time limit)
__global__ __launch_bounds__(768) void place_kernel(unsigned* arrived, unsigned target, unsigned* hw, unsigned* xcc,
                                                    unsigned* saw_all, unsigned long long limit_ticks) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned id, xc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
  if (lane == 0) {
    hw[blockIdx.x * 12 + wave] = id;
    xcc[blockIdx.x * 12 + wave] = xc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(arrived, 1u);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned ok = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < limit_ticks) {
      if (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(32);
    }
    saw_all[blockIdx.x] = ok;
  }
  __syncthreads();
}

// the chip is full of waiting 768-thread workgroups; does a workgroup with a trajectory workgroup's footprint get in?
__global__ __launch_bounds__(768) void hold_kernel(unsigned* arrived, const unsigned* release, unsigned long long limit_ticks) {
  if (threadIdx.x == 0) {
    atomicAdd(arrived, 1u);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < limit_ticks) {
      if (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      __builtin_amdgcn_s_sleep(32);
    }
  }
  __syncthreads();
}
__global__ __launch_bounds__(256) void big_kernel(unsigned* got_in, unsigned* release, unsigned n_wgs, double* sink) {
  __shared__ double lds[80 * 1024 / 8];
  // ~224 registers per lane: 100 live doubles
  double r[100];
#pragma unroll
  for (int i = 0; i < 100; ++i) r[i] = (double)(threadIdx.x + i) * 1.000001;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 100; ++i) r[i] = r[i] * r[(i + 7) % 100] + 0.5;
  lds[threadIdx.x] = r[0];
  __syncthreads();
  double s = lds[(threadIdx.x + 1) & 255];
#pragma unroll
  for (int i = 0; i < 100; ++i) s += r[i];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0 && atomicAdd(got_in, 1u) + 1 == n_wgs)
    __hip_atomic_store(release, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  int khz = 100000;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  unsigned *d_arr, *d_hw, *d_xcc, *d_saw, *d_rel, *d_in;
  double* d_sink;
  const int maxwg = 2048;
  CHECK(hipMalloc(&d_arr, 4)); CHECK(hipMalloc(&d_rel, 4)); CHECK(hipMalloc(&d_in, 4)); CHECK(hipMalloc(&d_sink, 8));
  CHECK(hipMalloc(&d_hw, maxwg * 12 * 4)); CHECK(hipMalloc(&d_xcc, maxwg * 12 * 4)); CHECK(hipMalloc(&d_saw, maxwg * 4));
  std::vector<unsigned> hw(maxwg * 12), xcc(maxwg * 12), saw(maxwg);
  // (1) + (2): n workgroups that all wait for each other: completes at once iff all n are resident together
  for (int n : {256, 512, 513, 640, 768}) {
    CHECK(hipMemset(d_arr, 0, 4));
    CHECK(hipMemset(d_saw, 0, maxwg * 4));
    hipLaunchKernelGGL(place_kernel, dim3(n), dim3(768), 0, 0, d_arr, (unsigned)n, d_hw, d_xcc, d_saw, (unsigned long long)khz * 20ull);  // 20 ms
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hw.data(), d_hw, n * 12 * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(xcc.data(), d_xcc, n * 12 * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(saw.data(), d_saw, n * 4, hipMemcpyDeviceToHost));
    int all = 0, even = 0;
    for (int w = 0; w < n; ++w) {
      all += saw[w];
      int per_simd[4] = {0, 0, 0, 0};
      for (int k = 0; k < 12; ++k) per_simd[(hw[w * 12 + k] >> 4) & 3]++;
      even += per_simd[0] == 3 && per_simd[1] == 3 && per_simd[2] == 3 && per_simd[3] == 3;
    }
    printf("%4d workgroups of 768 threads: %4d saw all %d resident at once (%s); %4d / %d have 3 waves on each SIMD\n", n, all, n,
           all == n ? "ALL co-resident" : "NOT co-resident: capacity exceeded", even, n);
  }
  // (3) the chip full of held 768-thread workgroups (more than fit), then 128 trajectory-sized workgroups on another stream
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CHECK(hipMemset(d_arr, 0, 4)); CHECK(hipMemset(d_rel, 0, 4)); CHECK(hipMemset(d_in, 0, 4));
  hipLaunchKernelGGL(hold_kernel, dim3(2048), dim3(768), 0, s1, d_arr, d_rel, (unsigned long long)khz * 200ull);  // held up to 200 ms
  unsigned arrived = 0;
  for (int i = 0; i < 200 && arrived < 512; ++i) {
    CHECK(hipMemcpy(&arrived, d_arr, 4, hipMemcpyDeviceToHost));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s2));
  hipLaunchKernelGGL(big_kernel, dim3(128), dim3(256), 0, s2, d_in, d_rel, 128u, d_sink);
  CHECK(hipEventRecord(e1, s2));
  CHECK(hipStreamSynchronize(s2));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned got = 0;
  CHECK(hipMemcpy(&got, d_in, 4, hipMemcpyDeviceToHost));
  CHECK(hipDeviceSynchronize());
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, (const void*)big_kernel));
  printf("chip full of held 768-thread workgroups (%u resident when probed): 128 workgroups of 256 threads, %d registers, %zu B LDS "
         "were placed and finished in %.3f ms (%u / 128) -> %s\n", arrived, fa.numRegs, (size_t)fa.sharedSizeBytes, ms, got,
         (got == 128 && ms < 50.0f) ? "PLACED while the holders were resident" : "NOT placed until the holders left");
  return 0;
}
