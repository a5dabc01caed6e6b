// Store patterns of a PERSISTENT rate kernel (standalone experiment behind csrc/riab_rates.hip: rate_stream_kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o tools/stream_bench && tools/stream_bench
// out[t][c][b], T x n x B floats (2 GiB at 128 x 1024 x 4096); every variant writes every byte once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));

// 0: the non-persistent shape of rate_kernel_wide (grid = every workgroup, x fastest): the 6.4 TB/s reference
__global__ __launch_bounds__(256) void k_wide(float* d, int n, int B, float x) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  int64_t off = ((int64_t)blockIdx.z * n + blockIdx.y * 4) * B + 4 * (int64_t)q;
  for (int j = 0; j < 4; ++j) { *reinterpret_cast<v4f*>(d + off) = v4f{x, x, x, (float)j}; off += B; }
}

// persistent: a wave owns items i = gw, gw + NW, ...; item -> (t, chunk, q) with q fastest; the wave writes ROWS rows
// (1 KB each, stride B floats) of chunk `chunk`.  ROT: start row rotated by a per-item offset.
template <int ROWS, bool ROT>
__global__ __launch_bounds__(256) void k_persist_rows(float* d, int T, int n, int B, float x) {
  const int lane = threadIdx.x & 63;
  const uint32_t gw = blockIdx.x * 4u + (threadIdx.x >> 6), NW = gridDim.x * 4u;
  const uint32_t Q = B / 256, G = n / ROWS;
  const uint32_t total = (uint32_t)T * G * Q;
  for (uint32_t i = gw; i < total; i += NW) {
    const uint32_t q = i % Q, g = (i / Q) % G, t = i / (Q * G);
    const int64_t base = ((int64_t)t * n + g * ROWS) * B + q * 256 + 4 * lane;
    const int rot = ROT ? (int)((g * 4u + t) & (ROWS - 1)) : 0;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int r = (j + rot) & (ROWS - 1);
      *reinterpret_cast<v4f*>(d + base + (int64_t)r * B) = v4f{x, x, x, (float)j};
    }
  }
}

// persistent, a wave writes ROWS rows x 4 KB: lane covers 4 float4 per row (1024 agents per wave-item)
template <int ROWS>
__global__ __launch_bounds__(256) void k_persist_wide(float* d, int T, int n, int B, float x) {
  const int lane = threadIdx.x & 63;
  const uint32_t gw = blockIdx.x * 4u + (threadIdx.x >> 6), NW = gridDim.x * 4u;
  const uint32_t Q = B / 1024, G = n / ROWS;
  const uint32_t total = (uint32_t)T * G * Q;
  for (uint32_t i = gw; i < total; i += NW) {
    const uint32_t q = i % Q, g = (i / Q) % G, t = i / (Q * G);
    const int64_t base = ((int64_t)t * n + g * ROWS) * B + q * 1024 + 4 * lane;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<v4f*>(d + base + (int64_t)j * B + k * 256) = v4f{x, x, x, (float)j};
    }
  }
}

// persistent, purely linear: wave-item = KB consecutive KiB
template <int KB>
__global__ __launch_bounds__(256) void k_persist_linear(float* d, int64_t n4, float x) {
  const int lane = threadIdx.x & 63;
  const uint64_t gw = blockIdx.x * 4u + (threadIdx.x >> 6), NW = gridDim.x * 4u;
  const uint64_t items = n4 / (64 * KB);
  for (uint64_t i = gw; i < items; i += NW) {
    v4f* p = reinterpret_cast<v4f*>(d) + i * 64 * KB + lane;
#pragma unroll
    for (int j = 0; j < KB; ++j) p[j * 64] = v4f{x, x, x, (float)j};
  }
}

// persistent, WORKGROUP-granular items like the wide kernel: workgroup-item = 4 rows x 4 KB (x fastest), the 4 waves
// of a workgroup write 1 KB each of every row
__global__ __launch_bounds__(256) void k_persist_wg(float* d, int T, int n, int B, float x) {
  const uint32_t S = B / 1024, G = n / 4;
  const uint32_t total = (uint32_t)T * G * S;
  for (uint32_t i = blockIdx.x; i < total; i += gridDim.x) {
    const uint32_t s = i % S, g = (i / S) % G, t = i / (S * G);
    int64_t off = ((int64_t)t * n + g * 4) * B + s * 1024 + 4 * threadIdx.x;
    for (int j = 0; j < 4; ++j) { *reinterpret_cast<v4f*>(d + off) = v4f{x, x, x, (float)j}; off += B; }
  }
}

// ---- the same with the PlaceCells arithmetic per store (4 exp2 + ~26 VALU per 1 KB): does compute cost bandwidth?
typedef const __attribute__((address_space(4))) float* cptr;
__device__ __forceinline__ v4f pc_eval(v4f X, v4f Y, float cx, float cy, float k) {
  v4f r;
  r.x = __builtin_amdgcn_exp2f(k * ((X.x - cx) * (X.x - cx) + (Y.x - cy) * (Y.x - cy)));
  r.y = __builtin_amdgcn_exp2f(k * ((X.y - cx) * (X.y - cx) + (Y.y - cy) * (Y.y - cy)));
  r.z = __builtin_amdgcn_exp2f(k * ((X.z - cx) * (X.z - cx) + (Y.z - cy) * (Y.z - cy)));
  r.w = __builtin_amdgcn_exp2f(k * ((X.w - cx) * (X.w - cx) + (Y.w - cy) * (Y.w - cy)));
  return r;
}
__global__ __launch_bounds__(256) void k_wide_c(float* d, const float* tab, const float* pos, int n, int B) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  const v4f X = *reinterpret_cast<const v4f*>(pos + (int64_t)blockIdx.z * 2 * B + 4 * q);
  const v4f Y = *reinterpret_cast<const v4f*>(pos + (int64_t)blockIdx.z * 2 * B + B + 4 * q);
  const cptr t = (cptr)(const void*)tab;
  int64_t off = ((int64_t)blockIdx.z * n + blockIdx.y * 4) * B + 4 * (int64_t)q;
  for (int j = 0; j < 4; ++j) {
    const int c = blockIdx.y * 4 + j;
    *reinterpret_cast<v4f*>(d + off) = pc_eval(X, Y, t[3 * c], t[3 * c + 1], t[3 * c + 2]);
    off += B;
  }
}
// non-persistent + a readiness poll per WAVE before the position loads (flags always ready here): what does the
// extra dependent round trip cost?  POLL 0 none, 1 poll then load;  SC1: positions through agent-scope loads
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
__device__ __forceinline__ v4f ld_sc1(const float* p) {
  gu64* g = (gu64*)(uintptr_t)p;
  const unsigned long long lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v4f{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32))};
}
template <int CPB, int POLL, bool SC1>
__global__ __launch_bounds__(256) void k_gated(float* d, const float* tab, const float* pos, const unsigned* flags, int n, int B) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  const uint32_t t = blockIdx.z;
  if (POLL == 1) {
    const uint32_t wq = (blockIdx.x * 4u + (threadIdx.x >> 6));  // 256-agent sub-segment of this wave
    for (;;) {
      unsigned v = 0xffffffffu;
      if (lane < 4) v = __hip_atomic_load((gu32*)(uintptr_t)(flags + 4 * wq + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_ballot_w64(v <= t) == 0) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
  const int64_t po = (int64_t)t * 2 * B + 4 * q;
  const v4f X = SC1 ? ld_sc1(pos + po) : *reinterpret_cast<const v4f*>(pos + po);
  const v4f Y = SC1 ? ld_sc1(pos + po + B) : *reinterpret_cast<const v4f*>(pos + po + B);
  const cptr tb = (cptr)(const void*)tab;
  int64_t off = ((int64_t)t * n + blockIdx.y * CPB) * B + 4 * (int64_t)q;
#pragma unroll
  for (int j = 0; j < CPB; ++j) {
    const int c = blockIdx.y * CPB + j;
    *reinterpret_cast<v4f*>(d + off) = pc_eval(X, Y, tb[3 * c], tb[3 * c + 1], tb[3 * c + 2]);
    off += B;
  }
}

// persistent with compute: wave-item = ROWS cells x QPL quads of 256 agents (QPL KB contiguous per row)
template <int ROWS, int QPL>
__global__ __launch_bounds__(256) void k_persist_c(float* d, const float* tab, const float* pos, int T, int n, int B) {
  const int lane = threadIdx.x & 63;
  const uint32_t gw = blockIdx.x * 4u + (threadIdx.x >> 6), NW = gridDim.x * 4u;
  const uint32_t Q = B / (256 * QPL), G = n / ROWS;
  const uint32_t total = (uint32_t)T * G * Q;
  const cptr tb = (cptr)(const void*)tab;
  for (uint32_t i = gw; i < total; i += NW) {
    const uint32_t q = i % Q, g = (i / Q) % G, t = i / (Q * G);
    v4f X[QPL], Y[QPL];
#pragma unroll
    for (int k = 0; k < QPL; ++k) {
      X[k] = *reinterpret_cast<const v4f*>(pos + (int64_t)t * 2 * B + q * 256 * QPL + k * 256 + 4 * lane);
      Y[k] = *reinterpret_cast<const v4f*>(pos + (int64_t)t * 2 * B + B + q * 256 * QPL + k * 256 + 4 * lane);
    }
    const int64_t base = ((int64_t)t * n + g * ROWS) * B + q * 256 * QPL + 4 * lane;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int c = g * ROWS + j;
      const float cx = tb[3 * c], cy = tb[3 * c + 1], kk = tb[3 * c + 2];
#pragma unroll
      for (int k = 0; k < QPL; ++k)
        *reinterpret_cast<v4f*>(d + base + (int64_t)j * B + k * 256) = pc_eval(X[k], Y[k], cx, cy, kk);
    }
  }
}

template <class F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}

int main() {
  const int T = 128, n = 1024, B = 4096;
  const int64_t bytes = (int64_t)T * n * B * 4;  // 2 GiB
  float* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const int64_t n4 = bytes / 16;
  auto rep = [&](const char* name, int wgs, float ms) { printf("%-40s wgs/cu %d  %8.3f ms  %7.0f GB/s\n", name, wgs, ms, bytes / ms / 1e6); fflush(stdout); };
  rep("wide (non-persistent, 4 rows x 4 KB/WG)", 0, timeit([&] { hipLaunchKernelGGL(k_wide, dim3(B / 1024, n / 4, T), dim3(256), 0, 0, d, n, B, 1.f); }));
  for (int w : {7}) {
    const dim3 g(256 * w), b(256);
    rep("persist rows 64 x 1KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_rows<64, false>), g, b, 0, 0, d, T, n, B, 1.f); }));
    rep("persist rows 64 x 1KB rotated", w, timeit([&] { hipLaunchKernelGGL((k_persist_rows<64, true>), g, b, 0, 0, d, T, n, B, 1.f); }));
    rep("persist rows 16 x 1KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_rows<16, false>), g, b, 0, 0, d, T, n, B, 1.f); }));
    rep("persist rows 8 x 1KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_rows<8, false>), g, b, 0, 0, d, T, n, B, 1.f); }));
    rep("persist rows 4 x 1KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_rows<4, false>), g, b, 0, 0, d, T, n, B, 1.f); }));
    rep("persist wide 16 rows x 4KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_wide<16>), g, b, 0, 0, d, T, n, B, 1.f); }));
    rep("persist wide 4 rows x 4KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_wide<4>), g, b, 0, 0, d, T, n, B, 1.f); }));
    rep("persist linear 8 KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_linear<8>), g, b, 0, 0, (float*)d, n4, 1.f); }));
    rep("persist linear 64 KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_linear<64>), g, b, 0, 0, (float*)d, n4, 1.f); }));
    rep("persist WG items (4 rows x 4 KB)", w, timeit([&] { hipLaunchKernelGGL(k_persist_wg, g, b, 0, 0, d, T, n, B, 1.f); }));
  }
  float *tab, *pos;
  hipMalloc(&tab, n * 3 * 4); hipMalloc(&pos, (int64_t)T * 2 * B * 4);
  {
    std::vector<float> h(n * 3), hp((size_t)T * 2 * B);
    for (int c = 0; c < n; ++c) { h[3 * c] = (c % 32) / 32.f; h[3 * c + 1] = (c / 32) / 32.f; h[3 * c + 2] = -18.f; }
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = (float)rand() / RAND_MAX;
    hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(pos, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
  }
  unsigned* flags; hipMalloc(&flags, 4096); hipMemset(flags, 0x7f, 4096);
#define GATED(CPB, POLL, SC1, NAME) rep(NAME, 0, timeit([&] { hipLaunchKernelGGL((k_gated<CPB, POLL, SC1>), dim3(B / 1024, n / CPB, T), dim3(256), 0, 0, d, tab, pos, flags, n, B); }))
  for (int r = 0; r < 2; ++r) {
    GATED(4, 0, false, "GATED cpb4 nopoll plain");
    GATED(4, 0, true, "GATED cpb4 nopoll sc1");
    GATED(4, 1, false, "GATED cpb4 poll plain");
    GATED(4, 1, true, "GATED cpb4 poll sc1");
    GATED(8, 0, false, "GATED cpb8 nopoll plain");
    GATED(8, 1, false, "GATED cpb8 poll plain");
    GATED(8, 1, true, "GATED cpb8 poll sc1");
    GATED(16, 1, true, "GATED cpb16 poll sc1");
  }
  rep("COMPUTE wide (non-persistent)", 0, timeit([&] { hipLaunchKernelGGL(k_wide_c, dim3(B / 1024, n / 4, T), dim3(256), 0, 0, d, tab, pos, n, B); }));
  for (int w : {7}) {
    const dim3 g(256 * w), b(256);
    rep("COMPUTE persist 64 rows x 1KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_c<64, 1>), g, b, 0, 0, d, tab, pos, T, n, B); }));
    rep("COMPUTE persist 16 rows x 1KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_c<16, 1>), g, b, 0, 0, d, tab, pos, T, n, B); }));
    rep("COMPUTE persist 32 rows x 2KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_c<32, 2>), g, b, 0, 0, d, tab, pos, T, n, B); }));
    rep("COMPUTE persist 16 rows x 4KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_c<16, 4>), g, b, 0, 0, d, tab, pos, T, n, B); }));
    rep("COMPUTE persist 8 rows x 4KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_c<8, 4>), g, b, 0, 0, d, tab, pos, T, n, B); }));
    rep("COMPUTE persist 4 rows x 4KB", w, timeit([&] { hipLaunchKernelGGL((k_persist_c<4, 4>), g, b, 0, 0, d, tab, pos, T, n, B); }));
  }
  return 0;
}
