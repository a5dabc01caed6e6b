// Store-pattern experiments for the rate kernels' write stream (standalone; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o tools/store_bench && tools/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 plain, 1 nontemporal
__device__ __forceinline__ void st(v4f* p, v4f v) {
  if (MODE == 0) *p = v; else __builtin_nontemporal_store(v, p);
}

// A: flat, one float4 per thread (torch-like)
template <int MODE> __global__ __launch_bounds__(256) void k_flat(v4f* d, int64_t n4, float x) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) st<MODE>(d + i, v4f{x, x, x, x});
}
// B: flat grid-stride
template <int MODE> __global__ __launch_bounds__(256) void k_stride(v4f* d, int64_t n4, float x) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) st<MODE>(d + i, v4f{x, x, x, x});
}
// C: rate-kernel pattern: out[t][c][b]; lane owns 4 agents, loops over CPB cells (row stride B floats)
template <int MODE> __global__ __launch_bounds__(256) void k_rows(float* d, int T, int n, int B, int cpb, float x) {
  const uint32_t qrow = B / 4;
  const uint32_t p4 = blockIdx.x * 256u + threadIdx.x;
  const uint32_t t = p4 / qrow, q = p4 - t * qrow;
  if (t >= (uint32_t)T) return;
  const int c0 = blockIdx.y * cpb;
  int64_t off = ((int64_t)t * n + c0) * B + 4 * (int64_t)q;
  for (int c = 0; c < cpb; ++c) { st<MODE>(reinterpret_cast<v4f*>(d + off), v4f{x, x, x, (float)c}); off += B; }
}
// C2: as C but linear block id in ADDRESS order: agent segment fastest, then cell chunk, then t
template <int MODE> __global__ __launch_bounds__(256) void k_rows2(float* d, int T, int n, int B, int cpb, float x) {
  const uint32_t S = B / 1024, NC = n / cpb;
  const uint32_t L = blockIdx.x;
  const uint32_t seg = L % S, chunk = (L / S) % NC, t = L / (S * NC);
  const int c0 = chunk * cpb;
  int64_t off = ((int64_t)t * n + c0) * B + 1024 * seg + 4 * threadIdx.x;
  for (int c = 0; c < cpb; ++c) { st<MODE>(reinterpret_cast<v4f*>(d + off), v4f{x, x, x, (float)c}); off += B; }
}
// C3: as C2 plus the position loads of the real kernel (x,y float4 per thread, L2-resident)
template <int MODE> __global__ __launch_bounds__(256) void k_rows3(float* d, const float* px, const float* py, int T, int n, int B, int cpb) {
  const uint32_t S = B / 1024, NC = n / cpb;
  const uint32_t L = blockIdx.x;
  const uint32_t seg = L % S, chunk = (L / S) % NC, t = L / (S * NC);
  const int c0 = chunk * cpb;
  const int64_t po = (int64_t)t * 8 * B + 1024 * seg + 4 * threadIdx.x;
  const v4f X = *reinterpret_cast<const v4f*>(px + po), Y = *reinterpret_cast<const v4f*>(py + po);
  int64_t off = ((int64_t)t * n + c0) * B + 1024 * seg + 4 * threadIdx.x;
  for (int c = 0; c < cpb; ++c) {
    const float cx = 0.001f * (c0 + c), cy = 0.002f * (c0 + c);
    v4f r;
    r.x = __builtin_amdgcn_exp2f(-18.f * ((X.x - cx) * (X.x - cx) + (Y.x - cy) * (Y.x - cy)));
    r.y = __builtin_amdgcn_exp2f(-18.f * ((X.y - cx) * (X.y - cx) + (Y.y - cy) * (Y.y - cy)));
    r.z = __builtin_amdgcn_exp2f(-18.f * ((X.z - cx) * (X.z - cx) + (Y.z - cy) * (Y.z - cy)));
    r.w = __builtin_amdgcn_exp2f(-18.f * ((X.w - cx) * (X.w - cx) + (Y.w - cy) * (Y.w - cy)));
    st<MODE>(reinterpret_cast<v4f*>(d + off), r); off += B; }
}
// D: block covers a whole (t, cell-chunk) slab contiguously: thread loops over slab linearly
template <int MODE> __global__ __launch_bounds__(256) void k_slab(float* d, int64_t slab4, float x) {
  v4f* base = reinterpret_cast<v4f*>(d) + (int64_t)blockIdx.x * slab4;
  for (int64_t i = threadIdx.x; i < slab4; i += 256) st<MODE>(base + i, v4f{x, x, x, x});
}

template <class F> float timeit(F f, int reps = 7) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f(); hipDeviceSynchronize();
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}

int main() {
  const int T = 128, n = 1024, B = 4096;
  const int64_t bytes = (int64_t)T * n * B * 4;  // 2 GiB
  float* d; hipMalloc(&d, bytes);
  const int64_t n4 = bytes / 16;
  auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
  rep("flat plain", timeit([&] { k_flat<0><<<(unsigned)((n4 + 255) / 256), 256>>>((v4f*)d, n4, 1.f); }));
  rep("flat nt", timeit([&] { k_flat<1><<<(unsigned)((n4 + 255) / 256), 256>>>((v4f*)d, n4, 1.f); }));
  for (int g : {1024, 2048, 4096, 8192, 16384}) {
    char nm[64];
    snprintf(nm, 64, "grid-stride plain grid=%d", g); rep(nm, timeit([&] { k_stride<0><<<g, 256>>>((v4f*)d, n4, 1.f); }));
    snprintf(nm, 64, "grid-stride nt grid=%d", g); rep(nm, timeit([&] { k_stride<1><<<g, 256>>>((v4f*)d, n4, 1.f); }));
  }
  for (int cpb : {16, 64, 128, 256, 1024}) {
    dim3 grid(T * (B / 4) / 256, n / cpb);
    char nm[64];
    snprintf(nm, 64, "rows plain cpb=%d (grid %ux%u)", cpb, grid.x, grid.y); rep(nm, timeit([&] { k_rows<0><<<grid, 256>>>(d, T, n, B, cpb, 1.f); }));
    snprintf(nm, 64, "rows nt cpb=%d", cpb); rep(nm, timeit([&] { k_rows<1><<<grid, 256>>>(d, T, n, B, cpb, 1.f); }));
  }
  float *px; hipMalloc(&px, (int64_t)T * 8 * B * 4); hipMemset(px, 0, (int64_t)T * 8 * B * 4); float* py = px + B;
  for (int cpb : {1, 2, 4, 8, 16, 64}) {
    const unsigned g = (unsigned)((int64_t)T * (n / cpb) * (B / 1024));
    char nm[64];
    snprintf(nm, 64, "rows2 (addr order) plain cpb=%d (grid %u)", cpb, g); rep(nm, timeit([&] { k_rows2<0><<<g, 256>>>(d, T, n, B, cpb, 1.f); }));
    snprintf(nm, 64, "rows2 (addr order) nt cpb=%d", cpb); rep(nm, timeit([&] { k_rows2<1><<<g, 256>>>(d, T, n, B, cpb, 1.f); }));
    snprintf(nm, 64, "rows3 (+loads+math) plain cpb=%d", cpb); rep(nm, timeit([&] { k_rows3<0><<<g, 256>>>(d, px, py, T, n, B, cpb); }));
    snprintf(nm, 64, "rows3 (+loads+math) nt cpb=%d", cpb); rep(nm, timeit([&] { k_rows3<1><<<g, 256>>>(d, px, py, T, n, B, cpb); }));
  }
  for (int64_t slabKB : {64, 256, 1024, 4096}) {
    const int64_t slab4 = slabKB * 1024 / 16; const unsigned g = (unsigned)(n4 / slab4);
    char nm[64];
    snprintf(nm, 64, "slab plain %lldKB (grid %u)", (long long)slabKB, g); rep(nm, timeit([&] { k_slab<0><<<g, 256>>>(d, slab4, 1.f); }));
    snprintf(nm, 64, "slab nt %lldKB", (long long)slabKB); rep(nm, timeit([&] { k_slab<1><<<g, 256>>>(d, slab4, 1.f); }));
  }
  hipFree(d);
  return 0;
}
