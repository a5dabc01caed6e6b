// Device-side helpers shared by the gfx950 kernels: Philox4x32-10, vector
// load/store wrappers, error codes.  CDNA4 only (wave64); no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "riab_hip.h"

#define RIAB_TAG_MOTION 0x4D4F5449u
#define RIAB_TAG_SPIKES 0x53504B00u
#define RIAB_TAG_NOISE 0x4E4F4900u
#define RIAB_MAX_RESAMPLES 64  // bound on the rejection loop of the resample boundary condition

namespace riab {

// riab_set_option's storage (defined in riab_rates.hip): plain ints, read per call
extern int g_options[RIAB_OPT_COUNT];

// the one-launch closed-loop step (riab_step1.hip): a population whose update() rides in the agent step's launch, and
// its rows of this step
#define RIAB_STEP1_MAX_POPS 4
struct Step1PopRef {
  const RiabPopulation* pop;
  float* rates_row;
  uint8_t* spikes_row;
};

struct u32x4 {
  uint32_t x, y, z, w;
};

// Philox4x32-10 (Salmon et al., SC'11).  32x32->64 products map to
// v_mad_u64_u32 on gfx950.  Counter-based: no state, any (step, cell, agent)
// can be regenerated on the host (oracle/riab_oracle.py: philox4x32_10).
template <int ROUNDS>
__device__ __forceinline__ u32x4 philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return philox4x32_r<10>(c0, c1, c2, c3, k0, k1);
}
// The spike streams: Philox4x32-7 — the smallest round count the generator's authors found to pass BigCrush (SC'11, table
// 2; 10 is their default with a safety margin).  One call per (cell, four agents) in the epilogue of kernels that are
// otherwise bound by their stores: at ten rounds the generator was two thirds of the epilogue's instructions and took a
// PlaceCells kernel from 6.5 to 4.3 TB/s (round 6).  oracle/riab_oracle.py: spike_uniforms restates it.
__device__ __forceinline__ u32x4 philox4x32_spikes(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return philox4x32_r<7>(c0, c1, c2, c3, k0, k1);
}

// Rows that are written once and not read again by the call that writes them (rates, spikes): streamed.  POLICY
// RIAB_STORE_NT: nontemporal stores — the lines pass through the L2 marked for early eviction, write-combined there, but
// what is still dirty when the kernel ends is written back by the dispatch's closing release before the next kernel may
// start; RIAB_STORE_WT: write-through as well (sc1) — the lines leave for memory while the kernel still runs and nothing
// is left to flush.  [MI355X, round 6] a kernel that runs for microseconds and is followed by one that needs its results
// (the one-launch closed-loop step: 16.8 MB of rates in 1.5 us, then the next step) gains 12 % from WT (9.67 -> 8.50 us
// per step); a kernel that streams for milliseconds (rate_kernel_gated, 1024 rows) loses 9 % to it (1.45 -> 1.32 G
// agent-steps/s: the L2 no longer combines the lanes' quads into full lines ahead of the memory channel).
#ifndef RIAB_WT_MODE
#define RIAB_WT_MODE 3
#endif
#if RIAB_WT_MODE == 1
#define RIAB_WT_BITS "sc1"
#elif RIAB_WT_MODE == 2
#define RIAB_WT_BITS "sc0 sc1"
#else
#define RIAB_WT_BITS "sc1 nt"
#endif
#define RIAB_STORE_NT 0
#define RIAB_STORE_WT 1
typedef float riab_v4f __attribute__((ext_vector_type(4)));
template <int POLICY>
__device__ __forceinline__ void store_stream(float* p, riab_v4f v) {
  // (s_nop BEHIND the store: a vector instruction must not write the data registers of a store of more than 64 bits in
  // the two wait states behind it — the store reads them late.  The compiler pads its own stores and cannot see inside an
  // asm statement: without the padding the z and w elements of a quad went to memory as whatever the next instruction
  // made of their registers — tests/test_gpu_parity.py, spike shapes, and test_gpu_fused.py caught it.)
  if (POLICY == RIAB_STORE_WT)
    asm volatile("global_store_dwordx4 %0, %1, off " RIAB_WT_BITS "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else __builtin_nontemporal_store(v, reinterpret_cast<riab_v4f*>(p));
}
template <int POLICY>
__device__ __forceinline__ void store_stream(uint32_t* p, uint32_t v) {
  if (POLICY == RIAB_STORE_WT) asm volatile("global_store_dword %0, %1, off " RIAB_WT_BITS ::"v"(p), "v"(v) : "memory");
  else __builtin_nontemporal_store(v, p);
}

// Read-only, wave-uniform tables: a pointer in the constant address space tells the compiler the
// memory is invariant, so uniform-index reads become scalar loads (s_load_dwordx4 -> SGPR operands)
// instead of per-lane vector loads.
typedef const __attribute__((address_space(4))) float* const_f32_ptr;
__device__ __forceinline__ const_f32_ptr as_const_table(const float* p) { return (const_f32_ptr)(const void*)p; }

// fp32 uniform in [0,1) with 24 random bits — exactly representable, so the
// host can regenerate it bit for bit.
__device__ __forceinline__ float u01_24(uint32_t w) { return (float)(w >> 8) * 0x1.0p-24f; }

// Strict segment/segment intersection test of utils.vector_intercepts
// (utils.py:74-106: 0 < l_a < 1 and 0 < l_b < 1), evaluated with sign logic on the
// float64 cross products instead of the two divisions (equal up to the last ulp of
// the quotient; parallel segments give den == 0 -> no hit, like +-inf/NaN in NumPy).
__device__ __forceinline__ bool seg_hit(double p0x, double p0y, double p1x, double p1y, double ax, double ay,
                                        double bx, double by) {
  const double sax = p1x - p0x, say = p1y - p0y;  // list a = line of sight
  const double sbx = bx - ax, sby = by - ay;      // list b = wall
  const double d0x = ax - p0x, d0y = ay - p0y;
  const double den_a = sax * (-sby) + say * sbx;
  const double num_a = d0x * (-sby) + d0y * sbx;
  const double den_b = sbx * (-say) + sby * sax;
  const double num_b = (-d0x) * (-say) + (-d0y) * sax;
  const bool ia = (den_a > 0) ? (num_a > 0 && num_a < den_a) : (den_a < 0 ? (num_a < 0 && num_a > den_a) : false);
  const bool ib = (den_b > 0) ? (num_b > 0 && num_b < den_b) : (den_b < 0 ? (num_b < 0 && num_b > den_b) : false);
  return ia && ib;
}
// Strict point-in-polygon over the edges flagged in `mask`, read through `edge(k, ax, ay, bx, by)`: even-odd
// crossings of the ray towards +x; a point ON an edge is not contained (shapely's `Polygon.contains`, which the
// reference calls in Environment.check_if_position_is_in_environment, Environment.py:808-816).  The edges may
// belong to several disjoint polygons (the holes): the parity then says "inside one of them".
template <class EdgeFn>
__device__ __forceinline__ bool polygon_contains_strict(double px, double py, uint64_t mask, EdgeFn edge) {
  // (no fused multiply-adds: whether a point is ON an edge is `cross == 0` in the reference's plain float64 arithmetic)
#pragma clang fp contract(off)
  bool odd = false, on_edge = false;
  for (uint64_t rest = mask; rest; rest &= rest - 1) {
    const int k = __ffsll((long long)rest) - 1;
    // a wall of the table runs from corner i+1 to corner i (Environment.py:139-142); the polygon's own edge, as the
    // reference's geometry library walks it, from corner i to corner i+1: same segment, but `cross == 0` and the
    // crossing abscissa round differently for the two orientations, so the edge is read back to front
    double ax, ay, bx, by;
    edge(k, bx, by, ax, ay);
    const double cross = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
    if (cross == 0.0 && fmin(ax, bx) <= px && px <= fmax(ax, bx) && fmin(ay, by) <= py && py <= fmax(ay, by)) on_edge = true;
    if ((ay > py) != (by > py)) {
      const double x_cross = ax + (py - ay) * (bx - ax) / (by - ay);
      if (px < x_cross) odd = !odd;
    }
  }
  return odd && !on_edge;
}

// lo + u * (hi - lo) with every operation rounded on its own (the host restates these draws bit for bit; HIP's
// __dmul_rn / __dadd_rn are plain operators and contract like any other)
__device__ __forceinline__ double uniform_between(double lo, double hi, float u) {
#pragma clang fp contract(off)
  const double span = hi - lo;
  const double prod = (double)u * span;
  return lo + prod;
}

// Environment.check_if_position_is_in_environment (Environment.py:781-818) for the geometry of a RiabEnv
struct EnvShape {
  double e0, e1, e2, e3;
  uint64_t boundary_mask;  // edges of a polygonal boundary (0: the boundary is the rectangle e0..e3)
  uint64_t hole_mask;      // edges of the holes
};
__host__ __device__ __forceinline__ EnvShape make_env_shape(const RiabEnv* env) {
  EnvShape s;
  s.e0 = env->extent[0]; s.e1 = env->extent[1]; s.e2 = env->extent[2]; s.e3 = env->extent[3];
  s.boundary_mask = env->polygon ? (env->n_boundary >= 64 ? ~0ull : ((1ull << env->n_boundary) - 1ull)) : 0ull;
  s.hole_mask = env->hole_mask;
  return s;
}
template <class EdgeFn>
__device__ __forceinline__ bool env_contains(const EnvShape& s, double px, double py, EdgeFn edge) {
  bool in = s.boundary_mask ? polygon_contains_strict(px, py, s.boundary_mask, edge)
                            : (px > s.e0 && px < s.e1 && py > s.e2 && py < s.e3);
  if (in && s.hole_mask) in = !polygon_contains_strict(px, py, s.hole_mask, edge);
  return in;
}
// argument checks shared by the entry points
static inline int check_env_shape(const RiabEnv* env) {
  const uint64_t all = env->n_walls >= 64 ? ~0ull : ((1ull << (env->n_walls < 0 ? 0 : env->n_walls)) - 1ull);
  if (env->hole_mask & ~all) return RIAB_EINVAL;
  if (env->polygon) {
    if (env->periodic || env->n_boundary < 3 || env->n_boundary > env->n_walls) return RIAB_EINVAL;
    if (env->hole_mask & ((1ull << env->n_boundary) - 1ull)) return RIAB_EINVAL;
  }
  return RIAB_OK;
}
}  // namespace riab
