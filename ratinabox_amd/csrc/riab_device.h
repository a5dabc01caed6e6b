// Device-side helpers shared by the gfx950 kernels: Philox4x32-10, vector
// load/store wrappers, error codes.  CDNA4 only (wave64); no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "riab_hip.h"

#define RIAB_TAG_MOTION 0x4D4F5449u
#define RIAB_TAG_SPIKES 0x53504B00u
#define RIAB_TAG_NOISE 0x4E4F4900u

namespace riab {

struct u32x4 {
  uint32_t x, y, z, w;
};

// Philox4x32-10 (Salmon et al., SC'11).  32x32->64 products map to
// v_mad_u64_u32 on gfx950.  Counter-based: no state, any (step, cell, agent)
// can be regenerated on the host (oracle/riab_oracle.py: philox4x32_10).
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
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
}  // namespace riab
