#pragma once
// The firing-rate functors of the store-bound populations (PlaceCells, GridCells, HeadDirectionCells / VelocityCells /
// SpeedCell), the argument block of their kernels and the Poisson-spike epilogue.  Shared by the rate kernels
// (riab_rates.hip) and the one-launch closed-loop step (riab_step1.hip): the same inlined code on the same operands,
// so a rate is the same bits whichever kernel evaluates it.
#include "riab_device.h"

namespace riab {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct RateArgs {
  const float* pos_x;
  const float* pos_y;
  const float* hd_x;
  const float* hd_y;
  int64_t pos_ld;
  int64_t nquads;  // T * (B/4)
  int64_t qrow;    // B/4
  int64_t B;
  float* rates;
  uint8_t* spikes;
  const float* u_in;
  float dt, fr_scale, fr_min;
  uint32_t k0, k1;       // Philox key (seed)
  uint32_t step0;
  uint32_t tag;          // RIAB_TAG_SPIKES | pop_id
  uint32_t group0;       // agent_id0 / 4
  int32_t n;
  int32_t cells_per_block;
};

struct PosQuad {
  v4f x, y;
};

__device__ __forceinline__ v4f ldv4(const float* p) { return *reinterpret_cast<const v4f*>(p); }
// 16 bytes another kernel has published write-through: two 8-byte relaxed agent-scope loads
// (global_load_dwordx2 ... sc1: served by L2, never by this CU's L1)
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ v4f ldv4_agent(const float* p) {
  gu64* g = (gu64*)(uintptr_t)p;
  const unsigned long long lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v4f{__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
             __uint_as_float((uint32_t)(hi >> 32))};
}

// finish_rate: per-position epilogue on the rate already scaled to [min_fr, max_fr].
// Neurons.update returns zeros while the agent's position is NaN (reference Neurons.py:163-164)
__device__ __forceinline__ v4f finish_rate(v4f r, const PosQuad& P) {
  r.x = (P.x.x == P.x.x) ? r.x : 0.0f;
  r.y = (P.x.y == P.x.y) ? r.y : 0.0f;
  r.z = (P.x.z == P.x.z) ? r.z : 0.0f;
  r.w = (P.x.w == P.x.w) ? r.w : 0.0f;
  return r;
}
template <class P_>
__device__ __forceinline__ v4f finish_rate(v4f r, const P_&) { return r; }

// ---- spike epilogue: Neurons.save_to_history (reference Neurons.py:681-687) -------------
template <bool EXPLICIT_U, int POLICY = RIAB_STORE_NT>
__device__ __forceinline__ void spike_store(const RateArgs& a, v4f r, int64_t off, uint32_t step, uint32_t c,
                                            uint32_t group) {
  v4f u;
  if (EXPLICIT_U) {
    u = ldv4(a.u_in + off);
  } else {
    const u32x4 w = philox4x32_spikes(step, c, group, a.tag, a.k0, a.k1);
    u = v4f{u01_24(w.x), u01_24(w.y), u01_24(w.z), u01_24(w.w)};
  }
  // one fp32 multiply, one fp32 compare: the exactly-specified spike rule
  const uint32_t s = (u.x < a.dt * r.x ? 1u : 0u) | (u.y < a.dt * r.y ? 0x100u : 0u) |
                     (u.z < a.dt * r.z ? 0x10000u : 0u) | (u.w < a.dt * r.w ? 0x1000000u : 0u);
  store_stream<POLICY>(reinterpret_cast<uint32_t*>(a.spikes + off), s);
}

// ---- PlaceCells (reference Neurons.py:936-981, Environment.py:677-779) -------------------
// GX: 0 euclidean, 1 line_of_sight, 2 geodesic, 3 euclidean + periodic wrap.
template <int DESC, int GX>
struct PlaceCell {
  typedef PosQuad Pos;
  static constexpr int LDS_DOUBLES = (GX == 1 || GX == 2) ? 4 * RIAB_MAX_WALLS + 8 : 1;
  const float* tab;  // [n][3] = (centre x, centre y, k = -log2(e) / (2 w^2))
  float scale, half_scale;  // periodic wrap (Environment.py:670-674)
  float top_hat_w2;
  const double* walls;  // device [n_walls][4]; internal walls = walls[4:] (Environment.py:715-717)
  int n_internal;
  double e0, e1, e2, e3;  // extent, for the geodesic "endpoint inside the env" test
  EnvShape shape;         // (and the boundary polygon / holes for the same test)
  const double* lds;      // set by stage()

  __device__ __forceinline__ void stage(double* s) {
    lds = s;
    if (GX == 1 || GX == 2) {
      for (int i = threadIdx.x; i < 4 * n_internal; i += 256) s[i] = walls[16 + i];
      __syncthreads();
    }
  }
  __device__ __forceinline__ Pos load(const RateArgs& a, int64_t off) const {
    return Pos{ldv4(a.pos_x + off), ldv4(a.pos_y + off)};
  }
  __device__ __forceinline__ Pos load_agent(const RateArgs& a, int64_t off) const {  // rows another kernel is publishing
    return Pos{ldv4_agent(a.pos_x + off), ldv4_agent(a.pos_y + off)};
  }
  // (riab_step1.hip: the same workgroup has just computed these positions and hands them over in LDS)
  __device__ __forceinline__ Pos from_rows(v4f x, v4f y, v4f, v4f) const { return Pos{x, y}; }
  static constexpr bool NEEDS_HD = false;
  static constexpr bool NEEDS_POS = true;
  __device__ __forceinline__ float wrap(float v) const {
    const float av = fabsf(v);
    return (av > half_scale) ? -copysignf(scale - av, v) : v;
  }
  __device__ __forceinline__ float dist2(float cxs, float cys, float px, float py) const {
    float dx = cxs - px, dy = cys - py;
    if (GX == 3) {
      dx = wrap(dx);
      dy = wrap(dy);
    }
    float d2 = fmaf(dy, dy, dx * dx);
    if (GX == 1) {
      bool blocked = false;
      for (int w = 0; w < n_internal; ++w)
        blocked |= seg_hit(cxs, cys, px, py, lds[4 * w], lds[4 * w + 1], lds[4 * w + 2], lds[4 * w + 3]);
      if (blocked) d2 = 1.0e6f;  // distance 1000 (Environment.py:730)
    }
    if (GX == 2) {
      if (n_internal > 0 && seg_hit(cxs, cys, px, py, lds[0], lds[1], lds[2], lds[3])) {
        // Environment.py:744-774: shortest route via a wall endpoint strictly inside the env
        float best = INFINITY;
        bool any = false;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const double exd = lds[2 * e], eyd = lds[2 * e + 1];
          if (env_contains(shape, exd, eyd, [&](int k, double& ax, double& ay, double& bx, double& by) {
                ax = walls[4 * k]; ay = walls[4 * k + 1]; bx = walls[4 * k + 2]; by = walls[4 * k + 3];
              })) {
            const float ex = (float)exd, ey = (float)eyd;
            const float d1 = sqrtf(fmaf(cys - ey, cys - ey, (cxs - ex) * (cxs - ex)));
            const float d2e = sqrtf(fmaf(ey - py, ey - py, (ex - px) * (ex - px)));
            best = fminf(best, d1 + d2e);
            any = true;
          }
        }
        if (any) d2 = best * best;
      }
    }
    return d2;
  }
  __device__ __forceinline__ float fr(float d2, float ks) const {
    if (DESC == RIAB_PC_GAUSSIAN) return __builtin_amdgcn_exp2f(d2 * ks);
    if (DESC == RIAB_PC_GAUSSIAN_THRESHOLD) {
      const float e12 = 0.60653065971263342f;  // exp(-1/2)
      return fmaxf(__builtin_amdgcn_exp2f(d2 * ks) - e12, 0.0f) * (1.0f / (1.0f - e12));
    }
    if (DESC == RIAB_PC_DIFF_OF_GAUSSIANS) {
      const float g1 = __builtin_amdgcn_exp2f(d2 * ks);
      const float g2 = __builtin_amdgcn_exp2f(d2 * ks * (1.0f / 2.25f));
      return (g1 - (1.0f / 2.25f) * g2) * (2.25f / 1.25f);
    }
    if (DESC == RIAB_PC_TOP_HAT) return (d2 < top_hat_w2) ? 1.0f : 0.0f;
    return 0.0f;
  }
  static constexpr int NP = 3;
  static constexpr int CPB = 4;  // cells per lane in the wide kernel
  __device__ __forceinline__ v4f eval(const float* p, const Pos& P) const {
    const float cxs = p[0], cys = p[1], ks = p[2];
    v4f r;
    r.x = fr(dist2(cxs, cys, P.x.x, P.y.x), ks);
    r.y = fr(dist2(cxs, cys, P.x.y, P.y.y), ks);
    r.z = fr(dist2(cxs, cys, P.x.z, P.y.z), ks);
    r.w = fr(dist2(cxs, cys, P.x.w, P.y.w), ks);
    return r;
  }
  template <int D2>
  __host__ PlaceCell<D2, GX> as() const {
    PlaceCell<D2, GX> c;
    c.tab = tab; c.scale = scale; c.half_scale = half_scale; c.top_hat_w2 = top_hat_w2;
    c.walls = walls; c.n_internal = n_internal; c.e0 = e0; c.e1 = e1; c.e2 = e2; c.e3 = e3; c.shape = shape; c.lds = nullptr;
    return c;
  }
};

// ---- GridCells (reference Neurons.py:1172-1236) -------------------------------------------
// phase of cosine i in revolutions: a_i - (x*bx_i + y*by_i); v_cos_f32 takes revolutions.
template <int DESC>
struct GridCell {
  typedef PosQuad Pos;
  static constexpr int LDS_DOUBLES = 1;
  __device__ __forceinline__ void stage(double*) {}
  const float* tab;  // [n][9]
  float f0, inv_1mf0;
  __device__ __forceinline__ Pos load(const RateArgs& a, int64_t off) const {
    return Pos{ldv4(a.pos_x + off), ldv4(a.pos_y + off)};
  }
  __device__ __forceinline__ Pos load_agent(const RateArgs& a, int64_t off) const {
    return Pos{ldv4_agent(a.pos_x + off), ldv4_agent(a.pos_y + off)};
  }
  __device__ __forceinline__ Pos from_rows(v4f x, v4f y, v4f, v4f) const { return Pos{x, y}; }
  static constexpr bool NEEDS_HD = false;
  static constexpr bool NEEDS_POS = true;
  // two agents per instruction: the phase arithmetic and the final affine map are packed fp32
  // (v_pk_mul / v_pk_fma / v_pk_add); v_fract and v_cos stay one per term
  __device__ __forceinline__ v2f two(const float* p, v2f x, v2f y) const {
    v2f s = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const v2f p0 = {p[3 * i], p[3 * i]}, p1 = {p[3 * i + 1], p[3 * i + 1]}, p2 = {p[3 * i + 2], p[3 * i + 2]};
      v2f rev = p0 - __builtin_elementwise_fma(x, p1, y * p2);
      rev.x -= floorf(rev.x);  // v_fract: keep the hardware cosine in its accurate range
      rev.y -= floorf(rev.y);
      s += v2f{__builtin_amdgcn_cosf(rev.x), __builtin_amdgcn_cosf(rev.y)};
    }
    s *= v2f{1.0f / 3.0f, 1.0f / 3.0f};
    if (DESC == RIAB_GC_RECTIFIED) {
      const v2f r = (s - v2f{f0, f0}) * v2f{inv_1mf0, inv_1mf0};
      return v2f{fmaxf(r.x, 0.0f), fmaxf(r.y, 0.0f)};
    }
    return v2f{2.0f / 3.0f, 2.0f / 3.0f} * (s + v2f{0.5f, 0.5f});
  }
  static constexpr int NP = 9;
  static constexpr int CPB = 4;
  __device__ __forceinline__ v4f eval(const float* p, const Pos& P) const {
    const v2f lo = two(p, v2f{P.x.x, P.x.y}, v2f{P.y.x, P.y.y});
    const v2f hi = two(p, v2f{P.x.z, P.x.w}, v2f{P.y.z, P.y.w});
    return v4f{lo.x, lo.y, hi.x, hi.y};
  }
};

// ---- HeadDirectionCells / VelocityCells / SpeedCell ------------------------------------------
// (reference Neurons.py:2466-2483, 2577-2583, 2632-2651; utils.py:231-273, 441-457)
// MODE 0: von Mises of utils.get_angle(head direction).
// MODE 1: the same of the NORMALISED velocity, times |v| / one_sigma_speed after the [min_fr, max_fr] scaling.
// MODE 2: one cell, |v| / one_sigma_speed.
template <int MODE>
struct HDCell {
  struct Pos {
    v4f cs, sn;  // cosine / sine of utils.get_angle(direction) = atan2(y, x + 1e-6)
    v4f speed;   // |v| / one_sigma_speed (MODE >= 1)
  };
  static constexpr int LDS_DOUBLES = 1;
  __device__ __forceinline__ void stage(double*) {}
  const float* tab;  // [n][3] = (cos, sin of the preferred angle, kappa * log2(e))
  float speed_inv;   // 1 / one_sigma_speed
  const double* vx64;  // MODE 1 at the agent: rows RIAB_S_VEL_X / _Y of the float64 state (T = 1), or NULL
  const double* vy64;
  __device__ __forceinline__ Pos load_agent(const RateArgs& a, int64_t off) const {
    return from_dirs(ldv4_agent(a.hd_x + off), ldv4_agent(a.hd_y + off));
  }
  __device__ __forceinline__ Pos from_rows(v4f, v4f, v4f hx, v4f hy) const { return from_dirs(hx, hy); }
  static constexpr bool NEEDS_HD = true;
  static constexpr bool NEEDS_POS = false;
  __device__ __forceinline__ Pos load(const RateArgs& a, int64_t off) const {
    v4f hx, hy;
    if (MODE == 1 && vx64) {
      const double* px = vx64 + off;
      const double* py = vy64 + off;
      hx = v4f{(float)px[0], (float)px[1], (float)px[2], (float)px[3]};
      hy = v4f{(float)py[0], (float)py[1], (float)py[2], (float)py[3]};
    } else {
      hx = ldv4(a.hd_x + off);
      hy = ldv4(a.hd_y + off);
    }
    return from_dirs(hx, hy);
  }
  __device__ __forceinline__ Pos from_dirs(v4f hx, v4f hy) const {
    Pos P;
    if (MODE >= 1) {
      const v4f sp{sqrtf(fmaf(hy.x, hy.x, hx.x * hx.x)), sqrtf(fmaf(hy.y, hy.y, hx.y * hx.y)),
                   sqrtf(fmaf(hy.z, hy.z, hx.z * hx.z)), sqrtf(fmaf(hy.w, hy.w, hx.w * hx.w))};
      P.speed = sp * speed_inv;
      if (MODE == 1) {  // direction = vel / |vel| before the 1e-6 of get_angle
        hx = hx / sp;
        hy = hy / sp;
      }
    } else {
      P.speed = v4f{1.0f, 1.0f, 1.0f, 1.0f};
    }
    // No trigonometry: the tuning only needs cos(angle - preferred) = cos a cos p + sin a sin p, and
    // (cos a, sin a) of a = atan2(y, x + 1e-6) is the vector (x + 1e-6, y) normalised.  (An fp32 atan2 +
    // cos pair costs 2e-5 of relative accuracy in the tails of narrow tunings; this form stays at the
    // rounding of the cosine itself.)
    const v4f bx = hx + 1e-6f;
    const v4f inv = {1.0f / sqrtf(fmaf(hy.x, hy.x, bx.x * bx.x)), 1.0f / sqrtf(fmaf(hy.y, hy.y, bx.y * bx.y)),
                     1.0f / sqrtf(fmaf(hy.z, hy.z, bx.z * bx.z)), 1.0f / sqrtf(fmaf(hy.w, hy.w, bx.w * bx.w))};
    P.cs = bx * inv;
    P.sn = hy * inv;
    return P;
  }
  static constexpr int NP = 3;
  static constexpr int CPB = 16;
  __device__ __forceinline__ v4f eval(const float* p, const Pos& P) const {
    if (MODE == 2) return P.speed;
    const float cp = p[0], sp = p[1], k2 = p[2];
    const v4f e = (P.cs * cp + P.sn * sp - 1.0f) * k2;
    return v4f{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y), __builtin_amdgcn_exp2f(e.z),
               __builtin_amdgcn_exp2f(e.w)};
  }
};

// VelocityCells: HDC_firingrates * speed_scale, AFTER the scaling to [min_fr, max_fr] (Neurons.py:2580-2582)
__device__ __forceinline__ v4f finish_rate(v4f r, const HDCell<1>::Pos& P) { return r * P.speed; }

}  // namespace riab
