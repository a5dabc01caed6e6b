#pragma once
// Agent.update() for B independent agents, T steps fused in one launch (gfx950).
//
// One lane = one agent; the whole recurrent state (12 values) lives in VGPRs for
// the T steps, wall segments are staged once per workgroup in LDS (broadcast
// reads: every lane walks the same wall list), the per-step output is one
// coalesced row per history field.  The kernel is a latency-bound recurrence
// (64 waves at B = 4096), not a bandwidth kernel: per step it moves 8 floats per
// agent.  It runs on its own stream underneath the firing-rate kernels of the
// previous chunk (see ratinabox_amd/Agent.py: simulate()).
//
// Arithmetic is templated on the real type R: R = double reproduces the
// reference's float64 NumPy path to ~1e-12 per step, including the discrete
// decisions (collision yes/no, which wall, boundary clamp); R = float is the
// throughput variant.
#include <cstdlib>
#include "riab_device.h"

// The same contraction rule wherever this header is compiled (riab_agent.hip; riab_plan.hip next to the task
// kernel, which switches contraction off for its own code): the stand-alone and the fused launch must agree
// bit for bit.
#pragma clang fp contract(fast)

#define RIAB_TABLE_QUAL static __device__ const
#include "riab_rayleigh_tables.h"

namespace riab {

typedef float v4f __attribute__((ext_vector_type(4)));

// LDS row strides (in doubles), padded so that lanes reading the same coefficient index of
// different segments spread over the banks
#define RIAB_G_STRIDE (RIAB_G_DEG + 4)
#define RIAB_H_STRIDE (RIAB_H_DEG + 4)

// The speed update of Agent._stochastic_velocity_update (reference Agent.py:302-309 with
// utils.rayleigh_to_normal / normal_to_rayleigh, utils.py:409-421), float64, table-driven:
//   G(t) = Phi^-1(clip(1 - exp(-t^2/2), 1e-6, 1-1e-6)),  H(n) = sqrt(-2 ln(1 - Phi(n))),  t = speed/sigma
// (tools/gen_rayleigh_tables.py; max error 1.6e-16 vs 50-digit arithmetic).  Every lane runs
// the same instruction stream on its own segment's coefficients: no exp/log/erfc/ndtri and
// no divergent range branches.  Arguments of H beyond the table use the library functions.
// lookup (issue the per-lane LDS reads of one segment's row) and evaluation are split so that
// independent work can be placed between them: the wave is alone on its SIMD, nothing else hides
// the LDS latency.
template <int DEG>
struct SegRow {
  double c[DEG + 3];
};
// The row pointer carries its address space: LDS rows are read with ds_read (lgkmcnt), rows of the
// global tables with global_load.  A generic pointer would make these flat loads, whose completion
// is tracked by vmcnt as well — each step would then wait for the previous step's history stores
// to be acknowledged before its polynomial could start.
typedef const __attribute__((address_space(3))) double* lds_cf64_ptr;
typedef const __attribute__((address_space(1))) double* glb_cf64_ptr;
template <int DEG, class Ptr>
__device__ __forceinline__ SegRow<DEG> seg_fetch(Ptr row) {
  SegRow<DEG> r;
#pragma unroll
  for (int i = 0; i < DEG + 3; ++i) r.c[i] = row[i];
  return r;
}
// p(x), x = (arg - mid) * inv_halfwidth, split into even and odd parts p = E(x^2) + x O(x^2): two
// independent Horner recurrences of half the length (dependent-chain depth is what costs here).
template <int DEG>
__device__ __forceinline__ double seg_eval(const SegRow<DEG>& r, double arg) {
#pragma clang fp contract(off)
  const double x = (arg - r.c[0]) * r.c[1];
  const double x2 = x * x;
  constexpr int KE = DEG & ~1, KO = (DEG - 1) | 1;  // highest even / odd power
  double pe = r.c[2 + KE], po = r.c[2 + KO];
#pragma unroll
  for (int k = KE - 2; k >= 0; k -= 2) pe = fma(pe, x2, r.c[2 + k]);
#pragma unroll
  for (int k = KO - 2; k >= 1; k -= 2) po = fma(po, x2, r.c[2 + k]);
  return fma(po, x, pe);
}
__device__ __forceinline__ double clamp_G_arg(double t) {
  t = (t < RIAB_G_TLO) ? RIAB_G_TLO : t;
  return (t > RIAB_G_THI) ? RIAB_G_THI : t;
}
__device__ __forceinline__ int seg_G(double t_clamped) {
  return (int)(((unsigned long long)__double_as_longlong(t_clamped) >> 49) - RIAB_G_KEY0);
}
__device__ __forceinline__ int seg_H(double n) {
  const int seg = (int)((n + RIAB_H_NMAX) * RIAB_H_INV_SEG);
  return seg < 0 ? 0 : (seg > RIAB_H_SEGS - 1 ? RIAB_H_SEGS - 1 : seg);
}

struct AgentArgs {
  RiabMotion m;
  double e0, e1, e2, e3;  // extent
  double scale;
  int periodic;
  int n_walls;
  const double* walls;  // device [n_walls][4]
  double* state;        // [12][B]
  int64_t B;
  int64_t agent_id0;
  const double* drift;  // [2][B] or null
  const double* z_in;   // [T][2][B] or null
  double* z_out;        // [T][2][B] or null
  const double* forced; // [T][2][B] or null: imported / forced positions (Agent.py:229-238)
  const double* resample;  // [T][2][B] or null: explicit replacement positions of the resample boundary condition
  EnvShape shape;          // boundary polygon / holes (Environment.py:781-818)
  uint32_t k0, k1;
  uint64_t step0;
  int T;
  float* hist;  // [T][8][B] or null
  int* diag;
  uint32_t* ctrl;  // PUB kernels only: control words of the flag-coupled pipeline (riab_hip.h RIAB_CTRL_*)
  int pub_single;  // PUB kernels only: the launch's first `pub_single` rows are published one by one, then blocks of four
};

// ---- math wrappers ---------------------------------------------------------------------------
__device__ __forceinline__ double r_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float r_sqrt(float x) { return sqrtf(x); }
// 1/sqrt(x) for positive finite x: the hardware estimate (v_rsq_f64, ~26 bits) and one third-order
// correction — the library routine's arithmetic without its special-case selects (x = 0 or inf
// only occur on paths whose result is discarded or replaced, see the call sites).
__device__ __forceinline__ double r_rsqrt(double x) {
#pragma clang fp contract(off)
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}
__device__ __forceinline__ float r_rsqrt(float x) { return rsqrtf(x); }
// sqrt(x) for x >= 0 in the normal range (squared distances, 1 - t^2 of the conveyor term): the hardware rsq estimate,
// one coupled Goldschmidt step and one residual correction — 9 instructions, no range scaling or special-case selects
// (the library routine: ~20, three of them per step on the stepping wave).  Within 1 ulp; exact for exact squares'
// neighbours 0 and 1 (x = 0 gives 1e-154, not 0: only reached by an agent exactly ON a wall).
__device__ __forceinline__ double r_sqrt_fast(double x) {
#pragma clang fp contract(off)
  x = fmax(x, 0x1.0p-1000);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, x);
  return fma(d, h, g);
}
__device__ __forceinline__ float r_sqrt_fast(float x) { return sqrtf(x); }
__device__ __forceinline__ double r_exp(double x) { return exp(x); }
__device__ __forceinline__ float r_exp(float x) { return expf(x); }
__device__ __forceinline__ double r_log(double x) { return log(x); }
__device__ __forceinline__ float r_log(float x) { return logf(x); }
__device__ __forceinline__ double r_ndtri(double u) { return normcdfinv(u); }
__device__ __forceinline__ float r_ndtri(float u) { return normcdfinvf(u); }
__device__ __forceinline__ double r_ndtr(double x) { return normcdf(x); }
__device__ __forceinline__ float r_ndtr(float x) { return normcdff(x); }

// clamp to [0, 1] / minimum as single v_max / v_min instructions.  A NaN argument (NaN position)
// comes out as a number where the reference's comparisons keep the NaN; the position stays NaN
// through `px - (...)` regardless, so nothing downstream differs.
__device__ __forceinline__ double r_clamp01(double l) { return fmin(fmax(l, 0.0), 1.0); }
__device__ __forceinline__ float r_clamp01(float l) { return fminf(fmaxf(l, 0.0f), 1.0f); }
__device__ __forceinline__ double r_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float r_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double r_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float r_max(float a, float b) { return fmaxf(a, b); }

// ---- arithmetic of one step, written once -------------------------------------------------------------------------
// Every trajectory kernel (one wave per 64 agents: agent_step_body; four specialised waves: riab_traj4_kernel.h; the
// fused motion + task launch) evaluates a step through the functions below and must produce the SAME BITS whatever
// code surrounds the call: with `fp contract(fast)` the compiler decides per basic block which multiply feeds which
// add, so the same expression could round differently in two kernels (it did: 1e-14 after 100 steps).  Here
// contraction is OFF and every fused multiply-add is spelled out.
#define RIAB_EXACT_FP _Pragma("clang fp contract(off)")

__device__ __forceinline__ double r_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float r_fma(float a, float b, float c) { return fmaf(a, b, c); }

template <class R>
__device__ __forceinline__ R norm2(R x, R y) {
  RIAB_EXACT_FP
  return r_fma(x, x, y * y);
}
// utils.ornstein_uhlenbeck (utils.py:347-368): x + theta * (drift - x) * dt + sigma * (dt * z)
// (sigma = sqrt(2 noise_scale^2 / (coherence_time * dt)) is prepared by the host)
template <class R>
__device__ __forceinline__ R ou_step(R x, R theta, R drift, R sigma, R dt, R z) {
  RIAB_EXACT_FP
  const R pull = theta * (drift - x);
  const R kick = sigma * (dt * z);
  return x + r_fma(pull, dt, kick);
}
// utils.rotate (utils.py:293-301) by the angle whose (cos, sin) are given
template <class R>
__device__ __forceinline__ void rotate_by(R cs, R sn, R& vx, R& vy) {
  RIAB_EXACT_FP
  const R nx = r_fma(cs, vx, (-sn) * vy);
  const R ny = r_fma(sn, vx, cs * vy);
  vx = nx;
  vy = ny;
}
// Agent._drift_velocity_update (Agent.py:331-341): an OU step towards the drift velocity without noise
template <class R>
__device__ __forceinline__ void drift_update(R theta, R drx, R dry, R dt, R& vx, R& vy) {
  RIAB_EXACT_FP
  vx = r_fma(theta * (drx - vx), dt, vx);
  vy = r_fma(theta * (dry - vy), dt, vy);
}
// pos += velocity * dt (Agent.py:216)
template <class R>
__device__ __forceinline__ void propose_step(R vx, R vy, R dt, R& px, R& py) {
  RIAB_EXACT_FP
  px = r_fma(vx, dt, px);
  py = r_fma(vy, dt, py);
}
// the displacement of the step as Agent._measure_velocity_of_step_taken sees it (Agent.py:456-458,
// Environment.get_vectors_between___accounting_for_environment, Environment.py:657-675: periodic-aware)
template <class R>
__device__ __forceinline__ void step_displacement(const AgentArgs& a, R px, R py, R ppx, R ppy, R& dpx, R& dpy) {
  RIAB_EXACT_FP
  dpx = px - ppx;
  dpy = py - ppy;
  if (a.periodic) {
    const R sc = (R)a.scale, hs = (R)(a.scale / 2);
    if (fabs(dpx) > hs) dpx = -copysign(sc - fabs(dpx), dpx);
    if (fabs(dpy) > hs) dpy = -copysign(sc - fabs(dpy), dpy);
  }
}

// atan2 in fp32 for the measured rotational velocity (an output: it does not feed back into the
// motion): Cephes-style argument reduction to [0, tan(pi/8)] and a degree-9 odd polynomial (~2 ulp),
// about half the instructions of the library routine, which also handles infinities.
__device__ __forceinline__ float atan2_fast(float y, float x) {
  RIAB_EXACT_FP
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float t = mn * __builtin_amdgcn_rcpf(mx);          // in [0, 1]; 0/0 -> NaN, fixed below
  const bool hi = t > 0.4142135623730950f;            // tan(pi/8)
  const float tr = (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f);
  t = hi ? tr : t;
  const float z = t * t;
  float p = 8.05374449538e-2f;
  p = fmaf(p, z, -1.38776856032e-1f);
  p = fmaf(p, z, 1.99777106478e-1f);
  p = fmaf(p, z, -3.33329491539e-1f);
  float r = fmaf(p * z, t, t);
  r = hi ? r + 0.78539816339744831f : r;
  r = (ay > ax) ? 1.57079632679489662f - r : r;
  r = (x < 0.0f) ? 3.14159265358979324f - r : r;
  r = (mx == 0.0f) ? 0.0f : r;
  return copysignf(r, y);
}

// sin/cos of the per-step heading increment rot*dt (|x| is a few 1e-2): Taylor in x^2, the library
// routine for |x| >= 0.5.
__device__ __forceinline__ void sincos_small(double x, double* s, double* c) {
  RIAB_EXACT_FP
  // the branch is wave-uniform (ballot): a wave runs exactly one of the three variants.  Each Taylor
  // coefficient costs two instructions here (the loop-invariant constant has to be copied into the
  // accumulator of a v_fmac), so the common case gets the shortest series that is exact to 1e-17.
  const double ax = fabs(x);
  if (__builtin_amdgcn_ballot_w64(ax >= 0.125) == 0) {  // |x| < 1/8: truncation < 3e-17 (sin), 3e-20 (cos)
    const double x2 = x * x;
    double sp = 1.0 / 362880.0;                        //  1/9!
    sp = fma(sp, x2, -1.0 / 5040.0);                   // -1/7!
    sp = fma(sp, x2, 1.0 / 120.0);                     //  1/5!
    sp = fma(sp, x2, -1.0 / 6.0);                      // -1/3!
    *s = fma(sp * x2, x, x);
    double cp = -1.0 / 3628800.0;                      // -1/10!
    cp = fma(cp, x2, 1.0 / 40320.0);                   //  1/8!
    cp = fma(cp, x2, -1.0 / 720.0);                    // -1/6!
    cp = fma(cp, x2, 1.0 / 24.0);                      //  1/4!
    cp = fma(cp, x2, -0.5);
    *c = fma(cp, x2, 1.0);
  } else if (__builtin_amdgcn_ballot_w64(ax >= 0.5) == 0) {
    const double x2 = x * x;
    double sp = -1.0 / 1307674368000.0;                // -1/15!
    sp = fma(sp, x2, 1.0 / 6227020800.0);              //  1/13!
    sp = fma(sp, x2, -1.0 / 39916800.0);               // -1/11!
    sp = fma(sp, x2, 1.0 / 362880.0);                  //  1/9!
    sp = fma(sp, x2, -1.0 / 5040.0);                   // -1/7!
    sp = fma(sp, x2, 1.0 / 120.0);                     //  1/5!
    sp = fma(sp, x2, -1.0 / 6.0);                      // -1/3!
    *s = fma(sp * x2, x, x);
    double cp = 1.0 / 20922789888000.0;                //  1/16!
    cp = fma(cp, x2, -1.0 / 87178291200.0);            // -1/14!
    cp = fma(cp, x2, 1.0 / 479001600.0);               //  1/12!
    cp = fma(cp, x2, -1.0 / 3628800.0);                // -1/10!
    cp = fma(cp, x2, 1.0 / 40320.0);                   //  1/8!
    cp = fma(cp, x2, -1.0 / 720.0);                    // -1/6!
    cp = fma(cp, x2, 1.0 / 24.0);                      //  1/4!
    cp = fma(cp, x2, -0.5);
    *c = fma(cp, x2, 1.0);
  } else {
    sincos(x, s, c);
  }
}
__device__ __forceinline__ void sincos_small(float x, float* s, float* c) { sincosf(x, s, c); }

template <class R>
struct Wall {  // staged in LDS
  R ax, ay, sx, sy;  // start point and direction (b - a)
  R inv_ss;          // 1 / |s|^2
  R inv_len;         // 1 / |s|
};

// IN: 0 = in-kernel Philox noise, 1 = explicit normals z_in, 2 = forced positions.  The Philox
// variant has NO global load inside the step loop, so no s_waitcnt vmcnt(0) ever makes a step
// wait for the previous step's history stores to land (measured: 37 % of the wave's cycles were
// spent in such waits when all modes shared one kernel).
//
// PC (helper wave, Philox mode only): the workgroup has a SECOND wave that takes everything off the stepping
// wave that is not part of the recurrence:
//   * it draws the normals — Philox + Box-Muller do not depend on the state — a batch of RIAB_Z_BATCH steps
//     ahead into a double-buffered LDS tile (the stepping wave reads two floats per step);
//   * it computes the output-only tail of every step (step_tail: measured velocities, head direction, distance)
//     from the displacement the stepping wave hands over in LDS, and owns those state rows;
//   * it writes the history rows to HBM, so the stepping wave never issues a global store inside the step
//     loop and never queues behind the rate kernels' store stream.
// The waves meet at one workgroup barrier per four steps.  Values are bit-identical to the single-wave
// kernel (the same inlined functions on the same operands).
//
// PUB (with PC): the trajectory is consumed by a firing-rate kernel that runs CONCURRENTLY (riab_simulate_fused,
// rate_kernel_gated / stream_gate_kernel in riab_rates.hip).  The helper wave then writes the history rows write-through (agent-scope
// `sc1` stores: the consumer sits on other CUs / XCDs whose L2s are not coherent with this one), drains them
// (`s_waitcnt vmcnt(0)`) and publishes "steps done" in ctrl[RIAB_CTRL_PROGRESS + workgroup] with one relaxed
// agent-scope store per four-step block.  Values are bit-identical to the other variants.
#define RIAB_Z_BATCH 16

typedef __attribute__((address_space(1))) unsigned long long riab_gu64;
typedef __attribute__((address_space(1))) uint32_t riab_gu32;
// 16 bytes write-through (two 8-byte agent-scope relaxed stores = global_store_dwordx2 ... sc1)
__device__ __forceinline__ void store_v4f_agent(void* p, v4f v) {
  riab_gu64* g = (riab_gu64*)(uintptr_t)p;
  const unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
  const unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
  __hip_atomic_store(g, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the two standard normals of `step` for agent `aid` as floats; `pw` carries the Philox block that
// serves an (even, odd) pair of steps
// The part of a step that only produces OUTPUTS (measured velocity, measured rotational velocity, head
// direction, distance travelled: Agent.py:456-507): nothing in it feeds the next step's motion, so the helper
// wave can run it (PC variant) — from the step's displacement alone.
template <class R>
struct StepTail {
  R mvx, mvy, mrot, hx, hy, dist;
  int n_still;
};
template <class R>
struct TailConst {
  R dt, inv_dt, hd_keep, hd_gain;
  bool hd_instant;
};
template <class R>
__device__ __forceinline__ StepTail<R> step_tail(StepTail<R> s, R dpx, R dpy, const TailConst<R> c, uint64_t stp,
                                                 uint32_t aid, uint32_t k0, uint32_t k1) {
  RIAB_EXACT_FP
  // ---- _measure_velocity_of_step_taken (Agent.py:456-471) -------------------------------
  const R pmvx = s.mvx, pmvy = s.mvy;  // prev_measured_velocity (Agent.py:201)
  R mvx = dpx * c.inv_dt;
  R mvy = dpy * c.inv_dt;
  R dp2 = norm2(dpx, dpy);
  R idp = r_rsqrt(dp2);          // one reciprocal square root serves |d_pos|, |mv| and 1/|mv|
  R dstep = dp2 * idp;
  R imv = idp * c.dt;            // 1 / |mv|
  if (dp2 == (R)0) {
    // 1e-8 * randn(2) (Agent.py:459-460): never reached in practice; its own Philox stream
    const u32x4 zw = philox4x32_10((uint32_t)stp, (uint32_t)(stp >> 32), aid, RIAB_TAG_MOTION ^ 1u, k0, k1);
    const float u3 = ((float)(zw.x >> 8) + 0.5f) * 0x1.0p-24f, u4 = (float)(zw.y >> 8) * 0x1.0p-24f;
    const float r2 = sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u3));
    mvx = (R)1e-8 * (R)(r2 * __builtin_amdgcn_cosf(u4));
    mvy = (R)1e-8 * (R)(r2 * __builtin_amdgcn_sinf(u4));
    imv = r_rsqrt(norm2(mvx, mvy));
    dstep = (R)0;
    ++s.n_still;
  }
  {
    // measured rotational velocity (Agent.py:465-468): pi_domain(get_angle(mv) - get_angle(prev_mv)) / dt.
    // The wrapped difference of the two angles IS the signed angle between the two vectors
    // (x + 1e-6 is utils.get_angle's quirk), taken directly from their cross / dot products;
    // the arctangent runs in fp32 on that DIFFERENCE (relative error 1e-7 of a small angle;
    // the quantity is an output, it does not feed back into the motion).
    const R ax_ = pmvx + (R)1e-6, bx_ = mvx + (R)1e-6;
    const R crs = r_fma(ax_, mvy, -(pmvy * bx_)), dotp = r_fma(ax_, bx_, pmvy * mvy);
    s.mrot = (R)atan2_fast((float)crs, (float)dotp) * c.inv_dt;
  }
  s.mvx = mvx;
  s.mvy = mvy;
  // ---- _update_head_direction (Agent.py:488-500) ----------------------------------------
  {
    const R ix = mvx * imv, iy = mvy * imv;
    if (c.hd_instant) {
      s.hx = ix;
      s.hy = iy;
    } else {
      const R nx = r_fma(s.hx, c.hd_keep, c.hd_gain * ix);
      const R ny = r_fma(s.hy, c.hd_keep, c.hd_gain * iy);
      const R inn = r_rsqrt(norm2(nx, ny));
      s.hx = nx * inn;
      s.hy = ny * inn;
    }
  }
  // ---- _update_distance_travelled (Agent.py:507) ----------------------------------------
  s.dist += dstep;
  return s;
}

struct MotionDraw {
  u32x4 pw;
  float z_rot, z_spd;
};
__device__ __forceinline__ MotionDraw motion_normals(uint64_t step, bool first, uint32_t aid, uint32_t k0, uint32_t k1,
                                                     u32x4 pw) {
  RIAB_EXACT_FP
  // one Philox4x32-10 call serves TWO steps: counter = step >> 1, words (x, y) on even steps
  // and (z, w) on odd ones
  if (first || (step & 1) == 0) {
    const uint64_t pair = step >> 1;
    // (the key is made opaque per call: otherwise the ten round keys are hoisted out of the step
    // loop as 20 loop-invariant SGPRs, spilled to VGPR lanes, and read back with a v_readlane each)
    uint32_t kk0 = k0, kk1 = k1;
    asm volatile("" : "+s"(kk0), "+s"(kk1));
    pw = philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), aid, RIAB_TAG_MOTION, kk0, kk1);
  }
  const uint32_t wa = (step & 1) ? pw.z : pw.x, wb = (step & 1) ? pw.w : pw.y;
  // Box-Muller on the Philox words with the hardware log2 / sin / cos (the draws only need
  // to be N(0,1) and a pure function of (seed, step, agent); `z_out` records them).
  const float u1 = ((float)(wa >> 8) + 0.5f) * 0x1.0p-24f, u2 = (float)(wb >> 8) * 0x1.0p-24f;
  const float rr = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1), hardware sqrt
  return MotionDraw{pw, rr * __builtin_amdgcn_cosf(u2), rr * __builtin_amdgcn_sinf(u2)};
}

// ---- the pieces of one step that both trajectory kernels share ------------------------------------------------
// (agent_step_body below: one wave does everything; riab_traj4_kernel.h: the same functions spread over four
// specialised waves.  Same inlined code on the same operands: the two kernels agree bit for bit.)
template <class R>
struct MotionConst {
  R dt, inv_dt, sm_kw, sm, inv_2s2, wd, v0, kspring, inv_wd2, cvel, cpos, wd2;
  double inv_sm;
  R e0, e1, e2, e3;
  bool repel, simple_box, box_fast;
  R bxl, bxr, byb, byt;  // box fast path: the room's edges
  int nw;
  Wall<R> w4[4];         // the first four walls (the box itself when boundaries are solid), in registers
  // the broad phase of the wall loops (RiabMotion.wall_grid), staged in LDS by the kernel; null: every wall, every step
  const uint64_t* grid;
  int gn;
  R g_ix, g_iy, g_lmax2;
};
#define RIAB_WALL_GRID_WORDS (2 * RIAB_WALL_GRID_MAX * RIAB_WALL_GRID_MAX)

// RiabMotion.wall_grid -> LDS (4 KB), by `nthreads` threads; the caller's barrier follows
__device__ __forceinline__ void stage_wall_grid(const AgentArgs& a, uint64_t* s_grid, int tid, int nthreads) {
  if (!a.m.wall_grid) return;
  const int n = 2 * a.m.wall_grid_n * a.m.wall_grid_n;
  for (int i = tid; i < n && i < RIAB_WALL_GRID_WORDS; i += nthreads) s_grid[i] = a.m.wall_grid[i];
}
// ... and into the step's constants (after motion_const_scalars): used only where it is valid for this launch — float64
// steps, a repel distance the masks were built for, a solid (non-periodic-wrapping is fine: walls do not wrap) extent
template <class R>
__device__ __forceinline__ void motion_const_grid(MotionConst<R>& k, const AgentArgs& a, const uint64_t* s_grid) {
  const bool ok = sizeof(R) == 8 && s_grid && a.m.wall_grid && a.m.wall_grid_n > 0 && a.m.wall_grid_n <= RIAB_WALL_GRID_MAX &&
                  (double)k.wd <= a.m.wall_grid_wd && k.nw > 4;  // (the masks' margin covers the near test's own: wd2 = wd^2 * 1.000001)
  k.grid = ok ? s_grid : nullptr;
  k.gn = a.m.wall_grid_n;
  k.g_ix = (R)((double)a.m.wall_grid_n / (a.e1 - a.e0));
  k.g_iy = (R)((double)a.m.wall_grid_n / (a.e3 - a.e2));
  k.g_lmax2 = (R)(a.m.wall_grid_lmax * a.m.wall_grid_lmax);
}
// the walls cell (px, py)'s mask `which` names (0: nearest / within the repel distance, 1: within a short step); a point
// outside the extent (or NaN): every wall
template <class R>
__device__ __forceinline__ uint64_t wall_grid_mask(const MotionConst<R>& k, R px, R py, int which) {
  const uint64_t all = k.nw >= 64 ? ~0ull : ((1ull << k.nw) - 1ull);
  const bool inside = px >= k.e0 && px <= k.e1 && py >= k.e2 && py <= k.e3;
  int ix = (int)((px - k.e0) * k.g_ix), iy = (int)((py - k.e2) * k.g_iy);
  ix = ix < 0 ? 0 : (ix >= k.gn ? k.gn - 1 : ix);
  iy = iy < 0 ? 0 : (iy >= k.gn ? k.gn - 1 : iy);
  const uint64_t m = k.grid[2 * (iy * k.gn + ix) + which];
  return inside ? (m & all) : all;
}

// The constants of a launch in two parts: the scalars, functions of the motion parameters and the extent alone — also
// evaluated on the HOST for the one-launch step (riab_step1.hip), where a per-thread evaluation would be paid on every
// step: plain IEEE double arithmetic without a multiply-add in it, the same bits on either side —, and what is read off
// the staged wall table.
template <class R>
__host__ __device__ __forceinline__ void motion_const_scalars(MotionConst<R>& k, const AgentArgs& a) {
  const RiabMotion& m = a.m;
  k.nw = a.n_walls;
  k.dt = (R)m.dt;
  k.sm_kw = (R)m.speed_mean_kw;
  k.sm = (R)m.speed_mean;
  k.inv_2s2 = (R)1 / ((R)2 * k.sm_kw * k.sm_kw);
  k.inv_sm = 1.0 / m.speed_mean_kw;
  k.wd = (R)m.wall_repel_distance_kw;
  k.v0 = (R)m.wall_repel_strength_kw * k.sm;
  k.kspring = (k.v0 * k.v0) / (k.wd * k.wd);
  k.inv_wd2 = (R)1 / (k.wd * k.wd);
  const R g = (R)m.thigmotaxis_kw;
  k.cvel = (R)3 * (((R)1 - g) * ((R)1 - g));
  k.cpos = (R)6 * (g * g);
  k.repel = (m.wall_repel_strength_kw != 0.0) && k.nw > 0;
  k.e0 = (R)a.e0; k.e1 = (R)a.e1; k.e2 = (R)a.e2; k.e3 = (R)a.e3;
  k.simple_box = !a.shape.boundary_mask && !a.shape.hole_mask;  // (the rectangle test alone decides)
  // divisions by loop constants become multiplications (<= 1 ulp from the reference's quotient)
  k.inv_dt = (R)(1.0 / m.dt);
  k.wd2 = k.wd * k.wd * (R)1.000001;
  k.grid = nullptr;
  k.gn = 0;
  k.g_ix = k.g_iy = k.g_lmax2 = (R)0;
}

template <class R>
__device__ __forceinline__ void motion_const_walls(MotionConst<R>& k, const AgentArgs& a, const Wall<R>* s_w) {
#pragma unroll
  for (int w = 0; w < 4; ++w) k.w4[w] = s_w[w < k.nw ? w : 0];
  // Box fast path.  In a solid rectangular room the first four walls are the room's own edges (Environment.py:128-163)
  // and every agent is inside them, so for those four the general point-to-segment arithmetic collapses: the nearest
  // point of an axis-aligned edge that spans the room is the foot of the perpendicular, the distance is a coordinate
  // difference, the unit normal is +-x or +-y, and of two opposite edges at most one is within the repel distance
  // (room wider than twice that distance).  Same quantities as the general formulas below up to their own rounding
  // (those carry ~1e-16 residues in the components that are exactly 0 / 1 here).  Checked once per launch, on the
  // wall table itself; any other geometry takes the general path.
  k.box_fast = false;
  k.bxl = 0; k.bxr = 0; k.byb = 0; k.byt = 0;
  if (k.nw >= 4 && k.simple_box && !a.periodic) {
    int nh = 0, nv = 0;
    R xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    bool ok = true;
    const R tol = (R)1e-9 * ((k.e1 - k.e0) + (k.e3 - k.e2));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const Wall<R> W = k.w4[w];
      if (W.sy == (R)0 && W.sx != (R)0) {  // horizontal edge: spans [e0, e1]
        ++nh;
        ylo = r_min(ylo, W.ay);
        yhi = r_max(yhi, W.ay);
        ok = ok && r_min(W.ax, W.ax + W.sx) <= k.e0 + tol && r_max(W.ax, W.ax + W.sx) >= k.e1 - tol;
      } else if (W.sx == (R)0 && W.sy != (R)0) {  // vertical edge: spans [e2, e3]
        ++nv;
        xlo = r_min(xlo, W.ax);
        xhi = r_max(xhi, W.ax);
        ok = ok && r_min(W.ay, W.ay + W.sy) <= k.e2 + tol && r_max(W.ay, W.ay + W.sy) >= k.e3 - tol;
      } else {
        ok = false;
      }
    }
    ok = ok && nh == 2 && nv == 2 && xlo == k.e0 && xhi == k.e1 && ylo == k.e2 && yhi == k.e3;
    ok = ok && (k.e1 - k.e0) > (R)2.01 * k.wd && (k.e3 - k.e2) > (R)2.01 * k.wd;
    k.box_fast = __builtin_amdgcn_readfirstlane((int)ok) != 0;  // (the same value in every lane: a scalar branch)
    k.bxl = xlo; k.bxr = xhi; k.byb = ylo; k.byt = yhi;
  }
}

template <class R>
__device__ __forceinline__ MotionConst<R> make_motion_const(const AgentArgs& a, const Wall<R>* s_w, const uint64_t* s_grid = nullptr) {
  MotionConst<R> k;
  motion_const_scalars<R>(k, a);
  motion_const_walls<R>(k, a, s_w);
  motion_const_grid<R>(k, a, s_grid);
  return k;
}

// ---- _wall_velocity_update, pass 1 (Agent.py:357-415, utils.py:121-184) ---------------
// squared distances first: the sqrt / normalisation only for walls inside the repel
// distance, and ONE sqrt for distance_to_closest_wall (sqrt is monotone: same value)
template <class R>
struct NearWalls {
  R x2min;
  R box_tx, box_ty, box_nx, box_ny;  // box fast path: (repel distance - distance)+ and normal per axis
  R box_dmin;                        // box fast path: signed distance to the nearest of the four edges
  uint64_t near_mask;                // bit w: wall w is within the repel distance (pass 2 walks the set bits in order)
};
// the four edges of a solid rectangular room (box fast path, see make_motion_const): distance = coordinate difference;
// of two opposite edges only the nearer one can repel
template <class R>
__device__ __forceinline__ void box_pass1(const MotionConst<R>& k, R px, R py, NearWalls<R>& n) {
  RIAB_EXACT_FP
  const R dl = px - k.bxl, dr = k.bxr - px, db = py - k.byb, dtp = k.byt - py;
  const R dxm = r_min(dl, dr), dym = r_min(db, dtp);
  n.box_tx = r_max(k.wd - fabs(dxm), (R)0);
  n.box_ty = r_max(k.wd - fabs(dym), (R)0);
  // normal = from the edge to the agent: into the room for an agent inside it (the other sign only for a
  // position handed in from outside the room, where the general formula points outwards as well)
  n.box_nx = ((dl < dr) == (dxm >= (R)0)) ? (R)1 : (R)-1;
  n.box_ny = ((db < dtp) == (dym >= (R)0)) ? (R)1 : (R)-1;
  const R dm = r_min(dxm, dym);
  n.box_dmin = dm;
  n.x2min = dm * dm;
}
// spring + conveyor of the nearer vertical and the nearer horizontal edge: exact zeros beyond the repel distance
// (t = 0), so the sums are the reference's sums over the four edges
template <class R>
struct WallPush {
  R ax, ay, sx, sy;
};
template <class R>
__device__ __forceinline__ WallPush<R> box_pass2_terms(const MotionConst<R>& k, const NearWalls<R>& n) {
  RIAB_EXACT_FP
  const R spx = k.v0 * ((R)1 - r_sqrt_fast(r_fma(-(n.box_tx * n.box_tx), k.inv_wd2, (R)1)));
  const R spy = k.v0 * ((R)1 - r_sqrt_fast(r_fma(-(n.box_ty * n.box_ty), k.inv_wd2, (R)1)));
  return WallPush<R>{(k.kspring * n.box_tx) * n.box_nx, (k.kspring * n.box_ty) * n.box_ny, spx * n.box_nx, spy * n.box_ny};
}
// Agent.distance_to_closest_wall (Agent.py:415): the open solid box knows it exactly (|nearest coordinate difference|)
template <class R>
__device__ __forceinline__ R closest_wall_distance(const MotionConst<R>& k, const NearWalls<R>& n) {
  return (k.box_fast && k.nw == 4) ? (R)fabs(n.box_dmin) : r_sqrt_fast(n.x2min);
}
// the rectangle's own safety net (Environment.py:871-885, solid boundaries): clamp to [min + 0.01, max - 0.01]
template <class R>
__device__ __forceinline__ void box_clamp(const MotionConst<R>& k, R& px, R& py) {
  RIAB_EXACT_FP
  const R lo_x = k.e0 + (R)0.01, hi_x = k.e1 - (R)0.01, lo_y = k.e2 + (R)0.01, hi_y = k.e3 - (R)0.01;
  px = (lo_x > px) ? lo_x : px;  // python max(pos, lo): NaN stays NaN
  px = (hi_x < px) ? hi_x : px;
  py = (lo_y > py) ? lo_y : py;
  py = (hi_y < py) ? hi_y : py;
}
template <class R>
__device__ __forceinline__ NearWalls<R> walls_pass1(const MotionConst<R>& k, const Wall<R>* s_w, R px, R py) {
  RIAB_EXACT_FP
  NearWalls<R> n;
  n.x2min = INFINITY;
  n.box_tx = 0; n.box_ty = 0; n.box_nx = 0; n.box_ny = 0; n.box_dmin = 0;
  n.near_mask = 0;
  const int nw = k.nw;
  if (nw > 0) {
    // pass 1 (cheap, every wall): squared distance to the nearest point of the wall; remember the
    // walls inside the repel distance.  pass 2 (expensive: sqrt, 1/x, the spring / conveyor terms)
    // runs only over those, in wall order, so the sums are the reference's sums (the skipped terms
    // are exact zeros).  Instruction count matters here (one wave per SIMD issues an fp64
    // instruction every 7.5 cycles): the clamps are v_min / v_max, the near set is a bit mask.
    auto pass1 = [&](const Wall<R>& W, int w) {
      RIAB_EXACT_FP
      const R dxw = px - W.ax, dyw = py - W.ay;
      R l = r_fma(dxw, W.sx, dyw * W.sy) * W.inv_ss;
      l = r_clamp01(l);
      const R qx = px - r_fma(l, W.sx, W.ax), qy = py - r_fma(l, W.sy, W.ay);
      const R x2 = norm2(qx, qy);
      n.x2min = r_min(x2, n.x2min);
      n.near_mask |= (uint64_t)(x2 <= k.wd2) << w;
    };
    // the first four walls (the box itself when boundaries are solid) live in registers for the
    // whole launch: no LDS round trip per step for the common open-box case
    if (k.grid) {
      // Wall-heavy rooms (RiabMotion.wall_grid): only the walls that can be the nearest one of, or within the repel
      // distance of, a point of this lane's cell — a lane walks the set bits of ITS mask in ascending order (its wall from
      // LDS by a per-lane index); the minimum and the near set are those of the full loop, bit for bit (a minimum and a
      // union do not depend on the order, and the walls left out can contribute to neither).
      uint64_t m = wall_grid_mask<R>(k, px, py, 0);
      if (k.box_fast) {
        box_pass1<R>(k, px, py, n);
        m &= ~0xFull;
      }
      while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
        if (m) {
          const int w = __ffsll((long long)m) - 1;
          m &= m - 1;
          pass1(s_w[w], w);
        }
      }
    } else if (k.box_fast) {
      box_pass1<R>(k, px, py, n);
      for (int w = 4; w < nw; ++w) pass1(s_w[w], w);
    } else if (nw >= 4) {  // one uniform test instead of four (each a spilled 64-bit mask read back per step)
#pragma unroll
      for (int w = 0; w < 4; ++w) pass1(k.w4[w], w);
      for (int w = 4; w < nw; ++w) pass1(s_w[w], w);
    } else {
      for (int w = 0; w < nw; ++w) pass1(s_w[w], w);
    }
  }
  return n;
}

// ---- _wall_velocity_update, pass 2: the spring acceleration and the conveyor speed summed over the near walls
// (they depend on the position only), and their application to velocity and position
template <class R>
__device__ __forceinline__ WallPush<R> walls_pass2_terms(const MotionConst<R>& k, const Wall<R>* s_w, const NearWalls<R>& n,
                                                         R px, R py) {
  RIAB_EXACT_FP
  R ax_ = 0, ay_ = 0, sx_ = 0, sy_ = 0;
  const R wd = k.wd, v0 = k.v0, kspring = k.kspring, inv_wd2 = k.inv_wd2;
  if (k.box_fast) {
    const WallPush<R> bx = box_pass2_terms<R>(k, n);
    ax_ = bx.ax;
    ay_ = bx.ay;
    sx_ = bx.sx;
    sy_ = bx.sy;
  }
  for (uint64_t rest = n.near_mask; rest; rest &= rest - 1) {
    const int w = __ffsll((long long)rest) - 1;
    const Wall<R> W = s_w[w];
    const R dxw = px - W.ax, dyw = py - W.ay;
    R l = r_fma(dxw, W.sx, dyw * W.sy) * W.inv_ss;
    l = r_clamp01(l);
    const R qx = px - r_fma(l, W.sx, W.ax), qy = py - r_fma(l, W.sy, W.ay);
    const R xx = norm2(qx, qy);
    const R ix = r_rsqrt(xx);  // 1/x and x from one reciprocal square root
    const R x = xx * ix;
    if (x <= wd) {
      const R nx = qx * ix, ny = qy * ix;
      const R acc = kspring * (wd - x);
      const R spd = v0 * ((R)1 - r_sqrt_fast(r_fma(-((wd - x) * (wd - x)), inv_wd2, (R)1)));
      ax_ = r_fma(acc, nx, ax_);
      ay_ = r_fma(acc, ny, ay_);
      sx_ = r_fma(spd, nx, sx_);
      sy_ = r_fma(spd, ny, sy_);
    }
  }
  return WallPush<R>{ax_, ay_, sx_, sy_};
}
template <class R>
__device__ __forceinline__ void walls_pass2_apply(const MotionConst<R>& k, const WallPush<R>& w, R& px, R& py, R& vx, R& vy) {
  RIAB_EXACT_FP
  const R dt = k.dt;
  vx = r_fma(k.cvel, w.ax * dt, vx);
  vy = r_fma(k.cvel, w.ay * dt, vy);
  px = r_fma(k.cpos, w.sx * dt, px);
  py = r_fma(k.cpos, w.sy * dt, py);
}

// ---- _check_and_handle_wall_collisions (Agent.py:426-441, utils.py:74-106, 304-328) ---
// A step shorter than the distance from prev_pos to the nearest wall cannot cross any wall:
// skip the per-wall segment tests (x2min was measured at prev_pos).
template <class R>
__device__ __forceinline__ void handle_collisions(const MotionConst<R>& k, const Wall<R>* s_w, R x2min, R ppx, R ppy, R& px,
                                                  R& py, R& vx, R& vy, int& n_bounce, int& n_sat) {
  RIAB_EXACT_FP
  const int nw = k.nw;
  const R dt = k.dt, sm = k.sm;
  const R step2 = norm2(px - ppx, py - ppy);
  if (nw > 0 && !(step2 < (R)0.998 * x2min)) {
    int it = 0;
    for (; it < RIAB_MAX_BOUNCES; ++it) {
      const R sbx = px - ppx, sby = py - ppy;  // the step (list b), walls are list a
      int hit = -1;
      auto crosses = [&](const Wall<R>& W) -> bool {
        RIAB_EXACT_FP
        const R d0x = ppx - W.ax, d0y = ppy - W.ay;
        const R den_a = r_fma(W.sx, -sby, W.sy * sbx);
        const R num_a = r_fma(d0x, -sby, d0y * sbx);
        const R den_b = r_fma(sbx, -W.sy, sby * W.sx);
        const R num_b = r_fma(-d0x, -W.sy, (-d0y) * W.sx);
        // 0 < num/den < 1 by sign logic (den == 0: +-inf / NaN in NumPy -> no hit)
        const bool ia = (den_a > 0) ? (num_a > 0 && num_a < den_a) : (den_a < 0 ? (num_a < 0 && num_a > den_a) : false);
        const bool ib = (den_b > 0) ? (num_b > 0 && num_b < den_b) : (den_b < 0 ? (num_b < 0 && num_b > den_b) : false);
        return ia && ib;
      };
      if (k.grid) {
        // (wall-heavy rooms: a step of at most the masks' length from prev_pos can only cross the walls of prev_pos's
        // cell's second mask; a longer step looks at every wall.  Ascending order: the first hit IS the lowest index)
        uint64_t m = norm2(sbx, sby) <= k.g_lmax2 ? wall_grid_mask<R>(k, ppx, ppy, 1) : (nw >= 64 ? ~0ull : ((1ull << nw) - 1ull));
        while (__builtin_amdgcn_ballot_w64(m != 0 && hit < 0) != 0) {
          if (m != 0 && hit < 0) {
            const int w = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (crosses(s_w[w])) hit = w;
          }
        }
      } else {
        for (int w = 0; w < nw; ++w)
          if (crosses(s_w[w]) && hit < 0) hit = w;  // first colliding wall = lowest index
      }
      if (hit < 0) break;
      const Wall<R> W = s_w[hit];
      // utils.wall_bounce
      R parx = W.sx * W.inv_len, pary = W.sy * W.inv_len;
      R perx = -W.sy * W.inv_len, pery = W.sx * W.inv_len;
      if (r_fma(-W.sy, vx, W.sx * vy) <= (R)0) {
        perx = -perx;
        pery = -pery;
      }
      if (r_fma(W.sx, vx, W.sy * vy) <= (R)0) {
        parx = -parx;
        pary = -pary;
      }
      const R vpar = r_fma(vx, parx, vy * pary), vper = r_fma(vx, perx, vy * pery);
      R nvx = r_fma(parx, vpar, -(perx * vper)), nvy = r_fma(pary, vpar, -(pery * vper));
      const R f = ((R)0.5 * sm) / r_sqrt(norm2(nvx, nvy));
      vx = f * nvx;
      vy = f * nvy;
      px = r_fma(vx, dt, ppx);
      py = r_fma(vy, dt, ppy);
      ++n_bounce;
    }
    if (it == RIAB_MAX_BOUNCES) ++n_sat;
  }
}

// ---- boundary safety net (Agent.py:221-222, Environment.py:781-894) -------------------
template <class R>
__device__ __forceinline__ void boundary_net(const MotionConst<R>& k, const AgentArgs& a, const Wall<R>* s_w, int t, int64_t b,
                                             uint32_t aid, R& px, R& py, int& n_bc, int& n_sat) {
  RIAB_EXACT_FP
  const R e0 = k.e0, e1 = k.e1, e2 = k.e2, e3 = k.e3;
  const int64_t B = a.B;
  const bool in_box = px > e0 && px < e1 && py > e2 && py < e3;
  bool inside = in_box;
  if (!k.simple_box) {  // polygonal boundary and / or holes: the strict interior of the one minus those of the others
    auto edge = [&](int kk, double& ax, double& ay, double& bx, double& by) {
      const Wall<R> W = s_w[kk];
      ax = (double)W.ax; ay = (double)W.ay; bx = (double)(W.ax + W.sx); by = (double)(W.ay + W.sy);
    };
    inside = env_contains(a.shape, (double)px, (double)py, edge);
  }
  if (!inside) {
    ++n_bc;
    if (!a.shape.boundary_mask && !in_box) {  // outside the box itself (Environment.py:871-885)
      if (a.periodic) {
        px = r_fma(-e1, floor(px / e1), px);  // np.mod(pos, extent)
        py = r_fma(-e3, floor(py / e3), py);
      } else {
        box_clamp<R>(k, px, py);
      }
    } else if (a.resample) {  // in a hole / outside the polygon: a new random position (Environment.py:886-893)
      px = (R)a.resample[((int64_t)t * 2 + 0) * B + b];
      py = (R)a.resample[((int64_t)t * 2 + 1) * B + b];
    } else {
      // uniform over the extent, rejected until inside (the reference's recursive sample_positions(1, "random"),
      // Environment.py:584-600); its own Philox stream, one call per attempt
      auto edge = [&](int kk, double& ax, double& ay, double& bx, double& by) {
        const Wall<R> W = s_w[kk];
        ax = (double)W.ax; ay = (double)W.ay; bx = (double)(W.ax + W.sx); by = (double)(W.ay + W.sy);
      };
      const uint64_t stp = a.step0 + (uint64_t)t;
      int attempt = 0;
      for (; attempt < RIAB_MAX_RESAMPLES; ++attempt) {
        const u32x4 w = philox4x32_10((uint32_t)stp, (uint32_t)(stp >> 32) ^ ((uint32_t)attempt << 24), aid,
                                      RIAB_TAG_MOTION ^ 2u, a.k0, a.k1);
        const double cx = uniform_between(a.shape.e0, a.shape.e1, u01_24(w.x));
        const double cy = uniform_between(a.shape.e2, a.shape.e3, u01_24(w.y));
        px = (R)cx;
        py = (R)cy;
        if (env_contains(a.shape, cx, cy, edge)) break;
      }
      if (attempt == RIAB_MAX_RESAMPLES) ++n_sat;
    }
  }
}

// one wall of the table as the kernels keep it: every operation rounded on its own and the one multiply-add spelled
// out, so that whichever kernel prepares a wall (a trajectory kernel while staging; walls_prepare_kernel of
// riab_step1.hip once per plan) produces the same bits
template <class R>
__device__ __forceinline__ Wall<R> make_wall(double ax, double ay, double bx, double by) {
  RIAB_EXACT_FP
  const double sx = bx - ax, sy = by - ay;
  const double ss = fma(sx, sx, sy * sy);
  Wall<R> w;
  w.ax = (R)ax;
  w.ay = (R)ay;
  w.sx = (R)sx;
  w.sy = (R)sy;
  w.inv_ss = (R)(1.0 / ss);
  w.inv_len = (R)(1.0 / sqrt(ss));
  return w;
}
// walls of the launch into LDS (any number of threads of the workgroup)
template <class R>
__device__ __forceinline__ void stage_walls(const AgentArgs& a, Wall<R>* s_w, int tid, int nthreads) {
  for (int w = tid; w < a.n_walls; w += nthreads)
    s_w[w] = make_wall<R>(a.walls[4 * w], a.walls[4 * w + 1], a.walls[4 * w + 2], a.walls[4 * w + 3]);
}

// the Rayleigh <-> normal tables into LDS (rows padded to RIAB_G_STRIDE / RIAB_H_STRIDE doubles): all of a thread's
// table loads are issued before the first LDS write (one memory round trip for the whole staging; a load -> store
// loop serialises ~36 of them, 37 us per launch)
template <int NT>
__device__ __forceinline__ void stage_rayleigh_tables(double* s_g, double* s_h, int tid) {
  constexpr int GN = RIAB_G_SEGS * (RIAB_G_DEG + 3), HN = RIAB_H_SEGS * (RIAB_H_DEG + 3);
  constexpr int GI = (GN + NT - 1) / NT, HI = (HN + NT - 1) / NT;
  double gv[GI], hv[HI];
#pragma unroll
  for (int k = 0; k < GI; ++k) {
    const int i = k * NT + tid;
    gv[k] = i < GN ? (&riab_g_table[0][0])[i] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < HI; ++k) {
    const int i = k * NT + tid;
    hv[k] = i < HN ? (&riab_h_table[0][0])[i] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < GI; ++k) {
    const int i = k * NT + tid;
    if (i < GN) s_g[(i / (RIAB_G_DEG + 3)) * RIAB_G_STRIDE + i % (RIAB_G_DEG + 3)] = gv[k];
  }
#pragma unroll
  for (int k = 0; k < HI; ++k) {
    const int i = k * NT + tid;
    if (i < HN) s_h[(i / (RIAB_H_DEG + 3)) * RIAB_H_STRIDE + i % (RIAB_H_DEG + 3)] = hv[k];
  }
}

template <class R, int IN, bool PC, bool PUB = false>
__device__ __forceinline__ void agent_step_body(const AgentArgs& a) {
  RIAB_EXACT_FP
  static_assert(!PC || IN == 0, "the producer wave only exists in Philox mode");
  static_assert(!PUB || PC, "rows are published by the helper wave");
  const int lane = (int)(threadIdx.x & 63);
  const int wave = PC ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  __shared__ float s_z[PC ? 2 : 1][PC ? RIAB_Z_BATCH : 1][2][PC ? 64 : 1];
  // history rows of four steps of this workgroup's 64 agents (written by whichever wave stores them)
  __shared__ __align__(16) float s_hist[1][4][RIAB_HIST_ROWS][64];
  const int hist_lds_lane = (lane >> 4) * 64 + (lane & 15) * 4;                    // floats
  const uint32_t hist_glb_lane = (uint32_t)(((int64_t)(lane >> 4) * a.B + (lane & 15) * 4) * 4);  // bytes
  // rows of the four-step block that starts at step t0 (n_steps of them), LDS -> HBM as float4 rows:
  // store j covers the (step, row) pairs 4j .. 4j+3: step j/2, rows 4(j&1) + lane/16, agents 4(lane&15)..+3.
  // Everything that depends on j or t0 is wave-uniform (scalar registers).
  auto flush_hist = [&](int buf, int t0, int n_steps) {
    const int n2 = 2 * n_steps;
    float* const g0 = a.hist + (int64_t)t0 * RIAB_HIST_ROWS * a.B + (int64_t)blockIdx.x * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < n2) {
        const v4f v = *reinterpret_cast<const v4f*>(&s_hist[buf][j >> 1][(j & 1) * 4][0] + hist_lds_lane);
        char* const gj = reinterpret_cast<char*>(g0 + (int64_t)((j >> 1) * RIAB_HIST_ROWS + (j & 1) * 4) * a.B);
        if (PUB) store_v4f_agent(gj + hist_glb_lane, v);
        else *reinterpret_cast<v4f*>(gj + hist_glb_lane) = v;
      }
    }
  };
  // what the stepping wave hands over per step in the PC variant: the displacement (float64: the tail is
  // float64 arithmetic) and the position as the history keeps it; two four-step blocks
  __shared__ double s_dp[PC ? 2 : 1][PC ? 4 : 1][2][PC ? 64 : 1];
  __shared__ float s_pp[PC ? 2 : 1][PC ? 4 : 1][2][PC ? 64 : 1];
  if (PC && wave == 1) {
    // ---- helper wave.  Barrier schedule (both waves): staging, "noise batch 0 ready", then one barrier
    // after every four-step block k.  While the stepping wave computes block k the helper (a) draws noise
    // batch k/4 + 1 when k is a multiple of 4 — into the buffer the stepping wave left at the previous
    // barrier — and (b) finishes block k - 1: measured velocities, head direction, distance and the
    // history rows of its four steps — the stepping wave fills the other hand-over buffer.
    const int64_t b = (int64_t)blockIdx.x * 64 + lane;
    const uint32_t aid = (uint32_t)(a.agent_id0 + b);
    const RiabMotion& m = a.m;
    const TailConst<R> tail_c = {(R)m.dt, (R)(1.0 / m.dt), (R)(1.0 - m.dt / m.hd_tau), (R)(m.dt / m.hd_tau),
                                 m.hd_tau <= m.dt};
    double* st = a.state + b;
    const int64_t B = a.B;
    StepTail<R> tl{(R)st[5 * B], (R)st[6 * B], (R)st[7 * B], (R)st[8 * B], (R)st[9 * B], (R)st[10 * B], 0};
    u32x4 pw = {0u, 0u, 0u, 0u};
    auto draw_batch = [&](int batch) {
      const int t0 = batch * RIAB_Z_BATCH;
      const int tn = min(RIAB_Z_BATCH, a.T - t0);
      for (int i = 0; i < tn; ++i) {
        const MotionDraw d = motion_normals(a.step0 + (uint64_t)(t0 + i), t0 + i == 0, aid, a.k0, a.k1, pw);
        pw = d.pw;
        s_z[batch & 1][i][0][lane] = d.z_rot;
        s_z[batch & 1][i][1][lane] = d.z_spd;
      }
    };
    // PUB: rows of steps < n have left this wave write-through and been acknowledged: the consumer may read them
    auto publish = [&](int n) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0)
        __hip_atomic_store((riab_gu32*)(uintptr_t)(a.ctrl + RIAB_CTRL_PROGRESS_WORD(blockIdx.x)), (uint32_t)a.step0 + (uint32_t)n,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // Short launches publish a block as soon as the stepping wave has been released for the next one (latency
    // matters, and the rate kernel has not saturated HBM for long); long ones one block later (see finish_block).
    const bool early_pub = a.T <= 64;
    auto finish_block = [&](int buf, int t0, int n_steps) {
      for (int i = 0; i < n_steps; ++i) {
        tl = step_tail<R>(tl, (R)s_dp[buf][i][0][lane], (R)s_dp[buf][i][1][lane], tail_c, a.step0 + (uint64_t)(t0 + i),
                          aid, a.k0, a.k1);
        if (a.hist) {
          float* sh = &s_hist[0][i][0][lane];
          sh[0 * 64] = s_pp[buf][i][0][lane];
          sh[1 * 64] = s_pp[buf][i][1][lane];
          sh[2 * 64] = (float)tl.mvx;
          sh[3 * 64] = (float)tl.mvy;
          sh[4 * 64] = (float)tl.hx;
          sh[5 * 64] = (float)tl.hy;
          sh[6 * 64] = (float)tl.mrot;
          sh[7 * 64] = (float)tl.dist;
        }
      }
      // The previous block's rows were stored one block ago (a whole block of the stepping wave's time plus this
      // block's tails): by now they are acknowledged and the wait below is free.  Waiting right after the stores
      // instead put the write-through latency under a saturated HBM on the stepping wave's barrier (the pair of
      // kernels then ran at 3.9-4.3 us per step [MI355X]).
      if (PUB && !early_pub && t0 > 0) publish(t0);
      if (a.hist) {
        __builtin_amdgcn_wave_barrier();  // (LDS serves a wave's requests in order: the reads below see the writes)
        flush_hist(0, t0, n_steps);
        __builtin_amdgcn_wave_barrier();
      }
    };
    if (PUB && lane == 0) {  // this workgroup is resident; its progress word first goes back to "no row of this launch yet"
      __hip_atomic_store((riab_gu32*)(uintptr_t)(a.ctrl + RIAB_CTRL_PROGRESS_WORD(blockIdx.x)), (uint32_t)a.step0, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      atomicAdd(a.ctrl + RIAB_CTRL_STARTED, 1u);
    }
    __syncthreads();  // (the table-staging barrier of the stepping wave)
    draw_batch(0);
    __syncthreads();  // noise batch 0 ready
    const int n_blocks = (a.T + 3) >> 2;
    for (int k = 0; k < n_blocks; ++k) {
      if (PUB && early_pub && k >= 2) publish(4 * (k - 1));  // the rows flushed before the last barrier
      if ((k & 3) == 0 && (k / 4 + 1) * RIAB_Z_BATCH < a.T) draw_batch(k / 4 + 1);
      if (k > 0) finish_block((k - 1) & 1, 4 * (k - 1), 4);
      __syncthreads();  // block k is handed over; noise for block k + 1 is ready
    }
    if (PUB && early_pub && n_blocks >= 2) publish(4 * (n_blocks - 1));
    finish_block((n_blocks - 1) & 1, 4 * (n_blocks - 1), a.T - 4 * (n_blocks - 1));
    if (PUB) publish(a.T);
    st[5 * B] = (double)tl.mvx;
    st[6 * B] = (double)tl.mvy;
    st[7 * B] = (double)tl.mrot;
    st[8 * B] = (double)tl.hx;
    st[9 * B] = (double)tl.hy;
    st[10 * B] = (double)tl.dist;
    if (a.diag && tl.n_still) atomicAdd(a.diag + 3, tl.n_still);
    return;
  }
  __shared__ Wall<R> s_w[RIAB_MAX_WALLS];
  __shared__ double s_g[sizeof(R) == 8 ? RIAB_G_SEGS * RIAB_G_STRIDE : 1];
  __shared__ double s_h[sizeof(R) == 8 ? RIAB_H_SEGS * RIAB_H_STRIDE : 1];
  // Long launches stage the tables in LDS (per-lane gathers every step); a launch of a few steps
  // (the closed-loop path, T = 1) reads its two rows per step straight from the L2-resident
  // global tables instead of paying the 19 KB staging each time.
  const bool use_lds = sizeof(R) == 8 && a.T >= 16;
  if (use_lds) stage_rayleigh_tables<64>(s_g, s_h, lane);
  const lds_cf64_ptr lds_g = (lds_cf64_ptr)s_g, lds_h = (lds_cf64_ptr)s_h;
  const glb_cf64_ptr glb_g = (glb_cf64_ptr)&riab_g_table[0][0], glb_h = (glb_cf64_ptr)&riab_h_table[0][0];
  stage_walls<R>(a, s_w, lane, 64);
  __shared__ uint64_t s_grid[RIAB_WALL_GRID_WORDS];  // (wall-heavy rooms: RiabMotion.wall_grid)
  stage_wall_grid(a, s_grid, lane, 64);
  __syncthreads();
  const int64_t b = (int64_t)blockIdx.x * 64 + lane;
  if (b >= a.B) return;
  // full wave and a multi-step launch: history rows go through LDS (below); single steps store directly
  const bool hist_staged = a.hist && a.T >= 4 && ((int64_t)blockIdx.x * 64 + 64 <= a.B);
  // a latency-bound recurrence sharing its CU with bandwidth-bound rate kernels: win the
  // SIMD's issue arbitration whenever this wave is ready
  __builtin_amdgcn_s_setprio(3);
  const RiabMotion& m = a.m;
  const int nw = a.n_walls;
  const R dt = (R)m.dt;

  double* st = a.state + b;
  const int64_t B = a.B;
  R px = (R)st[0 * B], py = (R)st[1 * B];
  R vx = (R)st[2 * B], vy = (R)st[3 * B];
  R rot = (R)st[4 * B];
  R mvx = (R)st[5 * B], mvy = (R)st[6 * B];
  R mrot = (R)st[7 * B];
  R hx = (R)st[8 * B], hy = (R)st[9 * B];
  R dist = (R)st[10 * B];
  R dwall = (R)st[11 * B];

  R drx = 0, dry = 0;
  if (m.has_drift) {
    drx = (R)a.drift[b];
    dry = (R)a.drift[B + b];
  }

  int n_bounce = 0, n_sat = 0, n_bc = 0, n_still = 0;
  const uint32_t aid = (uint32_t)(a.agent_id0 + b);

  // constants of the step
  const MotionConst<R> K = make_motion_const<R>(a, s_w, s_grid);
  const R sm_kw = K.sm_kw;
  const R inv_2s2 = K.inv_2s2;
  const double inv_sm = K.inv_sm;
  const R inv_dt = K.inv_dt;
  const TailConst<R> tail_c = {dt, inv_dt, (R)(1.0 - m.dt / m.hd_tau), (R)(m.dt / m.hd_tau), m.hd_tau <= m.dt};

  u32x4 pw = {0u, 0u, 0u, 0u};

  for (int t = 0; t < a.T; ++t) {
    // ---- the step's standard normals -------------------------------------------------------
    R z_rot, z_spd;
    if (IN == 1) {
      z_rot = (R)a.z_in[((int64_t)t * 2 + 0) * B + b];
      z_spd = (R)a.z_in[((int64_t)t * 2 + 1) * B + b];
    } else if (IN == 2) {
      z_rot = (R)0;
      z_spd = (R)0;
    } else {
      float zr, zs;
      if (PC) {
        if (t == 0) __syncthreads();  // noise batch 0 is ready (later batches: the barrier that ends a block)
        zr = s_z[(t / RIAB_Z_BATCH) & 1][t % RIAB_Z_BATCH][0][lane];
        zs = s_z[(t / RIAB_Z_BATCH) & 1][t % RIAB_Z_BATCH][1][lane];
      } else {
        const MotionDraw d = motion_normals(a.step0 + (uint64_t)t, t == 0, aid, a.k0, a.k1, pw);
        pw = d.pw;
        zr = d.z_rot;
        zs = d.z_spd;
      }
      z_rot = (R)zr;
      z_spd = (R)zs;
    }
    if (a.z_out) {
      a.z_out[((int64_t)t * 2 + 0) * B + b] = (double)z_rot;
      a.z_out[((int64_t)t * 2 + 1) * B + b] = (double)z_spd;
    }
    const R ppx = px, ppy = py;  // prev_pos (Agent.py:199)
    if (IN == 2) {
      // imported / forced trajectory (Agent.py:229-238): the position is given, the motion
      // model, wall handling and boundary conditions are skipped
      px = (R)a.forced[((int64_t)t * 2 + 0) * B + b];
      py = (R)a.forced[((int64_t)t * 2 + 1) * B + b];
    } else {

    // ---- _stochastic_velocity_update (Agent.py:287-312) -----------------------------------
    // float64 path, ordered for latency: the rotation does not change |v|, so the speed and the
    // LDS fetch of its G-segment come first; the H-segment fetch is followed by wall pass 1
    // (which only needs the position) before the H polynomial is evaluated.
    R v2 = norm2(vx, vy);
    const bool zero_v = (v2 == (R)0);
    if (zero_v) v2 = (R)1e-16;  // the reference replaces a zero velocity by (1e-8, 0) (Agent.py:299-300)
    const R ispeed = r_rsqrt(v2);
    const R speed = v2 * ispeed;
    SegRow<RIAB_G_DEG> grow;
    double tG = 0.0;
    if (sizeof(R) == 8) {
      tG = clamp_G_arg((double)speed * inv_sm);
      const int sg = seg_G(tG);
      grow = use_lds ? seg_fetch<RIAB_G_DEG>(lds_g + sg * RIAB_G_STRIDE) : seg_fetch<RIAB_G_DEG>(glb_g + sg * (RIAB_G_DEG + 3));
    }
    rot = ou_step<R>(rot, (R)m.rot_theta_kw, (R)m.rot_drift_kw, (R)m.rot_sigma_kw, dt, z_rot);
    {
      R sn, cs;
      sincos_small(rot * dt, &sn, &cs);
      rotate_by<R>(cs, sn, vx, vy);
      vx = zero_v ? (R)1e-8 : vx;
      vy = zero_v ? (R)0 : vy;
    }
    // utils.rayleigh_to_normal / normal_to_rayleigh (utils.py:409-421), sigma = speed_mean
    R speed_new = sm_kw;
    SegRow<RIAB_H_DEG> hrow;
    double nv64 = 0.0;
    bool h_in_table = true;
    if (sizeof(R) == 8) {
      nv64 = seg_eval<RIAB_G_DEG>(grow, tG);
      nv64 = ou_step<double>(nv64, m.speed_theta_kw, 0.0, m.speed_sigma_kw, m.dt, (double)z_spd);
      h_in_table = fabs(nv64) < RIAB_H_NMAX;
      const int sh = seg_H(nv64);
      hrow = use_lds ? seg_fetch<RIAB_H_DEG>(lds_h + sh * RIAB_H_STRIDE) : seg_fetch<RIAB_H_DEG>(glb_h + sh * (RIAB_H_DEG + 3));
    } else {
      R u = (R)1 - r_exp(-v2 * inv_2s2);
      u = (u < (R)1e-6) ? (R)1e-6 : u;
      u = (u > (R)(1 - 1e-6)) ? (R)(1 - 1e-6) : u;
      R nv = r_ndtri(u);
      nv = ou_step<R>(nv, (R)m.speed_theta_kw, (R)0, (R)m.speed_sigma_kw, dt, z_spd);
      const R x = r_ndtr(nv);
      speed_new = sm_kw * r_sqrt((R)-2 * r_log((R)1 - x));
    }
    // ---- _wall_velocity_update, pass 1 (walls_pass1 above) ----------------------------------
    const NearWalls<R> near = walls_pass1<R>(K, s_w, px, py);
    // ---- finish the speed update ---------------------------------------------------------------
    if (sizeof(R) == 8) {
      const double tnew = h_in_table ? seg_eval<RIAB_H_DEG>(hrow, nv64) : sqrt(-2.0 * log(1.0 - normcdf(nv64)));
      speed_new = (R)(m.speed_mean_kw * tnew);
    }
    if (m.speed_std_is_zero) speed_new = sm_kw;
    {
      const R f = speed_new * ispeed;
      vx *= f;
      vy *= f;
    }
    // ---- _drift_velocity_update (Agent.py:331-341) ----------------------------------------
    if (m.has_drift) drift_update<R>((R)m.drift_theta, drx, dry, dt, vx, vy);
    // ---- _wall_velocity_update, pass 2 -----------------------------------------------------
    if (nw > 0 && K.repel) {
      const WallPush<R> push = walls_pass2_terms<R>(K, s_w, near, px, py);
      dwall = closest_wall_distance<R>(K, near);
      walls_pass2_apply<R>(K, push, px, py, vx, vy);
    }
    // ---- propose (Agent.py:216) -----------------------------------------------------------
    propose_step<R>(vx, vy, dt, px, py);
    // ---- _check_and_handle_wall_collisions, boundary safety net -----------------------------
    handle_collisions<R>(K, s_w, near.x2min, ppx, ppy, px, py, vx, vy, n_bounce, n_sat);
    boundary_net<R>(K, a, s_w, t, b, aid, px, py, n_bc, n_sat);
    }  // random-motion branch
    // ---- _measure_velocity_of_step_taken (Agent.py:456-471) -------------------------------
    R dpx, dpy;
    step_displacement<R>(a, px, py, ppx, ppy, dpx, dpy);
    if (PC) {
      // the rest of the step only produces outputs: the helper wave computes it from the displacement
      s_dp[(t >> 2) & 1][t & 3][0][lane] = (double)dpx;
      s_dp[(t >> 2) & 1][t & 3][1][lane] = (double)dpy;
      s_pp[(t >> 2) & 1][t & 3][0][lane] = (float)px;
      s_pp[(t >> 2) & 1][t & 3][1][lane] = (float)py;
      if ((t & 3) == 3 || t == a.T - 1) __syncthreads();  // hand the block over; the next noise batch is ready
      continue;
    }
    {
      StepTail<R> tl{mvx, mvy, mrot, hx, hy, dist, n_still};
      tl = step_tail<R>(tl, dpx, dpy, tail_c, a.step0 + (uint64_t)t, aid, a.k0, a.k1);
      mvx = tl.mvx; mvy = tl.mvy; mrot = tl.mrot; hx = tl.hx; hy = tl.hy; dist = tl.dist; n_still = tl.n_still;
    }
    if (IN == 2) {  // overwrite_velocity=True (Agent.py:461-462, 469-470)
      vx = mvx;
      vy = mvy;
      rot = mrot;
    }
    // ---- save_to_history (Agent.py:514-520) ------------------------------------------------
    if (hist_staged) {
      // Eight dword stores per step are eight places to queue behind the rate kernels' store stream.
      // The rows of four steps are parked in LDS and written out as float4 rows: eight store instructions
      // per FOUR steps, each covering four (step, row) pairs.
      float* sh = &s_hist[0][t & 3][0][lane];
      sh[0 * 64] = (float)px;
      sh[1 * 64] = (float)py;
      sh[2 * 64] = (float)mvx;
      sh[3 * 64] = (float)mvy;
      sh[4 * 64] = (float)hx;
      sh[5 * 64] = (float)hy;
      sh[6 * 64] = (float)mrot;
      sh[7 * 64] = (float)dist;
      if ((t & 3) == 3 || t == a.T - 1) {
        __builtin_amdgcn_wave_barrier();  // (LDS serves a wave's requests in order: the reads below see the writes)
        flush_hist(0, t - (t & 3), (t & 3) + 1);
        __builtin_amdgcn_wave_barrier();
      }
    } else if (a.hist) {
      float* h = a.hist + (int64_t)t * RIAB_HIST_ROWS * B + b;
      h[0 * B] = (float)px;
      h[1 * B] = (float)py;
      h[2 * B] = (float)mvx;
      h[3 * B] = (float)mvy;
      h[4 * B] = (float)hx;
      h[5 * B] = (float)hy;
      h[6 * B] = (float)mrot;
      h[7 * B] = (float)dist;
    }
  }
  st[0 * B] = (double)px;
  st[1 * B] = (double)py;
  st[2 * B] = (double)vx;
  st[3 * B] = (double)vy;
  st[4 * B] = (double)rot;
  if (!PC) {  // (the helper wave owns these rows)
    st[5 * B] = (double)mvx;
    st[6 * B] = (double)mvy;
    st[7 * B] = (double)mrot;
    st[8 * B] = (double)hx;
    st[9 * B] = (double)hy;
    st[10 * B] = (double)dist;
  }
  st[11 * B] = (double)dwall;
  if (a.diag) {
    if (n_bounce) atomicAdd(a.diag + 0, n_bounce);
    if (n_sat) atomicAdd(a.diag + 1, n_sat);
    if (n_bc) atomicAdd(a.diag + 2, n_bc);
    if (n_still) atomicAdd(a.diag + 3, n_still);
  }
}

template <class R, int IN, bool PC, bool PUB = false>
__global__ __launch_bounds__(PC ? 128 : 64) void agent_step_kernel(const AgentArgs a) {
  agent_step_body<R, IN, PC, PUB>(a);
}

// argument checks + the kernel's argument block (shared by riab_agent_step and the step plan's fused
// motion + task launch)
static inline int fill_agent_args(AgentArgs& a, const RiabEnv* env, const RiabMotion* motion, double* state, int64_t B,
                                  int64_t agent_id0, const double* drift, const double* z_in, double* z_out,
                                  const double* forced_pos, uint64_t seed, uint64_t step0, int32_t T, float* hist,
                                  int32_t* diag, const double* resample_pos = nullptr) {
  if (!env || !motion || !state || B <= 0 || T <= 0 || agent_id0 < 0) return RIAB_EINVAL;
  if (env->n_walls < 0 || (env->n_walls > 0 && !env->walls)) return RIAB_EINVAL;
  if (env->n_walls > RIAB_MAX_WALLS) return RIAB_ETOOBIG;
  if (check_env_shape(env)) return RIAB_EINVAL;
  a.resample = resample_pos;
  a.shape = make_env_shape(env);
  if (motion->has_drift && !drift) return RIAB_EINVAL;
  a.m = *motion;
  a.e0 = env->extent[0];
  a.e1 = env->extent[1];
  a.e2 = env->extent[2];
  a.e3 = env->extent[3];
  a.scale = env->scale;
  a.periodic = env->periodic;
  a.n_walls = env->n_walls;
  a.walls = env->walls;
  a.state = state;
  a.B = B;
  a.agent_id0 = agent_id0;
  a.drift = drift;
  a.z_in = z_in;
  a.z_out = z_out;
  a.forced = forced_pos;
  a.k0 = (uint32_t)seed;
  a.k1 = (uint32_t)(seed >> 32);
  a.step0 = step0;
  a.T = T;
  a.hist = hist;
  a.diag = diag;
  a.ctrl = nullptr;
  a.pub_single = 4;
  return RIAB_OK;
}

}  // namespace riab

