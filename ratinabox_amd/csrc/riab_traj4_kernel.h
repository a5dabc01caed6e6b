#pragma once
// The trajectory kernel of the open-loop pipelines: Agent.update() for T fused steps with ONE AGENT'S STEP SPREAD OVER
// FOUR SPECIALISED WAVES (gfx950).  Reference: Agent.update, ratinabox/Agent.py:160-242 (pieces cited in
// riab_agent_kernel.h, whose inlined functions this kernel calls — the two kernels agree bit for bit).
//
// Why.  4096 agents are 64 wavefronts: one per compute unit, alone on its SIMD.  A lone wave issues one float64
// instruction every ~7.5 cycles whatever its dependencies (tools/lane_bench.hip), so the single-wave kernel is bound
// by its own instruction count (~650 per step), and the two-wave kernel of round 1 (a helper wave for noise, the
// output-only tail and the history stores) by the ~450 that were left on the stepping wave: 1.8 us per step alone,
// 2.6 us next to the firing-rate kernel — 52 us for the 20 steps of the driver's bench line, the critical path of
// that region.  SIMD lanes cannot run different instruction streams, but the four SIMDs of a compute unit can: the
// step is cut along its data dependencies into four instruction streams that run concurrently.
//
//   wave G  "geometry"   owns position and velocity: rotation by the step's (cos, sin), wall repulsion (both passes),
//                        drift, the proposed step, collisions / bounces, the boundary safety net.  Needs, per step,
//                        the speed factor f = |v_new| / |v| from wave S; hands |v|^2 of the NEXT step to wave S as
//                        soon as the velocity is final, and the displacement / position to wave T.
//   wave S  "speed"      the Rayleigh-speed Ornstein-Uhlenbeck update (Agent.py:302-309): |v| -> G-table polynomial
//                        -> OU step in normal space -> H-table polynomial -> f.  The longest dependent chain of a
//                        step (two per-lane LDS gathers and two Horner recurrences); nothing else runs on this wave.
//   wave N  "noise"      everything that does not depend on the state: Philox + Box-Muller (or the explicit normals
//                        of parity runs), the rotational-velocity OU — a recurrence on its own, Agent.py:287-297 —
//                        and sin / cos of the heading increment, a ring of steps ahead.
//   wave T  "tail"       the output-only rest of a step (measured velocity, measured rotational velocity, head
//                        direction low-pass, distance travelled: Agent.py:456-507), the history rows (LDS ->
//                        float4 row stores) and, in the publishing variant, the progress words the firing-rate
//                        kernels wait on.
//
// Coupling.  No workgroup barrier inside the step loop: a barrier needs all four waves, and wave T must be free to
// sit in `s_waitcnt vmcnt(0)` behind a saturated HBM without holding the others up.  Instead:
//   * G <-> S, twice per step, through ONE 8-byte LDS slot per lane and direction: the consumer polls its own lane's
//     slot until no lane holds the sentinel any more (the producer fills all 64 lanes with one ds_write_b64, LDS
//     executes a compute unit's instructions in order), takes the value and puts the sentinel back.  A hand-over is
//     one LDS write, ~one read latency of polling and nothing else;
//   * N -> G, S and G -> T through rings of RIAB_T4_RING steps with "steps produced" / "steps consumed" counters in
//     LDS, which the consumers re-read only when their cached copy runs out (once per ring length in steady state).
// Every wait is bounded (a stuck partner sets the abort word and every wave leaves its loop: the launch ends).
//
// Per step the critical path is now: S's chain (~75 issue slots + two LDS gathers) + G's ~25 instructions between
// f and the next |v|^2 + two hand-overs.
#include "riab_agent_kernel.h"

namespace riab {

#ifndef RIAB_T4_ABLATE
#define RIAB_T4_ABLATE 0  // timing experiments only (tools/traj_probe.py): bits switch parts of the step off
#endif
#ifdef RIAB_T4_PROFILE  // timing experiments only: shader-clock stamps of workgroup 0's G and S waves into a.z_out
#define T4_STAMP(k) do { if (blockIdx.x == 0 && t < 256) { const long long c_ = clock64(); if (lane == 0) ((long long*)a.z_out)[t * 8 + (k)] = c_; } } while (0)
#else
#define T4_STAMP(k) do { } while (0)
#endif
#define RIAB_T4_RING 16  // steps of noise ahead of the stepping waves / of hand-over slack in front of the tail wave
#define RIAB_T4_SENT 0x7FF8DEADBEEF0001ull  // a NaN payload no arithmetic produces: "slot empty"
#define RIAB_T4_SPIN_LIMIT (1u << 24)       // ~70 ns per poll: about a second

enum { T4_C_NOISE = 0, T4_C_GDONE = 1, T4_C_TDONE = 2, T4_C_STATE = 3, T4_C_WORDS = 4 };

template <bool V>
struct T4Tag {
  static constexpr bool value = V;
};
typedef volatile __attribute__((address_space(3))) unsigned long long* t4_slot_ptr;
typedef volatile __attribute__((address_space(3))) uint32_t* t4_cnt_ptr;

// wait until counter `which` exceeds `need`; returns the counter (> need), or 0 when the wait gave up.  The common
// path is one LDS read and one scalar compare; nothing but the spin count is tested inside the loop.
__device__ __forceinline__ uint32_t t4_wait_counter(t4_cnt_ptr cnt, int which, uint32_t need, bool nap) {
  uint32_t v, spins = 0;
  do {
    v = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[which]);
    if (nap && v <= need) __builtin_amdgcn_s_sleep(1);
  } while (v <= need && ++spins < RIAB_T4_SPIN_LIMIT);
  asm volatile("" ::: "memory");  // (what the counter announces is read after it)
  return v > need ? v : 0u;
}

// take the value another wave puts into this lane's slot; false when the wait gave up
__device__ __forceinline__ bool t4_take(t4_slot_ptr slot, double* out) {
  unsigned long long bits;
  uint32_t spins = 0;
  bool empty;
  do {
    bits = *slot;
    empty = __builtin_amdgcn_ballot_w64(bits == RIAB_T4_SENT) != 0;
  } while (empty && ++spins < RIAB_T4_SPIN_LIMIT);
  *slot = RIAB_T4_SENT;
  asm volatile("" ::: "memory");
  *out = __longlong_as_double((long long)bits);
  return !empty;
}

// rows of the block of `n_steps` (<= 4) steps that starts at step t0, LDS -> HBM as float4 rows: store j covers the
// (step, row) pairs 4j .. 4j+3: step j/2, rows 4(j&1) + lane/16, agents 4(lane&15)..+3.  Everything that depends on
// j or t0 is wave-uniform.  PUB: write-through (agent-scope) stores, see riab_agent_kernel.h.
template <bool PUB>
__device__ __forceinline__ void t4_flush_hist(const AgentArgs& a, const float* s_hist, int lane, int t0, int n_steps) {
  const int hist_lds_lane = (lane >> 4) * 64 + (lane & 15) * 4;                                    // floats
  const uint32_t hist_glb_lane = (uint32_t)(((int64_t)(lane >> 4) * a.B + (lane & 15) * 4) * 4);  // bytes
  const int n2 = 2 * n_steps;
  float* const g0 = a.hist + (int64_t)t0 * RIAB_HIST_ROWS * a.B + (int64_t)blockIdx.x * 64;
  if ((int64_t)blockIdx.x * 64 + (lane & 15) * 4 >= a.B) return;  // (a quad of agents beyond the batch: last workgroup only)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < n2) {
      const v4f v = *reinterpret_cast<const v4f*>(s_hist + ((j >> 1) * RIAB_HIST_ROWS + (j & 1) * 4) * 64 + hist_lds_lane);
      char* const gj = reinterpret_cast<char*>(g0 + (int64_t)((j >> 1) * RIAB_HIST_ROWS + (j & 1) * 4) * a.B);
      if (PUB) store_v4f_agent(gj + hist_glb_lane, v);
      else *reinterpret_cast<v4f*>(gj + hist_glb_lane) = v;
    }
  }
}

// One element of the float64 state.  PUB: written through (agent scope), like the history rows: the launch's LAST
// publication comes after these stores have been acknowledged, so whoever has waited for it — every form of the rate
// stage does, on the caller's stream — finds the state in memory too, and riab_simulate needs no event between the two
// streams (the event's record + wait cost 4.7 us of host time per call and its barrier packet, processed after the
// rate kernel, 3-4 us of the 11 us between that kernel's end and the return of a host synchronisation [MI355X]).
template <bool PUB>
__device__ __forceinline__ void t4_store_state(double* p, double v) {
  if (PUB) __hip_atomic_store((riab_gu64*)(uintptr_t)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// (PUB) this wave's state stores and counters have been acknowledged: tell the tail wave
template <bool PUB>
__device__ __forceinline__ void t4_state_stored(uint32_t* s_cnt, int lane) {
  if (PUB) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&s_cnt[T4_C_STATE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// IN: 0 = in-kernel Philox noise, 1 = explicit normals a.z_in.  float64.  Any B that is a multiple of 4: the lanes of
// the last workgroup beyond B shadow the workgroup's first agent (same loads, same arithmetic, same values) and store
// nothing.
template <int IN, bool PUB>
__global__ __launch_bounds__(256) void traj4_kernel(const AgentArgs a) {
  RIAB_EXACT_FP
  typedef double R;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __shared__ Wall<R> s_w[RIAB_MAX_WALLS];
  __shared__ double s_g[RIAB_G_SEGS * RIAB_G_STRIDE];
  __shared__ double s_h[RIAB_H_SEGS * RIAB_H_STRIDE];
  __shared__ double s_cs[RIAB_T4_RING][64], s_sn[RIAB_T4_RING][64], s_zs[RIAB_T4_RING][64];  // N -> G (cos, sin), N -> S (z_speed)
  __shared__ __align__(16) double s_dp[RIAB_T4_RING][64][2];                                  // G -> T: displacement (one 16-byte write)
  __shared__ __align__(8) float s_pp[RIAB_T4_RING][64][2];                                   //         position as the history keeps it
  __shared__ __align__(16) float s_hist[4 * RIAB_HIST_ROWS * 64];                             // T: rows of one block
  __shared__ unsigned long long s_v2[64], s_f[64];                                            // G -> S: |v|^2;  S -> G: f
  __shared__ uint32_t s_cnt[T4_C_WORDS];
  const RiabMotion& m = a.m;
  const int64_t B = a.B;
  const bool live = (int64_t)blockIdx.x * 64 + lane < B;
  const int64_t b = live ? (int64_t)blockIdx.x * 64 + lane : (int64_t)blockIdx.x * 64;
  const uint32_t aid = (uint32_t)(a.agent_id0 + b);
  double* st = a.state + b;
  const int T = a.T;
  // ---- every wave asks for ITS part of the state first: the loads travel while the tables are staged (the state was
  // written through by the previous launch and comes from memory: one round trip that used to sit between the barrier
  // and the first step — and the first row's publication is what the rate stage of a short call waits for) ----
  R pre0 = 0, pre1 = 0, pre2 = 0, pre3 = 0, pre4 = 0, pre5 = 0;
  if (wave == 0) {
    pre0 = st[0 * B]; pre1 = st[1 * B]; pre2 = st[2 * B]; pre3 = st[3 * B]; pre4 = st[11 * B];
  } else if (wave == 2) {
    pre0 = st[4 * B];
  } else if (wave == 3) {
    pre0 = st[5 * B]; pre1 = st[6 * B]; pre2 = st[7 * B]; pre3 = st[8 * B]; pre4 = st[9 * B]; pre5 = st[10 * B];
  }
  // ---- staging by all four waves, one barrier ----
  stage_rayleigh_tables<256>(s_g, s_h, tid);
  stage_walls<R>(a, s_w, tid, 256);
  __shared__ uint64_t s_grid[RIAB_WALL_GRID_WORDS];  // (wall-heavy rooms: RiabMotion.wall_grid)
  stage_wall_grid(a, s_grid, tid, 256);
  if (tid < 64) {
    s_v2[tid] = RIAB_T4_SENT;
    s_f[tid] = RIAB_T4_SENT;
  }
  if (tid < T4_C_WORDS) s_cnt[tid] = 0u;
  __syncthreads();
  const t4_cnt_ptr cnt = (t4_cnt_ptr)s_cnt;
  const t4_slot_ptr slot_v2 = (t4_slot_ptr)&s_v2[lane], slot_f = (t4_slot_ptr)&s_f[lane];
  bool gave_up = false;  // a wait of this wave timed out (a partner is stuck): the launch ends, the error is reported
  // Latency-bound waves sharing their SIMDs with the bandwidth-bound rate kernels' waves: win the issue arbitration
  // whenever ready — all four (a noise or tail wave that falls behind stalls the other two through the rings).
  __builtin_amdgcn_s_setprio(3);

  if (wave == 0) {
    // ================================ wave G: position and velocity ================================================
    const MotionConst<R> K = make_motion_const<R>(a, s_w, s_grid);
    const R dt = K.dt;
    R px = pre0, py = pre1;
    R vx = pre2, vy = pre3;
    R dwall = pre4;
    R drx = 0, dry = 0;
    if (m.has_drift) {
      drx = a.drift[b];
      dry = a.drift[B + b];
    }
    int n_bounce = 0, n_sat = 0, n_bc = 0;
    R v2 = norm2(vx, vy);
    *slot_v2 = (unsigned long long)__double_as_longlong(v2);
    // The loop is compiled twice.  OPEN: a solid rectangular room with no wall inside it and no drift (BASELINE
    // configs 2, 4, 5) — the four edges by coordinate differences, nothing else: no wall loops, no near-wall mask, no
    // polygon tests.  A lone wave pays ~7.5 cycles for every instruction it issues, scalar branches around unused
    // paths included: the general loop is ~370 issue slots per step in this room, the specialised one ~130.
    auto steps = [&](auto open_tag) {
      RIAB_EXACT_FP
      constexpr bool OPEN = decltype(open_tag)::value;
      uint32_t noise_ready = 0, tail_done = 0;
      const bool repel = K.nw > 0 && K.repel && !(RIAB_T4_ABLATE & 2);
      for (int t = 0; t < T; ++t) {
        if ((uint32_t)t >= noise_ready) {
          noise_ready = t4_wait_counter(cnt, T4_C_NOISE, (uint32_t)t, false);
          if (!noise_ready) { gave_up = true; break; }
        }
        T4_STAMP(0);
        const R cs = s_cs[t % RIAB_T4_RING][lane], sn = s_sn[t % RIAB_T4_RING][lane];
        const R ppx = px, ppy = py;  // prev_pos (Agent.py:199)
        // ---- _stochastic_velocity_update: the rotation (Agent.py:287-300) ----
        const bool zero_v = (v2 == (R)0);  // the reference replaces a zero velocity by (1e-8, 0) (Agent.py:299-300)
        rotate_by<R>(cs, sn, vx, vy);
        vx = zero_v ? (R)1e-8 : vx;
        vy = zero_v ? (R)0 : vy;
        // ---- wall repulsion: everything that depends on the position only ----
        NearWalls<R> near = {INFINITY, 0, 0, 0, 0, 0, 0};
        WallPush<R> push = {0, 0, 0, 0};
        if (!(RIAB_T4_ABLATE & 2)) {
          if (OPEN) box_pass1<R>(K, px, py, near);
          else near = walls_pass1<R>(K, s_w, px, py);
        }
        if (repel) {
          push = OPEN ? box_pass2_terms<R>(K, near) : walls_pass2_terms<R>(K, s_w, near, px, py);
          dwall = closest_wall_distance<R>(K, near);
        }
        // ---- the speed factor of this step from wave S ----
        T4_STAMP(1);
        R f;
        if (!t4_take(slot_f, &f)) { gave_up = true; break; }
        T4_STAMP(2);
        vx *= f;
        vy *= f;
        // ---- _drift_velocity_update (Agent.py:331-341) ----
        if (!OPEN && m.has_drift) drift_update<R>((R)m.drift_theta, drx, dry, dt, vx, vy);
        if (repel) walls_pass2_apply<R>(K, push, px, py, vx, vy);
        // ---- propose (Agent.py:216), collisions ----
        propose_step<R>(vx, vy, dt, px, py);
        if (!(RIAB_T4_ABLATE & 2)) handle_collisions<R>(K, s_w, near.x2min, ppx, ppy, px, py, vx, vy, n_bounce, n_sat);
        // the velocity of this step is final: wave S can start on the next one
        v2 = norm2(vx, vy);
        if (t + 1 < T) *slot_v2 = (unsigned long long)__double_as_longlong(v2);
        T4_STAMP(3);
        // ---- boundary safety net; the displacement as wave T needs it (Agent.py:456-458) ----
        R dpx, dpy;
        if (OPEN) {
          if (!(px > K.e0 && px < K.e1 && py > K.e2 && py < K.e3)) {
            ++n_bc;
            box_clamp<R>(K, px, py);
          }
          dpx = px - ppx;
          dpy = py - ppy;
        } else {
          if (!(RIAB_T4_ABLATE & 2)) boundary_net<R>(K, a, s_w, t, b, aid, px, py, n_bc, n_sat);
          step_displacement<R>(a, px, py, ppx, ppy, dpx, dpy);
        }
        if ((uint32_t)t >= tail_done + RIAB_T4_RING) {  // the ring slot still holds a step wave T has not taken
          const uint32_t v = t4_wait_counter(cnt, T4_C_TDONE, (uint32_t)t - RIAB_T4_RING, true);
          if (!v) { gave_up = true; break; }
          tail_done = v;
        }
        typedef double v2d __attribute__((ext_vector_type(2)));
        typedef float v2f32 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<v2d*>(&s_dp[t % RIAB_T4_RING][lane][0]) = v2d{dpx, dpy};
        *reinterpret_cast<v2f32*>(&s_pp[t % RIAB_T4_RING][lane][0]) = v2f32{(float)px, (float)py};
        asm volatile("" ::: "memory");  // (LDS executes a wave's instructions in order: data, then the counter)
        cnt[T4_C_GDONE] = (uint32_t)(t + 1);
        T4_STAMP(4);
      }
    };
    const bool open_room = K.box_fast && K.nw == 4 && !m.has_drift && !a.resample;
    if (open_room) steps(T4Tag<true>{});
    else steps(T4Tag<false>{});
    if (live) {
      t4_store_state<PUB>(st + 0 * B, px);
      t4_store_state<PUB>(st + 1 * B, py);
      t4_store_state<PUB>(st + 2 * B, vx);
      t4_store_state<PUB>(st + 3 * B, vy);
      t4_store_state<PUB>(st + 11 * B, dwall);
    }
    if (a.diag && live) {
      if (n_bounce) atomicAdd(a.diag + 0, n_bounce);
      if (n_sat) atomicAdd(a.diag + 1, n_sat);
      if (n_bc) atomicAdd(a.diag + 2, n_bc);
    }
    t4_state_stored<PUB>(s_cnt, lane);
  } else if (wave == 1) {
    // ================================ wave S: the speed chain ======================================================
    const lds_cf64_ptr lds_g = (lds_cf64_ptr)s_g, lds_h = (lds_cf64_ptr)s_h;
    const R sm_kw = (R)m.speed_mean_kw;
    const double inv_sm = 1.0 / m.speed_mean_kw;
    uint32_t noise_ready = 0;
    for (int t = 0; t < T; ++t) {
      R v2;
      if (!t4_take(slot_v2, &v2)) { gave_up = true; break; }
      T4_STAMP(5);
      if ((uint32_t)t >= noise_ready) {
        noise_ready = t4_wait_counter(cnt, T4_C_NOISE, (uint32_t)t, false);
        if (!noise_ready) { gave_up = true; break; }
      }
      const R z_spd = s_zs[t % RIAB_T4_RING][lane];
      // utils.rayleigh_to_normal / normal_to_rayleigh (utils.py:409-421), sigma = speed_mean: the expressions of
      // agent_step_body's float64 path
      if (RIAB_T4_ABLATE & 1) {
        *slot_f = (unsigned long long)__double_as_longlong(1.0 + 1e-9 * z_spd);
        continue;
      }
      if (v2 == (R)0) v2 = (R)1e-16;
      const R ispeed = r_rsqrt(v2);
      const R speed = v2 * ispeed;
      const double tG = clamp_G_arg((double)speed * inv_sm);
      const int sg = seg_G(tG);
      const SegRow<RIAB_G_DEG> grow = seg_fetch<RIAB_G_DEG>(lds_g + sg * RIAB_G_STRIDE);
      double nv64 = seg_eval<RIAB_G_DEG>(grow, tG);
      nv64 = ou_step<double>(nv64, m.speed_theta_kw, 0.0, m.speed_sigma_kw, m.dt, (double)z_spd);
      const bool h_in_table = fabs(nv64) < RIAB_H_NMAX;
      const int sh = seg_H(nv64);
      const SegRow<RIAB_H_DEG> hrow = seg_fetch<RIAB_H_DEG>(lds_h + sh * RIAB_H_STRIDE);
      const double tnew = h_in_table ? seg_eval<RIAB_H_DEG>(hrow, nv64) : sqrt(-2.0 * log(1.0 - normcdf(nv64)));
      R speed_new = (R)(m.speed_mean_kw * tnew);
      if (m.speed_std_is_zero) speed_new = sm_kw;
      const R f = speed_new * ispeed;
      *slot_f = (unsigned long long)__double_as_longlong(f);
      T4_STAMP(6);
    }
  } else if (wave == 2) {
    // ================================ wave N: noise and the rotational-velocity OU =================================
    const R dt = (R)m.dt;
    R rot = pre0;
    u32x4 pw = {0u, 0u, 0u, 0u};
    uint32_t g_done = 0;
    // explicit normals: the loads of four steps are issued together (one memory round trip per four steps)
    double zin[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int t = 0; t < T; ++t) {
      if ((uint32_t)t >= g_done + RIAB_T4_RING) {  // the ring slot still holds a step the stepping waves have not finished
        const uint32_t v = t4_wait_counter(cnt, T4_C_GDONE, (uint32_t)t - RIAB_T4_RING, true);
        if (!v) { gave_up = true; break; }
        g_done = v;
      }
      R z_rot, z_spd;
      if (IN == 1) {
        if ((t & 3) == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (t + i < T) {
              zin[i][0] = a.z_in[((int64_t)(t + i) * 2 + 0) * B + b];
              zin[i][1] = a.z_in[((int64_t)(t + i) * 2 + 1) * B + b];
            }
          }
        }
        z_rot = (t & 3) == 0 ? zin[0][0] : (t & 3) == 1 ? zin[1][0] : (t & 3) == 2 ? zin[2][0] : zin[3][0];
        z_spd = (t & 3) == 0 ? zin[0][1] : (t & 3) == 1 ? zin[1][1] : (t & 3) == 2 ? zin[2][1] : zin[3][1];
      } else if (RIAB_T4_ABLATE & 8) {
        z_rot = 0.25;
        z_spd = -0.5;
      } else {
        const MotionDraw d = motion_normals(a.step0 + (uint64_t)t, t == 0, aid, a.k0, a.k1, pw);
        pw = d.pw;
        z_rot = (R)d.z_rot;
        z_spd = (R)d.z_spd;
      }
#ifndef RIAB_T4_PROFILE
      if (a.z_out && live) {
        a.z_out[((int64_t)t * 2 + 0) * B + b] = (double)z_rot;
        a.z_out[((int64_t)t * 2 + 1) * B + b] = (double)z_spd;
      }
#endif
      // utils.ornstein_uhlenbeck on the rotational velocity (Agent.py:287-294), then sin / cos of the heading increment
      rot = ou_step<R>(rot, (R)m.rot_theta_kw, (R)m.rot_drift_kw, (R)m.rot_sigma_kw, dt, z_rot);
      R sn, cs;
      if (RIAB_T4_ABLATE & 8) {
        sn = rot * dt;
        cs = 1.0;
      } else {
        sincos_small(rot * dt, &sn, &cs);
      }
      s_cs[t % RIAB_T4_RING][lane] = cs;
      s_sn[t % RIAB_T4_RING][lane] = sn;
      s_zs[t % RIAB_T4_RING][lane] = z_spd;
      asm volatile("" ::: "memory");
      cnt[T4_C_NOISE] = (uint32_t)(t + 1);
    }
    if (live) t4_store_state<PUB>(st + 4 * B, rot);
    t4_state_stored<PUB>(s_cnt, lane);
  } else {
    // ================================ wave T: output-only tail, history rows, publication ==========================
    const TailConst<R> tail_c = {(R)m.dt, (R)(1.0 / m.dt), (R)(1.0 - m.dt / m.hd_tau), (R)(m.dt / m.hd_tau),
                                 m.hd_tau <= m.dt};
    StepTail<R> tl{pre0, pre1, pre2, pre3, pre4, pre5, 0};
#ifdef RIAB_PIPE_PROFILE
    if (PUB && lane == 0 && blockIdx.x == 0) ((unsigned long long*)(a.ctrl + 2048))[0] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
#endif
    if (PUB && lane == 0) {
      // This workgroup is resident.  Before it says so its progress word goes back to "no row of THIS launch yet":
      // the words hold absolute step counts, and a launch that replays earlier steps (the same argument block run
      // again) would otherwise let a consumer take the earlier run's count for its own.  Whoever has waited for the
      // announcement (the started gate) finds the word reset; launches that continue where the last one ended never
      // see a word above their own rows anyway.
      __hip_atomic_store((riab_gu32*)(uintptr_t)(a.ctrl + RIAB_CTRL_PROGRESS_WORD(blockIdx.x)), (uint32_t)a.step0, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      if (blockIdx.x == 0)   // (device clock at the start of the trajectory: RIAB_STREAMER_OPT_STEP_NS measures with it)
        __hip_atomic_store((riab_gu64*)(uintptr_t)(a.ctrl + RIAB_CTRL_TRAJ_STAMPS), (unsigned long long)__builtin_amdgcn_s_memrealtime(),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      atomicAdd(a.ctrl + RIAB_CTRL_STARTED, 1u);
    }
    // PUB: rows of steps < n have left this wave write-through and been acknowledged: the consumer may read them
    auto publish = [&](int n) {
#ifdef RIAB_PIPE_PROFILE  // (tools/pipe_profile.py)
      if (lane == 0 && (blockIdx.x == 0 || blockIdx.x + 1 == gridDim.x) && n <= 64)
        ((unsigned long long*)(a.ctrl + 2048))[(blockIdx.x == 0 ? 0 : 4) * 64 + n] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
#endif
      if (lane == 0)
        __hip_atomic_store((riab_gu32*)(uintptr_t)(a.ctrl + RIAB_CTRL_PROGRESS_WORD(blockIdx.x)), (uint32_t)a.step0 + (uint32_t)n,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // Blocks of rows: the first four steps of a launch leave one by one (a consumer that is already waiting gets its
    // first row after one step, not four), then four steps per block (eight float4 row stores).  Short launches
    // publish a block as soon as its stores are acknowledged; long ones (the rate stage saturates HBM, a write-through
    // then takes microseconds) one block later, when that wait is free.
    const bool early_pub = T <= 64;
    uint32_t g_done = 0;
    int t0 = 0, pending = -1;  // pending: steps covered by flushed but unpublished blocks (-1: none)
    bool ok = true;
    while (t0 < T && ok) {
      const int n = (t0 < a.pub_single) ? 1 : min(4, T - t0);
      for (int i = 0; i < n; ++i) {
        const int t = t0 + i;
        if ((uint32_t)t >= g_done) {
          g_done = t4_wait_counter(cnt, T4_C_GDONE, (uint32_t)t, true);
          if (!g_done) {
            ok = false;
            gave_up = true;
            break;
          }
        }
        const int r = t % RIAB_T4_RING;
        if (!(RIAB_T4_ABLATE & 16))
          tl = step_tail<R>(tl, s_dp[r][lane][0], s_dp[r][lane][1], tail_c, a.step0 + (uint64_t)t, aid, a.k0, a.k1);
        if (a.hist) {
          float* sh = &s_hist[i * RIAB_HIST_ROWS * 64 + lane];
          sh[0 * 64] = s_pp[r][lane][0];
          sh[1 * 64] = s_pp[r][lane][1];
          sh[2 * 64] = (float)tl.mvx;
          sh[3 * 64] = (float)tl.mvy;
          sh[4 * 64] = (float)tl.hx;
          sh[5 * 64] = (float)tl.hy;
          sh[6 * 64] = (float)tl.mrot;
          sh[7 * 64] = (float)tl.dist;
        }
      }
      if (!ok) break;
      asm volatile("" ::: "memory");
      cnt[T4_C_TDONE] = (uint32_t)(t0 + n);  // (the ring slots of this block have been read)
      if (PUB && !early_pub && pending >= 0) {
        // the previous block's rows were stored a whole block ago: acknowledged by now, the wait is free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(pending);
        pending = -1;
      }
      if (a.hist && !(RIAB_T4_ABLATE & 4)) {
        __builtin_amdgcn_wave_barrier();  // (LDS serves a wave's requests in order: the reads below see the writes)
        t4_flush_hist<PUB>(a, s_hist, lane, t0, n);
        __builtin_amdgcn_wave_barrier();
      }
      t0 += n;
      if (PUB && t0 < T) {   // (the launch's last publication follows the state, below)
        if (early_pub) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          publish(t0);
        } else {
          pending = t0;
        }
      }
    }
    if (live) {
      t4_store_state<PUB>(st + 5 * B, (double)tl.mvx);
      t4_store_state<PUB>(st + 6 * B, (double)tl.mvy);
      t4_store_state<PUB>(st + 7 * B, (double)tl.mrot);
      t4_store_state<PUB>(st + 8 * B, (double)tl.hx);
      t4_store_state<PUB>(st + 9 * B, (double)tl.hy);
      t4_store_state<PUB>(st + 10 * B, (double)tl.dist);
      if (a.diag && tl.n_still) atomicAdd(a.diag + 3, tl.n_still);
    }
    if (PUB && ok) {
      // every row, this wave's part of the state, and — once the position / velocity wave and the noise wave have
      // reported theirs — the whole state of these 64 agents is in memory: the publication that ends the launch
      // (device clock at the end of workgroup 0's work, stored BEFORE the wait below so that whoever sees the last
      // publication finds this launch's stamp, not the previous one's)
      if (lane == 0 && blockIdx.x == 0)
        __hip_atomic_store((riab_gu64*)(uintptr_t)(a.ctrl + RIAB_CTRL_TRAJ_STAMPS + 2), (unsigned long long)__builtin_amdgcn_s_memrealtime(),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (t4_wait_counter(cnt, T4_C_STATE, 1u, true)) publish(T);
      else gave_up = true;
    }
  }
  if (gave_up && lane == 0) {
    // report it like the consumers' timeouts (Agent.diagnostics["pipeline_timeouts"]; diag[1]: saturations)
    if (PUB) {
      atomicAdd(a.ctrl + RIAB_CTRL_TIMEOUTS, 1u);
      __hip_atomic_store((riab_gu32*)(uintptr_t)(a.ctrl + RIAB_CTRL_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.diag) atomicAdd(a.diag + 1, 1);
  }
}

}  // namespace riab
