// The closed-loop step in ONE launch (gfx950): Agent.update() + Neurons.update() of one store-bound population.
//
// The reference's API is the per-step loop `Ag.update(); PCs.update()` (reference Agent.py:160-242, Neurons.py:145-171;
// contribs/TaskEnvironment.py:399-408).  As two kernels a step costs two dependent, latency-bound launches: 6.4 us for
// a motion kernel of 64 waves + 6.8 us for a one-row rate kernel that is all ramp and drain.  One grid
// cannot hand rows from "motion workgroups" to "rate workgroups" without a round trip through memory and a poll, so
// this kernel does not hand anything over: EVERY workgroup advances the 256 agents whose rates it is about to write —
// the same instruction stream on the same inputs in every workgroup of an agent segment, hence the same bits — keeps
// their new positions in LDS, and then writes its own (256 agents) x (cell chunk) tile of the population's row with the
// rate kernels' store pattern (a lane owns four agents, 1 KiB per wave store).  The motion arithmetic costs the same
// wall time whether one workgroup or thirty-two run it (a step is a dependent chain, not throughput), and the chip is
// otherwise idle while it runs.
//
//   grid  = (B / 256 agent segments, cell chunks), x fastest; block = 256 threads = 4 waves
//   wave w of workgroup (x, y) evaluates the cell groups (4 y + w) * reps .. + reps - 1 (CPB cells each) for the
//   segment's 256 agents; thread i advances agent 256 x + i.
//
// Only the workgroups with blockIdx.y == 0 (the "writers") store what a step leaves behind: the history row, the
// float64 state, the diagnostics.  The state is read by all of a segment's workgroups and written in place by one, so
// the writer stores it LAST and only after every other workgroup of its segment has reported — one word per workgroup
// in `sync`, holding the launch's epoch — that its state loads have returned.  Nobody waits for the writer, the
// writer waits for workgroups that wait for nobody: no forward-progress assumption beyond "every workgroup of a grid
// is eventually dispatched", no co-residency requirement, capturable, re-entrant per plan.
//
// Values: the motion step is assembled from the functions of riab_agent_kernel.h in the order of agent_step_body (the
// kernel `riab_agent_step(T = 1)` launches), contraction off; the rates come from the functors of riab_rate_cells.h as
// rate_kernel_wide evaluates them.  Bit-identical to the two-launch step (tests/test_gpu_step1.py).
#include "riab_agent_kernel.h"
#include "riab_rate_cells.h"

namespace riab {

struct Step1Sync {
  uint32_t* words;     // [segments][RIAB_STEP1_SYNC_STRIDE] arrival words + [RIAB_STEP1_SYNC_TAIL] counters behind them
  uint32_t epoch;      // this launch's tag (never 0, never repeated on the same words)
  uint32_t spin_limit;
  uint32_t n_segments;
};

typedef __attribute__((address_space(1))) uint32_t s1_gu32;

template <class Cell, int SPK, int CPB, bool NT>
__global__ __launch_bounds__(256, 2) void step1_kernel(const AgentArgs a, const RateArgs ra, Cell cell, const Step1Sync sy,
                                                       const int reps) {
  RIAB_EXACT_FP
  constexpr int NP = Cell::NP;
  static_assert(NP * CPB <= 64, "a cell group's parameters must fit one wave");
  __shared__ Wall<double> s_w[RIAB_MAX_WALLS];
  __shared__ double s_g[RIAB_G_SEGS * RIAB_G_STRIDE];
  __shared__ double s_h[RIAB_H_SEGS * RIAB_H_STRIDE];
  __shared__ __align__(16) float s_row[4][256];  // x, y, head direction x, y of the segment's agents as the history keeps them
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool writer = blockIdx.y == 0;
  const int64_t B = a.B;
  const int64_t b = (int64_t)blockIdx.x * 256 + tid;  // (B is a multiple of 256: whole workgroups)
  const RiabMotion& m = a.m;

  // ---- everything that does not depend on anything: the tables into LDS, the state into registers — one round trip
  double* const st = a.state + b;
  double px = st[0 * B], py = st[1 * B];
  double vx = st[2 * B], vy = st[3 * B];
  double rot = st[4 * B];
  double mvx = st[5 * B], mvy = st[6 * B];
  double mrot = st[7 * B];
  double hx = st[8 * B], hy = st[9 * B];
  double dist = st[10 * B];
  double dwall = st[11 * B];
  double drx = 0, dry = 0;
  if (m.has_drift) {
    drx = a.drift[b];
    dry = a.drift[B + b];
  }
  const int g0 = (int)(blockIdx.y * 4u + (uint32_t)wave) * reps;  // this wave's first cell group
  auto group_params = [&](int g) -> float {
    const int pi = g * CPB * NP + lane;
    return (lane < NP * CPB && pi < ra.n * NP) ? cell.tab[pi] : 0.0f;
  };
  float mine = group_params(g0);
  stage_rayleigh_tables<256>(s_g, s_h, tid);
  stage_walls<double>(a, s_w, tid, 256);
  __syncthreads();

  // ---- Agent.update for agent b (agent_step_body's step, T = 1, Philox noise, tables in LDS) -----------------------
  const lds_cf64_ptr lds_g = (lds_cf64_ptr)s_g, lds_h = (lds_cf64_ptr)s_h;
  const double dt = m.dt;
  const int nw = a.n_walls;
  int n_bounce = 0, n_sat = 0, n_bc = 0, n_still = 0;
  const uint32_t aid = (uint32_t)(a.agent_id0 + b);
  const MotionConst<double> K = make_motion_const<double>(a, s_w);
  const TailConst<double> tail_c = {dt, K.inv_dt, 1.0 - m.dt / m.hd_tau, m.dt / m.hd_tau, m.hd_tau <= m.dt};
  {
    u32x4 pw = {0u, 0u, 0u, 0u};
    const MotionDraw d = motion_normals(a.step0, true, aid, a.k0, a.k1, pw);
    const double z_rot = (double)d.z_rot, z_spd = (double)d.z_spd;
    const double ppx = px, ppy = py;  // prev_pos (Agent.py:199)
    // ---- _stochastic_velocity_update (Agent.py:287-312)
    double v2 = norm2(vx, vy);
    const bool zero_v = (v2 == 0.0);
    if (zero_v) v2 = 1e-16;  // the reference replaces a zero velocity by (1e-8, 0) (Agent.py:299-300)
    const double ispeed = r_rsqrt(v2);
    const double speed = v2 * ispeed;
    const double tG = clamp_G_arg(speed * K.inv_sm);
    const SegRow<RIAB_G_DEG> grow = seg_fetch<RIAB_G_DEG>(lds_g + seg_G(tG) * RIAB_G_STRIDE);
    rot = ou_step<double>(rot, m.rot_theta_kw, m.rot_drift_kw, m.rot_sigma_kw, dt, z_rot);
    {
      double sn, cs;
      sincos_small(rot * dt, &sn, &cs);
      rotate_by<double>(cs, sn, vx, vy);
      vx = zero_v ? 1e-8 : vx;
      vy = zero_v ? 0.0 : vy;
    }
    // utils.rayleigh_to_normal / normal_to_rayleigh (utils.py:409-421), sigma = speed_mean
    double nv64 = seg_eval<RIAB_G_DEG>(grow, tG);
    nv64 = ou_step<double>(nv64, m.speed_theta_kw, 0.0, m.speed_sigma_kw, m.dt, z_spd);
    const bool h_in_table = fabs(nv64) < RIAB_H_NMAX;
    const SegRow<RIAB_H_DEG> hrow = seg_fetch<RIAB_H_DEG>(lds_h + seg_H(nv64) * RIAB_H_STRIDE);
    // ---- _wall_velocity_update, pass 1
    const NearWalls<double> near = walls_pass1<double>(K, s_w, px, py);
    // ---- finish the speed update
    double speed_new;
    {
      const double tnew = h_in_table ? seg_eval<RIAB_H_DEG>(hrow, nv64) : sqrt(-2.0 * log(1.0 - normcdf(nv64)));
      speed_new = m.speed_mean_kw * tnew;
    }
    if (m.speed_std_is_zero) speed_new = K.sm_kw;
    {
      const double f = speed_new * ispeed;
      vx *= f;
      vy *= f;
    }
    // ---- _drift_velocity_update (Agent.py:331-341)
    if (m.has_drift) drift_update<double>(m.drift_theta, drx, dry, dt, vx, vy);
    // ---- _wall_velocity_update, pass 2
    if (nw > 0 && K.repel) {
      const WallPush<double> push = walls_pass2_terms<double>(K, s_w, near, px, py);
      dwall = closest_wall_distance<double>(K, near);
      walls_pass2_apply<double>(K, push, px, py, vx, vy);
    }
    // ---- propose (Agent.py:216), collisions, boundary safety net
    propose_step<double>(vx, vy, dt, px, py);
    handle_collisions<double>(K, s_w, near.x2min, ppx, ppy, px, py, vx, vy, n_bounce, n_sat);
    boundary_net<double>(K, a, s_w, 0, b, aid, px, py, n_bc, n_sat);
    // ---- the output-only tail (Agent.py:456-507)
    double dpx, dpy;
    step_displacement<double>(a, px, py, ppx, ppy, dpx, dpy);
    StepTail<double> tl{mvx, mvy, mrot, hx, hy, dist, n_still};
    tl = step_tail<double>(tl, dpx, dpy, tail_c, a.step0, aid, a.k0, a.k1);
    mvx = tl.mvx; mvy = tl.mvy; mrot = tl.mrot; hx = tl.hx; hy = tl.hy; dist = tl.dist; n_still = tl.n_still;
  }
  s_row[0][tid] = (float)px;
  s_row[1][tid] = (float)py;
  if (Cell::NEEDS_HD) {
    s_row[2][tid] = (float)hx;
    s_row[3][tid] = (float)hy;
  }
  if (writer && a.hist) {  // save_to_history (Agent.py:514-520)
    float* h = a.hist + b;
    h[0 * B] = (float)px;
    h[1 * B] = (float)py;
    h[2 * B] = (float)mvx;
    h[3 * B] = (float)mvy;
    h[4 * B] = (float)hx;
    h[5 * B] = (float)hy;
    h[6 * B] = (float)mrot;
    h[7 * B] = (float)dist;
  }
  __syncthreads();  // the row is in LDS; every state value of this workgroup has been consumed, i.e. loaded
  if (!writer && tid == 0)
    __hip_atomic_store((s1_gu32*)(uintptr_t)(sy.words + (int64_t)blockIdx.x * RIAB_STEP1_SYNC_STRIDE + blockIdx.y), sy.epoch,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- Neurons.update of the population: this wave's cell groups for the segment's 256 agents ----------------------
  {
    const v4f rx = *reinterpret_cast<const v4f*>(&s_row[0][4 * lane]), ry = *reinterpret_cast<const v4f*>(&s_row[1][4 * lane]);
    v4f rhx = {0.0f, 0.0f, 0.0f, 0.0f}, rhy = rhx;
    if (Cell::NEEDS_HD) {
      rhx = *reinterpret_cast<const v4f*>(&s_row[2][4 * lane]);
      rhy = *reinterpret_cast<const v4f*>(&s_row[3][4 * lane]);
    }
    const typename Cell::Pos P = cell.from_rows(rx, ry, rhx, rhy);
    const uint32_t q = blockIdx.x * 64u + (uint32_t)lane;  // the lane's quad of agents within the row
    const uint32_t group = ra.group0 + q;
    for (int r = 0; r < reps; ++r) {
      const int c0 = (g0 + r) * CPB;
      if (c0 >= ra.n) break;  // wave-uniform
      const float cur = mine;
      if (r + 1 < reps) mine = group_params(g0 + r + 1);  // (requested before this group's stores are issued)
      int64_t off = (int64_t)c0 * B + 4 * (int64_t)q;
#pragma unroll
      for (int j = 0; j < CPB; ++j) {
        if (c0 + j < ra.n) {  // wave-uniform
          float p[NP];
#pragma unroll
          for (int i = 0; i < NP; ++i)
            p[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur), j * NP + i));
          v4f rr = cell.eval(p, P);
          rr = finish_rate(rr * ra.fr_scale + ra.fr_min, P);  // [0,1] -> [min_fr, max_fr]
          if (NT) __builtin_nontemporal_store(rr, reinterpret_cast<v4f*>(ra.rates + off));
          else *reinterpret_cast<v4f*>(ra.rates + off) = rr;
          if (SPK == 1) spike_store<false>(ra, rr, off, ra.step0, (uint32_t)(c0 + j), group);
          off += B;
        }
      }
    }
  }
  if (!writer) return;

  // ---- the writer: the state in place, once nobody can read the old one any more -------------------------------------
  if (wave == 0) {
    const uint32_t others = gridDim.y - 1u;  // (<= RIAB_STEP1_SYNC_STRIDE - 1: one lane per word)
    bool timed_out = false;
    for (uint32_t spins = 0;; ++spins) {
      uint32_t v = sy.epoch;
      if ((uint32_t)lane < others)
        v = __hip_atomic_load((s1_gu32*)(uintptr_t)(sy.words + (int64_t)blockIdx.x * RIAB_STEP1_SYNC_STRIDE + 1 + lane), __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_ballot_w64(v != sy.epoch) == 0) break;
      if (spins >= sy.spin_limit) {  // (a workgroup of this grid that never ran: nothing sane to do but to say so)
        timed_out = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    if (timed_out && lane == 0)
      atomicAdd(sy.words + (int64_t)sy.n_segments * RIAB_STEP1_SYNC_STRIDE + RIAB_STEP1_SYNC_TIMEOUTS, 1u);
  }
  __syncthreads();
  st[0 * B] = px;
  st[1 * B] = py;
  st[2 * B] = vx;
  st[3 * B] = vy;
  st[4 * B] = rot;
  st[5 * B] = mvx;
  st[6 * B] = mvy;
  st[7 * B] = mrot;
  st[8 * B] = hx;
  st[9 * B] = hy;
  st[10 * B] = dist;
  st[11 * B] = dwall;
  if (a.diag) {
    if (n_bounce) atomicAdd(a.diag + 0, n_bounce);
    if (n_sat) atomicAdd(a.diag + 1, n_sat);
    if (n_bc) atomicAdd(a.diag + 2, n_bc);
    if (n_still) atomicAdd(a.diag + 3, n_still);
  }
}

// ---- host side --------------------------------------------------------------------------------------------------
// how a (B, n) problem is cut: `reps` cell groups per wave so that the grid is about one wave of workgroups (two per
// compute unit: 512 on MI355X) — a second round of workgroups would run the motion step a second time — and a
// segment's workgroups fit its line of arrival words
static void step1_shape(int64_t B, int n, int cpb, dim3* grid, int* reps) {
  const int64_t segs = B / 256;
  const int64_t groups = (n + cpb - 1) / cpb;
  int64_t want_y = 512 / segs;  // workgroups per segment in one resident round
  if (want_y < 1) want_y = 1;
  if (want_y > RIAB_STEP1_SYNC_STRIDE) want_y = RIAB_STEP1_SYNC_STRIDE;
  int64_t r = (groups + 4 * want_y - 1) / (4 * want_y);
  if (r < 1) r = 1;
  int64_t gy = (groups + 4 * r - 1) / (4 * r);
  *reps = (int)r;
  *grid = dim3((unsigned)segs, (unsigned)gy, 1);
}

template <class Cell>
static int launch_step1_cell(const AgentArgs& a, const RateArgs& ra, const Cell& cell, const Step1Sync& sy, bool spikes,
                             bool nt, hipStream_t s) {
  constexpr int CPB = (Cell::NP * 2 * Cell::CPB <= 64) ? 2 * Cell::CPB : Cell::CPB;  // (as the row-following kernel)
  dim3 grid;
  int reps;
  step1_shape(a.B, ra.n, CPB, &grid, &reps);
  if (spikes) {
    if (nt) hipLaunchKernelGGL((step1_kernel<Cell, 1, CPB, true>), grid, dim3(256), 0, s, a, ra, cell, sy, reps);
    else hipLaunchKernelGGL((step1_kernel<Cell, 1, CPB, false>), grid, dim3(256), 0, s, a, ra, cell, sy, reps);
  } else {
    if (nt) hipLaunchKernelGGL((step1_kernel<Cell, 0, CPB, true>), grid, dim3(256), 0, s, a, ra, cell, sy, reps);
    else hipLaunchKernelGGL((step1_kernel<Cell, 0, CPB, false>), grid, dim3(256), 0, s, a, ra, cell, sy, reps);
  }
  return (int)hipGetLastError();
}

// 0 when the one-launch step covers this population (the store-bound kinds whose functor needs no LDS of its own)
int step1_supported(const RiabEnv* env, const RiabPopulation* pop, int64_t B) {
  if (!env || !pop || pop->n <= 0 || B <= 0 || B % 256 != 0 || !pop->table) return RIAB_EUNSUPPORTED;
  if (pop->noise_state) return RIAB_EUNSUPPORTED;  // (the OU noise pass and the spikes drawn on its result are kernels of their own)
  if (B / 256 > 65535) return RIAB_EUNSUPPORTED;
  switch (pop->kind) {
    case RIAB_POP_PLACE:
      if (pop->description == RIAB_PC_ONE_HOT) return RIAB_EUNSUPPORTED;
      if (pop->geometry != RIAB_GEOM_EUCLIDEAN) return RIAB_EUNSUPPORTED;  // (line of sight / geodesic: 88-96 registers + walls in LDS)
      return RIAB_OK;
    case RIAB_POP_GRID: return (pop->description == RIAB_GC_RECTIFIED || pop->description == RIAB_GC_SHIFTED) ? RIAB_OK : RIAB_EUNSUPPORTED;
    case RIAB_POP_HDC: return RIAB_OK;
    default: return RIAB_EUNSUPPORTED;
  }
}

// one Agent.update() (the arguments of riab_agent_step(T = 1), Philox noise) + the population's update() on the row it
// writes: `rates_row` / `spikes_row` are the population's rows of this step, `step_after` the number of agent steps
// taken once this one is done (Neurons.update's spike counter, as riab_plan_step passes it)
int launch_step1(const AgentArgs& a, const RiabEnv* env, const RiabPopulation* pop, float* rates_row, uint8_t* spikes_row,
                 uint64_t seed, uint64_t step_after, uint32_t* sync_words, uint32_t epoch, hipStream_t s) {
  const int rc = step1_supported(env, pop, a.B);
  if (rc) return rc;
  if (a.z_in || a.z_out || a.forced || a.T != 1 || !sync_words || !rates_row || epoch == 0u) return RIAB_EINVAL;
  if ((((uintptr_t)rates_row) & 15) || (((uintptr_t)spikes_row) & 3) || a.agent_id0 % 4) return RIAB_EALIGN;
  RateArgs ra;
  ra.pos_x = ra.pos_y = ra.hd_x = ra.hd_y = nullptr;  // (the row comes through LDS)
  ra.pos_ld = 0;
  ra.qrow = a.B / 4;
  ra.nquads = ra.qrow;
  ra.B = a.B;
  ra.rates = rates_row;
  ra.spikes = spikes_row;
  ra.u_in = nullptr;
  ra.dt = (float)a.m.dt;
  ra.fr_scale = pop->io.max_fr - pop->io.min_fr;
  ra.fr_min = pop->io.min_fr;
  ra.k0 = (uint32_t)seed;
  ra.k1 = (uint32_t)(seed >> 32);
  ra.step0 = (uint32_t)step_after;
  ra.tag = RIAB_TAG_SPIKES | ((uint32_t)pop->io.pop_id & 0xFFu);
  ra.group0 = (uint32_t)(a.agent_id0 / 4);
  ra.n = pop->n;
  ra.cells_per_block = 0;
  Step1Sync sy;
  sy.words = sync_words;
  sy.epoch = epoch;
  sy.spin_limit = 1u << 22;  // x ~0.3 us: about a second
  sy.n_segments = (uint32_t)(a.B / 256);
  const bool spikes = spikes_row != nullptr;
  const bool nt = g_options[RIAB_OPT_FUSED_STEP] != 2;
  switch (pop->kind) {
    case RIAB_POP_PLACE: {
      PlaceCell<RIAB_PC_GAUSSIAN, 0> c;
      c.tab = pop->table;
      c.scale = (float)env->scale;
      c.half_scale = (float)(env->scale / 2);
      c.top_hat_w2 = pop->top_hat_width * pop->top_hat_width;
      c.walls = env->walls;
      c.n_internal = 0;
      c.e0 = env->extent[0]; c.e1 = env->extent[1]; c.e2 = env->extent[2]; c.e3 = env->extent[3];
      c.shape = make_env_shape(env);
      c.lds = nullptr;
      if (env->periodic) {
        PlaceCell<RIAB_PC_GAUSSIAN, 3> w;
        w.tab = c.tab; w.scale = c.scale; w.half_scale = c.half_scale; w.top_hat_w2 = c.top_hat_w2; w.walls = c.walls;
        w.n_internal = 0; w.e0 = c.e0; w.e1 = c.e1; w.e2 = c.e2; w.e3 = c.e3; w.shape = c.shape; w.lds = nullptr;
        switch (pop->description) {
          case RIAB_PC_GAUSSIAN: return launch_step1_cell(a, ra, w, sy, spikes, nt, s);
          case RIAB_PC_GAUSSIAN_THRESHOLD: return launch_step1_cell(a, ra, w.as<RIAB_PC_GAUSSIAN_THRESHOLD>(), sy, spikes, nt, s);
          case RIAB_PC_DIFF_OF_GAUSSIANS: return launch_step1_cell(a, ra, w.as<RIAB_PC_DIFF_OF_GAUSSIANS>(), sy, spikes, nt, s);
          case RIAB_PC_TOP_HAT: return launch_step1_cell(a, ra, w.as<RIAB_PC_TOP_HAT>(), sy, spikes, nt, s);
          default: return RIAB_EUNSUPPORTED;
        }
      }
      switch (pop->description) {
        case RIAB_PC_GAUSSIAN: return launch_step1_cell(a, ra, c, sy, spikes, nt, s);
        case RIAB_PC_GAUSSIAN_THRESHOLD: return launch_step1_cell(a, ra, c.as<RIAB_PC_GAUSSIAN_THRESHOLD>(), sy, spikes, nt, s);
        case RIAB_PC_DIFF_OF_GAUSSIANS: return launch_step1_cell(a, ra, c.as<RIAB_PC_DIFF_OF_GAUSSIANS>(), sy, spikes, nt, s);
        case RIAB_PC_TOP_HAT: return launch_step1_cell(a, ra, c.as<RIAB_PC_TOP_HAT>(), sy, spikes, nt, s);
        default: return RIAB_EUNSUPPORTED;
      }
    }
    case RIAB_POP_GRID:
      if (pop->description == RIAB_GC_RECTIFIED) {
        GridCell<RIAB_GC_RECTIFIED> c{pop->table, pop->f0, 1.0f / (1.0f - pop->f0)};
        return launch_step1_cell(a, ra, c, sy, spikes, nt, s);
      } else {
        GridCell<RIAB_GC_SHIFTED> c{pop->table, pop->f0, 1.0f};
        return launch_step1_cell(a, ra, c, sy, spikes, nt, s);
      }
    case RIAB_POP_HDC: {
      HDCell<0> c{pop->table, 0.0f, nullptr, nullptr};
      return launch_step1_cell(a, ra, c, sy, spikes, nt, s);
    }
    default: return RIAB_EUNSUPPORTED;
  }
}

}  // namespace riab
