// The closed-loop step in ONE launch (gfx950): Agent.update() + Neurons.update() of one store-bound population.
//
// The reference's API is the per-step loop `Ag.update(); PCs.update()` (reference Agent.py:160-242, Neurons.py:145-171;
// contribs/TaskEnvironment.py:399-408).  As two kernels a step costs two dependent, latency-bound launches: 6.4 us for
// a motion kernel of 64 waves + 6.8 us for a one-row rate kernel that is all ramp and drain (13.7 us per step at cfg 2).
// One grid cannot hand rows from "motion workgroups" to "rate workgroups" without a round trip through memory and a
// poll, so this kernel does not hand anything over: EVERY workgroup advances the 256 agents whose rates it is about to
// write — the same instruction stream on the same inputs in every workgroup of an agent segment, hence the same bits —
// keeps their new positions in LDS, and then writes its own (256 agents) x (cell chunk) tile of the population's row
// with the rate kernels' store pattern (a lane owns four agents, 1 KiB per wave store).  A step is a dependent chain,
// not throughput: it costs the same wall time whether one workgroup or sixteen run it, and the chip is otherwise idle.
//
//   grid  = (B / 256 agent segments, cell chunks), x fastest; block = 512 threads = 8 waves, one workgroup per
//   compute unit (cfg 2: 16 x 16 = 256 workgroups).  Waves 0-3 ("movers") advance the segment's 256 agents, one mover
//   wave per SIMD — with all eight waves moving, two float64 chains shared every SIMD and the step took 0.7 us longer —;
//   wave w of workgroup (x, y) then evaluates the cell groups (8 y + w) * reps .. + reps - 1 (CPB cells each).
//
// Only the workgroups with blockIdx.y == 0 (the "writers") store what a step leaves behind: the history row, the
// float64 state, the diagnostics.  The state is read by all of a segment's workgroups and written in place by one, so
// the writer stores it LAST and only after every other workgroup of its segment has reported — one word per workgroup
// in `sync`, holding the launch's epoch — that its state loads have returned.  Nobody waits for the writer, the
// writer waits for workgroups that wait for nobody: no forward-progress assumption beyond "every workgroup of a grid
// is eventually dispatched", no co-residency requirement, capturable, re-entrant per plan.
//
// What a step must not pay per thread is prepared outside it: the launch's scalar constants by the host (six float64
// divisions), the wall table in the kernels' form and the box fast path's verdict once per plan (walls_prepare_kernel).
//
// Values: the motion step is assembled from the functions of riab_agent_kernel.h in the order of agent_step_body (the
// kernel `riab_agent_step(T = 1)` launches), contraction off; the rates come from the functors of riab_rate_cells.h as
// rate_kernel_wide evaluates them.  Bit-identical to the two-launch step (tests/test_gpu_step1.py).
// [MI355X] cfg 2 (4096 agents x 1024 PlaceCells): 9.0-9.2 us per step against 13.7; where it goes and what was tried:
// DESIGN.md 3.9, docs/EXPERIMENTS.md r05.
#include <map>
#include <mutex>
#include <utility>
#include "riab_agent_kernel.h"
#include "riab_rate_cells.h"
#include "riab_task_world_kernel.h"  // (last: riab_task_kernel.h, which it includes, turns fp contraction off for its own code)

namespace riab {

struct Step1Sync {
  uint32_t* words;     // [segments][RIAB_STEP1_SYNC_STRIDE] arrival words + [RIAB_STEP1_SYNC_TAIL] counters behind them
  uint32_t epoch;      // this launch's tag (never 0, never repeated on the same words)
  uint32_t spin_limit;
  uint32_t n_segments;
  const Wall<double>* walls;  // [n_walls] the wall table as the kernels keep it, prepared once per plan (walls_prepare_kernel)
};

// The populations whose update() rides in the launch (round 6: every store-bound one of the plan, not only the
// largest).  The cell-group axis of the grid runs over the populations one after the other: population i owns the global
// groups [group0, group0 + n_groups), a group being `cpb` consecutive cells of it (cpb a compile-time property of the
// population's functor, s1_cpb); a wave finds the population of each of its groups (wave-uniform) and switches into the
// functor's pass — the same inlined code on the same operands as the population's own kernel, hence the same bits.
// functor ids: PlaceCells 0 .. 7 = 4 * periodic + (gaussian 0, gaussian_threshold 1, diff_of_gaussians 2, top_hat 3);
// GridCells 8 + description; HeadDirectionCells 10
enum { S1_KIND_PC = 0, S1_KIND_GC = 8, S1_KIND_HDC = 10 };
struct Step1Pop {
  const float* tab;   // the population's parameter table [n][NP]
  float* rates;       // its row of this step
  uint8_t* spikes;    // ... or null
  int32_t n, kind;
  int32_t group0, n_groups;
  float fr_scale, fr_min;
  float p0, p1, p2;   // PlaceCells: scale, half_scale (periodic wrap), top_hat_w2; GridCells: f0, 1 / (1 - f0)
  uint32_t tag;       // RIAB_TAG_SPIKES | pop_id
};
struct Step1Pops {
  int32_t n_pops, total_groups;
  int32_t needs_hd;      // a population reads the head direction: the movers leave it in LDS too
  float dt;
  uint32_t k0, k1;       // Philox key (seed)
  uint32_t step0;        // agent steps taken once this one is done (Neurons.update's spike counter)
  uint32_t quad0;        // agent_id0 / 4
  Step1Pop pop[RIAB_STEP1_MAX_POPS];
};
// cells per group of a functor: as the row-following kernel, twice the wide kernel's where a wave's 64 lanes hold the
// parameters; a task's step (15 of a segment's 16 workgroups write rates) a sixteenth more, so that a population that
// filled one round of workgroups stays one round: cfg 2, 1024 cells: 114 groups of 9 on 15 x 8 waves instead of 128 of 8
// (SPK — the kernel draws spikes —: a cell then costs its Philox block whatever its functor costs, and a wave's time is its
// CELLS: HeadDirectionCells, 32 to a group where only stores count, would make the waves that hold them four times as long
// as a wave of PlaceCells groups — the step kernel of BASELINE configs[4]'s closed loop ran 31.9 us for that —: 8)
template <class Cell, bool TASK, bool SPK>
struct S1Cpb {
  static constexpr int WIDE = (Cell::NP * 2 * Cell::CPB <= 64) ? 2 * Cell::CPB : Cell::CPB;
  static constexpr int BASE = (SPK && WIDE > 8) ? 8 : WIDE;
  static constexpr int T16 = (BASE * 16 + 14) / 15;
  static constexpr int value = (TASK && Cell::NP * T16 <= 64) ? T16 : BASE;
};
template <class Cell>
__host__ __device__ static inline int s1_cpb_of(bool task, bool spk) {
  return task ? (spk ? S1Cpb<Cell, true, true>::value : S1Cpb<Cell, true, false>::value)
              : (spk ? S1Cpb<Cell, false, true>::value : S1Cpb<Cell, false, false>::value);
}
__host__ __device__ static inline int s1_cpb(int kind, bool task, bool spk) {
  if (kind < S1_KIND_GC) return s1_cpb_of<PlaceCell<0, 0>>(task, spk);
  if (kind < S1_KIND_HDC) return s1_cpb_of<GridCell<0>>(task, spk);
  return s1_cpb_of<HDCell<0>>(task, spk);
}
__host__ __device__ static inline int s1_np(int kind) { return kind < S1_KIND_GC ? PlaceCell<0, 0>::NP : kind < S1_KIND_HDC ? GridCell<0>::NP : HDCell<0>::NP; }

// Once per plan (its first one-launch step): Wall<double>[n_walls] from the plan's wall table — a wall costs two float64
// divisions and a square root, which a staging wave would otherwise pay in front of every step's first barrier — and,
// behind the walls, the verdict of the box fast path (make_motion_const's check of the first four walls, ~600
// instructions and thirty branches that every thread of every step would otherwise repeat): one word, 1 = box.
__global__ __launch_bounds__(64) void walls_prepare_kernel(const AgentArgs a, Wall<double>* out) {
  __shared__ Wall<double> s_w[RIAB_MAX_WALLS];
  stage_walls<double>(a, s_w, (int)threadIdx.x, 64);
  __syncthreads();
  for (int w = (int)threadIdx.x; w < a.n_walls; w += 64) out[w] = s_w[w];
  const MotionConst<double> k = make_motion_const<double>(a, s_w);
  if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(out + RIAB_MAX_WALLS) = k.box_fast ? 1u : 0u;
}

typedef __attribute__((address_space(1))) uint32_t s1_gu32;
// A wait that gave up (one lane): counted, and the step it happened in remembered (first / last, as "agent steps taken
// once the step is done") — the host recomputes the fused populations' rows of those steps from the history rows
// (plan.py: settle_fused; what a give-up can leave wrong is rates only: the writer computes the state from its own
// loads and waits for the others only before it STORES; nobody but a writer stores state, history or task rows)
__device__ __forceinline__ void step1_note_timeout(const Step1Sync& sy, uint32_t step_after) {
  uint32_t* const tail = sy.words + (int64_t)sy.n_segments * RIAB_STEP1_SYNC_STRIDE;
  atomicAdd(tail + RIAB_STEP1_SYNC_TIMEOUTS, 1u);
  atomicCAS(tail + RIAB_STEP1_SYNC_FIRST_BAD, 0u, step_after);
  atomicMax(tail + RIAB_STEP1_SYNC_LAST_BAD, step_after);
}
// A workgroup barrier for hand-overs through LDS: __syncthreads() also waits for every global store and atomic the wave
// has in flight (its fence covers global memory) — a round trip in the middle of the writer's dependent chain, where the
// bookkeeping's stores and the episode table's atomic are meant to stay in flight.  Nothing in global memory is handed
// over at these barriers.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef __attribute__((address_space(1))) unsigned long long s1_gu64;

// The rest of TaskEnvironment.step in the same launch (TASK = task_kernel's MODE: 1 step, | 2 the caller's `if terminal:
// reset()`, | 4 the scripted action of the next step) — what the step plan's motion + task launch did behind the motion
// step, for a plan whose lead population can be fused as well (one kernel per closed-loop step instead of two).
//   * The segment's writer workgroup (blockIdx.y == 0) keeps the books of its 256 lanes (the pieces of
//     riab_task_kernel.h: the same code on the same operands as task_kernel), its task rows fetched with the state at
//     the top, a whole motion step ahead of their use.  It writes no rates: the cell groups are dealt to the
//     workgroups y >= 1.  The bookkeeping is a long, branchy per-lane instruction stream (7 us behind the motion step
//     when one wave does all of it), so it is cut by function: what does not depend on where the agent went — the
//     reward cache's update, everything a reset would draw — is the helper waves' (4-7), beside the movers' second
//     half; the movers keep the goal checks and the reset; the next action is the helper waves' again, while the movers
//     store rows, history and state.  Hand-overs go through LDS (lds_barrier).
//   * A reset that teleports changes the position the rates are a function of.  The other workgroups cannot know — the
//     goal lists are the writer's — so each mover wave of the writer posts six verdict entries per step as soon as its
//     lanes' positions are final (its mask of moved lanes in halves, the new positions of its first two movers; every
//     entry carries the launch's epoch, none has to be ordered against another), and the workgroups y >= 1 read the 24
//     entries after their last rate store: the wave runs its rate pass again on the patched row, stored by the lanes
//     whose quad of agents has a mover (own stores acknowledged first).  The rows end up as the population's kernel
//     would have written them from the history row the reset patched.
//   * What a lane's bookkeeping writes where other workgroups read (the position, the next action in the drift
//     buffer) is stored behind the arrival words.
// Workgroups y >= 1 now wait for the writer's verdict, which the writer posts without waiting for anybody: no cycle, but
// the grid has to be resident at once (launch_step1_cell refuses shapes that are not); bounded by the same spin limit
// and counter as the state write-back.
struct Step1Task {
  TaskArgs a;
  ResetArgs r;  // (pos_x / hist_x null: the writer stores what it is handed back)
  double t_env;
  double* reward_out;
  uint8_t* terminal_out;
  double gv_scale;
  double* gv_x;
  double* gv_y;
  int32_t* diag;
  uint32_t* mail;  // [segments][RIAB_STEP1_MAIL_STRIDE]
  // the lanes are the agents of ONE world (TASK & 8; riab_task_world.hip): its shared state, the step's scratch
  double* world;
  uint64_t* met;
  int32_t* cand;
  int32_t* ctl;
};

// tools/step1_profile.py (a -DRIAB_STEP1_PROFILE build): three workgroups — the first writer, its segment's second
// workgroup, the grid's last — leave the device's constant clock at the phase boundaries of the LAST step, behind the
// arrival words' tail (the Python layer allocates 256 words of slack there)
#ifdef RIAB_STEP1_PROFILE
#define RIAB_S1_STAMP(k)                                                                                                  \
  if (prof_slot >= 0 && tid == 0)                                                                                         \
    ((unsigned long long*)(sy.words + RIAB_STEP1_SYNC_WORDS((int64_t)sy.n_segments * 256)))[prof_slot * 16 + (k)] = \
        (unsigned long long)__builtin_amdgcn_s_memrealtime();
#define RIAB_S1_STAMP_H(k) /* (the same, by the first helper wave) */                                                     \
  if (prof_slot >= 0 && tid == 256)                                                                                       \
    ((unsigned long long*)(sy.words + RIAB_STEP1_SYNC_WORDS((int64_t)sy.n_segments * 256)))[prof_slot * 16 + (k)] = \
        (unsigned long long)__builtin_amdgcn_s_memrealtime();
#else
#define RIAB_S1_STAMP(k)
#define RIAB_S1_STAMP_H(k)
#endif
// timing experiments (tools/build_step1_variants.sh): bit 0 no motion step, bit 1 no rate stores, bit 2 no write-back
#ifndef RIAB_S1_ABLATE
#define RIAB_S1_ABLATE 0
#endif
// ... of the task modes: task_kernel MODE bits left out (1 the step's bookkeeping, 2 resets, 4 the next action), 8: the
// workgroups y >= 1 do not wait for the writer's verdict
#ifndef RIAB_S1_TASK_DROP
#define RIAB_S1_TASK_DROP 0
#endif
#ifndef RIAB_S1_STORE
#define RIAB_S1_STORE RIAB_STORE_WT
#endif
#ifndef RIAB_S1_FEW
#define RIAB_S1_FEW 1  // a task's second pass over the quads a reset moved: s1_group_few (0: the whole pass again, stored by those quads)
#endif
#ifndef RIAB_S1_WAVES_PER_EU
#define RIAB_S1_WAVES_PER_EU 2
#endif
#ifndef RIAB_S1_STAGE
#define RIAB_S1_STAGE 8  // cell groups per wave, at most (the parameters of all but the first wait in LDS: 14 KB, a workgroup stays below 64 KB of LDS)
#endif
#define RIAB_S1_WAVES 8  // waves per workgroup: 0-3 advance the 256 agents, 4-7 draw their normals, all of them write rates
// one cell group of population `q` (functor `Cell`, CPB cells) for the lane's quad of agents: rate_kernel_wide's inner
// loop.  `cur`: the group's parameters, one per lane (s1_group_params); `store`: the lanes that write their values.
template <class Cell, int CPB, int SPK, bool NT>
__device__ __forceinline__ void s1_group(const Cell& cell, const Step1Pops& ps, const Step1Pop& q, const int gl, const float cur,
                                         const v4f rx, const v4f ry, const v4f rhx, const v4f rhy, const int64_t B,
                                         const uint32_t quad, const bool store) {
  constexpr int NP = Cell::NP;
  static_assert(NP * CPB <= 64, "a cell group's parameters must fit one wave");
  const typename Cell::Pos P = cell.from_rows(rx, ry, rhx, rhy);
  const int c0 = gl * CPB;
  int64_t off = (int64_t)c0 * B + 4 * (int64_t)quad;
  RateArgs sa;  // (what spike_store reads)
  sa.u_in = nullptr;
  sa.spikes = q.spikes;
  sa.tag = q.tag;
  sa.k0 = ps.k0;
  sa.k1 = ps.k1;
  sa.dt = ps.dt;
#pragma unroll
  for (int j = 0; j < CPB; ++j) {
    if (c0 + j < q.n) {  // wave-uniform
      float p[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) p[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cur), j * NP + i));
      v4f rr = cell.eval(p, P);
      rr = finish_rate(rr * q.fr_scale + q.fr_min, P);  // [0,1] -> [min_fr, max_fr]
      if (store) {
        if (RIAB_S1_ABLATE & 2) {
          if (rr.x == 123.0f) *reinterpret_cast<v4f*>(q.rates + off) = rr;
        } else if (NT) store_stream<RIAB_S1_STORE>(q.rates + off, rr);
        else *reinterpret_cast<v4f*>(q.rates + off) = rr;
        if (SPK && q.spikes) spike_store<false, RIAB_S1_STORE>(sa, rr, off, ps.step0, (uint32_t)(c0 + j), ps.quad0 + quad);  // (wave-uniform)
      }
      off += B;
    }
  }
}
// ... for a FEW of the segment's quads only (a task's reset has moved an agent or two of the segment: the quads `mq`, a
// wave-uniform mask): the other way round — a lane takes ONE cell of the group and one of those quads, 64 / CPB quads to
// a round, instead of every lane all CPB cells for a quad nobody moved.  The same functor on the same operands, element
// for element: the same bits (the cell's parameters arrive in a vector register instead of a scalar one).  `rows`: the
// segment's row in LDS (x, y, head direction x, y: 256 values each) with the moved agents' new positions in;
// `quad0`: the segment's first quad within the history row.
// (The quads' positions come from LDS, not from the lanes that hold them: four v_readlane of the elements of one vector
// value inside this loop came out of the compiler — ROCm 7.2 — as four copies of the first element.)
typedef const __attribute__((address_space(3))) float* lds_cf32_ptr;
template <class Cell, int CPB, int SPK, bool NT>
__device__ __forceinline__ void s1_group_few(const Cell& cell, const Step1Pops& ps, const Step1Pop& q, const int gl, const float cur,
                                             const lds_cf32_ptr rows, const int64_t B, const uint32_t quad0, unsigned long long mq) {
  constexpr int NP = Cell::NP;
  constexpr int QPR = 64 / CPB;  // quads per round
  typedef const __attribute__((address_space(3))) v4f* lds_cv4f_ptr;
  const int lane = (int)__lane_id();
  const int slot = lane / CPB, j = lane - slot * CPB;  // the lane's quad of the round, its cell of the group
  float p[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i)
    p[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (j * NP + i), __builtin_bit_cast(int, cur)));
  const int c = gl * CPB + j;
  RateArgs sa;
  sa.u_in = nullptr;
  sa.spikes = q.spikes;
  sa.tag = q.tag;
  sa.k0 = ps.k0;
  sa.k1 = ps.k1;
  sa.dt = ps.dt;
  while (mq) {  // (wave-uniform) a round of up to QPR quads
    int my_quad = -1;
    for (int t = 0; t < QPR && mq; ++t) {
      const int l = __ffsll((long long)mq) - 1;
      mq &= mq - 1;
      my_quad = (slot == t) ? l : my_quad;
    }
    const int rq = my_quad < 0 ? 0 : my_quad;
    const v4f cx = *(lds_cv4f_ptr)(rows + 4 * rq), cy = *(lds_cv4f_ptr)(rows + 256 + 4 * rq);
    v4f chx = {1.0f, 1.0f, 1.0f, 1.0f}, chy = {0.0f, 0.0f, 0.0f, 0.0f};
    if (ps.needs_hd) {
      chx = *(lds_cv4f_ptr)(rows + 512 + 4 * rq);
      chy = *(lds_cv4f_ptr)(rows + 768 + 4 * rq);
    }
    const typename Cell::Pos P = cell.from_rows(cx, cy, chx, chy);
    v4f rr = cell.eval(p, P);
    rr = finish_rate(rr * q.fr_scale + q.fr_min, P);
    if (my_quad >= 0 && c < q.n) {
      const uint32_t quad = quad0 + (uint32_t)my_quad;
      const int64_t off = (int64_t)c * B + 4 * (int64_t)quad;
      if (NT) store_stream<RIAB_S1_STORE>(q.rates + off, rr);
      else *reinterpret_cast<v4f*>(q.rates + off) = rr;
      if (SPK && q.spikes) spike_store<false, RIAB_S1_STORE>(sa, rr, off, ps.step0, (uint32_t)c, ps.quad0 + quad);
    }
  }
}

// ... switched into by the population's functor id (wave-uniform)
// (KIND >= 0: the functor is known when the kernel is compiled — the plan's only population is of that kind)
// (FEW: `quad` is the segment's first quad, `mq` the quads to evaluate, `rows` the segment's row in LDS — s1_group_few;
// otherwise the lane's own quad, its row values and whether it stores)
template <int SPK, bool NT, bool TASK, int KIND, bool FEW = false>
__device__ __forceinline__ void s1_group_any(const Step1Pops& ps, const Step1Pop& q, const int gl, const float cur, const v4f rx,
                                             const v4f ry, const v4f rhx, const v4f rhy, const int64_t B, const uint32_t quad,
                                             const bool store, const unsigned long long mq = 0ull,
                                             const lds_cf32_ptr rows = nullptr) {
#define RIAB_S1_RUN(CELL, c, hx_, hy_)                                                                                   \
  if (FEW) s1_group_few<CELL, S1Cpb<CELL, TASK, SPK != 0>::value, SPK, NT>(c, ps, q, gl, cur, rows, B, quad, mq);          \
  else s1_group<CELL, S1Cpb<CELL, TASK, SPK != 0>::value, SPK, NT>(c, ps, q, gl, cur, rx, ry, hx_, hy_, B, quad, store);
#define RIAB_S1_PC(DESC, GX, ID)                                                                                        \
  case ID: {                                                                                                            \
    typedef PlaceCell<DESC, GX> CellT;                                                                                  \
    CellT c;                                                                                                            \
    c.tab = q.tab; c.scale = q.p0; c.half_scale = q.p1; c.top_hat_w2 = q.p2;                                            \
    c.walls = nullptr; c.n_internal = 0; c.lds = nullptr;                                                               \
    RIAB_S1_RUN(CellT, c, rhx, rhy)                                                                                     \
    break;                                                                                                              \
  }
  switch (KIND >= 0 ? KIND : q.kind) {
    RIAB_S1_PC(RIAB_PC_GAUSSIAN, 0, 0)
    RIAB_S1_PC(RIAB_PC_GAUSSIAN_THRESHOLD, 0, 1)
    RIAB_S1_PC(RIAB_PC_DIFF_OF_GAUSSIANS, 0, 2)
    RIAB_S1_PC(RIAB_PC_TOP_HAT, 0, 3)
    RIAB_S1_PC(RIAB_PC_GAUSSIAN, 3, 4)
    RIAB_S1_PC(RIAB_PC_GAUSSIAN_THRESHOLD, 3, 5)
    RIAB_S1_PC(RIAB_PC_DIFF_OF_GAUSSIANS, 3, 6)
    RIAB_S1_PC(RIAB_PC_TOP_HAT, 3, 7)
    case S1_KIND_GC + RIAB_GC_RECTIFIED: {
      typedef GridCell<RIAB_GC_RECTIFIED> CellT;
      const CellT c{q.tab, q.p0, q.p1};
      RIAB_S1_RUN(CellT, c, rhx, rhy)
      break;
    }
    case S1_KIND_GC + RIAB_GC_SHIFTED: {
      typedef GridCell<RIAB_GC_SHIFTED> CellT;
      const CellT c{q.tab, q.p0, q.p1};
      RIAB_S1_RUN(CellT, c, rhx, rhy)
      break;
    }
    default: {
      typedef HDCell<0> CellT;
      const CellT c{q.tab, 0.0f, nullptr, nullptr};
      // (the head direction's normalisation — square roots, divisions — is loop-invariant and free of side effects: left
      // alone, the compiler hoists it in front of the switch, where every wave of every population pays for it)
      v4f hx = rhx, hy = rhy;
      asm volatile("" : "+v"(hx.x), "+v"(hx.y), "+v"(hx.z), "+v"(hx.w), "+v"(hy.x), "+v"(hy.y), "+v"(hy.z), "+v"(hy.w));
      RIAB_S1_RUN(CellT, c, hx, hy)
      break;
    }
  }
#undef RIAB_S1_PC
#undef RIAB_S1_RUN
}

template <int SPK, bool NT, int TASK, int KIND>
__device__ __forceinline__ void step1_body(const AgentArgs& a, const Step1Pops& ps, const Step1Sync& sy,
                                           const int reps, const MotionConst<double>& hk, const TailConst<double>& tail_c,
                                           const Step1Task& tk) {
  RIAB_EXACT_FP
  constexpr bool MULTI = KIND < 0;  // (several populations, or one of a kind without a kernel of its own)
  constexpr int NT_ = 64 * RIAB_S1_WAVES;
  __shared__ Wall<double> s_w[RIAB_MAX_WALLS];
  __shared__ double s_g[RIAB_G_SEGS * RIAB_G_STRIDE];
  __shared__ double s_h[RIAB_H_SEGS * RIAB_H_STRIDE];
  __shared__ __align__(16) float s_row[4][256];  // x, y, head direction x, y of the segment's agents as the history keeps them
  __shared__ float s_z[2][256];                  // the step's two standard normals per agent (drawn by waves 4-7)
  __shared__ uint64_t s_grid[RIAB_WALL_GRID_WORDS];  // (wall-heavy rooms: RiabMotion.wall_grid)
  __shared__ float s_par[RIAB_S1_WAVES][RIAB_S1_STAGE][64];  // each wave's cell groups' parameters, one per lane and group
  // TASK & 8: the lanes are the agents of ONE world (the writers keep the world's books, below); otherwise every lane is
  // its own replica of the task.  RT / WT: the mode bits of the one or the other (0: not that kind of task)
  constexpr int RT = (TASK & 8) ? 0 : TASK, WT = (TASK & 8) ? (TASK & 7) : 0;
  __shared__ WorldShared s_world;
  __shared__ double s_goals[TASK ? RIAB_TASK_MAX_POOL * RIAB_GOAL_COLS : 1];  // the task's goal pool (writer)
  // (writer) what its noise-drawing waves work out for the lanes' books while the movers move: the reward cache's update
  // (rewards alive, their total) and what a reset of the lane would draw (position, next episode's goals)
  __shared__ double s_rw_total[RT ? 256 : 1];
  __shared__ int s_rw_n[RT ? 256 : 1];
  __shared__ double s_draw_xy[2][(RT & 6) ? 256 : 1];
  __shared__ unsigned long long s_draw_list[2][(RT & 6) ? 256 : 1];
  // ... and, the same arrays later, what the movers hand BACK for the next action (TASK & 4): the lane's list and position
  __shared__ int s_fin_n[(RT & 4) ? 256 : 1];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool writer = blockIdx.y == 0;
  const bool mover = wave < 4;  // (wave-uniform) the waves that advance the segment's 256 agents
#ifdef RIAB_STEP1_PROFILE
  const int prof_slot = (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : (blockIdx.x == 0 && blockIdx.y == 1) ? 1
                        : (blockIdx.x + 1 == gridDim.x && blockIdx.y + 1 == gridDim.y) ? 2 : -1;
#endif
  RIAB_S1_STAMP(0)
  const int64_t B = a.B;
  const int64_t b = (int64_t)blockIdx.x * 256 + (tid & 255);  // (B is a multiple of 256: whole segments)
  const RiabMotion& m = a.m;

  // ---- everything that does not depend on anything: the tables into LDS, the state into registers — one round trip
  double* const st = a.state + b;
  double px = 0, py = 0, vx = 0, vy = 0, rot = 0, mvx = 0, mvy = 0, mrot = 0, hx = 0, hy = 0, dist = 0, dwall = 0, drx = 0, dry = 0;
  if (mover) {
    px = st[0 * B]; py = st[1 * B];
    vx = st[2 * B]; vy = st[3 * B];
    rot = st[4 * B];
    mvx = st[5 * B]; mvy = st[6 * B];
    mrot = st[7 * B];
    hx = st[8 * B]; hy = st[9 * B];
    dist = st[10 * B];
    dwall = st[11 * B];
    if (m.has_drift) {
      drx = a.drift[b];
      dry = a.drift[B + b];
    }
  }
  // (task: the writer's lanes ask for their bookkeeping rows in the same batch; the pool goes to LDS with the tables)
  // The bookkeeping is one long, branchy instruction stream per lane (~7 us behind a 4 us motion step when one wave does
  // it all), and most of it does not depend on where the agent went: the reward cache's update and the draws of a reset
  // are the helper waves' (4-7, idle once the normals are drawn), lane for lane beside the movers' second half; the
  // movers keep what needs the new position — the goal checks, the reset itself, the next action.
  // (dealing the lanes to all eight waves, 32 each, was tried: no faster — docs/EXPERIMENTS.md r05-5)
  const bool tlive = RT && writer && mover && b < tk.a.B;
  const bool hlive = RT && writer && !mover && b < tk.a.B;
  const bool wlive = WT && writer && mover && b < tk.a.B;  // (one world: a mover lane keeps its agent's books)
  LaneIn tin;
  RewardsIn trin;
  WorldLaneIn win;
  double w_pad_start0 = 0.0;
  uint8_t w_terminal_prev = 0;
  if (TASK) {
    if (tlive) tin = task_lane_load<RT & ~RIAB_S1_TASK_DROP>(tk.a, b);  // (TM, declared below)
    if (hlive && (RT & 1)) trin = load_rewards_in(tk.a, b);
    if (writer) task_stage_goals(tk.a, s_goals, tid, NT_);
    if (WT && writer) {  // the world's list as this step finds it, the lane's reward rows: one batch with the state
      if (tid < RIAB_TASK_MAX_GOALS) s_world.list[tid] = (uint8_t)((int)tk.world[RIAB_TW_GOAL_LIST + tid] & 0xFF);
      if (tid == 0) s_world.n = (int)tk.world[RIAB_TW_N_GOALS];
      w_pad_start0 = tk.world[RIAB_TW_PAD_START];
      w_terminal_prev = tk.world[RIAB_TW_TERMINAL] != 0.0 ? 1 : 0;
      if (wlive) win = world_lane_load(tk.a, b);
    }
  }
  // (the box fast path's verdict, worked out once per plan by walls_prepare_kernel)
  const uint32_t box_word = *reinterpret_cast<const uint32_t*>(sy.walls + RIAB_MAX_WALLS);
  // this wave's first cell group (task: the writer has none, the cell groups are dealt to the workgroups y >= 1)
  const bool rates_here = !(TASK && writer);
  const int g0 = (int)((blockIdx.y - (TASK ? 1u : 0u)) * (uint32_t)RIAB_S1_WAVES + (uint32_t)wave) * reps;
  // (wave-uniform) the population the global cell group g belongs to, and its record.  The records are read at STATIC
  // offsets of the argument block — population 0's with the kernel's other arguments, a later one's only by a wave that
  // has a group of it: a record picked by a computed index is fetched field by field, every field a scalar-memory round
  // trip of its own at its first use (a microsecond in front of the first store when tried).
  auto pop_of = [&](int g) -> int {
    int pi = 0;
    if (MULTI) {
#pragma unroll
      for (int k = 1; k < RIAB_STEP1_MAX_POPS; ++k) pi += (k < ps.n_pops && g >= ps.pop[k].group0) ? 1 : 0;
    }
    return pi;
  };
  auto pick = [&](int pi) -> Step1Pop {  // (field by field: a record copied as a whole is given a home in scratch memory)
    pi = __builtin_amdgcn_readfirstlane(pi);
    Step1Pop q;
    // (MULTI = false, one population — the closed loop of BASELINE configs[1] —: its record comes with the kernel's other
    // arguments at entry; a pick among several is a second scalar-memory round trip, in front of the parameter loads at
    // the top and in front of the first store after the motion step: 0.7 us per step when that was the only path — and
    // a run-time test of n_pops does not help: the compiler turns it into the same select of addresses)
#define RIAB_S1_FIELDS(F) F(tab) F(rates) F(spikes) F(n) F(kind) F(group0) F(n_groups) F(fr_scale) F(fr_min) F(p0) F(p1) F(p2) F(tag)
    // (every field passes through a scalar register of its own: neighbouring fields copied together become a block copy
    // into a private array, which the compiler then gives a home in LDS or scratch memory)
#define RIAB_S1_FIELD0(f) { auto v_ = ps.pop[0].f; asm("" : "+s"(v_)); q.f = v_; }
#define RIAB_S1_FIELD(f) { auto v_ = pi == 1 ? ps.pop[1].f : pi == 2 ? ps.pop[2].f : pi == 3 ? ps.pop[3].f : ps.pop[0].f; asm("" : "+s"(v_)); q.f = v_; }
    if (!MULTI) {
      RIAB_S1_FIELDS(RIAB_S1_FIELD0)
    } else {
      RIAB_S1_FIELDS(RIAB_S1_FIELD)
    }
#undef RIAB_S1_FIELD
#undef RIAB_S1_FIELD0
#undef RIAB_S1_FIELDS
    static_assert(RIAB_STEP1_MAX_POPS == 4, "pick() spells the records out");
    return q;
  };
  auto group_params = [&](const Step1Pop& q, int g) -> float {  // group g (of population q)'s parameters, one per lane
    const int kind = KIND >= 0 ? KIND : q.kind;
    const int np = s1_np(kind), width = np * s1_cpb(kind, TASK != 0, SPK != 0);
    const int pi = (g - q.group0) * width + lane;
    return (g < ps.total_groups && lane < width && pi < q.n * np) ? q.tab[pi] : 0.0f;
  };
  // Every group's parameters come with the state, a whole motion step ahead of their use, and wait in LDS: a load issued
  // between two groups' stores would have to be waited for with the one counter loads and stores share — in effect a
  // drain of the wave's stores per group, and of the writer's before its write-back.  (step1_shape keeps reps within the
  // staging's rows.)
  // (without a task the first group's stay in a register, as they always did — most plans have one group per wave —; a
  // task's step, whose rate workgroups may run their pass twice, measured faster with every group's in LDS)
  constexpr bool MINE_REG = TASK == 0;
  const int pi0 = pop_of(g0);
  float mine = 0.0f;
  if (rates_here) {
    int pi = pi0;
    Step1Pop q = pick(pi);
    if (MINE_REG) mine = group_params(q, g0);
    for (int r = MINE_REG ? 1 : 0; r < reps; ++r) {
      if (MULTI && g0 + r >= q.group0 + q.n_groups && pi + 1 < ps.n_pops) q = pick(++pi);  // (a population has at least one group)
      s_par[wave][r][lane] = group_params(q, g0 + r);
    }
  }
  stage_rayleigh_tables<NT_>(s_g, s_h, tid);
  for (int i = tid; i < a.n_walls * (int)(sizeof(Wall<double>) / sizeof(double)); i += NT_)
    reinterpret_cast<double*>(s_w)[i] = reinterpret_cast<const double*>(sy.walls)[i];
  stage_wall_grid(a, s_grid, tid, NT_);
  __syncthreads();
#ifdef RIAB_STEP1_PROFILE
  if (prof_slot >= 0) {  // (the state must have arrived: its first use would otherwise be timed with the motion step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RIAB_S1_STAMP(1)
  }
#endif

  // ---- Agent.update for agent b (agent_step_body's step, T = 1, Philox noise, tables in LDS) -----------------------
  // In two halves around one more barrier.  The step's two standard normals are a function of (seed, step, agent)
  // alone — Philox + Box-Muller, a fifth of a step's instructions — so the workgroup's OTHER four waves draw them (one
  // lane per agent, into LDS) while the movers run everything that does not need them: the speed, its G-table
  // polynomial, both wall passes (they read the position only).  The same functions on the same operands as
  // agent_step_body, in another order where the order does not matter.
  int n_bounce = 0, n_sat = 0, n_bc = 0, n_still = 0;
  const uint32_t aid = (uint32_t)(a.agent_id0 + b);
  const double dt = m.dt;
  const int nw = a.n_walls;
  MotionConst<double> K = hk;  // (the scalars were worked out by the host: six float64 divisions less per thread and step)
  NearWalls<double> near = {};
  WallPush<double> push = {};
  double ispeed = 0, nv64 = 0, ppx = 0, ppy = 0;
  bool zero_v = false;
  if (!mover) {
    u32x4 pw = {0u, 0u, 0u, 0u};
    const MotionDraw d = motion_normals(a.step0, true, aid, a.k0, a.k1, pw);
    s_z[0][tid & 255] = d.z_rot;
    s_z[1][tid & 255] = d.z_spd;
  } else if (!(RIAB_S1_ABLATE & 1)) {
    const lds_cf64_ptr lds_g = (lds_cf64_ptr)s_g;
#pragma unroll
    for (int w = 0; w < 4; ++w) K.w4[w] = s_w[w < K.nw ? w : 0];
    motion_const_grid<double>(K, a, s_grid);
    K.box_fast = __builtin_amdgcn_readfirstlane((int)box_word) != 0;  // motion_const_walls' verdict ...
    K.bxl = K.box_fast ? K.e0 : 0.0;                                   // ... and its edges: the extent itself when it holds
    K.bxr = K.box_fast ? K.e1 : 0.0;
    K.byb = K.box_fast ? K.e2 : 0.0;
    K.byt = K.box_fast ? K.e3 : 0.0;
    ppx = px;  // prev_pos (Agent.py:199)
    ppy = py;
    // ---- _stochastic_velocity_update (Agent.py:287-312), the part in front of the noise
    double v2 = norm2(vx, vy);
    zero_v = (v2 == 0.0);
    if (zero_v) v2 = 1e-16;  // the reference replaces a zero velocity by (1e-8, 0) (Agent.py:299-300)
    ispeed = r_rsqrt(v2);
    const double speed = v2 * ispeed;
    const double tG = clamp_G_arg(speed * K.inv_sm);
    const SegRow<RIAB_G_DEG> grow = seg_fetch<RIAB_G_DEG>(lds_g + seg_G(tG) * RIAB_G_STRIDE);
    // ---- _wall_velocity_update: pass 1 and the spring / conveyor terms of pass 2 (functions of the position)
    near = walls_pass1<double>(K, s_w, px, py);
    if (nw > 0 && K.repel) {
      push = walls_pass2_terms<double>(K, s_w, near, px, py);
      dwall = closest_wall_distance<double>(K, near);
    }
    // utils.rayleigh_to_normal (utils.py:416-421), sigma = speed_mean
    nv64 = seg_eval<RIAB_G_DEG>(grow, tG);
  }
  __syncthreads();  // the normals are in LDS
  ResetArgs tr = tk.r;  // (what this kernel never does, said so that the compiler can drop it: no positions handed in,
  tr.new_x = tr.new_y = nullptr;  // none stored by the lane's reset itself)
  tr.pos_x = tr.pos_y = nullptr;
  tr.hist_x = tr.hist_y = nullptr;
  constexpr int TM = RT & ~RIAB_S1_TASK_DROP;
  if (RT && writer && !mover) {  // (wave-uniform) the helper waves' share of the lanes' books, see above
    if (hlive && (TM & 1)) {
      const RewardsOut ro = rewards_step(tk.a, (lds_f64_ptr)s_goals, b, trin);
      s_rw_n[tid & 255] = ro.n_rw;
      s_rw_total[tid & 255] = ro.total;
    }
    lds_barrier();  // (writer) the reward caches are up to date
    if (hlive && (TM & 2)) {
      const ResetDraw d = reset_draw(tk.a, tr, b);
      s_draw_xy[0][tid & 255] = d.x;
      s_draw_xy[1][tid & 255] = d.y;
      s_draw_list[0][tid & 255] = (unsigned long long)d.list;
      s_draw_list[1][tid & 255] = (unsigned long long)(d.list >> 64);
    }
    lds_barrier();  // (writer) the resets' draws are in LDS
    RIAB_S1_STAMP_H(14)
  }
  if (mover && !(RIAB_S1_ABLATE & 1)) {
    const lds_cf64_ptr lds_h = (lds_cf64_ptr)s_h;
    const double z_rot = (double)s_z[0][tid], z_spd = (double)s_z[1][tid];
    rot = ou_step<double>(rot, m.rot_theta_kw, m.rot_drift_kw, m.rot_sigma_kw, dt, z_rot);
    {
      double sn, cs;
      sincos_small(rot * dt, &sn, &cs);
      rotate_by<double>(cs, sn, vx, vy);
      vx = zero_v ? 1e-8 : vx;
      vy = zero_v ? 0.0 : vy;
    }
    // the speed's OU step in normal space and normal_to_rayleigh (utils.py:409-413)
    nv64 = ou_step<double>(nv64, m.speed_theta_kw, 0.0, m.speed_sigma_kw, m.dt, z_spd);
    const bool h_in_table = fabs(nv64) < RIAB_H_NMAX;
    const SegRow<RIAB_H_DEG> hrow = seg_fetch<RIAB_H_DEG>(lds_h + seg_H(nv64) * RIAB_H_STRIDE);
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
    // ---- _wall_velocity_update, pass 2 applied
    if (nw > 0 && K.repel) walls_pass2_apply<double>(K, push, px, py, vx, vy);
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
  RIAB_S1_STAMP(2)
  // ---- the writer's part: the state in place, once nobody can read the old one any more.  Each of its mover waves asks
  // for its segment's arrival words BEFORE its share of the rates (task: before its lanes' bookkeeping) — the request
  // travels meanwhile — and looks at the answer after it; only a word that was not there yet costs a poll.  Bounded;
  // counted when it gives up.
  auto arrivals = [&]() -> uint32_t {
    const uint32_t others = gridDim.y - 1u;  // (<= RIAB_STEP1_SYNC_MAX_Y - 1: one lane per word)
    uint32_t v = sy.epoch;
    if ((uint32_t)lane < others)
      v = __hip_atomic_load((s1_gu32*)(uintptr_t)(sy.words + (int64_t)blockIdx.x * RIAB_STEP1_SYNC_STRIDE + 1 + lane), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  const bool wb = writer && mover && !(RIAB_S1_ABLATE & 4);  // (wave-uniform)
  uint32_t seen = sy.epoch;
  LaneMid tmid = {0, 0, false, false, {false, 0, 0.0, 0.0, 0.0}, 0};
  // a wait for the segment's other workgroups (their state / action loads have returned), bounded
  auto wait_arrivals = [&]() {
    bool timed_out = false;
    for (uint32_t spins = 0; __builtin_amdgcn_ballot_w64(seen != sy.epoch) != 0; ++spins) {
      if (spins >= sy.spin_limit) {  // (a workgroup of this grid that never ran: nothing sane to do but to say so)
        timed_out = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      seen = arrivals();
    }
    if (timed_out && lane == 0) step1_note_timeout(sy, ps.step0);
  };
  // The world's verdict on this step: five 8-byte entries at the head of the mail region, each tagged with the launch's
  // epoch (epoch << 32 | value): [0] bit 0 = the episode ended and the world was reset (the caller's `if terminal:
  // env.reset()`, decided here), bits 8.. = the length of the list the NEXT step finds; [1..4] that list, four entries
  // a word.  Posted by the writer workgroup that took the last ticket; read by every wave that needs it (lanes 0-4).
  auto world_verdict = [&](bool& stale) -> unsigned long long {  // -> lane l: entry l's value (lanes 0-4), once all are fresh
    const s1_gu64* const mail64 = (const s1_gu64*)(uintptr_t)tk.mail;
    unsigned long long e = (unsigned long long)sy.epoch << 32;
    const bool polls = lane < 5;
    if (polls) e = __hip_atomic_load((s1_gu64*)(mail64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stale = false;
    for (uint32_t spins = 0; __builtin_amdgcn_ballot_w64((uint32_t)(e >> 32) != sy.epoch) != 0; ++spins) {
      if (spins >= sy.spin_limit) {
        stale = true;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
      if (polls) e = __hip_atomic_load((s1_gu64*)(mail64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return e;
  };
  if (WT && writer) {
    // ---- ONE world (contribs/TaskEnvironment.py:1030, 1076-1172; riab_task_world.hip): every writer keeps the books of
    // its 256 agents (phase A), takes a ticket; the writer with the last ticket walks the shared list for everybody
    // (phase B), resets the world when its episode ended, and posts the verdict; every writer then teleports its agents
    // (a reset's draws are functions of (seed, counter, agent) alone), works out their next action against the list the
    // verdict carries and stores what it is handed back.  All eight waves take the barriers.
    const lds_f64_ptr goals = (lds_f64_ptr)s_goals;
    if (wb) seen = arrivals();
    if (wlive)
      world_phase_a(tk.a, goals, s_world, b, px, py, win, tk.t_env, w_pad_start0, w_terminal_prev, tk.reward_out, tk.terminal_out,
                    tk.met, tk.cand, tk.ctl);
    // (phase A's write-through stores must have been ACKNOWLEDGED before the ticket is taken: the workgroup with the last
    // ticket reads them.  A barrier does not wait for them — the compiler puts `lgkmcnt(0)` in front of it, a workgroup's
    // waves need no more of each other —: without this wait the last workgroup read, once in some millions of steps
    // beside a foreign load, a row that had not arrived yet; tools/task_world_soak.py found it)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) (an asm statement with a memory clobber here costs one instantiation a stack frame)
    __syncthreads();
    if (tid == 0) s_world.last = atomicAdd(tk.ctl, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (s_world.last) {  // (workgroup-uniform)
      const bool terminal_last = world_phase_b<NT_>(tk.a, goals, s_world, tk.world, tk.t_env, w_pad_start0, w_terminal_prev,
                                                    tk.reward_out, tk.terminal_out, tk.met, tk.cand, tk.ctl, tk.diag);
      const bool reset = (WT & 2) && terminal_last;
      __syncthreads();
      if (tid == 0) {
        if (reset) {  // GoalCache.reset (:1218-1252): one selection — the shared list —, the world's episode table
          const ResetDraw d = reset_draw_id(tk.a, tk.r, RIAB_WORLD_STREAM_ID);
          Lane nl;
          nl.list = d.list;
          nl.n_goals = tk.r.n_select < tk.a.n_pool ? tk.r.n_select : tk.a.n_pool;
          world_reset_books(tk.a, tk.r, tk.world, tk.t_env, nl, tk.diag);
          for (int i = 0; i < RIAB_TASK_MAX_GOALS; ++i) s_world.list[i] = (uint8_t)list_get(nl.list, i);
          s_world.n = nl.n_goals;
        }
        s_world.next_agent = reset ? 1 : 0;
      }
      __syncthreads();
      if (tid < 5) {
        uint32_t v;
        if (tid == 0) v = (uint32_t)s_world.next_agent | ((uint32_t)s_world.n << 8);
        else v = (uint32_t)s_world.list[4 * tid - 4] | ((uint32_t)s_world.list[4 * tid - 3] << 8) |
                 ((uint32_t)s_world.list[4 * tid - 2] << 16) | ((uint32_t)s_world.list[4 * tid - 1] << 24);
        __hip_atomic_store((s1_gu64*)(uintptr_t)tk.mail + tid, ((unsigned long long)sy.epoch << 32) | v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (mover) {
      bool stale;
      const unsigned long long e = world_verdict(stale);
      if (stale && lane == 0) {  // (a writer that does not learn whether the world was reset cannot keep its agents' state right)
        step1_note_timeout(sy, ps.step0);
        atomicAdd(sy.words + (int64_t)sy.n_segments * RIAB_STEP1_SYNC_STRIDE + RIAB_STEP1_SYNC_FATAL, 1u);
      }
      const int lo = (int)(uint32_t)e;
      const uint32_t flags = (uint32_t)__builtin_amdgcn_readlane(lo, 0);
      const bool reset = !stale && (flags & 1u);
      Lane L;
      L.n_goals = (int)((flags >> 8) & 0xFFu);
      L.list = (u128)(uint32_t)__builtin_amdgcn_readlane(lo, 1) | ((u128)(uint32_t)__builtin_amdgcn_readlane(lo, 2) << 32) |
               ((u128)(uint32_t)__builtin_amdgcn_readlane(lo, 3) << 64) | ((u128)(uint32_t)__builtin_amdgcn_readlane(lo, 4) << 96);
      if (wlive && reset && tk.r.teleport) {  // teleport_on_reset (:323-330)
        const ResetDraw d = reset_draw_id(tk.a, tk.r, (uint64_t)(tk.r.agent_id0 + b));
        px = d.x;
        py = d.y;
      }
      if (WT & 4) {  // the coming step's action (get_goal_vector, :1555-1584), into the drift buffer: behind the arrival words
        double gx = 0.0, gy = 0.0;
        if (wlive) {
          L.px = px;
          L.py = py;
          goal_vector(tk.a, goals, L, tk.gv_scale, gx, gy);
        }
        wait_arrivals();
        if (wlive) {
          tk.gv_x[b] = gx;
          tk.gv_y[b] = gy;
        }
      }
    }
  } else if (RT && writer && mover) {
    // ---- the rest of TaskEnvironment.step for the writer's lanes (contribs/TaskEnvironment.py:410-449), the caller's
    // reset of the lanes that ended an episode, the next scripted action: task_kernel's lane, on the position just made
    if (wb) seen = arrivals();
    double qx = px, qy = py;
    const double mx = qx, my = qy;
#ifdef RIAB_STEP1_PROFILE
    auto probe = [&](int k) { RIAB_S1_STAMP(k) };
#else
    const NoProbe probe;
#endif
    lds_barrier();  // (writer) the reward caches are up to date: the helper waves' rewards_step
    // (the arrival words are taken delivery of HERE, before the bookkeeping's first store: the wait counter is one for
    // loads and stores, and the compiler — which must assume the worst at every join of this branchy code — otherwise
    // drains the bookkeeping's stores, the write-through mail included, in the middle of the chain to keep that one
    // register safe; write_back() asks again in the rare case a word was not there yet)
    asm volatile("" : "+v"(seen));
    if (tlive) {
      const RewardsOut ro = {s_rw_n[tid], s_rw_total[tid]};
      tmid = task_lane_goals<TM>(tk.a, b, (lds_f64_ptr)s_goals, tin, ro, qx, qy, tk.t_env, tk.reward_out, tk.terminal_out, tk.diag, probe);
    }
    lds_barrier();  // (writer) the resets' draws are in LDS
    if (tlive)
      task_lane_reset<TM>(tk.a, tr, b, tin, tmid, qx, qy, tk.t_env, tk.diag, [&]() {
        return ResetDraw{s_draw_xy[0][tid], s_draw_xy[1][tid], ((u128)s_draw_list[1][tid] << 64) | (u128)s_draw_list[0][tid]};
      });
    const bool moved = tlive && !(qx == mx && qy == my);  // (a reset that teleported the lane)
    uint32_t* const mail = tk.mail + (int64_t)blockIdx.x * RIAB_STEP1_MAIL_STRIDE;
    // who moved and where to: 8-byte entries that carry the epoch themselves (epoch << 32 | value), so that none of them
    // has to be ordered against another — or against the bookkeeping's stores and the episode table's atomic, which are
    // still in flight.  This wave's two verdict entries (the halves of its lane mask) are posted every step.
    const unsigned long long lanes = __builtin_amdgcn_ballot_w64(moved);
    const unsigned long long tag = (unsigned long long)sy.epoch << 32;
    s1_gu64* const mail64 = (s1_gu64*)(uintptr_t)(tk.mail + (int64_t)blockIdx.x * RIAB_STEP1_MAIL_STRIDE);
    // this wave's six verdict entries, posted every step: the halves of its mask of moved lanes and — what the others
    // would otherwise come back for, one more round trip — the new positions of its first two movers (a wave has more
    // than two once in thousands of steps: the others' positions go to their per-agent entries)
    const unsigned long long lanes2 = lanes & (lanes - 1), lanes3 = lanes2 & (lanes2 - 1);
    const int i0 = lanes ? __ffsll((long long)lanes) - 1 : 0, i1 = lanes2 ? __ffsll((long long)lanes2) - 1 : 0;
    const int xb = (int)__float_as_uint((float)qx), yb = (int)__float_as_uint((float)qy);
    const uint32_t x0 = (uint32_t)__builtin_amdgcn_readlane(xb, i0), y0 = (uint32_t)__builtin_amdgcn_readlane(yb, i0);
    const uint32_t x1 = (uint32_t)__builtin_amdgcn_readlane(xb, i1), y1 = (uint32_t)__builtin_amdgcn_readlane(yb, i1);
    if (lane < 6) {
      const uint32_t v = lane == 0 ? (uint32_t)lanes : lane == 1 ? (uint32_t)(lanes >> 32) : lane == 2 ? x0 : lane == 3 ? y0 : lane == 4 ? x1 : y1;
      __hip_atomic_store(mail64 + 8 * wave + lane, tag | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lanes3 && moved && lane != i0 && lane != i1) {
      __hip_atomic_store(mail64 + 32 + tid, tag | (uint32_t)xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mail64 + 288 + tid, tag | (uint32_t)yb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RIAB_S1_STAMP(7)
    // (the others know; now what only this lane's books need: the ended episode's row, the new episode's goals, the next
    // action — the helper waves work it out from the list and the position while this wave stores —, the rows back)
    if (tlive) {
      task_lane_episode<TM>(tk.a, tr, b, tin, tmid, tk.t_env, tk.diag);
      task_lane_newgoals<TM>(tk.a, tr, tin, tmid);
      if (TM & 4) {
        s_draw_list[0][tid] = (unsigned long long)tin.L.list;
        s_draw_list[1][tid] = (unsigned long long)(tin.L.list >> 64);
        s_fin_n[tid] = tin.L.n_goals;
        s_draw_xy[0][tid] = qx;
        s_draw_xy[1][tid] = qy;
      }
    }
    RIAB_S1_STAMP(13)
    if (TM & 4) lds_barrier();  // (writer) the lanes' lists and positions are in LDS
    if (tlive) {
      task_lane_store<TM>(tk.a, b, tin, tmid);
      px = qx;
      py = qy;
    }
    RIAB_S1_STAMP(15)
  } else if (RT && (TM & 4) && writer) {
    // ---- the helper waves' last share: the coming step's action of every lane (get_goal_vector, :1555-1584), into the
    // drift buffer — which every workgroup of the segment read at the top: behind the arrival words, like the state
    uint32_t hseen = arrivals();
    lds_barrier();  // (writer) the lanes' lists and positions are in LDS
    double hx_ = 0.0, hy_ = 0.0;
    if (hlive) {
      Lane hl = {};
      hl.list = ((u128)s_draw_list[1][tid & 255] << 64) | (u128)s_draw_list[0][tid & 255];
      hl.n_goals = s_fin_n[tid & 255];
      hl.px = s_draw_xy[0][tid & 255];
      hl.py = s_draw_xy[1][tid & 255];
      goal_vector(tk.a, (lds_f64_ptr)s_goals, hl, tk.gv_scale, hx_, hy_);
    }
    bool late = false;
    for (uint32_t spins = 0; __builtin_amdgcn_ballot_w64(hseen != sy.epoch) != 0; ++spins) {
      if (spins >= sy.spin_limit) {
        late = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      hseen = arrivals();
    }
    if (late && lane == 0) step1_note_timeout(sy, ps.step0);
    if (hlive) {
      tk.gv_x[b] = hx_;
      tk.gv_y[b] = hy_;
    }
    RIAB_S1_STAMP_H(12)
  } else if (mover) {
    s_row[0][tid] = (float)px;
    s_row[1][tid] = (float)py;
    if (ps.needs_hd) {
      s_row[2][tid] = (float)hx;
      s_row[3][tid] = (float)hy;
    }
  }
  if (mover && writer && !TASK && a.hist) {  // save_to_history (Agent.py:514-520)
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
  // the row is in LDS; every state value of this workgroup has been consumed, i.e. loaded.  (A task's writer writes no
  // rates: its movers go on to the write-back while its helper waves are still busy with the next action.)
  if (!(TASK && writer)) __syncthreads();
  if (TASK && writer && mover) {  // (the lanes' books are kept: a reset may have moved them)
    if (a.hist) {  // save_to_history (Agent.py:514-520), agent.history["pos"][-1] = agent.pos of a teleporting reset included
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
  }
  if (!writer && tid == 0)
    __hip_atomic_store((s1_gu32*)(uintptr_t)(sy.words + (int64_t)blockIdx.x * RIAB_STEP1_SYNC_STRIDE + blockIdx.y), sy.epoch,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  if (wb && !TASK) seen = arrivals();
  auto write_back = [&]() {
    wait_arrivals();
    RIAB_S1_STAMP(5)
    {
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
  };

  // ---- Neurons.update of the population: this wave's cell groups for the segment's 256 agents ----------------------
  // (`store`: the lanes that write their quad's values — all of them, but for the pass that follows a reset, below)
  auto rates_pass = [&](const v4f rx, const v4f ry, const bool store) __attribute__((always_inline)) {
    v4f rhx = {0.0f, 0.0f, 0.0f, 0.0f}, rhy = rhx;
    if (ps.needs_hd) {
      rhx = *reinterpret_cast<const v4f*>(&s_row[2][4 * lane]);
      rhy = *reinterpret_cast<const v4f*>(&s_row[3][4 * lane]);
    }
    const uint32_t quad = blockIdx.x * 64u + (uint32_t)lane;  // the lane's quad of agents within the row
    float params = MINE_REG ? mine : s_par[wave][0][lane];
    int pi = pi0;
    Step1Pop q = pick(pi);
    for (int r = 0; r < reps; ++r) {
      const int g = g0 + r;
      if (g >= ps.total_groups) break;  // wave-uniform
      const float cur = params;
      if (r + 1 < reps) params = s_par[wave][r + 1][lane];  // (asked for before this group's stores are issued)
      if (MULTI && g >= q.group0 + q.n_groups && pi + 1 < ps.n_pops) q = pick(++pi);
      s1_group_any<SPK, NT, TASK != 0, KIND>(ps, q, g - q.group0, cur, rx, ry, rhx, rhy, B, quad, store);
    }
  };
  // (a task's step) this wave's cell groups again, for the quads `mq` of the segment alone (s1_group_few) — whose new
  // positions the lanes that hold them (`moved`) put into the row in LDS first: every wave of the workgroup writes the
  // same values there, and reads what it wrote itself
  auto rates_few = [&](const v4f rx, const v4f ry, const bool moved) __attribute__((always_inline)) {
    const unsigned long long mq = __builtin_amdgcn_ballot_w64(moved);
    if (moved) {
      *reinterpret_cast<v4f*>(&s_row[0][4 * lane]) = rx;
      *reinterpret_cast<v4f*>(&s_row[1][4 * lane]) = ry;
    }
    const v4f none = {0.0f, 0.0f, 0.0f, 0.0f};
    int pi = pi0;
    Step1Pop q = pick(pi);
    for (int r = 0; r < reps; ++r) {
      const int g = g0 + r;
      if (g >= ps.total_groups) break;  // wave-uniform
      const float cur = (MINE_REG && r == 0) ? mine : s_par[wave][r][lane];
      if (MULTI && g >= q.group0 + q.n_groups && pi + 1 < ps.n_pops) q = pick(++pi);
      s1_group_any<SPK, NT, TASK != 0, KIND, true>(ps, q, g - q.group0, cur, none, none, none, none, B, blockIdx.x * 64u, false, mq,
                                                   (lds_cf32_ptr)&s_row[0][0]);
    }
  };
  if (rates_here)
    rates_pass(*reinterpret_cast<const v4f*>(&s_row[0][4 * lane]), *reinterpret_cast<const v4f*>(&s_row[1][4 * lane]), true);
  RIAB_S1_STAMP(3)
#ifdef RIAB_STEP1_PROFILE
  if (prof_slot >= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RIAB_S1_STAMP(4)
  }
#endif
  // (a wave whose groups are all HeadDirectionCells' writes nothing a reset can change: no wait at all)
  bool wave_needs_pos = false;
  if (TASK && !writer) {
#pragma unroll
    for (int k = 0; k < (MULTI ? RIAB_STEP1_MAX_POPS : 1); ++k)  // (a population other than HeadDirectionCells with a group in this wave's range)
      wave_needs_pos = wave_needs_pos || (k < ps.n_pops && ps.pop[k].kind != S1_KIND_HDC && ps.pop[k].group0 < g0 + reps &&
                                          ps.pop[k].group0 + ps.pop[k].n_groups > g0);
  }
  if (WT && (WT & 2) && !writer && wave_needs_pos) {
    // ---- did the world's episode end?  Then every agent was teleported (where to is a function of (seed, counter,
    // agent) alone: this lane's quad works it out for itself) and the wave runs its pass again on the new row
    bool stale;
    const unsigned long long e = world_verdict(stale);
    const uint32_t flags = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e, 0);
    if (stale) {
      if (lane == 0) step1_note_timeout(sy, ps.step0);
    } else if ((flags & 1u) && tk.r.teleport) {
      v4f rx = *reinterpret_cast<const v4f*>(&s_row[0][4 * lane]), ry = *reinterpret_cast<const v4f*>(&s_row[1][4 * lane]);
      const int64_t b4 = (int64_t)blockIdx.x * 256 + 4 * lane;
      float nx[4], ny[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const ResetDraw d = reset_draw_id(tk.a, tk.r, (uint64_t)(tk.r.agent_id0 + b4 + k));
        nx[k] = (float)d.x;
        ny[k] = (float)d.y;
      }
      const bool l0 = b4 + 0 < tk.a.B, l1 = b4 + 1 < tk.a.B, l2 = b4 + 2 < tk.a.B, l3 = b4 + 3 < tk.a.B;  // (lanes of the task)
      rx.x = l0 ? nx[0] : rx.x; ry.x = l0 ? ny[0] : ry.x;
      rx.y = l1 ? nx[1] : rx.y; ry.y = l1 ? ny[1] : ry.y;
      rx.z = l2 ? nx[2] : rx.z; ry.z = l2 ? ny[2] : ry.z;
      rx.w = l3 ? nx[3] : rx.w; ry.w = l3 ? ny[3] : ry.w;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the values this wave stored a moment ago are in place)
      rates_pass(rx, ry, true);
    }
  }
  if (RT && !writer && wave_needs_pos && !(RIAB_S1_TASK_DROP & 8)) {
    // ---- did a reset move one of the segment's agents?  The writer's 4 x 6 verdict entries of this launch: one round
    // trip (they are usually there by now) says that they are posted, which agents moved and where (almost all of) them went.
    const s1_gu64* const mail64 = (const s1_gu64*)(uintptr_t)(tk.mail + (int64_t)blockIdx.x * RIAB_STEP1_MAIL_STRIDE);
    auto peek = [&](int at) -> unsigned long long {
      return __hip_atomic_load((s1_gu64*)(mail64 + at), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // entry `at`, last seen as `e`, once it carries this launch's epoch (bounded); lanes that do not `want` it pass
    auto fresh = [&](int at, bool want, unsigned long long e) -> unsigned long long {
      if (!want) e = (unsigned long long)sy.epoch << 32;
      for (uint32_t spins = 0; __builtin_amdgcn_ballot_w64((uint32_t)(e >> 32) != sy.epoch) != 0; ++spins) {
        if (spins >= sy.spin_limit) break;
        __builtin_amdgcn_s_sleep(4);
        if (want) e = peek(at);
      }
      return e;
    };
    const bool polls = lane < 32 && (lane & 7) < 6;  // lane 8 w + l: entry l of the writer's mover wave w
    const unsigned long long verdict = fresh(lane & 31, polls, polls ? peek(lane & 31) : 0ull);
    bool stale = __builtin_amdgcn_ballot_w64((uint32_t)(verdict >> 32) != sy.epoch) != 0;
    RIAB_S1_STAMP(7)
    // this lane's quad of agents 4 lane .. 4 lane + 3 of the segment: patched where the mail says an agent was moved
    v4f rx = *reinterpret_cast<const v4f*>(&s_row[0][4 * lane]), ry = *reinterpret_cast<const v4f*>(&s_row[1][4 * lane]);
    bool mine_moved = false;
    uint32_t extra4 = 0;  // this quad's agents whose positions did not ride in the verdict (third and later movers of a wave)
    auto patch = [&](int agent, uint32_t xbits, uint32_t ybits) {  // (uniform arguments)
      const bool here = lane == (agent >> 2);
      const int k = agent & 3;
      const float fx = __uint_as_float(xbits), fy = __uint_as_float(ybits);
      mine_moved = mine_moved || here;
      rx.x = (here && k == 0) ? fx : rx.x; ry.x = (here && k == 0) ? fy : ry.x;
      rx.y = (here && k == 1) ? fx : rx.y; ry.y = (here && k == 1) ? fy : ry.y;
      rx.z = (here && k == 2) ? fx : rx.z; ry.z = (here && k == 2) ? fy : ry.z;
      rx.w = (here && k == 3) ? fx : rx.w; ry.w = (here && k == 3) ? fy : ry.w;
    };
    if (!stale) {
      const int vlo = (int)(uint32_t)verdict;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned long long m = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(vlo, 8 * w) |
                                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(vlo, 8 * w + 1) << 32);
        if (m == 0) continue;  // (uniform)
        patch(w * 64 + __ffsll((long long)m) - 1, (uint32_t)__builtin_amdgcn_readlane(vlo, 8 * w + 2),
              (uint32_t)__builtin_amdgcn_readlane(vlo, 8 * w + 3));
        const unsigned long long m2 = m & (m - 1);
        if (m2 == 0) continue;
        patch(w * 64 + __ffsll((long long)m2) - 1, (uint32_t)__builtin_amdgcn_readlane(vlo, 8 * w + 4),
              (uint32_t)__builtin_amdgcn_readlane(vlo, 8 * w + 5));
        const unsigned long long m3 = m2 & (m2 - 1);
        if ((lane >> 4) == w) extra4 |= (uint32_t)(m3 >> (4 * (lane & 15))) & 15u;
      }
    }
    if (__builtin_amdgcn_ballot_w64(extra4 != 0u) != 0) {  // (rare) those positions: one batch per attempt, bounded
      unsigned long long ex[4] = {0, 0, 0, 0}, ey[4] = {0, 0, 0, 0};
      for (uint32_t spins = 0;; ++spins) {
        bool waiting = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          waiting = waiting || (((extra4 >> k) & 1u) && ((uint32_t)(ex[k] >> 32) != sy.epoch || (uint32_t)(ey[k] >> 32) != sy.epoch));
        if (__builtin_amdgcn_ballot_w64(waiting) == 0) break;
        if (spins >= sy.spin_limit) {
          stale = true;
          break;
        }
        if (spins) __builtin_amdgcn_s_sleep(4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((extra4 >> k) & 1u) {
            ex[k] = peek(32 + 4 * lane + k);
            ey[k] = peek(288 + 4 * lane + k);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool m = (extra4 >> k) & 1u;
        const float fx = __uint_as_float((uint32_t)ex[k]), fy = __uint_as_float((uint32_t)ey[k]);
        if (k == 0) { rx.x = m ? fx : rx.x; ry.x = m ? fy : ry.x; }
        if (k == 1) { rx.y = m ? fx : rx.y; ry.y = m ? fy : ry.y; }
        if (k == 2) { rx.z = m ? fx : rx.z; ry.z = m ? fy : ry.z; }
        if (k == 3) { rx.w = m ? fx : rx.w; ry.w = m ? fy : ry.w; }
      }
      mine_moved = mine_moved || extra4 != 0u;
    }
    if (!stale && __builtin_amdgcn_ballot_w64(mine_moved) != 0) {
      // (wave-uniform) this wave's cell groups again for the quads an agent of which was moved: the same pass — the
      // same instructions, hence the same bits as the population's own kernel gives on the patched history row — on
      // the row with the new positions in, stored by the lanes whose quad changed
      // (... by lanes that take one cell and one moved quad each — usually there is one, seldom more than three — instead
      // of every lane running all its cells for a quad that did not move: the pass, and the wave's wait for the writer, are
      // the tail of the step's critical path)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the values this wave stored for those quads a moment ago are in place)
      if (RIAB_S1_FEW) rates_few(rx, ry, mine_moved);
      else rates_pass(rx, ry, mine_moved);
    }
    RIAB_S1_STAMP(12)
    if (stale && lane == 0) step1_note_timeout(sy, ps.step0);
  }
  if (wb) write_back();
  // (the ended episodes' rows, last: their slots in the table were asked for a write-back ago)
  if ((TM & 2) && tlive) episode_log_store(tr, b, tmid.rec, tk.t_env, tk.diag);
#ifdef RIAB_STEP1_PROFILE
  if (prof_slot >= 0 && writer) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RIAB_S1_STAMP(6)
  }
#endif
}

// The kernel's arguments are ~1.4 KB (the populations' records, the motion step's constants, the task's tables): 23
// lines of the scalar cache, cold at every launch.  Left to the compiler, a workgroup meets them in five or six
// DEPENDENT rounds of scalar loads in front of its first barrier (load what the next test needs, wait, branch, load
// ...), each round with a miss in it; one word of every line asked for at entry makes that one round.  [MI355X] a task's
// step (whose writer reads the most of them): 13.15 -> 12.87 us; the plain step, which meets fewer: + 0.08 us — not there.
#ifndef RIAB_S1_KERNARG_WARM
#define RIAB_S1_KERNARG_WARM 1
#endif
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if RIAB_S1_KERNARG_WARM
  // (ONE statement: the loads' target is a register the compiler may hand to somebody else only behind the wait; the
  // values are not used — loads that return out of order into the same register harm nobody)
  static_assert(BYTES > 64 && BYTES <= 26 * 64, "one load per line of the argument block");
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();  // (a pointer into the constant address space: a scalar pair)
  uint32_t t;
#define RIAB_KA_OFF(i) "i"((64 * (i) < BYTES - 4) ? 64 * (i) : BYTES - 4)
  asm volatile(
      "s_load_dword %0, %1, %2\n s_load_dword %0, %1, %3\n s_load_dword %0, %1, %4\n s_load_dword %0, %1, %5\n"
      "s_load_dword %0, %1, %6\n s_load_dword %0, %1, %7\n s_load_dword %0, %1, %8\n s_load_dword %0, %1, %9\n"
      "s_load_dword %0, %1, %10\n s_load_dword %0, %1, %11\n s_load_dword %0, %1, %12\n s_load_dword %0, %1, %13\n"
      "s_load_dword %0, %1, %14\n s_load_dword %0, %1, %15\n s_load_dword %0, %1, %16\n s_load_dword %0, %1, %17\n"
      "s_load_dword %0, %1, %18\n s_load_dword %0, %1, %19\n s_load_dword %0, %1, %20\n s_load_dword %0, %1, %21\n"
      "s_load_dword %0, %1, %22\n s_load_dword %0, %1, %23\n s_load_dword %0, %1, %24\n s_load_dword %0, %1, %25\n"
      "s_load_dword %0, %1, %26\n s_load_dword %0, %1, %27\n s_waitcnt lgkmcnt(0)"
      : "=&s"(t)
      : "s"(ka), RIAB_KA_OFF(0), RIAB_KA_OFF(1), RIAB_KA_OFF(2), RIAB_KA_OFF(3), RIAB_KA_OFF(4), RIAB_KA_OFF(5), RIAB_KA_OFF(6),
        RIAB_KA_OFF(7), RIAB_KA_OFF(8), RIAB_KA_OFF(9), RIAB_KA_OFF(10), RIAB_KA_OFF(11), RIAB_KA_OFF(12), RIAB_KA_OFF(13),
        RIAB_KA_OFF(14), RIAB_KA_OFF(15), RIAB_KA_OFF(16), RIAB_KA_OFF(17), RIAB_KA_OFF(18), RIAB_KA_OFF(19), RIAB_KA_OFF(20),
        RIAB_KA_OFF(21), RIAB_KA_OFF(22), RIAB_KA_OFF(23), RIAB_KA_OFF(24), RIAB_KA_OFF(25)
      : "memory");
#undef RIAB_KA_OFF
#endif
}
struct Step1KernArgs {  // (the argument block of step1_kernel as the ABI lays it out)
  AgentArgs a;
  Step1Pops ps;
  Step1Sync sy;
  int reps;
  MotionConst<double> hk;
  TailConst<double> tail_c;
};
struct Step1TaskKernArgs {
  Step1KernArgs k;
  Step1Task tk;
};

template <int SPK, bool NT, int KIND>
__global__ __launch_bounds__(64 * RIAB_S1_WAVES, RIAB_S1_WAVES_PER_EU) void step1_kernel(const AgentArgs a, const Step1Pops ps,
                                                                                           const Step1Sync sy, const int reps,
                                                                                           const MotionConst<double> hk,
                                                                                           const TailConst<double> tail_c) {
  const Step1Task none = {};
  step1_body<SPK, NT, 0, KIND>(a, ps, sy, reps, hk, tail_c, none);
}

template <int SPK, int TASK, int KIND>
__global__ __launch_bounds__(64 * RIAB_S1_WAVES, RIAB_S1_WAVES_PER_EU) void step1_task_kernel(const AgentArgs a, const Step1Pops ps,
                                                                                                const Step1Sync sy, const int reps,
                                                                                                const MotionConst<double> hk,
                                                                                                const TailConst<double> tail_c,
                                                                                                const Step1Task tk) {
  // (not where the kernel draws spikes: there the statement's registers cost one instantiation — the one-world step of
  // PlaceCells with spikes, `<1, 15, 0>` — a 36-byte stack frame, and a kernel with a frame is not used at all)
  if (SPK == 0) kernarg_warm<(int)sizeof(Step1TaskKernArgs) + 32>();  // (+ the grid's size, the first of the implicit arguments)
  step1_body<SPK, true, TASK, KIND>(a, ps, sy, reps, hk, tail_c, tk);
}

// ---- host side --------------------------------------------------------------------------------------------------
// How a (B, groups) problem is cut: `reps` cell groups per wave so that the grid is one round of workgroups on the
// compute units the plan's launches can occupy (`resident`: workgroups of this kernel the device holds at once — a
// second round would run the motion step a second time, and a task's step must not have one at all, see below) and a
// segment's workgroups fit its line of arrival words.
// (task: one of a segment's workgroups — its writer — keeps the task's books instead of writing rates)
static int step1_shape(int64_t B, int64_t groups, bool task, int64_t resident, dim3* grid, int* reps) {
  const int64_t segs = B / 256;
  int64_t want_y = resident / segs;  // workgroups per segment in one resident round
  if (want_y > RIAB_STEP1_SYNC_MAX_Y) want_y = RIAB_STEP1_SYNC_MAX_Y;
  if (task) want_y -= 1;
  if (want_y < 1) {
    if (task) return RIAB_EUNSUPPORTED;  // (no room for a writer and a rate workgroup per segment: two launches)
    want_y = 1;
  }
  int64_t r = (groups + RIAB_S1_WAVES * want_y - 1) / (RIAB_S1_WAVES * want_y);
  if (r < 1) r = 1;
  if (r > RIAB_S1_STAGE) return RIAB_EUNSUPPORTED;  // (a device this much smaller than the problem: kernel by kernel)
  const int64_t gy = (groups + RIAB_S1_WAVES * r - 1) / (RIAB_S1_WAVES * r);
  *reps = (int)r;
  *grid = dim3((unsigned)segs, (unsigned)(gy + (task ? 1 : 0)), 1);
  return RIAB_OK;
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

static int step1_fill_pop(Step1Pop& q, const RiabEnv* env, const Step1PopRef& ref, bool task, bool spikes, int32_t group0) {
  const RiabPopulation* pop = ref.pop;
  if (!ref.rates_row) return RIAB_EINVAL;
  if ((((uintptr_t)ref.rates_row) & 15) || (((uintptr_t)ref.spikes_row) & 3)) return RIAB_EALIGN;
  q.tab = pop->table;
  q.rates = ref.rates_row;
  q.spikes = ref.spikes_row;
  q.n = pop->n;
  q.fr_scale = pop->io.max_fr - pop->io.min_fr;
  q.fr_min = pop->io.min_fr;
  q.p0 = q.p1 = q.p2 = 0.0f;
  q.tag = RIAB_TAG_SPIKES | ((uint32_t)pop->io.pop_id & 0xFFu);
  switch (pop->kind) {
    case RIAB_POP_PLACE: {
      int d;
      switch (pop->description) {
        case RIAB_PC_GAUSSIAN: d = 0; break;
        case RIAB_PC_GAUSSIAN_THRESHOLD: d = 1; break;
        case RIAB_PC_DIFF_OF_GAUSSIANS: d = 2; break;
        case RIAB_PC_TOP_HAT: d = 3; break;
        default: return RIAB_EUNSUPPORTED;
      }
      q.kind = S1_KIND_PC + (env->periodic ? 4 : 0) + d;
      q.p0 = (float)env->scale;
      q.p1 = (float)(env->scale / 2);
      q.p2 = pop->top_hat_width * pop->top_hat_width;
      break;
    }
    case RIAB_POP_GRID:
      q.kind = S1_KIND_GC + (pop->description == RIAB_GC_RECTIFIED ? RIAB_GC_RECTIFIED : RIAB_GC_SHIFTED);
      q.p0 = pop->f0;
      q.p1 = pop->description == RIAB_GC_RECTIFIED ? 1.0f / (1.0f - pop->f0) : 1.0f;
      break;
    case RIAB_POP_HDC: q.kind = S1_KIND_HDC; break;
    default: return RIAB_EUNSUPPORTED;
  }
  const int cpb = s1_cpb(q.kind, task, spikes);  // (`spikes`: of the LAUNCH — the kernel's SPK — not of this population)
  q.group0 = group0;
  q.n_groups = (pop->n + cpb - 1) / cpb;
  return RIAB_OK;
}

// one Agent.update() (the arguments of riab_agent_step(T = 1), Philox noise) + the update() of `n_pops` populations on the
// row it writes: `refs[i].rates_row` / `spikes_row` are population i's rows of this step, `step_after` the number of agent
// steps taken once this one is done (Neurons.update's spike counter, as riab_plan_step passes it).  `n_cus`: the compute
// units the launch can occupy (riab_plan_set_compute_units).
static int launch_step1_impl(const AgentArgs& a, const RiabEnv* env, const Step1PopRef* refs, int n_pops, uint64_t seed,
                             uint64_t step_after, uint32_t* sync_words, uint32_t epoch, bool* walls_ready, int n_cus, hipStream_t s,
                             const Step1Task* tk, int task_mode, bool query = false) {
  if (n_pops < 1 || n_pops > RIAB_STEP1_MAX_POPS || !refs) return RIAB_EINVAL;
  for (int i = 0; i < n_pops; ++i) {
    const int rc = step1_supported(env, refs[i].pop, a.B);
    if (rc) return rc;
  }
  if (a.z_in || a.z_out || a.forced || a.T != 1 || !sync_words || epoch == 0u || n_cus < 1) return RIAB_EINVAL;
  if (a.agent_id0 % 4) return RIAB_EALIGN;
  Step1Pops ps = {};
  ps.n_pops = n_pops;
  ps.dt = (float)a.m.dt;
  ps.k0 = (uint32_t)seed;
  ps.k1 = (uint32_t)(seed >> 32);
  ps.step0 = (uint32_t)step_after;
  ps.quad0 = (uint32_t)(a.agent_id0 / 4);
  int32_t groups = 0;
  bool spikes = false;
  for (int i = 0; i < n_pops; ++i) spikes = spikes || refs[i].spikes_row != nullptr;
  for (int i = 0; i < n_pops; ++i) {
    const int rc = step1_fill_pop(ps.pop[i], env, refs[i], tk != nullptr, spikes, groups);
    if (rc) return rc;
    groups += ps.pop[i].n_groups;
    if (ps.pop[i].kind == S1_KIND_HDC) ps.needs_hd = 1;
  }
  ps.total_groups = groups;
  Step1Sync sy;
  sy.words = sync_words;
  sy.epoch = epoch;
  sy.spin_limit = g_options[RIAB_OPT_STEP1_SPIN] ? 1u << g_options[RIAB_OPT_STEP1_SPIN] : 0u;  // x ~0.3 us: about a second by default
  sy.n_segments = (uint32_t)(a.B / 256);
  Wall<double>* const gw = reinterpret_cast<Wall<double>*>(sync_words + RIAB_STEP1_SYNC_WALLS_AT(a.B));
  sy.walls = gw;
  // the launch's scalar constants, once, here (float64 divisions the kernel would otherwise repeat per thread and step)
  MotionConst<double> hk = {};
  motion_const_scalars<double>(hk, a);
  const TailConst<double> tc = {a.m.dt, hk.inv_dt, 1.0 - a.m.dt / a.m.hd_tau, a.m.dt / a.m.hd_tau, a.m.hd_tau <= a.m.dt};
  const dim3 block(64 * RIAB_S1_WAVES);
  // the plan's only population, gaussian PlaceCells in a solid room (BASELINE configs[1]'s closed loop): the kernel compiled
  // for that functor, its record read at static offsets of the argument block; everything else: the generic kernel
#ifndef RIAB_S1_FORCE_GENERIC
#define RIAB_S1_FORCE_GENERIC 0
#endif
  const bool multi = n_pops > 1 || ps.pop[0].kind != S1_KIND_PC || RIAB_S1_FORCE_GENERIC;
  const bool check = g_options[RIAB_OPT_STEP1_RESIDENCY] != 0;  // (A/B, tests: 0 = the grid of a whole, idle MI355X whatever the device)
  dim3 grid;
  int reps = 1;
  auto prepare_walls = [&]() {
    if (walls_ready && !*walls_ready) {  // (the plan's first one-launch step, and the first after its motion parameters changed)
      hipLaunchKernelGGL(walls_prepare_kernel, dim3(1), dim3(64), 0, s, a, gw);
      *walls_ready = true;
    }
  };
  if (tk) {  // (non-temporal stores; TASK = task_kernel's MODE)
    // The task step's workgroups wait for each other in both directions (the writer for the others' arrival, the others
    // for the writer's verdict): the whole grid must be resident at once.  How many of its workgroups a compute unit
    // holds (one, at 160+ registers per lane) is asked of the runtime once per instantiation; how many compute units
    // the plan's launches can occupy is the plan's (`n_cus`: the device's count unless the caller measured fewer — a
    // CU-masked process, a partitioned device).  Where a segment has no room for a second workgroup the plan keeps its
    // two launches.
    // A kernel the register allocator gave a stack frame (a few bytes of spilled scalars, never touched) would have the
    // dispatcher set scratch memory up for every launch — more than the fusion saves: such an instantiation is not used.
    auto launch = [&](auto kernel) -> int {
      // (per KERNEL: the lambda's own statics would be shared by every instantiation — they all have one function type —
      // and the first kernel asked about would answer for all of them)
      static std::mutex info_lock;
      static std::map<const void*, std::pair<int, int>> info;
      std::lock_guard<std::mutex> guard(info_lock);
      std::pair<int, int>& fi = info.emplace(reinterpret_cast<const void*>(kernel), std::make_pair(-1, 0)).first->second;
      int& frame = fi.first;
      int& per_cu = fi.second;
      if (frame < 0) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) != hipSuccess) return RIAB_EUNSUPPORTED;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 64 * RIAB_S1_WAVES, 0) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb;
        frame = (int)fa.localSizeBytes;
      }
#ifndef RIAB_STEP1_PROFILE  // (a profiling build's extra stores may cost an instantiation its frame: the phases are what it is for)
      if (frame > 0) return RIAB_EUNSUPPORTED;
#endif
      const int64_t resident = check ? (int64_t)n_cus * per_cu : 2048 / RIAB_S1_WAVES;
      const int rc = step1_shape(a.B, groups, true, resident, &grid, &reps);
      if (rc) return rc;
      if ((int64_t)grid.x * grid.y > resident) return RIAB_EUNSUPPORTED;
      if (query) return RIAB_OK;
      prepare_walls();
      hipLaunchKernelGGL(kernel, grid, block, 0, s, a, ps, sy, reps, hk, tc, *tk);
      return (int)hipGetLastError();
    };
#define RIAB_S1_TASK(MODE)                                                                                      \
  return multi ? (spikes ? launch(step1_task_kernel<1, MODE, -1>) : launch(step1_task_kernel<0, MODE, -1>)) \
               : (spikes ? launch(step1_task_kernel<1, MODE, 0>) : launch(step1_task_kernel<0, MODE, 0>))
    switch (task_mode) {
      case 1: RIAB_S1_TASK(1);
      case 3: RIAB_S1_TASK(3);
      case 5: RIAB_S1_TASK(5);
      case 7: RIAB_S1_TASK(7);
      case 9: RIAB_S1_TASK(9);  // (| 8: the lanes are the agents of one world)
      case 11: RIAB_S1_TASK(11);
      case 13: RIAB_S1_TASK(13);
      case 15: RIAB_S1_TASK(15);
      default: return RIAB_EINVAL;
    }
#undef RIAB_S1_TASK
  }
  // (no workgroup of this form waits for one that waits: nothing has to be resident together — the shape only keeps the
  // motion step from being run more often than the device has room for)
  const int rc_shape = step1_shape(a.B, groups, false, check ? n_cus : 2048 / RIAB_S1_WAVES, &grid, &reps);
  if (rc_shape) return rc_shape;
  if (query) return RIAB_OK;
  prepare_walls();
  const bool nt = g_options[RIAB_OPT_FUSED_STEP] != 2;
#define RIAB_S1_PLAIN(SPK, NT)                                                                              \
  if (multi) hipLaunchKernelGGL((step1_kernel<SPK, NT, -1>), grid, block, 0, s, a, ps, sy, reps, hk, tc); \
  else hipLaunchKernelGGL((step1_kernel<SPK, NT, 0>), grid, block, 0, s, a, ps, sy, reps, hk, tc)
  if (spikes) {
    if (nt) { RIAB_S1_PLAIN(1, true); } else { RIAB_S1_PLAIN(1, false); }
  } else {
    if (nt) { RIAB_S1_PLAIN(0, true); } else { RIAB_S1_PLAIN(0, false); }
  }
#undef RIAB_S1_PLAIN
  return (int)hipGetLastError();
}

int launch_step1(const AgentArgs& a, const RiabEnv* env, const Step1PopRef* refs, int n_pops, uint64_t seed, uint64_t step_after,
                 uint32_t* sync_words, uint32_t epoch, bool* walls_ready, int n_cus, hipStream_t s, bool query) {
  return launch_step1_impl(a, env, refs, n_pops, seed, step_after, sync_words, epoch, walls_ready, n_cus, s, nullptr, 0, query);
}

// ... + the rest of TaskEnvironment.step for the first `task_B` lanes: the arguments of launch_motion_task (riab_agent.hip),
// whose two stages this replaces together with the fused populations' launches
// (`query`: nothing is launched; RIAB_OK when this plan's step has a kernel to be launched with)
int launch_step1_task(const AgentArgs& a, const RiabEnv* env, const Step1PopRef* refs, int n_pops, uint64_t seed,
                      uint64_t step_after, uint32_t* sync_words, uint32_t epoch, bool* walls_ready, int n_cus, const RiabTask* task,
                      double* task_state, int64_t task_B, double t_env, double* reward_out, uint8_t* terminal_out, int32_t* diag,
                      bool auto_reset, int32_t n_select, int32_t ordered, uint64_t task_seed, uint64_t counter, int32_t teleport,
                      double* ep_log, int64_t ep_log_cap, int32_t* ep_count, double gv_scale, double* gv_x, double* gv_y,
                      double* world, uint64_t* world_met, int32_t* world_cand, int32_t* world_ctl, hipStream_t s, bool query) {
  Step1Task tk = {};
  int rc = fill_args(tk.a, env, task, task_state, task_B);
  if (rc) return rc;
  if (task_B > a.B || !reward_out || !terminal_out || !diag || !sync_words) return RIAB_EINVAL;
  if (world && (!world_met || !world_cand || !world_ctl || task_B > 0x7FFFFFFF)) return RIAB_EINVAL;
  tk.world = world;  // (non-null: the lanes are the agents of ONE world)
  tk.met = world_met;
  tk.cand = world_cand;
  tk.ctl = world_ctl;
  if (auto_reset) {
    double* const pos_x = a.state + (int64_t)RIAB_S_POS_X * a.B;
    rc = fill_reset(tk.r, env, a.agent_id0, n_select, ordered, task_seed, counter, teleport, nullptr, nullptr, pos_x, pos_x + a.B,
                    nullptr, nullptr, ep_log, ep_log_cap, ep_count);
    if (rc) return rc;
    tk.r.pos_x = tk.r.pos_y = nullptr;  // (the writer stores the position the lane hands back, with the state)
  }
  tk.t_env = t_env;
  tk.reward_out = reward_out;
  tk.terminal_out = terminal_out;
  tk.gv_scale = gv_scale;
  tk.gv_x = gv_x;
  tk.gv_y = gv_y;
  tk.diag = diag;
  tk.mail = sync_words + RIAB_STEP1_SYNC_MAIL_AT(a.B);
  const int mode = 1 | (auto_reset ? 2 : 0) | (gv_x ? 4 : 0) | (world ? 8 : 0);
  return launch_step1_impl(a, env, refs, n_pops, seed, step_after, sync_words, epoch, walls_ready, n_cus, s, &tk, mode, query);
}

// The compute units a stream's workgroups land on, counted: every one-wave workgroup of a grid that fills any device
// several times over marks the unit it runs on (XCC_ID and HW_ID's shader engine / array / unit fields); a second, tiny
// kernel counts the marks into scratch[RIAB_CU_PROBE_WORDS - 1].  A process whose queues carry a CU mask (HSA_CU_MASK,
// ROC_GLOBAL_CU_MASK, a stream made with hipExtStreamCreateWithCUMask) reports the full device through
// hipGetDeviceProperties; this does not.
__global__ __launch_bounds__(64) void cu_probe_kernel(uint32_t* seen) {
  unsigned id, xc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
  if (threadIdx.x == 0) seen[((xc & 15u) << 8) | ((id >> 8) & 0xFFu)] = 1u;  // HW_ID [15:8]: CU_ID, SH_ID, SE_ID
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // (100 MHz: stay ~4 us, so that the grid spreads)
  while (__builtin_amdgcn_s_memrealtime() - t0 < 400ull) __builtin_amdgcn_s_sleep(16);
}
__global__ __launch_bounds__(256) void cu_probe_count_kernel(uint32_t* seen) {
  __shared__ uint32_t total;
  if (threadIdx.x == 0) total = 0u;
  __syncthreads();
  uint32_t mine = 0;
  for (int i = (int)threadIdx.x; i < RIAB_CU_PROBE_WORDS - 1; i += 256) mine += seen[i] ? 1u : 0u;
  atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) seen[RIAB_CU_PROBE_WORDS - 1] = total;
}

}  // namespace riab

extern "C" int riab_probe_compute_units(uint32_t* scratch, riab_stream_t stream) {
  if (!scratch) return RIAB_EINVAL;
  if (((uintptr_t)scratch) & 3) return RIAB_EALIGN;
  hipLaunchKernelGGL(riab::cu_probe_kernel, dim3(32768), dim3(64), 0, (hipStream_t)stream, scratch);
  hipLaunchKernelGGL(riab::cu_probe_count_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch);
  return (int)hipGetLastError();
}
