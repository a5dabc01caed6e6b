// TaskEnvironment whose lanes are the agents of ONE world, agentmode = "interact" (the reference's default,
// contribs/TaskEnvironment.py:1030): one clock, one episode, ONE goal list — a goal is consumed for everybody by the
// first agent, in agent order, found inside it (GoalCache.check :1076-1152 with GoalCache.pop :1165-1172) — and a
// reward cache per agent.  (riab_task.hip is the other batching: every lane a single-agent replica of the task.)
//
// The reference's check is serial over the agents: each takes its turn against the list as the agents before it left
// it.  What can be done per agent without looking at the others is done by every lane at once (phase A); the serial
// remainder is a handful of turns — only an agent standing in a goal of the list changes anything, and every such turn
// consumes at least one of the <= 16 entries — found by min-reductions over the work list of such agents (phase B).
//
//   phase A, all workgroups: RewardCache.update of the lane (:913-927), its mask of the list's goals it stands in
//     (SpatialGoal.check :1337-1360: line-of-sight distance < radius); a lane that stands in none is done — total
//     (:929-939), statistics —, the others ("candidates") put themselves on a work list;
//   phase B, the workgroup that finishes LAST (a ticket counter; no workgroup waits for another: nothing has to be
//     resident together, capturable): the step's check passes (:418-440) over the shared list among the candidates, the
//     awards appended to the winners' caches in award order, the candidates' totals, the shared state.  A quiet step —
//     nobody stands in anything — finds an empty work list: a ticket, three empty passes, nothing stored.
//
// Float64 like the reference; contraction off (riab_task_kernel.h).
#include "riab_agent_kernel.h"  // (the motion step: the plan launches it and the world step as one kernel)
#include "riab_task_world_kernel.h"  // (after it: its headers turn fp contraction off for their own code)

#pragma clang fp contract(off)

namespace riab {

constexpr int WORLD_BLOCK = 256;

// ctl: [0] ticket of the workgroups, [1] number of candidates (both zero between launches).  The body of the step for
// workgroups of BLOCK lanes: the stand-alone kernel (256) and the plan's motion + world step launch (64, behind the
// motion step's body) run the same code (the pieces: riab_task_world_kernel.h).
template <int BLOCK>
__device__ __forceinline__ void world_step_body(const TaskArgs& a, double* world, const double* pos_x, const double* pos_y,
                                                double t_env, double* reward_out, uint8_t* terminal_out, uint64_t* met,
                                                int32_t* cand, int32_t* ctl, int32_t* diag) {
  __shared__ double s_goals[RIAB_TASK_MAX_POOL * RIAB_GOAL_COLS];
  __shared__ WorldShared S;
  const int tid = (int)threadIdx.x;
  const int64_t b = (int64_t)blockIdx.x * BLOCK + tid;
  const bool live = b < a.B;
  // ---- phase A: what an agent's step needs of nobody else
  task_stage_goals(a, s_goals, tid, BLOCK);
  if (tid < RIAB_TASK_MAX_GOALS) S.list[tid] = (uint8_t)((int)world[RIAB_TW_GOAL_LIST + tid] & 0xFF);
  if (tid == 0) S.n = (int)world[RIAB_TW_N_GOALS];
  const double pad_start0 = world[RIAB_TW_PAD_START];
  const uint8_t terminal_prev = world[RIAB_TW_TERMINAL] != 0.0 ? 1 : 0;
  WorldLaneIn in;
  double px = 0.0, py = 0.0;
  if (live) {  // (one batch of independent loads)
    in = world_lane_load(a, b);
    px = pos_x[b];
    py = pos_y[b];
  }
  __syncthreads();
  const lds_f64_ptr goals = (lds_f64_ptr)s_goals;
  if (live) world_phase_a(a, goals, S, b, px, py, in, t_env, pad_start0, terminal_prev, reward_out, terminal_out, met, cand, ctl);
  // ---- the last workgroup to get here goes on: what phase B reads of the others' phase A was written through and has
  // been acknowledged — waited for HERE, wave by wave (a barrier alone does not wait for global stores: the compiler puts
  // `lgkmcnt(0)` in front of it)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) S.last = atomicAdd(ctl, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!S.last) return;
  // ---- phase B: the step's check passes over the shared list
  world_phase_b<BLOCK>(a, goals, S, world, t_env, pad_start0, terminal_prev, reward_out, terminal_out, met, cand, ctl, diag);
}

__global__ __launch_bounds__(WORLD_BLOCK) void task_world_step_kernel(TaskArgs a, double* world, const double* pos_x,
                                                                       const double* pos_y, double t_env, double* reward_out,
                                                                       uint8_t* terminal_out, uint64_t* met, int32_t* cand,
                                                                       int32_t* ctl, int32_t* diag) {
  world_step_body<WORLD_BLOCK>(a, world, pos_x, pos_y, t_env, reward_out, terminal_out, met, cand, ctl, diag);
}

// Agent.update and the world's step in ONE launch (the step plan): each one-wave workgroup moves its 64 agents (the
// stand-alone motion kernel's body), reads back the positions it has just written and goes on with the world's step; the
// last of them to finish walks the shared list.  What is saved is one dependent dispatch per step.
__global__ __launch_bounds__(64) void motion_world_kernel(const AgentArgs ma, TaskArgs a, double* world, const double* pos_x,
                                                          const double* pos_y, double t_env, double* reward_out,
                                                          uint8_t* terminal_out, uint64_t* met, int32_t* cand, int32_t* ctl,
                                                          int32_t* diag) {
  agent_step_body<double, 0, false>(ma);
  world_step_body<64>(a, world, pos_x, pos_y, t_env, reward_out, terminal_out, met, cand, ctl, diag);
}

// TaskEnvironment.reset (:307-351) of the world: the episode table and the goal selection once (workgroup 0's first
// thread), teleport_on_reset for every agent (:323-330).
// `only_if_terminal`: the caller's `if terminal: env.reset()` without a host round trip — nothing is reset unless the last
// step left the world's flag set.  (No reset touches RIAB_TW_TERMINAL — here other workgroups are still reading it —: it
// stays "what the last step wrote into terminal_out", and the next step's last workgroup rewrites both.)
// `gv_x`: the goal vector / scripted action of every agent AFTER the reset-or-not (the step plan's next action, in the
// same launch).  What the world draws at a reset is a function of (seed, counter) alone: every lane that needs the new
// list works it out for itself instead of waiting for the thread that stores it.
__global__ __launch_bounds__(WORLD_BLOCK) void task_world_reset_kernel(TaskArgs a, ResetArgs r, double* world, double t_env,
                                                                        int only_if_terminal, double gv_scale, double* gv_x,
                                                                        double* gv_y, int32_t* diag) {
  __shared__ double s_goals[RIAB_TASK_MAX_POOL * RIAB_GOAL_COLS];
  const int64_t b = (int64_t)blockIdx.x * WORLD_BLOCK + threadIdx.x;
  const bool gv = gv_x != nullptr;
  const bool reset = !only_if_terminal || world[RIAB_TW_TERMINAL] != 0.0;
  if (!reset && !gv) return;
  if (gv) task_stage_goals(a, s_goals, (int)threadIdx.x, WORLD_BLOCK);
  Lane L;
  L.px = L.py = 0.0;
  ResetDraw world_draw = {0.0, 0.0, 0};
  if (reset) {  // GoalCache.reset (:1218-1252): one selection, appended to every agent's list — the shared list
    if (gv || b == 0) world_draw = reset_draw_id(a, r, RIAB_WORLD_STREAM_ID);
    L.list = world_draw.list;
    L.n_goals = r.n_select < a.n_pool ? r.n_select : a.n_pool;
  } else {
    world_list_load(world, L);
  }
  if (b < a.B) {
    if (reset && r.teleport) {
      ResetDraw d = {0.0, 0.0, 0};
      if (!r.new_x) d = reset_draw_id(a, r, (uint64_t)(r.agent_id0 + b));
      reset_lane_teleport(r, b, L, d);
    } else if (gv) {
      L.px = r.pos_x[b];
      L.py = r.pos_y[b];
    }
  }
  if (gv) {
    __syncthreads();
    if (b < a.B) {
      double vx, vy;
      goal_vector(a, (lds_f64_ptr)s_goals, L, gv_scale, vx, vy);
      gv_x[b] = vx;
      gv_y[b] = vy;
    }
  }
  if (b != 0 || !reset) return;
  world_reset_books(a, r, world, t_env, L, diag);
}

// get_goal_vector (:1555-1584) of every agent against the shared list
__global__ __launch_bounds__(WORLD_BLOCK) void task_world_goal_vector_kernel(TaskArgs a, const double* world, const double* pos_x,
                                                                              const double* pos_y, double scale, double* out_x,
                                                                              double* out_y) {
  __shared__ double s_goals[RIAB_TASK_MAX_POOL * RIAB_GOAL_COLS];
  const int64_t b = (int64_t)blockIdx.x * WORLD_BLOCK + threadIdx.x;
  task_stage_goals(a, s_goals, (int)threadIdx.x, WORLD_BLOCK);
  Lane L;
  world_list_load(world, L);
  __syncthreads();
  if (b >= a.B) return;
  L.px = pos_x[b];
  L.py = pos_y[b];
  double vx, vy;
  goal_vector(a, (lds_f64_ptr)s_goals, L, scale, vx, vy);
  out_x[b] = vx;
  out_y[b] = vy;
}

static int fill_world_args(TaskArgs& a, const RiabEnv* env, const RiabTask* task, double* task_state, double* world, int64_t B) {
  if (!world) return RIAB_EINVAL;
  if (B > 0x7FFFFFFF) return RIAB_ETOOBIG;
  return fill_args(a, env, task, task_state, B);
}

// the step plan's motion + world step launch (riab_plan.hip); `ma`: the motion step of the plan's (padded) batch
int launch_motion_world(const AgentArgs& ma, const RiabEnv* env, const RiabTask* task, double* task_state, double* world,
                        const double* pos_x, const double* pos_y, int64_t task_B, double t_env, double* reward_out,
                        uint8_t* terminal_out, uint64_t* met, int32_t* cand, int32_t* ctl, int32_t* diag, hipStream_t s) {
  TaskArgs a;
  const int rc = fill_world_args(a, env, task, task_state, world, task_B);
  if (rc) return rc;
  if (!pos_x || !pos_y || !reward_out || !terminal_out || !met || !cand || !ctl || !diag) return RIAB_EINVAL;
  hipLaunchKernelGGL(motion_world_kernel, dim3((unsigned)((ma.B + 63) / 64)), dim3(64), 0, s, ma, a, world, pos_x, pos_y, t_env,
                     reward_out, terminal_out, met, cand, ctl, diag);
  return (int)hipGetLastError();
}

}  // namespace riab

using namespace riab;

extern "C" int riab_task_world_step(const RiabEnv* env, const RiabTask* task, double* task_state, double* world,
                                    const double* pos_x, const double* pos_y, int64_t B, double t_env, double* reward_out,
                                    uint8_t* terminal_out, uint64_t* met_scratch, int32_t* cand_scratch, int32_t* ctl,
                                    int32_t* diag, riab_stream_t stream) {
  TaskArgs a;
  const int rc = fill_world_args(a, env, task, task_state, world, B);
  if (rc) return rc;
  if (!pos_x || !pos_y || !reward_out || !terminal_out || !met_scratch || !cand_scratch || !ctl || !diag) return RIAB_EINVAL;
  hipLaunchKernelGGL(task_world_step_kernel, dim3((unsigned)((B + WORLD_BLOCK - 1) / WORLD_BLOCK)), dim3(WORLD_BLOCK), 0,
                     (hipStream_t)stream, a, world, pos_x, pos_y, t_env, reward_out, terminal_out, met_scratch, cand_scratch,
                     ctl, diag);
  return (int)hipGetLastError();
}

extern "C" int riab_task_world_reset(const RiabEnv* env, const RiabTask* task, double* task_state, double* world, int64_t B,
                                     int64_t agent_id0, double t_env, int32_t n_select, int32_t ordered, uint64_t seed,
                                     uint64_t counter, int32_t teleport, const double* new_x, const double* new_y,
                                     double* pos_x, double* pos_y, float* hist_x, float* hist_y, double* ep_log,
                                     int64_t ep_log_cap, int32_t* ep_count, int32_t only_if_terminal, double gv_scale,
                                     double* gv_x, double* gv_y, int32_t* diag, riab_stream_t stream) {
  TaskArgs a;
  int rc = fill_world_args(a, env, task, task_state, world, B);
  if (rc) return rc;
  if (!diag || (gv_x == nullptr) != (gv_y == nullptr) || (gv_x && (!pos_x || !pos_y))) return RIAB_EINVAL;
  ResetArgs r;
  rc = fill_reset(r, env, agent_id0, n_select, ordered, seed, counter, teleport, new_x, new_y, pos_x, pos_y, hist_x, hist_y,
                  ep_log, ep_log_cap, ep_count);
  if (rc) return rc;
  hipLaunchKernelGGL(task_world_reset_kernel, dim3((unsigned)((B + WORLD_BLOCK - 1) / WORLD_BLOCK)), dim3(WORLD_BLOCK), 0,
                     (hipStream_t)stream, a, r, world, t_env, (int)only_if_terminal, gv_scale, gv_x, gv_y, diag);
  return (int)hipGetLastError();
}

extern "C" int riab_task_world_goal_vector(const RiabEnv* env, const RiabTask* task, double* task_state, const double* world,
                                           const double* pos_x, const double* pos_y, int64_t B, double scale, double* out_x,
                                           double* out_y, riab_stream_t stream) {
  TaskArgs a;
  const int rc = fill_world_args(a, env, task, task_state, const_cast<double*>(world), B);
  if (rc) return rc;
  if (!pos_x || !pos_y || !out_x || !out_y) return RIAB_EINVAL;
  hipLaunchKernelGGL(task_world_goal_vector_kernel, dim3((unsigned)((B + WORLD_BLOCK - 1) / WORLD_BLOCK)), dim3(WORLD_BLOCK), 0,
                     (hipStream_t)stream, a, world, pos_x, pos_y, scale, out_x, out_y);
  return (int)hipGetLastError();
}
