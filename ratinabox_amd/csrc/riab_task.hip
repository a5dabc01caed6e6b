// Batched TaskEnvironment bookkeeping for gfx950 (reference contribs/TaskEnvironment.py).
//
// One lane = one agent of the batch = one independent single-agent replica of the reference's
// TaskEnvironment.  After the motion kernel has moved the agents, task_step_kernel does for every
// lane what TaskEnvironment.step does between Agent.update and its return (:410-449): decay the
// lane's active rewards, check and consume its goals, append the termination-delay goal, total the
// reward.  The reference's list semantics are kept, including its two iteration quirks (the goal
// after a popped goal is skipped in that pass, :1130-1141; the reward after an expired reward is
// skipped in that update, :919-922), because they change which step a reward starts or ends on.
//
// All arithmetic is float64 like the reference (goal tests are discontinuous predicates; there are a
// handful of operations per lane and step).  Per-lane state lives in one [RIAB_TS_ROWS][B] tensor:
// every access below is a unit-stride row access across the 64 lanes of a wave.
#include "riab_device.h"

// The reward recursions are compared bit for bit with the reference's float64 python arithmetic:
// no fused multiply-adds in this file.
#pragma clang fp contract(off)

#define RIAB_TAG_TASK 0x5441534Bu  // "TASK"

namespace riab {

struct TaskArgs {
  const double* walls;  // [n_walls][4]
  int n_walls;
  const double* goals;  // [n_pool][8]
  int n_pool;
  int goalorder;
  double terminate_delay;
  double pad_reward[5];
  double default_level;
  double* ts;  // [RIAB_TS_ROWS][B]
  int64_t B;
};

__device__ __forceinline__ double& ts_at(const TaskArgs& a, int row, int64_t b) { return a.ts[(int64_t)row * a.B + b]; }

// Reward.get_delta with no external drive (:823-832): -(decay(state)), presets of :732-737
__device__ __forceinline__ double reward_delta(int preset, double knob, double state) {
  switch (preset) {
    case RIAB_DECAY_CONSTANT: return -knob;
    case RIAB_DECAY_LINEAR: return -(knob * state);
    case RIAB_DECAY_EXPONENTIAL: return -(knob * exp(state));
    default: return -0.0;
  }
}

// the Reward template (init_state, dt, expire_clock, preset, knob) a goal hands out
__device__ __forceinline__ const double* reward_of(const TaskArgs& a, int src) {
  return src == RIAB_GOAL_TIME_ELAPSED ? a.pad_reward : a.goals + (int64_t)src * RIAB_GOAL_COLS + 3;
}

// SpatialGoal._in_goal_radius (:1319-1332): line_of_sight distance < radius, i.e. the euclidean
// distance unless a wall of walls[4:] crosses the segment agent -> goal (Environment.py:715-722: the
// distance becomes 1000, which no radius reaches in practice but is compared all the same).  The
// reference asserts solid boundaries for this geometry (Environment.py:710-713); so does the host.
__device__ bool in_goal_radius(const TaskArgs& a, double px, double py, const double* g) {
  const double gx = g[0], gy = g[1], radius = g[2];
  const double vx = px - gx, vy = py - gy;
  double dist = sqrt(vx * vx + vy * vy);
  for (int w = 4; w < a.n_walls; ++w) {
    const double* ww = a.walls + 4 * w;
    if (seg_hit(px, py, gx, gy, ww[0], ww[1], ww[2], ww[3])) {
      dist = 1000.0;
      break;
    }
  }
  return dist < radius;
}

// RewardCache.append (:902-911): a copy of the goal's reward joins the end of the cache
__device__ void award(const TaskArgs& a, int64_t b, int src, int32_t* diag) {
  const int n = (int)ts_at(a, RIAB_TS_N_REWARDS, b);
  if (n >= RIAB_TASK_MAX_REWARDS) {
    atomicAdd(diag + RIAB_TD_REWARD_OVERFLOW, 1);
    return;
  }
  const double* r = reward_of(a, src);
  ts_at(a, RIAB_TS_RW_STATE + n, b) = r[0];
  ts_at(a, RIAB_TS_RW_EXPIRE + n, b) = r[2];
  ts_at(a, RIAB_TS_RW_SRC + n, b) = (double)src;
  ts_at(a, RIAB_TS_N_REWARDS, b) = (double)(n + 1);
}

// GoalCache.pop (:1154-1172) on the lane's list
__device__ void pop_goal(const TaskArgs& a, int64_t b, int g, int n) {
  for (int k = g; k + 1 < n; ++k) ts_at(a, RIAB_TS_GOAL_LIST + k, b) = ts_at(a, RIAB_TS_GOAL_LIST + k + 1, b);
  ts_at(a, RIAB_TS_N_GOALS, b) = (double)(n - 1);
}

__device__ bool goal_met(const TaskArgs& a, int64_t b, int src, double px, double py, double t_env) {
  if (src == RIAB_GOAL_TIME_ELAPSED)  // TimeElapsedGoal.check (:1271-1278)
    return t_env - ts_at(a, RIAB_TS_PAD_START, b) >= a.terminate_delay;
  return in_goal_radius(a, px, py, a.goals + (int64_t)src * RIAB_GOAL_COLS);
}

// One GoalCache.check(remove_finished=True) for the lane (:1076-1152); returns goals consumed.
__device__ int check_pass(const TaskArgs& a, int64_t b, double px, double py, double t_env, int32_t* diag) {
  int n = (int)ts_at(a, RIAB_TS_N_GOALS, b);
  int done = 0;
  if (n == 0) return 0;
  if (a.goalorder == RIAB_GOALORDER_SEQUENTIAL) {
    // `this` = last achieved + 1 is always the head of the list: pop() rewinds the marker (:1161-1163)
    const int src = (int)ts_at(a, RIAB_TS_GOAL_LIST + 0, b);
    if (goal_met(a, b, src, px, py, t_env)) {
      award(a, b, src, diag);
      pop_goal(a, b, 0, n);
      done = 1;
    }
    return done;
  }
  int g = 0;
  while (g < n) {  // :1130-1141: g advances after a pop too, so the goal that slid into slot g waits a pass
    const int src = (int)ts_at(a, RIAB_TS_GOAL_LIST + g, b);
    if (goal_met(a, b, src, px, py, t_env)) {
      award(a, b, src, diag);
      pop_goal(a, b, g, n);
      n -= 1;
      done += 1;
    }
    g += 1;
  }
  return done;
}

__global__ __launch_bounds__(64) void task_step_kernel(TaskArgs a, const double* pos_x, const double* pos_y,
                                                       double t_env, double* reward_out, uint8_t* terminal_out,
                                                       int32_t* diag) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  // ---- RewardCache.update (:913-927)
  int nr = (int)ts_at(a, RIAB_TS_N_REWARDS, b);
  if (nr > 0) {
    ts_at(a, RIAB_TS_STEPS_ACTIVE, b) += 1.0;
    int i = 0;
    while (i < nr) {
      const int src = (int)ts_at(a, RIAB_TS_RW_SRC + i, b);
      const double* r = reward_of(a, src);
      const double rdt = r[1];
      double state = ts_at(a, RIAB_TS_RW_STATE + i, b);
      double expire = ts_at(a, RIAB_TS_RW_EXPIRE + i, b);
      state = state + reward_delta((int)r[3], r[4], state) * rdt;  // Reward.update (:817-821)
      expire -= rdt;
      if (expire <= 0.0) {  // cache.remove while iterating: the next reward is skipped this step
        for (int k = i; k + 1 < nr; ++k) {
          ts_at(a, RIAB_TS_RW_STATE + k, b) = ts_at(a, RIAB_TS_RW_STATE + k + 1, b);
          ts_at(a, RIAB_TS_RW_EXPIRE + k, b) = ts_at(a, RIAB_TS_RW_EXPIRE + k + 1, b);
          ts_at(a, RIAB_TS_RW_SRC + k, b) = ts_at(a, RIAB_TS_RW_SRC + k + 1, b);
        }
        nr -= 1;
      } else {
        ts_at(a, RIAB_TS_RW_STATE + i, b) = state;
        ts_at(a, RIAB_TS_RW_EXPIRE + i, b) = expire;
      }
      i += 1;
    }
    ts_at(a, RIAB_TS_N_REWARDS, b) = (double)nr;
  } else {
    ts_at(a, RIAB_TS_STEPS_INACTIVE, b) += 1.0;
  }
  // ---- goals: _is_terminal_state (:278-290) as step() calls it (:418-440)
  const double px = pos_x[b], py = pos_y[b];
  check_pass(a, b, px, py, t_env, diag);
  bool terminal = (int)ts_at(a, RIAB_TS_N_GOALS, b) == 0;
  if (terminal && a.terminate_delay != 0.0 && ts_at(a, RIAB_TS_DELAYED, b) == 0.0) {
    // :421-434: one unrewarded TimeElapsedGoal pads the episode
    ts_at(a, RIAB_TS_DELAYED, b) = 1.0;
    ts_at(a, RIAB_TS_PAD_START, b) = t_env;
    ts_at(a, RIAB_TS_GOAL_LIST + 0, b) = (double)RIAB_GOAL_TIME_ELAPSED;
    ts_at(a, RIAB_TS_N_GOALS, b) = 1.0;
    check_pass(a, b, px, py, t_env, diag);
    terminal = (int)ts_at(a, RIAB_TS_N_GOALS, b) == 0;
  }
  const int late = check_pass(a, b, px, py, t_env, diag);  // the pass of the `for agent, term in ...` loop (:438)
  const bool terminal_last = (int)ts_at(a, RIAB_TS_N_GOALS, b) == 0;
  if (late > 0 && terminal_last && !terminal) atomicAdd(diag + RIAB_TD_LATE_COMPLETIONS, 1);
  // ---- RewardCache.get_total (:929-939): python sum() left to right from 0, then + default level
  nr = (int)ts_at(a, RIAB_TS_N_REWARDS, b);
  double total = 0.0;
  for (int i = 0; i < nr; ++i) total += ts_at(a, RIAB_TS_RW_STATE + i, b);
  total += a.default_level;
  if (total > ts_at(a, RIAB_TS_R_MAX, b)) ts_at(a, RIAB_TS_R_MAX, b) = total;
  if (total < ts_at(a, RIAB_TS_R_MIN, b)) ts_at(a, RIAB_TS_R_MIN, b) = total;
  reward_out[b] = total;
  terminal_out[b] = terminal_last ? 1 : 0;
}

__global__ __launch_bounds__(64) void task_reset_kernel(TaskArgs a, const uint8_t* mask, int64_t agent_id0, double t_env,
                                                        int n_select, int ordered, uint64_t seed, uint64_t counter,
                                                        int teleport, const double* new_x, const double* new_y,
                                                        double* pos_x, double* pos_y, float* hist_x, float* hist_y,
                                                        double cx, double cy, double half,
                                                        double* ep_log, int64_t ep_log_cap, int32_t* ep_count,
                                                        int32_t* diag) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  if (mask && !mask[b]) return;
  atomicAdd(diag + RIAB_TD_RESETS, 1);
  const uint64_t id = (uint64_t)(agent_id0 + b);
  // ---- write_end_episode (:536-539) + the episode counter (:333-338)
  bool zero_duration = false;
  if (ts_at(a, RIAB_TS_STARTED, b) != 0.0) {
    const double start = ts_at(a, RIAB_TS_EP_START, b);
    const double duration = t_env - start;
    zero_duration = duration == 0.0;
    if (!zero_duration) {  // a zero-duration episode is popped again right away (:333-335)
      ts_at(a, RIAB_TS_EP_ANY_ENDED, b) = 1.0;
      if (ep_log) {
        const int slot = atomicAdd(ep_count, 1);
        if (slot < ep_log_cap) {
          double* e = ep_log + (int64_t)slot * 5;
          e[0] = (double)id;
          e[1] = ts_at(a, RIAB_TS_EPISODE, b);
          e[2] = start;
          e[3] = t_env;
          e[4] = duration;
        } else {
          atomicAdd(diag + RIAB_TD_EPLOG_OVERFLOW, 1);
        }
      }
    }
  }
  if (!zero_duration) ts_at(a, RIAB_TS_EPISODE, b) += 1.0;
  ts_at(a, RIAB_TS_STARTED, b) = 1.0;
  // _current_episode_start (:526-527): the end of the last kept episode, 0 before any
  ts_at(a, RIAB_TS_EP_START, b) = ts_at(a, RIAB_TS_EP_ANY_ENDED, b) != 0.0 ? t_env : 0.0;
  // ---- teleport_on_reset (:323-330)
  u32x4 rnd = philox4x32_10((uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)id, RIAB_TAG_TASK, (uint32_t)seed,
                            (uint32_t)(seed >> 32));
  if (teleport) {
    double x, y;
    if (new_x) {
      x = new_x[b];
      y = new_y[b];
    } else {  // sample_positions(1), "uniform_jitter": the centre of the box +- 0.45 * scale
      const double ux = ((double)rnd.x + 0.5) * 0x1.0p-32, uy = ((double)rnd.y + 0.5) * 0x1.0p-32;
      x = cx + (2.0 * ux - 1.0) * half;
      y = cy + (2.0 * uy - 1.0) * half;
    }
    pos_x[b] = x;
    pos_y[b] = y;
    if (hist_x) {  // agent.history["pos"][-1] = agent.pos
      hist_x[b] = (float)x;
      hist_y[b] = (float)y;
    }
  }
  // ---- GoalCache.reset (:1218-1252)
  const int n = n_select < a.n_pool ? n_select : a.n_pool;
  if (ordered) {
    for (int i = 0; i < n; ++i) ts_at(a, RIAB_TS_GOAL_LIST + i, b) = (double)i;
  } else {  // uniform sample without replacement: partial Fisher-Yates over the pool
    uint8_t perm[RIAB_TASK_MAX_POOL];
    for (int i = 0; i < a.n_pool; ++i) perm[i] = (uint8_t)i;
    u32x4 blk_words = rnd;
    int cur = 0;
    for (int i = 0; i < n; ++i) {
      const int blk = 1 + (i >> 2);  // draw i is word (i & 3) of Philox block 1 + i / 4 (block 0 = teleport)
      if (blk != cur) {
        blk_words = philox4x32_10((uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)id, RIAB_TAG_TASK + (uint32_t)blk,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
        cur = blk;
      }
      const int q = i & 3;
      const uint32_t w = q == 0 ? blk_words.x : (q == 1 ? blk_words.y : (q == 2 ? blk_words.z : blk_words.w));
      const int j = i + (int)(((uint64_t)w * (uint64_t)(a.n_pool - i)) >> 32);
      const uint8_t tmp = perm[i];
      perm[i] = perm[j];
      perm[j] = tmp;
      ts_at(a, RIAB_TS_GOAL_LIST + i, b) = (double)perm[i];
    }
  }
  ts_at(a, RIAB_TS_N_GOALS, b) = (double)n;
  ts_at(a, RIAB_TS_DELAYED, b) = 0.0;
}

static int fill_args(TaskArgs& a, const RiabEnv* env, const RiabTask* task, double* task_state, int64_t B) {
  if (!env || !task || !task_state || B <= 0) return RIAB_EINVAL;
  if (task->n_pool < 0 || task->n_pool > RIAB_TASK_MAX_POOL) return RIAB_ETOOBIG;
  if (task->n_pool > 0 && !task->goals) return RIAB_EINVAL;
  if (env->n_walls > 0 && !env->walls) return RIAB_EINVAL;
  if (env->periodic && task->n_pool > 0) return RIAB_EUNSUPPORTED;  // line_of_sight needs solid boundaries
  if (task->goalorder != RIAB_GOALORDER_NONSEQUENTIAL && task->goalorder != RIAB_GOALORDER_SEQUENTIAL)
    return RIAB_EUNSUPPORTED;
  a.walls = env->walls;
  a.n_walls = env->n_walls;
  a.goals = task->goals;
  a.n_pool = task->n_pool;
  a.goalorder = task->goalorder;
  a.terminate_delay = task->terminate_delay;
  for (int i = 0; i < 5; ++i) a.pad_reward[i] = task->pad_reward[i];
  a.default_level = task->default_reward_level;
  a.ts = task_state;
  a.B = B;
  return RIAB_OK;
}

}  // namespace riab

using namespace riab;

extern "C" int riab_task_step(const RiabEnv* env, const RiabTask* task, double* task_state, const double* pos_x,
                              const double* pos_y, int64_t B, double t_env, double* reward_out, uint8_t* terminal_out,
                              int32_t* diag, riab_stream_t stream) {
  TaskArgs a;
  const int rc = fill_args(a, env, task, task_state, B);
  if (rc) return rc;
  if (!pos_x || !pos_y || !reward_out || !terminal_out || !diag) return RIAB_EINVAL;
  hipLaunchKernelGGL(task_step_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a, pos_x, pos_y,
                     t_env, reward_out, terminal_out, diag);
  return (int)hipGetLastError();
}

extern "C" int riab_task_reset(const RiabEnv* env, const RiabTask* task, double* task_state, const uint8_t* mask, int64_t B,
                               int64_t agent_id0, double t_env, int32_t n_select, int32_t ordered, uint64_t seed,
                               uint64_t counter, int32_t teleport, const double* new_x, const double* new_y,
                               double* pos_x, double* pos_y, float* hist_x, float* hist_y, double* ep_log,
                               int64_t ep_log_cap, int32_t* ep_count, int32_t* diag, riab_stream_t stream) {
  TaskArgs a;
  const int rc = fill_args(a, env, task, task_state, B);
  if (rc) return rc;
  if (!diag || n_select < 0) return RIAB_EINVAL;
  if (n_select > RIAB_TASK_MAX_GOALS - 1) return RIAB_ETOOBIG;  // one slot stays free for the termination-delay goal
  if (teleport && (!pos_x || !pos_y)) return RIAB_EINVAL;
  if ((new_x == nullptr) != (new_y == nullptr) || (hist_x == nullptr) != (hist_y == nullptr)) return RIAB_EINVAL;
  if (ep_log && (!ep_count || ep_log_cap <= 0)) return RIAB_EINVAL;
  const double cx = 0.5 * (env->extent[0] + env->extent[1]), cy = 0.5 * (env->extent[2] + env->extent[3]);
  const double w = env->extent[1] - env->extent[0], h = env->extent[3] - env->extent[2];
  if (teleport && !new_x && w != h) return RIAB_EUNSUPPORTED;  // sample_positions(1) only works for a square box
  hipLaunchKernelGGL(task_reset_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a, mask,
                     agent_id0, t_env, n_select, ordered, seed, counter, teleport, new_x, new_y, pos_x, pos_y, hist_x, hist_y, cx,
                     cy, 0.45 * sqrt(w * h), ep_log, ep_log_cap, ep_count, diag);
  return (int)hipGetLastError();
}
