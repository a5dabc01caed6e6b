#pragma once
// Batched TaskEnvironment bookkeeping for gfx950 (reference contribs/TaskEnvironment.py).
//
// One lane = one agent of the batch = one independent single-agent replica of the reference's
// TaskEnvironment.  After the motion kernel has moved the agents, task_kernel does for every lane
// what TaskEnvironment.step does between Agent.update and its return (:410-449): decay the lane's
// active rewards, check and consume its goals, append the termination-delay goal, total the reward.
// The reference's list semantics are kept, including its two iteration quirks (the goal after a
// popped goal is skipped in that pass, :1130-1141; the reward after an expired reward is skipped in
// that update, :919-922), because they change which step a reward starts or ends on.
//
// All arithmetic is float64 like the reference (goal tests are discontinuous predicates; there are a
// handful of operations per lane and step).  The kernel is latency-bound, not bandwidth-bound: at
// 4096 lanes it is 64 wavefronts.  So it is organised around round trips, not bytes:
//   * everything a lane needs is fetched by ONE batch of independent loads at the top (counts, the 16
//     list rows, position, stats), the goal list then lives in registers as 16 packed bytes
//     (pops are 128-bit shifts — no dynamically indexed arrays, hence no scratch memory, whose
//     per-dispatch set-up alone costs ~10 us), and every row is written back once at the end;
//   * the reward total is accumulated while the cache is walked (no read-back of what was just stored);
//   * the caller's `if terminal: reset()` and the scripted goal-seeking action of the NEXT step are
//     fused into the same launch for the step plan, which also puts the one-step motion launch in front of it
//     (motion_task_kernel, riab_agent.hip: 2 kernels per closed-loop step) or, where the plan's lead population
//     can be fused as well, runs the whole step as ONE kernel that deals the pieces below to different waves
//     (step1_task_kernel, riab_step1.hip).
#include "riab_device.h"

// The reward recursions are compared bit for bit with the reference's float64 python arithmetic:
// no fused multiply-adds in this file.
#pragma clang fp contract(off)

#define RIAB_TAG_TASK 0x5441534Bu  // "TASK"

namespace riab {

typedef unsigned __int128 u128;
// the staged goal pool is read with LDS instructions proper (ds_read): a generic pointer would compile
// to flat loads, whose completion is tracked by vmcnt as well — every goal-row read would then also
// wait for the reward rows stored just before it
typedef const __attribute__((address_space(3))) double* lds_f64_ptr;

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

struct ResetArgs {
  int64_t agent_id0;
  int n_select, ordered, teleport;
  uint64_t seed, counter;
  const double* new_x;
  const double* new_y;
  double* pos_x;
  double* pos_y;
  float* hist_x;
  float* hist_y;
  double cx, cy, half;
  double* ep_log;
  int64_t ep_log_cap;
  int32_t* ep_count;
};

__device__ __forceinline__ double& ts_at(const TaskArgs& a, int row, int64_t b) { return a.ts[(int64_t)row * a.B + b]; }

// ---- the lane's goal list: 16 bytes in registers (pool index, 0xFE = termination-delay goal) ----
__device__ __forceinline__ int list_get(u128 l, int i) {
  const int v = (int)((uint32_t)(l >> (8 * i)) & 0xFFu);
  return v == 0xFE ? RIAB_GOAL_TIME_ELAPSED : v;
}
__device__ __forceinline__ u128 list_set(u128 l, int i, int src) {
  const u128 m = (u128)0xFF << (8 * i);
  return (l & ~m) | ((u128)(uint32_t)(src & 0xFF) << (8 * i));
}
__device__ __forceinline__ u128 list_pop(u128 l, int g) {  // GoalCache.pop (:1154-1172)
  const u128 low = g ? (l & ((((u128)1) << (8 * g)) - 1)) : (u128)0;
  const u128 high = g < 15 ? ((l >> (8 * (g + 1))) << (8 * g)) : (u128)0;
  return low | high;
}

struct Lane {
  u128 list;
  int n_goals, n_rw;
  bool delayed, list_dirty;
  double pad_start;
  double px, py;
  double new_total;  // states of the rewards awarded this step, in award order
  uint64_t met;      // bit `pool index`: the lane stands inside that goal's radius (goals_met, once per step)
};

// (only as many rows as the longest list of the wave are packed / written back: a step is one dependent instruction
// stream, every instruction it does not issue is time.  The compiler sinks each row's load behind the test that guards
// its use — a chain of round trips as long as that list, two entries in the benchmark's task; forcing all 16 loads into
// ONE batch in front of the loop measured SLOWER, 12.9 -> 13.25 us per one-launch step: docs/EXPERIMENTS.md r06-9)
__device__ __forceinline__ void load_list(const TaskArgs& a, int64_t b, Lane& L) {
  double rows[RIAB_TASK_MAX_GOALS];
#pragma unroll
  for (int i = 0; i < RIAB_TASK_MAX_GOALS; ++i) rows[i] = ts_at(a, RIAB_TS_GOAL_LIST + i, b);
  u128 l = 0;
#pragma unroll
  for (int i = 0; i < RIAB_TASK_MAX_GOALS; ++i) {
    if (__builtin_amdgcn_ballot_w64(i < L.n_goals) == 0) break;  // (wave-uniform; entries past a lane's count are never read)
    l |= (u128)(uint32_t)((int)rows[i] & 0xFF) << (8 * i);
  }
  L.list = l;
  L.list_dirty = false;
}

__device__ __forceinline__ void store_list(const TaskArgs& a, int64_t b, const Lane& L) {
#pragma unroll
  for (int i = 0; i < RIAB_TASK_MAX_GOALS; ++i) {
    if (__builtin_amdgcn_ballot_w64(i < L.n_goals) == 0) break;
    if (i < L.n_goals) ts_at(a, RIAB_TS_GOAL_LIST + i, b) = (double)list_get(L.list, i);
  }
}

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
struct RewardTpl {
  double init, dt, expire, knob;
  int preset;
};
__device__ __forceinline__ RewardTpl reward_of(const TaskArgs& a, lds_f64_ptr goals, int src) {
  if (src == RIAB_GOAL_TIME_ELAPSED) return {a.pad_reward[0], a.pad_reward[1], a.pad_reward[2], a.pad_reward[4], (int)a.pad_reward[3]};
  const lds_f64_ptr r = goals + src * RIAB_GOAL_COLS + 3;
  return {r[0], r[1], r[2], r[4], (int)r[3]};
}

// SpatialGoal._in_goal_radius (:1319-1332): line_of_sight distance < radius, i.e. the euclidean
// distance unless a wall of walls[4:] crosses the segment agent -> goal (Environment.py:715-722: the
// distance becomes 1000, which no radius reaches in practice but is compared all the same).  The
// reference asserts solid boundaries for this geometry (Environment.py:710-713); so does the host.
__device__ bool in_goal_radius(const TaskArgs& a, double px, double py, lds_f64_ptr g) {
  const double gx = g[0], gy = g[1], radius = g[2];
  const double vx = px - gx, vy = py - gy;
  double dist = sqrt(vx * vx + vy * vy);
  if (!(dist < radius)) return false;  // (a blocked line of sight can only turn a hit into a miss)
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
__device__ void award(const TaskArgs& a, lds_f64_ptr goals, int64_t b, Lane& L, int src, int32_t* diag) {
  if (L.n_rw >= RIAB_TASK_MAX_REWARDS) {
    atomicAdd(diag + RIAB_TD_REWARD_OVERFLOW, 1);
    return;
  }
  const RewardTpl r = reward_of(a, goals, src);
  ts_at(a, RIAB_TS_RW_STATE + L.n_rw, b) = r.init;
  ts_at(a, RIAB_TS_RW_EXPIRE + L.n_rw, b) = r.expire;
  ts_at(a, RIAB_TS_RW_SRC + L.n_rw, b) = (double)src;
  L.n_rw += 1;
  L.new_total = L.new_total + r.init;
}

// The spatial tests of a step, once: the passes below ask about the same goals at the same position up to three times
// (each answer a square root and a division-free comparison on float64, a line-of-sight loop behind it), and which of
// them a pass asks about is list logic that does not change the answers.  Sequential order only ever looks at the head,
// and at most two heads per step (the step's pass and the late pass).
__device__ __forceinline__ uint64_t goals_met(const TaskArgs& a, lds_f64_ptr goals, const Lane& L) {
  uint64_t met = 0;
  const int n = (a.goalorder == RIAB_GOALORDER_SEQUENTIAL && L.n_goals > 2) ? 2 : L.n_goals;
  u128 l = L.list;
  for (int g = 0; g < n; ++g) {
    const int v = (int)((uint32_t)l & 0xFFu);
    l >>= 8;
    if (v != 0xFE && in_goal_radius(a, L.px, L.py, goals + v * RIAB_GOAL_COLS)) met |= 1ull << v;
  }
  return met;
}

__device__ __forceinline__ bool goal_met(const TaskArgs& a, lds_f64_ptr goals, const Lane& L, int src, double t_env) {
  if (src == RIAB_GOAL_TIME_ELAPSED)  // TimeElapsedGoal.check (:1271-1278)
    return t_env - L.pad_start >= a.terminate_delay;
  return (L.met >> src) & 1ull;  // SpatialGoal.check: goals_met's answer
}

// One GoalCache.check(remove_finished=True) for the lane (:1076-1152); returns goals consumed.
__device__ int check_pass(const TaskArgs& a, lds_f64_ptr goals, int64_t b, Lane& L, double t_env, int32_t* diag) {
  int done = 0;
  if (L.n_goals == 0) return 0;
  if (a.goalorder == RIAB_GOALORDER_SEQUENTIAL) {
    // `this` = last achieved + 1 is always the head of the list: pop() rewinds the marker (:1161-1163)
    const int src = list_get(L.list, 0);
    if (goal_met(a, goals, L, src, t_env)) {
      award(a, goals, b, L, src, diag);
      L.list = list_pop(L.list, 0);
      L.n_goals -= 1;
      L.list_dirty = true;
      done = 1;
    }
    return done;
  }
  // (shifts of a 128-bit value by a variable count are a dozen instructions each; this walk only shifts by constants:
  // `rest` = the list from slot g on, `below` = one bit at slot g's byte)
  int g = 0;
  u128 rest = L.list, below = 1;
  while (g < L.n_goals) {  // :1130-1141: g advances after a pop too, so the goal that slid into slot g waits a pass
    const int v = (int)((uint32_t)rest & 0xFFu);
    const int src = v == 0xFE ? RIAB_GOAL_TIME_ELAPSED : v;
    if (goal_met(a, goals, L, src, t_env)) {
      award(a, goals, b, L, src, diag);
      L.list = (L.list & (below - 1)) | ((L.list >> 8) & ~(below - 1));  // GoalCache.pop (:1154-1172) of slot g
      L.n_goals -= 1;
      L.list_dirty = true;
      done += 1;
      rest >>= 8;  // (the goal that slid into slot g)
    }
    rest >>= 8;
    below <<= 8;
    g += 1;
  }
  return done;
}

// RewardCache.update (:913-927) as an out-of-place compaction; returns the python sum() of the
// surviving states (left to right from 0).  `cache.remove` while iterating makes the iterator skip
// the reward after an expired one: it is carried over untouched.  The first RW_PRE entries were
// fetched with the lane's first batch of loads (`pre`): walking the cache costs no further global
// round trips unless more than RW_PRE rewards are active.
constexpr int RW_PRE = 4;
struct RewardRows {
  double state[RW_PRE], expire[RW_PRE], src[RW_PRE];
};

__device__ __forceinline__ void load_rewards(const TaskArgs& a, int64_t b, RewardRows& pre) {
#pragma unroll
  for (int i = 0; i < RW_PRE; ++i) {
    pre.state[i] = ts_at(a, RIAB_TS_RW_STATE + i, b);
    pre.expire[i] = ts_at(a, RIAB_TS_RW_EXPIRE + i, b);
    pre.src[i] = ts_at(a, RIAB_TS_RW_SRC + i, b);
  }
}

__device__ double rewards_update(const TaskArgs& a, lds_f64_ptr goals, int64_t b, int& n_rw, const RewardRows& pre) {
  double total = 0.0;
  int w = 0;
  bool skip = false;  // the previous reward expired: this one is carried over untouched
  const int nr = n_rw;
  auto visit = [&](int i, double state, double expire, double srcd) {
    if (skip) {
      ts_at(a, RIAB_TS_RW_STATE + w, b) = state;
      ts_at(a, RIAB_TS_RW_EXPIRE + w, b) = expire;
      ts_at(a, RIAB_TS_RW_SRC + w, b) = srcd;
      total = total + state;
      w += 1;
      skip = false;
      return;
    }
    const RewardTpl r = reward_of(a, goals, (int)srcd);
    const double rdt = r.dt;
    state = state + reward_delta(r.preset, r.knob, state) * rdt;  // Reward.update (:817-821)
    expire -= rdt;
    if (expire <= 0.0) {
      skip = true;
    } else {
      ts_at(a, RIAB_TS_RW_STATE + w, b) = state;
      ts_at(a, RIAB_TS_RW_EXPIRE + w, b) = expire;
      if (w != i) ts_at(a, RIAB_TS_RW_SRC + w, b) = srcd;
      total = total + state;
      w += 1;
    }
  };
#pragma unroll
  for (int i = 0; i < RW_PRE; ++i)
    if (i < nr) visit(i, pre.state[i], pre.expire[i], pre.src[i]);
  for (int i = RW_PRE; i < nr; ++i)
    visit(i, ts_at(a, RIAB_TS_RW_STATE + i, b), ts_at(a, RIAB_TS_RW_EXPIRE + i, b), ts_at(a, RIAB_TS_RW_SRC + i, b));
  n_rw = w;
  return total;
}

// The reward cache's share of TaskEnvironment.step — RewardCache.update and the step counters — as a unit of its own:
// it needs nothing of the step's motion (not the position, not the goals' state), so a caller with idle lanes runs it
// BESIDE the motion step (the one-launch step, riab_step1.hip: the workgroup's noise-drawing waves) and hands the lane
// that keeps the goals two numbers: how many rewards are still alive, and their total.
struct RewardsIn {
  int n_rw;
  RewardRows pre;
  double steps_active, steps_inactive;
};
__device__ __forceinline__ RewardsIn load_rewards_in(const TaskArgs& a, int64_t b) {
  RewardsIn in;
  in.n_rw = (int)ts_at(a, RIAB_TS_N_REWARDS, b);
  load_rewards(a, b, in.pre);
  in.steps_active = ts_at(a, RIAB_TS_STEPS_ACTIVE, b);
  in.steps_inactive = ts_at(a, RIAB_TS_STEPS_INACTIVE, b);
  return in;
}
struct RewardsOut {
  int n_rw;
  double total;
};
__device__ __forceinline__ RewardsOut rewards_step(const TaskArgs& a, lds_f64_ptr goals, int64_t b, const RewardsIn& in) {
  RewardsOut o = {in.n_rw, 0.0};
  if (in.n_rw > 0) {
    o.total = rewards_update(a, goals, b, o.n_rw, in.pre);
    ts_at(a, RIAB_TS_STEPS_ACTIVE, b) = in.steps_active + 1.0;
  } else {
    ts_at(a, RIAB_TS_STEPS_INACTIVE, b) = in.steps_inactive + 1.0;
  }
  return o;
}

// TaskEnvironment.reset for one lane (:307-351, GoalCache.reset :1218-1252): episode bookkeeping in
// memory, goal list / position in the lane's registers.  In three parts, so that nothing a reset needs from memory sits
// on a step's critical path (a closed-loop step is one dependent chain, and at 4096 lanes some lane ends an episode in
// nearly every step): what it reads was fetched with the lane's first batch of loads (ResetIn), the slot of the episode
// table — an atomic with a return value, a round trip of its own — is asked for in part 1 and used in part 3, which the
// caller runs last.  The parts' stores do not overlap: any order gives the same memory.
struct ResetIn {  // the lane's episode bookkeeping rows
  double any_ended, started, ep_start, episode;
};
__device__ __forceinline__ ResetIn load_reset(const TaskArgs& a, int64_t b) {
  return {ts_at(a, RIAB_TS_EP_ANY_ENDED, b), ts_at(a, RIAB_TS_STARTED, b), ts_at(a, RIAB_TS_EP_START, b), ts_at(a, RIAB_TS_EPISODE, b)};
}
// What a reset draws — where the lane is teleported to, which goals its next episode has — is a function of (seed, reset
// counter, agent id) alone: a caller with idle lanes draws it for every lane ahead of time (two Philox blocks and a
// sampling loop off the critical path of the lanes that do end an episode).
struct ResetDraw {
  double x, y;
  u128 list;
};
// (`id`: the global agent id — or RIAB_WORLD_STREAM_ID for what a whole world draws once, riab_task_world.hip)
__device__ __forceinline__ ResetDraw reset_draw_id(const TaskArgs& a, const ResetArgs& r, uint64_t id) {
  ResetDraw d = {0.0, 0.0, 0};
  if (r.teleport && !r.new_x) {  // sample_positions(1), "uniform_jitter": the centre of the box +- 0.45 * scale (Philox block 0)
    const u32x4 rnd = philox4x32_10((uint32_t)r.counter, (uint32_t)(r.counter >> 32), (uint32_t)id, RIAB_TAG_TASK,
                                    (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
    const double ux = ((double)rnd.x + 0.5) * 0x1.0p-32, uy = ((double)rnd.y + 0.5) * 0x1.0p-32;
    d.x = r.cx + (2.0 * ux - 1.0) * r.half;
    d.y = r.cy + (2.0 * uy - 1.0) * r.half;
  }
  // GoalCache.reset (:1218-1252)
  const int n = r.n_select < a.n_pool ? r.n_select : a.n_pool;
  u128 list = 0;
  if (r.ordered) {  // slots 0 .. n-1 hold goals 0 .. n-1
    const u128 iota = ((u128)0x0F0E0D0C0B0A0908ull << 64) | (u128)0x0706050403020100ull;
    list = n >= 16 ? iota : (iota & ((((u128)1) << (8 * n)) - 1));
  } else {
    // uniform sample without replacement: draw i picks the j-th goal still in the pool (ascending
    // order), j = floor(w_i * (n_pool - i) / 2^32), w_i = word i%4 of Philox block 1 + i/4
    uint64_t remaining = a.n_pool >= 64 ? ~0ull : ((1ull << a.n_pool) - 1ull);
    u32x4 words = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
      if ((i & 3) == 0)
        words = philox4x32_10((uint32_t)r.counter, (uint32_t)(r.counter >> 32), (uint32_t)id,
                              RIAB_TAG_TASK + 1u + (uint32_t)(i >> 2), (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
      const int q = i & 3;
      const uint32_t w = q == 0 ? words.x : (q == 1 ? words.y : (q == 2 ? words.z : words.w));
      int j = (int)(((uint64_t)w * (uint64_t)(a.n_pool - i)) >> 32);
      uint64_t m = remaining;
      while (j > 0) {  // drop the j lowest remaining goals
        m &= m - 1;
        j -= 1;
      }
      const int pick = __ffsll((long long)m) - 1;
      remaining &= ~(1ull << pick);
      list = (list >> 8) | ((u128)(uint32_t)pick << 120);  // (enters at the top: constant shifts; brought down once, below)
    }
    if (n > 0) list >>= 8 * (16 - n);
  }
  d.list = list;
  return d;
}
__device__ __forceinline__ ResetDraw reset_draw(const TaskArgs& a, const ResetArgs& r, int64_t b) {
  return reset_draw_id(a, r, (uint64_t)(r.agent_id0 + b));
}
struct EpisodeRecord {  // write_end_episode's row, on its way into the table
  bool pending;
  int slot;
  double episode, start, duration;
};

// part 1a: teleport_on_reset (:323-330) — what others may be waiting to hear
__device__ __forceinline__ void reset_lane_teleport(const ResetArgs& r, int64_t b, Lane& L, const ResetDraw& d) {
  if (r.teleport) {
    double x, y;
    if (r.new_x) {
      x = r.new_x[b];
      y = r.new_y[b];
    } else {
      x = d.x;
      y = d.y;
    }
    L.px = x;
    L.py = y;
    if (r.pos_x) {  // (null: the caller stores the position it is handed back — the one-launch step, riab_step1.hip)
      r.pos_x[b] = x;
      r.pos_y[b] = y;
    }
    if (r.hist_x) {  // agent.history["pos"][-1] = agent.pos
      r.hist_x[b] = (float)x;
      r.hist_y[b] = (float)y;
    }
  }
}

// part 1b: write_end_episode (:536-539), the episode counter (:333-338)
__device__ __forceinline__ EpisodeRecord reset_lane_episode(const TaskArgs& a, const ResetArgs& r, int64_t b, const ResetIn& in,
                                                            double t_env, int32_t* diag) {
  atomicAdd(diag + RIAB_TD_RESETS, 1);
  EpisodeRecord rec = {false, 0, in.episode, in.ep_start, 0.0};
  bool zero_duration = false;
  bool any_ended = in.any_ended != 0.0;
  if (in.started != 0.0) {
    const double duration = t_env - in.ep_start;
    zero_duration = duration == 0.0;
    if (!zero_duration) {  // a zero-duration episode is popped again right away (:333-335)
      any_ended = true;
      ts_at(a, RIAB_TS_EP_ANY_ENDED, b) = 1.0;
      if (r.ep_log) {
        rec.pending = true;
        rec.duration = duration;
        // (the counter's address goes through an opaque register: for an address it can prove uniform the compiler
        // folds the lanes' requests into one and waits for the answer on the spot — the round trip this split is for)
        uintptr_t counter = (uintptr_t)r.ep_count;
        asm volatile("" : "+v"(counter));
        rec.slot = __hip_atomic_fetch_add((__attribute__((address_space(1))) int*)counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (!zero_duration) ts_at(a, RIAB_TS_EPISODE, b) = in.episode + 1.0;
  ts_at(a, RIAB_TS_STARTED, b) = 1.0;
  // _current_episode_start (:526-527): the end of the last kept episode, 0 before any
  ts_at(a, RIAB_TS_EP_START, b) = any_ended ? t_env : 0.0;
  return rec;
}

// part 2: GoalCache.reset (:1218-1252)
__device__ __forceinline__ void reset_lane_goals(const TaskArgs& a, const ResetArgs& r, Lane& L, const ResetDraw& d) {
  L.list = d.list;
  L.n_goals = r.n_select < a.n_pool ? r.n_select : a.n_pool;
  L.list_dirty = true;
  L.delayed = false;
}

// part 3: the episode's row into the table (the slot has had the rest of the lane's work to arrive)
__device__ __forceinline__ void episode_log_store(const ResetArgs& r, int64_t b, const EpisodeRecord& rec, double t_env, int32_t* diag) {
  if (!rec.pending) return;
  if (rec.slot < r.ep_log_cap) {
    double* e = r.ep_log + (int64_t)rec.slot * 5;
    e[0] = (double)(uint64_t)(r.agent_id0 + b);
    e[1] = rec.episode;
    e[2] = rec.start;
    e[3] = t_env;
    e[4] = rec.duration;
  } else {
    atomicAdd(diag + RIAB_TD_EPLOG_OVERFLOW, 1);
  }
}

// get_goal_vector (:1555-1584): goal - position for the head of the list (sequential) or the nearest
// pending spatial goal; (0,0) when none is pending.  scale > 0: scale * unit vector instead (the
// scripted policy of the reference's test loop, :1599-1605, with its NaN -> 0 of :403-404).
__device__ __forceinline__ void goal_vector(const TaskArgs& a, lds_f64_ptr goals, const Lane& L, double scale, double& vx_out,
                                            double& vy_out) {
  double vx = 0.0, vy = 0.0, best = INFINITY;
  u128 rest = L.list;
  for (int g = 0; g < L.n_goals; ++g) {
    const int v = (int)((uint32_t)rest & 0xFFu);
    const int src = v == 0xFE ? RIAB_GOAL_TIME_ELAPSED : v;
    rest >>= 8;
    if (src < 0) {
      if (a.goalorder == RIAB_GOALORDER_SEQUENTIAL) break;
      continue;
    }
    const lds_f64_ptr gl = goals + src * RIAB_GOAL_COLS;
    const double dx = gl[0] - L.px, dy = gl[1] - L.py;
    const double d = sqrt(dx * dx + dy * dy);
    if (d < best) {  // strict: the first of equidistant goals, like argmin
      best = d;
      vx = dx;
      vy = dy;
    }
    if (a.goalorder == RIAB_GOALORDER_SEQUENTIAL) break;
  }
  if (scale > 0.0) {
    const double nrm = sqrt(vx * vx + vy * vy);
    vx = nrm > 0.0 ? scale * (vx / nrm) : 0.0;
    vy = nrm > 0.0 ? scale * (vy / nrm) : 0.0;
  }
  vx_out = vx;
  vy_out = vy;
}

// MODE bit 0: TaskEnvironment.step; bit 1: reset the lanes selected by `mask` (terminal lanes when
// fused with the step); bit 2: write the goal vector of the (possibly reset) lane.
//
// task_lane_load / load_rewards_in, rewards_step, task_lane_goals, task_lane_reset, task_lane_finish, episode_log_store:
// the bookkeeping of ONE lane `b` whose
// position is handed in (and handed back: a reset with teleport_on_reset moves it), with the goal pool already staged in
// LDS — no barrier, no dependence on how the caller's workgroup is shaped.  task_body (the stand-alone kernel and the
// motion + task launch) stages the pool and calls them back to back with the position the state holds; the one-launch
// closed-loop step (riab_step1.hip) calls them from the workgroup that writes a segment's state, with the position its
// motion step has just computed, and does its own work in between.  Same code, same operands: same bits.
// What a lane's bookkeeping reads before it can start is ONE batch of independent loads (counts, the 16 list rows, the
// first reward rows, the reward statistics, the episode rows a reset needs) — issued by the caller as early as it likes
// (the one-launch step: with its state loads, a whole motion step ahead of the use).
struct LaneIn {
  Lane L;
  double rmax, rmin;
  ResetIn ep;
};
template <int MODE>
__device__ __forceinline__ LaneIn task_lane_load(const TaskArgs& a, int64_t b) {
  constexpr bool STEP = MODE & 1, RESET = MODE & 2;
  LaneIn in;
  in.L.n_goals = (int)ts_at(a, RIAB_TS_N_GOALS, b);
  in.L.n_rw = (int)ts_at(a, RIAB_TS_N_REWARDS, b);
  in.L.delayed = ts_at(a, RIAB_TS_DELAYED, b) != 0.0;
  in.L.pad_start = ts_at(a, RIAB_TS_PAD_START, b);
  in.L.px = in.L.py = 0.0;
  in.L.new_total = 0.0;
  in.L.met = 0;
  load_list(a, b, in.L);
  in.rmax = in.rmin = 0.0;
  if (STEP) {
    in.rmax = ts_at(a, RIAB_TS_R_MAX, b);
    in.rmin = ts_at(a, RIAB_TS_R_MIN, b);
  }
  in.ep = ResetIn{0.0, 0.0, 0.0, 0.0};
  if (RESET) in.ep = load_reset(a, b);
  return in;
}
// ... in three calls: task_lane_goals (the step's goal checks and reward total; `ro`: rewards_step's result for the lane),
// task_lane_reset (leaves the lane's position final — what a caller that shares it with others wants to know first;
// `draw()`: the lane's ResetDraw, asked for only by a lane that resets), task_lane_finish (what is left); LaneMid
// carries the state between them.
struct NoProbe {  // (tools/step1_profile.py hands in one that reads the clock)
  __device__ __forceinline__ void operator()(int) const {}
};
struct LaneMid {
  int n_goals0, n_rw0;
  bool delayed0, reset;
  EpisodeRecord rec;
  u128 new_list;
};
template <int MODE, class Probe = NoProbe>
__device__ __forceinline__ LaneMid task_lane_goals(const TaskArgs& a, int64_t b, lds_f64_ptr goals, LaneIn& in, const RewardsOut& ro,
                                                   double px, double py, double t_env, double* reward_out, uint8_t* terminal_out,
                                                   int32_t* diag, const Probe& probe = Probe()) {
  constexpr bool STEP = MODE & 1, RESET = MODE & 2;
  Lane& L = in.L;
  L.px = px;
  L.py = py;
  LaneMid mid = {L.n_goals, L.n_rw, L.delayed, false, {false, 0, 0.0, 0.0, 0.0}, 0};
  bool terminal_last = false;
  if (STEP) {
    const double rmax = in.rmax, rmin = in.rmin;
    // ---- RewardCache.update (:913-927): done (rewards_step)
    double total = ro.total;
    L.n_rw = ro.n_rw;
    probe(8);
    // ---- goals: _is_terminal_state (:278-290) as step() calls it (:418-440)
    L.met = goals_met(a, goals, L);
    check_pass(a, goals, b, L, t_env, diag);
    probe(9);
    bool terminal = L.n_goals == 0;
    if (terminal && a.terminate_delay != 0.0 && !L.delayed) {
      // :421-434: one unrewarded TimeElapsedGoal pads the episode
      L.delayed = true;
      L.pad_start = t_env;
      L.list = list_set(L.list, 0, RIAB_GOAL_TIME_ELAPSED);
      L.n_goals = 1;
      L.list_dirty = true;
      check_pass(a, goals, b, L, t_env, diag);
      terminal = L.n_goals == 0;
    }
    probe(10);
    const int late = check_pass(a, goals, b, L, t_env, diag);  // the pass of the `for agent, term in ...` loop (:438)
    terminal_last = L.n_goals == 0;
    if (late > 0 && terminal_last && !terminal) atomicAdd(diag + RIAB_TD_LATE_COMPLETIONS, 1);
    // ---- RewardCache.get_total (:929-939): survivors, then this step's awards, then the default level
    total = total + L.new_total;
    total = total + a.default_level;
    if (total > rmax) ts_at(a, RIAB_TS_R_MAX, b) = total;
    if (total < rmin) ts_at(a, RIAB_TS_R_MIN, b) = total;
    reward_out[b] = total;
    terminal_out[b] = terminal_last ? 1 : 0;
  }
  probe(11);
  mid.reset = RESET && (STEP ? terminal_last : true);
  return mid;
}
template <int MODE, class Draw>
__device__ __forceinline__ void task_lane_reset(const TaskArgs& a, const ResetArgs& r, int64_t b, LaneIn& in, LaneMid& mid, double& px,
                                                double& py, double t_env, int32_t* diag, const Draw& draw) {
  constexpr bool RESET = MODE & 2;
  if (RESET && mid.reset) {
    const ResetDraw d = draw();
    mid.new_list = d.list;
    reset_lane_teleport(r, b, in.L, d);
    px = in.L.px;  // (moved by a reset that teleports)
    py = in.L.py;
  }
}
// ... between task_lane_reset and task_lane_finish, in either order with the latter: the ended episode's bookkeeping
template <int MODE>
__device__ __forceinline__ void task_lane_episode(const TaskArgs& a, const ResetArgs& r, int64_t b, const LaneIn& in, LaneMid& mid,
                                                  double t_env, int32_t* diag) {
  if ((MODE & 2) && mid.reset) mid.rec = reset_lane_episode(a, r, b, in.ep, t_env, diag);
}
// task_lane_finish = task_lane_newgoals, goal_vector, task_lane_store — in pieces for a caller that has other lanes work
// out the next action (it needs the lane's list, count and position after task_lane_newgoals: nothing else)
template <int MODE>
__device__ __forceinline__ void task_lane_newgoals(const TaskArgs& a, const ResetArgs& r, LaneIn& in, const LaneMid& mid) {
  if ((MODE & 2) && mid.reset) reset_lane_goals(a, r, in.L, ResetDraw{0.0, 0.0, mid.new_list});
}
template <int MODE>
__device__ __forceinline__ void task_lane_store(const TaskArgs& a, int64_t b, const LaneIn& in, const LaneMid& mid) {
  const Lane& L = in.L;
  if (MODE & 3) {  // ---- write back what changed
    if (L.n_goals != mid.n_goals0) ts_at(a, RIAB_TS_N_GOALS, b) = (double)L.n_goals;
    if (L.n_rw != mid.n_rw0) ts_at(a, RIAB_TS_N_REWARDS, b) = (double)L.n_rw;
    if (L.delayed != mid.delayed0) {
      ts_at(a, RIAB_TS_DELAYED, b) = L.delayed ? 1.0 : 0.0;
      if (L.delayed) ts_at(a, RIAB_TS_PAD_START, b) = L.pad_start;
    }
    if (L.list_dirty) store_list(a, b, L);
  }
}
template <int MODE, class Probe = NoProbe>
__device__ __forceinline__ void task_lane_finish(const TaskArgs& a, const ResetArgs& r, int64_t b, lds_f64_ptr goals, LaneIn& in,
                                                 const LaneMid& mid, double gv_scale, double& gvx, double& gvy,
                                                 const Probe& probe = Probe()) {
  task_lane_newgoals<MODE>(a, r, in, mid);
  probe(13);
  if (MODE & 4) goal_vector(a, goals, in.L, gv_scale, gvx, gvy);  // (handed back: the caller stores it)
  probe(14);
  task_lane_store<MODE>(a, b, in, mid);
}

// the goal pool (<= 4 KB) into LDS, by any number of threads of the workgroup; the caller synchronises
__device__ __forceinline__ void task_stage_goals(const TaskArgs& a, double* s_goals, int tid, int nthreads) {
  for (int i = tid; i < a.n_pool * RIAB_GOAL_COLS; i += nthreads) s_goals[i] = a.goals[i];
}

template <int MODE>
__device__ __forceinline__ void task_body(const TaskArgs& a, const ResetArgs& r, const double* pos_x, const double* pos_y,
                                          double t_env, double* reward_out, uint8_t* terminal_out, const uint8_t* mask,
                                          double gv_scale, double* gv_x, double* gv_y, int32_t* diag) {
  constexpr bool STEP = MODE & 1, RESET = MODE & 2;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  // the goal pool goes to LDS with the first batch of loads: goal rows and reward templates
  // are then ~100 ns away instead of one more global round trip per check pass / cached reward
  __shared__ double s_goals[RIAB_TASK_MAX_POOL * RIAB_GOAL_COLS];
  task_stage_goals(a, s_goals, (int)threadIdx.x, 64);
  const bool live = b < a.B && !(RESET && !STEP && mask && !mask[b]);
  double px = 0.0, py = 0.0;
  LaneIn in;
  RewardsIn rin;
  if (live) {  // (one batch of independent loads, in flight together with the pool's)
    in = task_lane_load<MODE>(a, b);
    if (STEP) rin = load_rewards_in(a, b);
    px = pos_x[b];
    py = pos_y[b];
  }
  __syncthreads();
  if (!live) return;
  double gvx = 0.0, gvy = 0.0;
  RewardsOut ro = {0, 0.0};
  if (STEP) ro = rewards_step(a, (lds_f64_ptr)s_goals, b, rin);
  LaneMid mid = task_lane_goals<MODE>(a, b, (lds_f64_ptr)s_goals, in, ro, px, py, t_env, reward_out, terminal_out, diag);
  task_lane_reset<MODE>(a, r, b, in, mid, px, py, t_env, diag, [&]() { return reset_draw(a, r, b); });
  task_lane_episode<MODE>(a, r, b, in, mid, t_env, diag);
  task_lane_finish<MODE>(a, r, b, (lds_f64_ptr)s_goals, in, mid, gv_scale, gvx, gvy);
  if (MODE & 4) {
    gv_x[b] = gvx;
    gv_y[b] = gvy;
  }
  if (RESET) episode_log_store(r, b, mid.rec, t_env, diag);
}

template <int MODE>
__global__ __launch_bounds__(64) void task_kernel(TaskArgs a, ResetArgs r, const double* pos_x, const double* pos_y,
                                                  double t_env, double* reward_out, uint8_t* terminal_out,
                                                  const uint8_t* mask, double gv_scale, double* gv_x, double* gv_y,
                                                  int32_t* diag) {
  task_body<MODE>(a, r, pos_x, pos_y, t_env, reward_out, terminal_out, mask, gv_scale, gv_x, gv_y, diag);
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

static int fill_reset(ResetArgs& r, const RiabEnv* env, int64_t agent_id0, int32_t n_select, int32_t ordered, uint64_t seed,
                      uint64_t counter, int32_t teleport, const double* new_x, const double* new_y, double* pos_x,
                      double* pos_y, float* hist_x, float* hist_y, double* ep_log, int64_t ep_log_cap, int32_t* ep_count) {
  if (n_select < 0) return RIAB_EINVAL;
  if (n_select > RIAB_TASK_MAX_GOALS - 1) return RIAB_ETOOBIG;  // one slot stays free for the termination-delay goal
  if (teleport && (!pos_x || !pos_y)) return RIAB_EINVAL;
  if ((new_x == nullptr) != (new_y == nullptr) || (hist_x == nullptr) != (hist_y == nullptr)) return RIAB_EINVAL;
  if (ep_log && (!ep_count || ep_log_cap <= 0)) return RIAB_EINVAL;
  const double w = env->extent[1] - env->extent[0], h = env->extent[3] - env->extent[2];
  if (teleport && !new_x && w != h) return RIAB_EUNSUPPORTED;  // sample_positions(1) only works for a square box
  r.agent_id0 = agent_id0;
  r.n_select = n_select;
  r.ordered = ordered;
  r.teleport = teleport;
  r.seed = seed;
  r.counter = counter;
  r.new_x = new_x;
  r.new_y = new_y;
  r.pos_x = pos_x;
  r.pos_y = pos_y;
  r.hist_x = hist_x;
  r.hist_y = hist_y;
  r.cx = 0.5 * (env->extent[0] + env->extent[1]);
  r.cy = 0.5 * (env->extent[2] + env->extent[3]);
  r.half = 0.45 * sqrt(w * h);
  r.ep_log = ep_log;
  r.ep_log_cap = ep_log_cap;
  r.ep_count = ep_count;
  return RIAB_OK;
}

}  // namespace riab

// (end of the no-contraction region: code that includes this header after it keeps the default)
#pragma clang fp contract(fast)
