#pragma once
// The pieces of the one-world task's step (riab_task_world.hip: semantics and references there) that more than one kernel
// runs: the stand-alone world step, the step plan's motion + world step, and the one-launch closed-loop step
// (riab_step1.hip), whose writer workgroups keep the world's books.
#include "riab_task_kernel.h"   // (turns fp contraction off for its own code)
#include "riab_task_world_logic.h"

#pragma clang fp contract(off)

#define RIAB_WORLD_STREAM_ID 0xFFFFFFFFull  // Philox "agent id" of what the world draws once (its goal selection)

namespace riab {

// What phase A hands to phase B across workgroups — a candidate's "stands inside" mask, its place on the work list, the
// survivors' sum of its rewards, the length of its reward cache — travels write-through and is read past the reader's
// L1 (relaxed agent-scope accesses: global_store / global_load ... sc1), the writer having waited for its stores'
// acknowledgement before it takes its ticket.  Two __threadfence() — a write-back of the L2's dirty lines on one side, an
// invalidation on the other, 3-7 us between them on this chip — used to stand where the ticket is taken; a quiet step
// (no candidate: nothing to hand over but the counter) paid them all the same.
__device__ __forceinline__ void st_agent(uint64_t* p, uint64_t v) {
  __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(uintptr_t)p, (unsigned long long)v, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) {
  return (uint64_t)__hip_atomic_load((__attribute__((address_space(1))) unsigned long long*)(uintptr_t)p, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int32_t* p, int32_t v) {
  __hip_atomic_store((__attribute__((address_space(1))) int*)(uintptr_t)p, (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int32_t ld_agent(const int32_t* p) {
  return (int32_t)__hip_atomic_load((__attribute__((address_space(1))) int*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) { st_agent(reinterpret_cast<uint64_t*>(p), (uint64_t)__double_as_longlong(v)); }
__device__ __forceinline__ double ld_agent(const double* p) {
  return __longlong_as_double((long long)ld_agent(reinterpret_cast<const uint64_t*>(p)));
}

struct WorldShared {  // phase B's state, in LDS
  uint8_t list[RIAB_TASK_MAX_GOALS];
  int n;
  WorldAward awards[RIAB_WL_MAX_AWARDS];
  int n_awards;
  int next_agent;  // min-reduction slot
  int last;
};

// One GoalCache.check(remove_finished=True) over all agents, by the whole workgroup (uniform control flow).  Only the
// step's CANDIDATES can take a turn that changes anything: the lanes that stand in a goal of the list as the step found
// it (the list only shrinks within a step) — phase A left their indices in `cand` — and, for a termination-delay goal whose
// time has elapsed, whoever's turn comes first.  Returns the number of goals consumed.
template <int BLOCK>
__device__ int world_pass(WorldShared& S, const uint64_t* met, const int32_t* cand, int n_cand, int64_t B, bool pad_elapsed,
                          bool sequential) {
  const int tid = (int)threadIdx.x;
  int64_t a_next = 0;
  int done = 0;
  for (;;) {
    __syncthreads();  // (the list as the last turn left it)
    WorldList l = {S.list, S.n};
    bool looks_at_pad;
    const uint64_t mask = world_turn_mask(l, sequential, looks_at_pad);
    const bool pad_now = looks_at_pad && pad_elapsed;
    if (l.n == 0 || a_next >= B || ((n_cand == 0 || mask == 0) && !pad_now)) break;
    int64_t who;
    if (pad_now) {
      who = a_next;  // whoever's turn it is takes the termination-delay goal
    } else {
      if (tid == 0) S.next_agent = 0x7FFFFFFF;
      __syncthreads();
      int mine = 0x7FFFFFFF;
      for (int i = tid; i < n_cand; i += BLOCK) {  // (the candidates are in no particular order)
        const int c = ld_agent(cand + i);
        if (c >= a_next && c < mine && (ld_agent(met + c) & mask)) mine = c;
      }
      if (mine != 0x7FFFFFFF) atomicMin(&S.next_agent, mine);
      __syncthreads();
      if (S.next_agent == 0x7FFFFFFF) break;
      who = S.next_agent;
    }
    const int before = S.n;
    __syncthreads();  // (everybody has read the slot and the list)
    if (tid == 0) {
      WorldList w = {S.list, S.n};
      world_agent_turn(w, ld_agent(met + who), pad_now, sequential, (int)who, S.awards, S.n_awards);
      S.n = w.n;
    }
    __syncthreads();
    done += before - S.n;
    a_next = who + 1;
  }
  return done;
}

// RewardCache.get_total (:929-939) and the cache's statistics of one agent, from the python sum() of its rewards
__device__ __forceinline__ void world_lane_total(const TaskArgs& a, int64_t b, double sum, double rmax, double rmin,
                                                 double* reward_out) {
  const double total = sum + a.default_level;
  if (total > rmax) ts_at(a, RIAB_TS_R_MAX, b) = total;
  if (total < rmin) ts_at(a, RIAB_TS_R_MIN, b) = total;
  reward_out[b] = total;
}

// what a lane's phase A reads, asked for in one batch (the one-launch step: with the agent's state, a motion step ahead)
struct WorldLaneIn {
  RewardsIn rin;
  double rmax, rmin;
};
__device__ __forceinline__ WorldLaneIn world_lane_load(const TaskArgs& a, int64_t b) {
  WorldLaneIn in;
  in.rin = load_rewards_in(a, b);
  in.rmax = ts_at(a, RIAB_TS_R_MAX, b);
  in.rmin = ts_at(a, RIAB_TS_R_MIN, b);
  return in;
}

// ---- phase A for one lane: what an agent's step needs of nobody else.  S: the shared list as the step found it;
// ctl[1]: the number of candidates (zero between launches)
__device__ __forceinline__ void world_phase_a(const TaskArgs& a, lds_f64_ptr goals, const WorldShared& S, int64_t b, double px,
                                              double py, const WorldLaneIn& in, double t_env, double pad_start0,
                                              uint8_t terminal_prev, double* reward_out, uint8_t* terminal_out, uint64_t* met,
                                              int32_t* cand, int32_t* ctl) {
  const RewardsOut ro = rewards_step(a, goals, b, in.rin);
  if (ro.n_rw != in.rin.n_rw) st_agent(&ts_at(a, RIAB_TS_N_REWARDS, b), (double)ro.n_rw);  // (phase B appends behind it)
  uint64_t m = 0;
  bool pad_in_list = false;
  for (int g = 0; g < S.n; ++g) {
    const int v = S.list[g];
    if (v == (int)RIAB_WL_PAD) pad_in_list = true;
    else if (in_goal_radius(a, px, py, goals + v * RIAB_GOAL_COLS)) m |= 1ull << v;
  }
  st_agent(met + b, m);
  // a termination-delay goal whose time has elapsed goes to the first agent (its pass starts with agent 0)
  const bool candidate = m != 0 || (b == 0 && pad_in_list && t_env - pad_start0 >= a.terminate_delay);
  if (candidate) {
    st_agent(reward_out + b, ro.total);  // (the survivors' sum: phase B adds this step's awards, then totals)
    const int slot = atomicAdd(ctl + 1, 1);
    if (slot < a.B) st_agent(cand + slot, (int32_t)b);  // (always, unless the caller's counter did not start at zero)
  } else {
    world_lane_total(a, b, ro.total, in.rmax, in.rmin, reward_out);
  }
  // (the world's flag; phase B rewrites the column when it changes — from another workgroup, possibly behind another L2:
  // written through here, so that the rewrite, which comes after this store's acknowledgement, is what stays)
  __hip_atomic_store((__attribute__((address_space(1))) uint8_t*)(uintptr_t)(terminal_out + b), terminal_prev, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// ---- phase B, by the whole workgroup that took the last ticket: the step's check passes over the shared list, the
// awards, the candidates' totals, the shared state.  Returns _is_terminal_state as the step's last pass left it; the
// list is left in S.
template <int BLOCK>
__device__ bool world_phase_b(const TaskArgs& a, lds_f64_ptr goals, WorldShared& S, double* world, double t_env,
                              double pad_start0, uint8_t terminal_prev, double* reward_out, uint8_t* terminal_out,
                              const uint64_t* met, const int32_t* cand, int32_t* ctl, int32_t* diag) {
  const int tid = (int)threadIdx.x;
  int n_cand = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n_cand > a.B) n_cand = (int)a.B;
  if (tid == 0) S.n_awards = 0;
  const bool sequential = a.goalorder == RIAB_GOALORDER_SEQUENTIAL;
  const int n0 = S.n;
  bool delayed = world[RIAB_TW_DELAYED] != 0.0;
  const bool delayed0 = delayed;
  double pad_start = pad_start0;
  // _is_terminal_state (:278-290) as step() calls it (:418-440)
  world_pass<BLOCK>(S, met, cand, n_cand, a.B, t_env - pad_start >= a.terminate_delay, sequential);
  bool terminal = S.n == 0;
  if (terminal && a.terminate_delay != 0.0 && !delayed) {  // :421-434: one unrewarded TimeElapsedGoal pads the episode
    delayed = true;
    pad_start = t_env;
    __syncthreads();
    if (tid == 0) {
      S.list[0] = (uint8_t)RIAB_WL_PAD;
      S.n = 1;
    }
    world_pass<BLOCK>(S, met, cand, n_cand, a.B, t_env - pad_start >= a.terminate_delay, sequential);
    terminal = S.n == 0;
  }
  const int late = world_pass<BLOCK>(S, met, cand, n_cand, a.B, t_env - pad_start >= a.terminate_delay, sequential);  // :438
  const bool terminal_last = S.n == 0;
  if (tid == 0) {
    if (late > 0 && terminal_last && !terminal) atomicAdd(diag + RIAB_TD_LATE_COMPLETIONS, 1);
    // _is_terminal_state :284-285 — RewardCache.append (:902-911) in award order
    for (int i = 0; i < S.n_awards; ++i) {
      const int64_t w = S.awards[i].agent;
      const int src = S.awards[i].entry == (int)RIAB_WL_PAD ? RIAB_GOAL_TIME_ELAPSED : S.awards[i].entry;
      const int n_rw = (int)ld_agent(&ts_at(a, RIAB_TS_N_REWARDS, w));
      if (n_rw >= RIAB_TASK_MAX_REWARDS) {
        atomicAdd(diag + RIAB_TD_REWARD_OVERFLOW, 1);
        continue;
      }
      const RewardTpl r = reward_of(a, goals, src);
      ts_at(a, RIAB_TS_RW_STATE + n_rw, w) = r.init;
      ts_at(a, RIAB_TS_RW_EXPIRE + n_rw, w) = r.expire;
      ts_at(a, RIAB_TS_RW_SRC + n_rw, w) = (double)src;
      st_agent(&ts_at(a, RIAB_TS_N_REWARDS, w), (double)(n_rw + 1));
      st_agent(reward_out + w, ld_agent(reward_out + w) + r.init);  // python sum(): left to right, the new rewards last
    }
    // the shared state
    if (S.n != n0) world[RIAB_TW_N_GOALS] = (double)S.n;
    if (delayed != delayed0) {
      world[RIAB_TW_DELAYED] = 1.0;
      world[RIAB_TW_PAD_START] = pad_start;
    }
    if (S.n != n0 || delayed != delayed0)
      for (int i = 0; i < RIAB_TASK_MAX_GOALS; ++i)
        world[RIAB_TW_GOAL_LIST + i] = i < S.n ? (S.list[i] == RIAB_WL_PAD ? (double)RIAB_GOAL_TIME_ELAPSED : (double)S.list[i]) : 0.0;
    if ((terminal_last ? 1 : 0) != terminal_prev) world[RIAB_TW_TERMINAL] = terminal_last ? 1.0 : 0.0;
    ctl[0] = 0;  // (the next launch is ordered behind this one)
    ctl[1] = 0;
  }
  __syncthreads();
  for (int i = tid; i < n_cand; i += BLOCK) {  // the candidates' totals, with what they were awarded
    const int64_t c = ld_agent(cand + i);
    world_lane_total(a, c, ld_agent(reward_out + c), ts_at(a, RIAB_TS_R_MAX, c), ts_at(a, RIAB_TS_R_MIN, c), reward_out);
  }
  if ((terminal_last ? 1 : 0) != terminal_prev)
    for (int64_t i = tid; i < a.B; i += BLOCK) terminal_out[i] = terminal_last ? 1 : 0;
  return terminal_last;
}

// ---- the world's own books at a reset (TaskEnvironment.reset :307-351): write_end_episode (:536-539), the episode
// counter (:333-338), the new list — by ONE thread (agent 0's).  L: the new list.
__device__ __forceinline__ void world_reset_books(const TaskArgs& a, const ResetArgs& r, double* world, double t_env,
                                                  const Lane& L, int32_t* diag) {
  atomicAdd(diag + RIAB_TD_RESETS, 1);
  bool zero_duration = false;
  bool any_ended = world[RIAB_TW_EP_ANY_ENDED] != 0.0;
  const double episode = world[RIAB_TW_EPISODE], ep_start = world[RIAB_TW_EP_START];
  if (world[RIAB_TW_STARTED] != 0.0) {
    const double duration = t_env - ep_start;
    zero_duration = duration == 0.0;
    if (!zero_duration) {  // a zero-duration episode is popped again right away (:333-335)
      any_ended = true;
      world[RIAB_TW_EP_ANY_ENDED] = 1.0;
      if (r.ep_log) {
        const int slot = atomicAdd(r.ep_count, 1);
        if (slot < r.ep_log_cap) {
          double* e = r.ep_log + (int64_t)slot * 5;
          e[0] = -1.0;  // (no lane: the world's episode)
          e[1] = episode;
          e[2] = ep_start;
          e[3] = t_env;
          e[4] = duration;
        } else {
          atomicAdd(diag + RIAB_TD_EPLOG_OVERFLOW, 1);
        }
      }
    }
  }
  if (!zero_duration) world[RIAB_TW_EPISODE] = episode + 1.0;
  world[RIAB_TW_STARTED] = 1.0;
  world[RIAB_TW_EP_START] = any_ended ? t_env : 0.0;  // _current_episode_start (:526-527)
  for (int i = 0; i < RIAB_TASK_MAX_GOALS; ++i) world[RIAB_TW_GOAL_LIST + i] = i < L.n_goals ? (double)list_get(L.list, i) : 0.0;
  world[RIAB_TW_N_GOALS] = (double)L.n_goals;
  world[RIAB_TW_DELAYED] = 0.0;
}

// the shared list as the world keeps it -> a lane's packed list
__device__ __forceinline__ void world_list_load(const double* world, Lane& L) {
  L.n_goals = (int)world[RIAB_TW_N_GOALS];
  u128 l = 0;
  for (int i = 0; i < RIAB_TASK_MAX_GOALS; ++i)
    if (i < L.n_goals) l |= (u128)(uint32_t)((int)world[RIAB_TW_GOAL_LIST + i] & 0xFF) << (8 * i);
  L.list = l;
}

}  // namespace riab
