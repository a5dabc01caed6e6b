#pragma once
// The list logic of a TaskEnvironment whose lanes are the agents of ONE world (agentmode = "interact",
// reference contribs/TaskEnvironment.py:1030, 1076-1172): integer code only, no device intrinsics, so that the
// same text is compiled into the kernel (riab_task_world.hip) and, by tests/test_task_cpu.py with g++, into a
// host harness that walks it over the reference's own GoalCache.check outcomes (tests/golden/taskworld_list_logic.npz).
//
// With "interact" GoalCache.pop removes a satisfied goal from EVERY agent's list (:1165-1172), and reset /
// append fill every list alike (:1204-1211, :1250-1252): the lists stay equal — one shared list.  A check pass
// gives the agents their turns in `agent_names` order against the list as the earlier agents of the pass left it
// (:1100-1142).
#include <stdint.h>

#if defined(__HIPCC__)
#define RIAB_WL_FN __host__ __device__ __forceinline__
#else
#define RIAB_WL_FN static inline
#endif

#define RIAB_WL_PAD 0xFEu  // the termination-delay goal's byte in a list (RIAB_GOAL_TIME_ELAPSED & 0xFF)
#define RIAB_WL_MAX_AWARDS 32

namespace riab {

struct WorldList {
  uint8_t* e;  // [16] pool indices in list order (LDS in the kernel: dynamically indexed)
  int n;
};
struct WorldAward {
  int agent;
  int entry;  // pool index, or RIAB_WL_PAD
};

// Which agents can change the list in their turn: the pool goals a turn looks at (nonsequential: every entry; sequential:
// the head — `this` = last achieved + 1 is the head for everybody, because pop() rewinds every agent's marker, :1167-1172),
// and whether a turn looks at the termination-delay goal (which is met by whoever looks, once its time has elapsed).
RIAB_WL_FN uint64_t world_turn_mask(const WorldList& l, bool sequential, bool& looks_at_pad) {
  uint64_t mask = 0;
  looks_at_pad = false;
  const int n = sequential ? (l.n > 0 ? 1 : 0) : l.n;
  for (int g = 0; g < n; ++g) {
    const uint8_t v = l.e[g];
    if (v == RIAB_WL_PAD) looks_at_pad = true;
    else mask |= 1ull << v;
  }
  return mask;
}

// One agent's turn of GoalCache.check(remove_finished=True).  `met`: bit p = the agent stands inside pool goal p
// (SpatialGoal.check, :1337-1360); `pad_elapsed`: TimeElapsedGoal.check (:1271-1278).  Awards are appended to `out`
// in the order the reference appends them to (rewards, agents).  Returns the number of goals consumed.
RIAB_WL_FN int world_agent_turn(WorldList& l, uint64_t met, bool pad_elapsed, bool sequential, int agent, WorldAward* out,
                                int& n_out) {
  int done = 0;
  if (l.n == 0) return 0;  // :1102 / :1126
  int g = 0;
  while (g < l.n) {
    const uint8_t v = l.e[g];
    const bool hit = v == RIAB_WL_PAD ? pad_elapsed : (bool)((met >> v) & 1ull);
    if (hit) {
      if (n_out < RIAB_WL_MAX_AWARDS) {
        out[n_out].agent = agent;
        out[n_out].entry = v;
        n_out += 1;
      }
      for (int i = g; i + 1 < l.n; ++i) l.e[i] = l.e[i + 1];  // GoalCache.pop: for everybody
      l.n -= 1;
      done += 1;
    }
    if (sequential) break;  // one look at the head per agent and pass (:1107-1116)
    g += 1;                 // also after a pop (:1141): the goal that slid into slot g is left to the later agents
  }
  return done;
}

}  // namespace riab
