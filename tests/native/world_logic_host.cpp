// Host harness around ratinabox_amd/csrc/riab_task_world_logic.h (test infrastructure, built by tests/test_task_cpu.py with
// g++): the control flow of the kernel's world_pass (riab_task_world.hip) — only an agent that stands in a goal its turn
// looks at takes a turn — as a serial loop over the same world_turn_mask / world_agent_turn.
#include "riab_task_world_logic.h"

extern "C" int world_pass_host(uint8_t* list, int* n, const uint64_t* met, int n_agents, int pad_elapsed, int sequential,
                               int* award_agent, int* award_entry) {
  riab::WorldList l = {list, *n};
  riab::WorldAward out[RIAB_WL_MAX_AWARDS];
  int n_out = 0, a_next = 0;
  for (;;) {
    bool looks_at_pad;
    const uint64_t mask = riab::world_turn_mask(l, sequential != 0, looks_at_pad);
    const bool pad_now = looks_at_pad && pad_elapsed;
    if (l.n == 0 || a_next >= n_agents) break;
    int cand = -1;
    if (pad_now) cand = a_next;
    else
      for (int i = a_next; i < n_agents; ++i)
        if (met[i] & mask) {
          cand = i;
          break;
        }
    if (cand < 0) break;
    riab::world_agent_turn(l, met[cand], pad_now, sequential != 0, cand, out, n_out);
    a_next = cand + 1;
  }
  *n = l.n;
  for (int i = 0; i < n_out; ++i) {
    award_agent[i] = out[i].agent;
    award_entry[i] = out[i].entry;
  }
  return n_out;
}
