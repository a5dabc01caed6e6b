// Step plans: the per-step (closed-loop) path as ONE native call.
//
// `Agent.update(); N.update() for N in neurons` costs, from Python, one ctypes transition,
// struct filling and history bookkeeping per kernel — 30+ us per step at cfg 2 against ~9 us of
// GPU time.  A plan records the agent and its populations once (the same arguments the
// per-kernel entry points take); riab_plan_step then advances the row cursors, RNG counters and
// pointers in C++ and enqueues the motion kernel and every population's rate kernel for each
// requested step.  No allocation, no synchronisation: histories are chunks handed in by the
// caller, and the call reports RIAB_EFULL (before launching anything) when a chunk is exhausted.
#include <cstdlib>
#include <new>
#include <vector>

#include "riab_agent_kernel.h"

namespace riab {
int launch_task_fused(const RiabEnv* env, const RiabTask* task, double* task_state, double* pos_x, double* pos_y, int64_t B,
                      double t_env, double* reward_out, uint8_t* terminal_out, int32_t* diag, bool auto_reset,
                      int64_t agent_id0, int32_t n_select, int32_t ordered, uint64_t seed, uint64_t counter,
                      int32_t teleport, float* hist_x, float* hist_y, double* ep_log, int64_t ep_log_cap,
                      int32_t* ep_count, double gv_scale, double* gv_x, double* gv_y, hipStream_t s);

int launch_motion_task(const AgentArgs& ma, const RiabEnv* env, const RiabTask* task, double* task_state, double* pos_x,
                       double* pos_y, int64_t task_B, double t_env, double* reward_out, uint8_t* terminal_out,
                       int32_t* diag, bool auto_reset, int64_t agent_id0, int32_t n_select, int32_t ordered, uint64_t seed,
                       uint64_t counter, int32_t teleport, float* hist_x, float* hist_y, double* ep_log,
                       int64_t ep_log_cap, int32_t* ep_count, double gv_scale, double* gv_x, double* gv_y, hipStream_t s);

int launch_motion_world(const AgentArgs& ma, const RiabEnv* env, const RiabTask* task, double* task_state, double* world,
                        const double* pos_x, const double* pos_y, int64_t task_B, double t_env, double* reward_out,
                        uint8_t* terminal_out, uint64_t* met, int32_t* cand, int32_t* ctl, int32_t* diag, hipStream_t s);

// boundary vector cells with the ray exchange of one-row launches (riab_bvc.hip)
int launch_bvc(const RiabEnv* env, const RiabRateIO* io, const double* test_dirs, const double* ray_rden, int32_t K,
               const float* cells, const float* vm_table, const float* inv_norm, int32_t n, int32_t egocentric,
               float* ray_out, const int32_t* cell_rows, const int32_t* windows, float* xch, uint32_t* xch_count,
               uint32_t* xch_arrivals, int n_cus, hipStream_t stream);

// the one-launch step (riab_step1.hip)
int step1_supported(const RiabEnv* env, const RiabPopulation* pop, int64_t B);
int launch_step1(const AgentArgs& a, const RiabEnv* env, const Step1PopRef* refs, int n_pops, uint64_t seed, uint64_t step_after,
                 uint32_t* sync_words, uint32_t epoch, bool* walls_ready, int n_cus, hipStream_t s, bool query);
int launch_step1_task(const AgentArgs& a, const RiabEnv* env, const Step1PopRef* refs, int n_pops, uint64_t seed,
                      uint64_t step_after, uint32_t* sync_words, uint32_t epoch, bool* walls_ready, int n_cus, const RiabTask* task,
                      double* task_state, int64_t task_B, double t_env, double* reward_out, uint8_t* terminal_out, int32_t* diag,
                      bool auto_reset, int32_t n_select, int32_t ordered, uint64_t task_seed, uint64_t counter, int32_t teleport,
                      double* ep_log, int64_t ep_log_cap, int32_t* ep_count, double gv_scale, double* gv_x, double* gv_y,
                      double* world, uint64_t* world_met, int32_t* world_cand, int32_t* world_ctl, hipStream_t s, bool query);
}  // namespace riab

struct RiabPlan {
  RiabEnv env;
  RiabMotion motion;
  double* state;
  int64_t B;
  int64_t agent_id0;
  uint64_t seed;
  uint64_t step;  // number of Agent.update() steps taken so far (the RNG counter)
  const double* drift;
  // imported / forced trajectory (riab_plan_set_forced): positions of the coming steps, [rows][2][B]; null = motion model
  const double* forced;
  int64_t forced_rows, forced_fill;
  float* hist_base;     // [cap][8][B]
  int64_t hist_cap, hist_fill;
  float* row_scratch;   // [8][B] used when no history chunk is attached
  int32_t* diag;
  std::vector<RiabPopulation> pops;
  std::vector<int64_t> pop_fill;
  // attached task (riab_plan_set_task)
  bool has_task;
  RiabTask task;
  double* task_state;
  int64_t task_B;
  double t_env, dt_env;
  double* reward_out;
  uint8_t* terminal_out;
  int32_t* task_diag;
  int32_t auto_reset, n_select, ordered, teleport;
  uint64_t task_seed, reset_counter;
  double* ep_log;
  int64_t ep_log_cap;
  int32_t* ep_count;
  double scripted_speed;
  // ... whose lanes are the agents of one world (riab_plan_set_task_world); null: every lane its own replica
  double* world;
  uint64_t* world_met;
  int32_t* world_cand;
  int32_t* world_ctl;
  bool action_ready;  // the drift buffer holds the scripted action of the coming step
  // the one-launch step (riab_plan_set_fused)
  uint32_t* sync_words;
  uint32_t epoch;        // tag of the last one-launch step on sync_words
  bool walls_ready;      // the wall table behind sync_words has been prepared (the plan's first one-launch step does it)
  int n_cus;             // compute units the plan's launches can occupy (riab_plan_set_compute_units)
  int fused_n;           // populations whose update() rides in the agent step's launch; -2: not worked out yet
  int fused[RIAB_STEP1_MAX_POPS];  // ... their indices, in list order
  bool fused_whole;      // ... worked out for whole-plan steps (riab_plan_step) / for the split entry points
  int64_t fused_steps, launches;
  // split entry points: riab_plan_step_agent wrote the rows of step `pre_step` of the populations flagged in
  // `pre_pending` ahead; a row nobody claimed (riab_plan_step_population) is a miss of its population
  std::vector<uint32_t> xch_arrivals;  // per population: arrivals its ray-exchange launches have asked of bvc_xch_count so far
  std::vector<char> pre_pending;  // per population
  uint64_t pre_step;
  std::vector<int> pre_misses;    // per population, in a row
};

static int fused_agent_step(RiabPlan* p, float* row, hipStream_t s, bool need_free_row, uint32_t* mask, bool query);

// this step's rows of the plan's fused populations (split: those whose chunk has a free row); returns how many
static int fused_refs(const RiabPlan* p, riab::Step1PopRef* refs, uint32_t* mask, bool need_free_row) {
  int n = 0;
  *mask = 0u;
  for (int k = 0; k < p->fused_n; ++k) {
    const int i = p->fused[k];
    const RiabPopulation& q = p->pops[i];
    if (need_free_row && q.capacity_rows > 0 && p->pop_fill[i] >= q.capacity_rows) continue;
    const int64_t r = q.capacity_rows > 0 ? p->pop_fill[i] : 0;
    const int64_t row_elems = (int64_t)q.n * p->B;
    refs[n].pop = &q;
    refs[n].rates_row = q.rates_base + r * row_elems;
    refs[n].spikes_row = q.spikes_base ? q.spikes_base + r * row_elems : nullptr;
    *mask |= 1u << k;
    ++n;
  }
  return n;
}

// One closed-loop step of a plan with a task as ONE kernel: Agent.update(), the rest of TaskEnvironment.step (+ the caller's
// `if terminal: reset()`, + the next scripted action) and the fused populations' update().  The caller has advanced the
// plan's step counter and clock.  `query`: nothing is launched; RIAB_OK when there is a kernel for this plan's step.
static int fused_task_step(RiabPlan* p, float* row, hipStream_t s, bool query) {
  riab::AgentArgs ma;
  const uint64_t step_before = query ? p->step : p->step - 1;
  int rc = riab::fill_agent_args(ma, &p->env, &p->motion, p->state, p->B, p->agent_id0, p->drift, nullptr, nullptr, nullptr, p->seed,
                                 step_before, 1, row, p->diag);
  if (rc) return rc;
  const bool scripted = p->scripted_speed > 0.0;
  double* act = const_cast<double*>(p->drift);
  if (scripted && (!act || !p->motion.has_drift)) return RIAB_EINVAL;
  riab::Step1PopRef refs[RIAB_STEP1_MAX_POPS];
  uint32_t mask;
  const int n = fused_refs(p, refs, &mask, false);
  uint32_t epoch = 1u;
  if (!query) {
    p->epoch += 1u;
    if (p->epoch == 0u) p->epoch = 1u;
    epoch = p->epoch;
  }
  rc = riab::launch_step1_task(ma, &p->env, refs, n, p->seed, step_before + 1, p->sync_words, epoch, &p->walls_ready, p->n_cus,
                               &p->task, p->task_state, p->task_B, p->t_env, p->reward_out, p->terminal_out, p->task_diag,
                               p->auto_reset != 0, p->n_select, p->ordered, p->task_seed, p->reset_counter, p->teleport, p->ep_log,
                               p->ep_log_cap, p->ep_count, p->scripted_speed, scripted ? act : nullptr,
                               scripted ? act + p->B : nullptr, p->world, p->world_met, p->world_cand, p->world_ctl, s, query);
  if (rc == RIAB_OK && !query) {
    p->fused_steps += 1;
    p->launches += 1;
  }
  return rc;
}

// The populations whose update() rides in the agent step's launch: every store-bound one without additive noise the
// one-launch step has a functor for (riab_step1.hip: step1_supported), up to RIAB_STEP1_MAX_POPS of them — the ones
// that write most bytes per row —, in list order.  Returns how many (0: the step is launched kernel by kernel).
// `whole_step`: for riab_plan_step; otherwise for the split entry points, where a population that keeps no history
// (its one row is what `firingrate` shows until ITS update() call) and one the caller's loop does not update after
// every agent step are left out.  (a plan with a task: whole-plan steps only, motion + task fused (RIAB_OPT_FUSED_TASK))
static int plan_fused(RiabPlan* p, bool whole_step = false) {
  if (!p->sync_words || riab::g_options[RIAB_OPT_FUSED_STEP] == 0 || p->forced) return 0;
  if (p->has_task && (!whole_step || riab::g_options[RIAB_OPT_FUSED_TASK] == 0)) return 0;
  if (p->fused_n == -2 || p->fused_whole != whole_step) {
    p->fused_whole = whole_step;
    p->pre_misses.resize(p->pops.size(), 0);
    p->pre_pending.resize(p->pops.size(), 0);
    if (p->n_cus <= 0) {  // (nobody said: the device's own count)
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 1;
      p->n_cus = n;
    }
    int cand[RIAB_STEP1_MAX_POPS];
    int64_t bytes[RIAB_STEP1_MAX_POPS];
    int n = 0;
    for (size_t i = 0; i < p->pops.size(); ++i) {
      const RiabPopulation& q = p->pops[i];
      if (riab::step1_supported(&p->env, &q, p->B) != RIAB_OK) continue;
      if (!whole_step && (q.capacity_rows == 0 || p->pre_misses[i] >= 2)) continue;
      const int64_t b = (int64_t)q.n * (q.spikes_base ? 5 : 4);
      if (n < RIAB_STEP1_MAX_POPS) {
        cand[n] = (int)i;
        bytes[n] = b;
        ++n;
        continue;
      }
      int least = 0;  // (more candidates than slots: the smallest gives way; the list order of the others is kept)
      for (int k = 1; k < n; ++k)
        if (bytes[k] < bytes[least]) least = k;
      if (bytes[least] >= b) continue;
      for (int k = least; k + 1 < n; ++k) {
        cand[k] = cand[k + 1];
        bytes[k] = bytes[k + 1];
      }
      cand[n - 1] = (int)i;
      bytes[n - 1] = b;
    }
    p->fused_n = n;
    for (int k = 0; k < n; ++k) p->fused[k] = cand[k];
    uint32_t mask;  // (is there a kernel and a grid for it on this device?)
    if (n > 0 && (p->has_task ? fused_task_step(p, p->row_scratch, nullptr, true)
                              : fused_agent_step(p, p->row_scratch, nullptr, false, &mask, true)) != RIAB_OK)
      p->fused_n = 0;
  }
  return p->fused_n;
}
static bool is_fused(const RiabPlan* p, int index) {
  for (int k = 0; k < p->fused_n; ++k)
    if (p->fused[k] == index) return true;
  return false;
}

// Agent.update() + the fused populations' update() of the same step as one kernel; cursors are the caller's business.
// `mask`: which of the fused populations took part (split entry points: the ones with a free row).
// (`query`: nothing is launched; RIAB_OK when there is a kernel and a grid for this plan's step)
static int fused_agent_step(RiabPlan* p, float* row, hipStream_t s, bool need_free_row, uint32_t* mask, bool query) {
  riab::Step1PopRef refs[RIAB_STEP1_MAX_POPS];
  const int n = fused_refs(p, refs, mask, need_free_row);
  if (n == 0) return RIAB_EUNSUPPORTED;
  riab::AgentArgs ma;
  int rc = riab::fill_agent_args(ma, &p->env, &p->motion, p->state, p->B, p->agent_id0, p->drift, nullptr, nullptr, nullptr,
                                 p->seed, p->step, 1, row, p->diag);
  if (rc) return rc;
  uint32_t epoch = 1u;
  if (!query) {
    p->epoch += 1u;
    if (p->epoch == 0u) p->epoch = 1u;
    epoch = p->epoch;
  }
  rc = riab::launch_step1(ma, &p->env, refs, n, p->seed, p->step + 1, p->sync_words, epoch, &p->walls_ready, p->n_cus, s, query);
  if (rc == RIAB_OK && !query) {
    p->fused_steps += 1;
    p->launches += 1;
  }
  return rc;
}

extern "C" RiabPlan* riab_plan_create(const RiabEnv* env, const RiabMotion* motion, double* state, int64_t B,
                                      int64_t agent_id0, uint64_t seed, uint64_t step, float* row_scratch,
                                      int32_t* diag) {
  if (!env || !motion || !state || B <= 0 || !row_scratch) return nullptr;
  RiabPlan* p = new (std::nothrow) RiabPlan();
  if (!p) return nullptr;
  p->env = *env;
  p->motion = *motion;
  p->state = state;
  p->B = B;
  p->agent_id0 = agent_id0;
  p->seed = seed;
  p->step = step;
  p->drift = nullptr;
  p->forced = nullptr;
  p->forced_rows = p->forced_fill = 0;
  p->hist_base = nullptr;
  p->hist_cap = p->hist_fill = 0;
  p->row_scratch = row_scratch;
  p->diag = diag;
  p->has_task = false;
  p->world = nullptr;
  p->action_ready = false;
  p->sync_words = nullptr;
  p->epoch = 0u;
  p->walls_ready = false;
  p->n_cus = 0;
  p->fused_n = -2;
  p->fused_whole = false;
  p->fused_steps = p->launches = 0;
  p->pre_step = 0;
  return p;
}

extern "C" int riab_plan_set_fused(RiabPlan* p, uint32_t* sync_words, int64_t n_words) {
  if (!p || n_words < 0) return RIAB_EINVAL;
  if (sync_words && n_words < (int64_t)RIAB_STEP1_SYNC_WORDS(p->B)) return RIAB_EINVAL;
  if (((uintptr_t)sync_words) & 7) return RIAB_EALIGN;  // (the mail's 8-byte entries, the prepared walls' float64)
  p->sync_words = sync_words;
  p->epoch = 0u;
  p->walls_ready = false;
  p->fused_n = -2;
  p->pre_pending.assign(p->pops.size(), 0);
  p->pre_misses.assign(p->pops.size(), 0);
  return RIAB_OK;
}

extern "C" int riab_plan_set_compute_units(RiabPlan* p, int32_t n_cus) {
  if (!p || n_cus < 0) return RIAB_EINVAL;
  p->n_cus = n_cus;  // (0: the device's own count, asked at the next one-launch step)
  p->fused_n = -2;
  return RIAB_OK;
}

extern "C" int64_t riab_plan_info(const RiabPlan* p, int32_t which) {
  if (!p) return 0;
  switch (which) {
    case 0: return p->fused_steps;
    case 1: return p->fused_n <= 0 ? -1 : p->fused[0];
    case 2: return p->launches;
    case 3: return p->sync_words ? 1 : 0;
    case 4: return p->fused_n < 0 ? 0 : p->fused_n;
    case 5: return p->n_cus;
    default:
      if (which >= 8 && which < 8 + RIAB_STEP1_MAX_POPS) return which - 8 < p->fused_n ? p->fused[which - 8] : -1;
      return 0;
  }
}

extern "C" void riab_plan_destroy(RiabPlan* p) { delete p; }

extern "C" int riab_plan_set_motion(RiabPlan* p, const RiabMotion* motion, const double* drift) {
  if (!p || !motion) return RIAB_EINVAL;
  if (motion->has_drift && !drift) return RIAB_EINVAL;
  if (motion->wall_repel_distance_kw != p->motion.wall_repel_distance_kw) p->walls_ready = false;  // (the box fast path's verdict depends on it)
  p->motion = *motion;
  p->drift = drift;
  p->action_ready = false;
  return RIAB_OK;
}

// Agent._update_position_along_imported_trajectory / forced_next_position (Agent.py:229-266) for the coming
// `n_rows` steps: every agent step of the plan then MOVES the agents to the next row of `forced` ([n_rows][2][B],
// float64, device) instead of running the motion model; RIAB_EFULL once the rows are used up (set the next ones).
extern "C" int riab_plan_set_forced(RiabPlan* p, const double* forced, int64_t n_rows) {
  if (!p || n_rows < 0 || (forced && n_rows == 0) || p->has_task) return RIAB_EINVAL;
  p->forced = forced;
  p->forced_rows = forced ? n_rows : 0;
  p->forced_fill = 0;
  return RIAB_OK;
}

extern "C" int riab_plan_set_agent_history(RiabPlan* p, float* hist_base, int64_t capacity_rows) {
  if (!p || capacity_rows < 0 || (capacity_rows > 0 && !hist_base)) return RIAB_EINVAL;
  p->hist_base = hist_base;
  p->hist_cap = capacity_rows;
  p->hist_fill = 0;
  return RIAB_OK;
}

extern "C" int riab_plan_add(RiabPlan* p, const RiabPopulation* pop) {
  if (!p || !pop || pop->n <= 0 || pop->kind < RIAB_POP_PLACE || pop->kind > RIAB_POP_RANDOM_SPATIAL) return RIAB_EINVAL;
  if (pop->kind == RIAB_POP_FF) {
    if (pop->n_inputs <= 0 || pop->n_inputs > RIAB_FF_MAX_INPUTS || !pop->bias) return RIAB_EINVAL;
    for (int l = 0; l < pop->n_inputs; ++l)  // feed-forward only: an input must already be in the plan
      if (pop->input_index[l] < 0 || pop->input_index[l] >= (int)p->pops.size() || !pop->input_wt[l]) return RIAB_EINVAL;
  }
  p->pops.push_back(*pop);
  p->pop_fill.push_back(0);
  p->pre_misses.push_back(0);
  p->pre_pending.push_back(0);
  p->fused_n = -2;
  return (int)p->pops.size() - 1;
}

extern "C" int riab_plan_set_population_history(RiabPlan* p, int32_t index, float* rates_base, uint8_t* spikes_base,
                                                int64_t capacity_rows) {
  if (!p || index < 0 || index >= (int)p->pops.size() || capacity_rows < 0) return RIAB_EINVAL;
  if (!rates_base) return RIAB_EINVAL;  // capacity 0 = a single-row scratch that is overwritten every step
  RiabPopulation& q = p->pops[index];
  q.rates_base = rates_base;
  q.spikes_base = spikes_base;
  q.capacity_rows = capacity_rows;
  p->pop_fill[index] = 0;
  p->pre_pending[index] = 0;  // (a row written ahead was in the old chunk)
  p->fused_n = -2;            // (spikes or not changes the bytes a row takes)
  return RIAB_OK;
}

extern "C" int riab_plan_set_task(RiabPlan* p, const RiabTask* task, double* task_state, int64_t task_B, double t_env,
                                  double dt_env, double* reward_out, uint8_t* terminal_out, int32_t* task_diag,
                                  int32_t auto_reset, int32_t n_select, int32_t ordered, uint64_t task_seed,
                                  uint64_t reset_counter, int32_t teleport, double* ep_log, int64_t ep_log_cap,
                                  int32_t* ep_count, double scripted_speed) {
  if (!p) return RIAB_EINVAL;
  p->fused_n = -2;
  p->pre_pending.assign(p->pops.size(), 0);
  p->world = nullptr;
  if (!task) {
    p->has_task = false;
    return RIAB_OK;
  }
  if (!task_state || task_B <= 0 || task_B > p->B || !reward_out || !terminal_out || !task_diag) return RIAB_EINVAL;
  if (n_select < 0 || n_select > RIAB_TASK_MAX_GOALS - 1) return RIAB_ETOOBIG;
  if (ep_log && (!ep_count || ep_log_cap <= 0)) return RIAB_EINVAL;
  p->has_task = true;
  p->task = *task;
  p->task_state = task_state;
  p->task_B = task_B;
  p->t_env = t_env;
  p->dt_env = dt_env;
  p->reward_out = reward_out;
  p->terminal_out = terminal_out;
  p->task_diag = task_diag;
  p->auto_reset = auto_reset;
  p->n_select = n_select;
  p->ordered = ordered;
  p->task_seed = task_seed;
  p->reset_counter = reset_counter;
  p->teleport = teleport;
  p->ep_log = ep_log;
  p->ep_log_cap = ep_log_cap;
  p->ep_count = ep_count;
  p->scripted_speed = scripted_speed;
  p->action_ready = false;
  return RIAB_OK;
}

extern "C" int riab_plan_set_task_world(RiabPlan* p, double* world, uint64_t* met_scratch, int32_t* cand_scratch, int32_t* ctl) {
  if (!p || !p->has_task) return RIAB_EINVAL;
  if (world && (!met_scratch || !cand_scratch || !ctl)) return RIAB_EINVAL;
  p->world = world;
  p->world_met = met_scratch;
  p->world_cand = cand_scratch;
  p->world_ctl = ctl;
  p->fused_n = -2;
  return RIAB_OK;
}

extern "C" double riab_plan_task_clock(const RiabPlan* p) { return p && p->has_task ? p->t_env : 0.0; }

extern "C" int64_t riab_plan_rows_free(const RiabPlan* p) {
  if (!p) return 0;
  int64_t free_rows = p->hist_base ? p->hist_cap - p->hist_fill : INT64_MAX;
  for (size_t i = 0; i < p->pops.size(); ++i) {
    if (p->pops[i].capacity_rows == 0) continue;  // single-row scratch (no history kept)
    const int64_t f = p->pops[i].capacity_rows - p->pop_fill[i];
    free_rows = f < free_rows ? f : free_rows;
  }
  return free_rows;
}

extern "C" uint64_t riab_plan_step_index(const RiabPlan* p) { return p ? p->step : 0; }

static int launch_population(RiabPlan* p, size_t i, const float* row, hipStream_t s) {
  RiabPopulation& q = p->pops[i];
  const int64_t B = p->B;
  RiabRateIO io = q.io;
  io.pos_x = row + RIAB_H_POS_X * B;
  io.pos_y = row + RIAB_H_POS_Y * B;
  io.hd_x = row + RIAB_H_HD_X * B;
  io.hd_y = row + RIAB_H_HD_Y * B;
  io.pos_ld = B;
  io.T = 1;
  io.B = B;
  const int64_t r = p->pop_fill[i];
  io.rates = q.rates_base + r * (int64_t)q.n * B;
  io.spikes = q.spikes_base ? q.spikes_base + r * (int64_t)q.n * B : nullptr;
  io.u_in = nullptr;
  io.dt = (float)p->motion.dt;
  io.seed = p->seed;
  io.step0 = p->step;  // Neurons.update after the p->step-th Agent.update (the cursor was already advanced)
  io.agent_id0 = p->agent_id0;
  const bool noisy = q.noise_state != nullptr;
  uint8_t* const spikes = io.spikes;
  if (noisy) io.spikes = nullptr;  // spikes are drawn on the final rate, after the noise has been added
  int rc = RIAB_EINVAL;
  switch (q.kind) {
    case RIAB_POP_PLACE:
      rc = riab_place_cells(&p->env, &io, q.table, q.n, q.description, q.geometry, q.top_hat_width, s);
      break;
    case RIAB_POP_GRID:
      rc = riab_grid_cells(&io, q.table, q.n, q.description, q.f0, s);
      break;
    case RIAB_POP_HDC:
      rc = riab_head_direction_cells(&io, q.table, q.n, s);
      break;
    case RIAB_POP_VELOCITY:  // Agent.velocity: rows of the float64 state, not of the history record
      rc = riab_velocity_cells(&io, q.table, q.n, q.one_sigma_speed, p->state + RIAB_S_VEL_X * B,
                               p->state + RIAB_S_VEL_Y * B, s);
      break;
    case RIAB_POP_SPEED:  // history["vel"][-1]: the measured velocity of the step just taken
      io.hd_x = row + RIAB_H_VEL_X * B;
      io.hd_y = row + RIAB_H_VEL_Y * B;
      rc = riab_speed_cell(&io, q.one_sigma_speed, s);
      break;
    case RIAB_POP_RANDOM_SPATIAL:
      rc = riab_random_spatial_neurons(&p->env, &io, q.table, q.n_anchors, q.targets, q.n, q.geometry, s);
      break;
    case RIAB_POP_BVC: {
      if (p->n_cus <= 0) {  // (nobody said: the device's own count)
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 1;
        p->n_cus = n;
      }
      if (p->xch_arrivals.size() < p->pops.size()) p->xch_arrivals.resize(p->pops.size(), 0u);
      rc = riab::launch_bvc(&p->env, &io, q.test_dirs, q.ray_rden, q.K, q.table, q.vm_table, q.inv_norm, q.n, q.egocentric,
                            nullptr, q.cell_rows, q.windows, q.bvc_xch, q.bvc_xch_count, &p->xch_arrivals[i], p->n_cus, s);
      break;
    }
    case RIAB_POP_OVC:
      rc = riab_object_vector_cells(&p->env, &io, q.objects, q.object_types, q.n_objects, q.table, q.n, q.walls_occlude,
                                    q.egocentric, s);
      break;
    case RIAB_POP_FF: {
      RiabFFInput in[RIAB_FF_MAX_INPUTS];
      for (int l = 0; l < q.n_inputs; ++l) {
        const int j = q.input_index[l];
        const RiabPopulation& src = p->pops[j];
        // population j < i has been launched this step already: its cursor points past the row it wrote
        const int64_t row_j = src.capacity_rows > 0 ? p->pop_fill[j] - 1 : 0;
        in[l].rates = src.rates_base + row_j * (int64_t)src.n * B;
        in[l].wt = q.input_wt[l];
        in[l].n_in = src.n;
      }
      rc = riab_feedforward(in, q.n_inputs, q.bias, q.n, 1, B, q.activation, q.act_params, io.rates, q.rates_prime, s);
      if (rc == RIAB_OK && io.spikes) rc = riab_spikes(&io, q.n, s);
      break;
    }
  }
  if (rc) return rc;
  p->launches += (q.kind == RIAB_POP_FF && io.spikes) ? 2 : 1;
  if (noisy) {
    p->launches += spikes ? 2 : 1;
    rc = riab_neuron_noise(q.noise_state, io.rates, nullptr, q.n, B, 1, q.noise_theta_dt, q.noise_sigma_dt, p->seed, p->step,
                           q.io.pop_id, p->agent_id0, s);
    if (rc) return rc;
    if (spikes) {
      io.spikes = spikes;
      rc = riab_spikes(&io, q.n, s);
    }
  }
  return rc;
}

// The two halves of a plan step, for callers that keep the reference's call structure — `Ag.update()` here,
// `N.update()` there (demos/simple_example.ipynb cell 4) — and only want each call to cost one native transition:
// the same kernels, arguments and counters as riab_plan_step, so the same results bit for bit.  No task attached.
extern "C" int riab_plan_step_agent(RiabPlan* p, riab_stream_t stream) {
  if (!p || p->has_task) return RIAB_EINVAL;
  if (p->hist_base && p->hist_fill >= p->hist_cap) return RIAB_EFULL;
  const double* forced = nullptr;
  if (p->forced) {
    if (p->forced_fill >= p->forced_rows) return RIAB_EFULL;
    forced = p->forced + p->forced_fill * 2 * p->B;
  }
  float* row = p->hist_base ? p->hist_base + p->hist_fill * (int64_t)RIAB_HIST_ROWS * p->B : p->row_scratch;
  // The one-launch step: the fused populations' rows of THIS step are written by the agent's launch, ahead of the
  // populations' own calls, which then only move their cursors (same inputs, same rows, same bits).  A row written ahead
  // that nobody claims (a loop that does not update that population after every agent step) is a miss; two in a row
  // leave the population out.
  for (size_t i = 0; i < p->pre_pending.size(); ++i) {
    if (!p->pre_pending[i]) continue;
    p->pre_pending[i] = 0;
    if (++p->pre_misses[i] == 2) p->fused_n = -2;
  }
  if (plan_fused(p) > 0) {
    uint32_t mask = 0u;
    const int rc = fused_agent_step(p, row, (hipStream_t)stream, true, &mask, false);
    if (rc == RIAB_OK) {
      p->step += 1;
      if (p->hist_base) p->hist_fill += 1;
      for (int k = 0; k < p->fused_n; ++k)
        if (mask & (1u << k)) p->pre_pending[p->fused[k]] = 1;
      p->pre_step = p->step;
      return RIAB_OK;
    }
    if (rc != RIAB_EUNSUPPORTED) return rc;  // (no fused population has a free row: the plain agent step)
  }
  const int rc = riab_agent_step(&p->env, &p->motion, p->state, p->B, p->agent_id0, p->drift, nullptr, nullptr, forced, nullptr,
                                 p->seed, p->step, 1, row, p->diag, (hipStream_t)stream);
  if (rc) return rc;
  p->launches += 1;
  if (forced) p->forced_fill += 1;
  p->step += 1;
  if (p->hist_base) p->hist_fill += 1;
  return RIAB_OK;
}

// Something the fused populations read was edited after riab_plan_step_agent wrote their rows ahead (a TaskEnvironment
// reset teleported agents and patched the newest history row): the rows are not claimed — each population's own call
// launches its kernel on the edited row, which overwrites them.  Counted like a miss: a loop that discards every step
// (`env.reset(mask)` after every step) pays the fused work AND the populations' kernels, so two in a row switch the
// one-launch step off for them; one claimed row switches it on again.
extern "C" int riab_plan_discard_ahead(RiabPlan* p) {
  if (!p) return RIAB_EINVAL;
  for (size_t i = 0; i < p->pre_pending.size(); ++i) {
    if (!p->pre_pending[i]) continue;
    p->pre_pending[i] = 0;
    if (++p->pre_misses[i] == 2) p->fused_n = -2;
  }
  return RIAB_OK;
}

// Neurons.update() of population `index` on the agent's newest history row
extern "C" int riab_plan_step_population(RiabPlan* p, int32_t index, riab_stream_t stream) {
  if (!p || p->has_task || index < 0 || index >= (int)p->pops.size()) return RIAB_EINVAL;
  const size_t i = (size_t)index;
  if (p->pops[i].capacity_rows > 0 && p->pop_fill[i] >= p->pops[i].capacity_rows) return RIAB_EFULL;
  if (p->hist_base && p->hist_fill == 0) return RIAB_EINVAL;  // no agent row written into this chunk yet
  if (i < p->pre_pending.size() && p->pre_pending[i] && p->pre_step == p->step) {  // written by this step's riab_plan_step_agent
    p->pre_pending[i] = 0;
    p->pre_misses[i] = 0;
    if (p->pops[i].capacity_rows > 0) p->pop_fill[i] += 1;
    return RIAB_OK;
  }
  const float* row = p->hist_base ? p->hist_base + (p->hist_fill - 1) * (int64_t)RIAB_HIST_ROWS * p->B : p->row_scratch;
  const int rc = launch_population(p, i, row, (hipStream_t)stream);
  if (rc) return rc;
  if (i < p->pre_misses.size() && p->pre_misses[i] >= 2 && ++p->pre_misses[i] >= 2 + 64) {  // (left out: looked at again every 64 updates)
    p->pre_misses[i] = 0;
    p->fused_n = -2;
  }
  if (p->pops[i].capacity_rows > 0) p->pop_fill[i] += 1;
  return RIAB_OK;
}

extern "C" int riab_plan_step(RiabPlan* p, int32_t n_steps, riab_stream_t stream) {
  if (!p || n_steps <= 0) return RIAB_EINVAL;
  if (riab_plan_rows_free(p) < n_steps) return RIAB_EFULL;
  if (p->forced && (p->has_task || p->forced_rows - p->forced_fill < n_steps)) return p->has_task ? RIAB_EINVAL : RIAB_EFULL;
  hipStream_t s = (hipStream_t)stream;
  p->pre_pending.assign(p->pops.size(), 0);
  const int n_fused = plan_fused(p, true);
  for (int32_t k = 0; k < n_steps; ++k) {
    float* row = p->hist_base ? p->hist_base + p->hist_fill * (int64_t)RIAB_HIST_ROWS * p->B : p->row_scratch;
    if (n_fused > 0 && !p->has_task) {  // Agent.update() and the fused populations' update() in one launch, the others after it
      uint32_t mask;
      int rc = fused_agent_step(p, row, s, false, &mask, false);
      if (rc) return rc;
      p->step += 1;
      if (p->hist_base) p->hist_fill += 1;
      for (size_t i = 0; i < p->pops.size(); ++i) {
        if (!is_fused(p, (int)i)) {
          rc = launch_population(p, i, row, s);
          if (rc) return rc;
        }
        if (p->pops[i].capacity_rows > 0) p->pop_fill[i] += 1;
      }
      continue;
    }
    double* pos_x = p->state + (int64_t)RIAB_S_POS_X * p->B;
    double* pos_y = p->state + (int64_t)RIAB_S_POS_Y * p->B;
    int rc;
    const bool scripted = p->has_task && p->scripted_speed > 0.0;
    double* act = const_cast<double*>(p->drift);
    if (p->has_task && p->world) {  // the lanes are the agents of ONE world
      if (scripted) {
        if (!act || !p->motion.has_drift) return RIAB_EINVAL;
        if (!p->action_ready) {  // first step / no auto-reset: later ones get their action from the previous step's reset launch
          rc = riab_task_world_goal_vector(&p->env, &p->task, p->task_state, p->world, pos_x, pos_y, p->task_B,
                                           p->scripted_speed, act, act + p->B, s);
          if (rc) return rc;
          p->launches += 1;
        }
      }
      if (n_fused > 0) {  // Agent.update, the world's step, its reset when the episode ended, the next action and the fused
        p->step += 1;     // populations' update(): ONE kernel (riab_step1.hip, TASK & 8)
        if (p->hist_base) p->hist_fill += 1;
        p->t_env += p->dt_env;
        if (p->auto_reset) p->reset_counter += 1;
        rc = fused_task_step(p, row, s, false);
        if (rc) return rc;
        p->action_ready = scripted;
        for (size_t i = 0; i < p->pops.size(); ++i) {
          if (!is_fused(p, (int)i)) {
            rc = launch_population(p, i, row, s);
            if (rc) return rc;
          }
          if (p->pops[i].capacity_rows > 0) p->pop_fill[i] += 1;
        }
        continue;
      }
      riab::AgentArgs ma;  // Agent.update and the world's step in one launch
      rc = riab::fill_agent_args(ma, &p->env, &p->motion, p->state, p->B, p->agent_id0, p->drift, nullptr, nullptr, nullptr,
                                 p->seed, p->step, 1, row, p->diag);
      if (rc) return rc;
      p->step += 1;
      if (p->hist_base) p->hist_fill += 1;
      p->t_env += p->dt_env;
      rc = riab::launch_motion_world(ma, &p->env, &p->task, p->task_state, p->world, pos_x, pos_y, p->task_B, p->t_env,
                                     p->reward_out, p->terminal_out, p->world_met, p->world_cand, p->world_ctl, p->task_diag, s);
      if (rc) return rc;
      p->launches += 1;
      p->action_ready = false;
      if (p->auto_reset) {  // the caller's `if terminal: env.reset()`, decided on the device, and the next scripted action
        p->reset_counter += 1;
        rc = riab_task_world_reset(&p->env, &p->task, p->task_state, p->world, p->task_B, p->agent_id0, p->t_env, p->n_select,
                                   p->ordered, p->task_seed, p->reset_counter, p->teleport, nullptr, nullptr, pos_x, pos_y,
                                   row + (int64_t)RIAB_H_POS_X * p->B, row + (int64_t)RIAB_H_POS_Y * p->B, p->ep_log,
                                   p->ep_log_cap, p->ep_count, 1, p->scripted_speed, scripted ? act : nullptr,
                                   scripted ? act + p->B : nullptr, p->task_diag, s);
        if (rc) return rc;
        p->launches += 1;
        p->action_ready = scripted;
      }
      for (size_t i = 0; i < p->pops.size(); ++i) {
        rc = launch_population(p, i, row, s);
        if (rc) return rc;
        if (p->pops[i].capacity_rows > 0) p->pop_fill[i] += 1;
      }
      continue;
    }
    if (scripted) {
      if (!act || !p->motion.has_drift) return RIAB_EINVAL;
      if (!p->action_ready) {  // first step: later ones get their action from the previous step's fused task kernel
        rc = riab_task_goal_vector(&p->env, &p->task, p->task_state, pos_x, pos_y, p->task_B, p->scripted_speed, act,
                                   act + p->B, s);
        if (rc) return rc;
        p->launches += 1;
      }
    }
    const bool fused = p->has_task && riab::g_options[RIAB_OPT_FUSED_TASK] != 0;  // (A/B: 0 = motion and task launched separately)
    if (fused) {  // motion + the rest of TaskEnvironment.step (+ the caller's `if terminal: reset()`) in one launch
      riab::AgentArgs ma;
      rc = riab::fill_agent_args(ma, &p->env, &p->motion, p->state, p->B, p->agent_id0, p->drift, nullptr, nullptr, nullptr,
                                 p->seed, p->step, 1, row, p->diag);
      if (rc) return rc;
      p->step += 1;
      if (p->hist_base) p->hist_fill += 1;
      p->t_env += p->dt_env;
      if (p->auto_reset) p->reset_counter += 1;
      if (n_fused > 0) {  // ... and the fused populations' update() as well: the whole closed-loop step is one kernel
        rc = fused_task_step(p, row, s, false);
        if (rc) return rc;
        p->action_ready = scripted;
        for (size_t i = 0; i < p->pops.size(); ++i) {
          if (!is_fused(p, (int)i)) {
            rc = launch_population(p, i, row, s);
            if (rc) return rc;
          }
          if (p->pops[i].capacity_rows > 0) p->pop_fill[i] += 1;
        }
        continue;
      }
      rc = riab::launch_motion_task(ma, &p->env, &p->task, p->task_state, pos_x, pos_y, p->task_B, p->t_env, p->reward_out,
                                    p->terminal_out, p->task_diag, p->auto_reset != 0, p->agent_id0, p->n_select,
                                    p->ordered, p->task_seed, p->reset_counter, p->teleport,
                                    row + (int64_t)RIAB_H_POS_X * p->B, row + (int64_t)RIAB_H_POS_Y * p->B, p->ep_log,
                                    p->ep_log_cap, p->ep_count, p->scripted_speed, scripted ? act : nullptr,
                                    scripted ? act + p->B : nullptr, s);
      if (rc) return rc;
      p->launches += 1;
      p->action_ready = scripted;
    } else {
    const double* forced = p->forced ? p->forced + p->forced_fill * 2 * p->B : nullptr;
    rc = riab_agent_step(&p->env, &p->motion, p->state, p->B, p->agent_id0, p->drift, nullptr, nullptr, forced, nullptr,
                         p->seed, p->step, 1, row, p->diag, s);
    if (rc) return rc;
    p->launches += 1;
    if (forced) p->forced_fill += 1;
    p->step += 1;
    if (p->hist_base) p->hist_fill += 1;
    if (p->has_task) {  // the rest of TaskEnvironment.step (+ the caller's `if terminal: reset()`)
      p->t_env += p->dt_env;
      if (p->auto_reset) p->reset_counter += 1;
      rc = riab::launch_task_fused(&p->env, &p->task, p->task_state, pos_x, pos_y, p->task_B, p->t_env, p->reward_out,
                                   p->terminal_out, p->task_diag, p->auto_reset != 0, p->agent_id0, p->n_select, p->ordered,
                                   p->task_seed, p->reset_counter, p->teleport, row + (int64_t)RIAB_H_POS_X * p->B,
                                   row + (int64_t)RIAB_H_POS_Y * p->B, p->ep_log, p->ep_log_cap, p->ep_count,
                                   p->scripted_speed, scripted ? act : nullptr, scripted ? act + p->B : nullptr, s);
      if (rc) return rc;
      p->launches += 1;
      p->action_ready = scripted;
    }
    }
    for (size_t i = 0; i < p->pops.size(); ++i) {
      rc = launch_population(p, i, row, s);
      if (rc) return rc;
      if (p->pops[i].capacity_rows > 0) p->pop_fill[i] += 1;
    }
  }
  return RIAB_OK;
}
