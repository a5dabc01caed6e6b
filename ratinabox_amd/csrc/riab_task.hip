// Batched TaskEnvironment on gfx950: the C ABI entry points of the task kernel (riab_task_kernel.h).
#include "riab_task_kernel.h"

namespace riab {

// The step plan's fused launch: TaskEnvironment.step, reset of the lanes that became terminal (when
// auto_reset) and the scripted action of the next step (when gv_x is given) in one kernel.
int launch_task_fused(const RiabEnv* env, const RiabTask* task, double* task_state, double* pos_x, double* pos_y, int64_t B,
                      double t_env, double* reward_out, uint8_t* terminal_out, int32_t* diag, bool auto_reset,
                      int64_t agent_id0, int32_t n_select, int32_t ordered, uint64_t seed, uint64_t counter,
                      int32_t teleport, float* hist_x, float* hist_y, double* ep_log, int64_t ep_log_cap,
                      int32_t* ep_count, double gv_scale, double* gv_x, double* gv_y, hipStream_t s) {
  TaskArgs a;
  int rc = fill_args(a, env, task, task_state, B);
  if (rc) return rc;
  if (!pos_x || !pos_y || !reward_out || !terminal_out || !diag) return RIAB_EINVAL;
  ResetArgs r = {};
  if (auto_reset) {
    rc = fill_reset(r, env, agent_id0, n_select, ordered, seed, counter, teleport, nullptr, nullptr, pos_x, pos_y, hist_x,
                    hist_y, ep_log, ep_log_cap, ep_count);
    if (rc) return rc;
  }
  const dim3 grid((unsigned)((B + 63) / 64)), block(64);
  const bool gv = gv_x != nullptr;
#define RIAB_TASK_LAUNCH(MODE)                                                                                  \
  hipLaunchKernelGGL(task_kernel<MODE>, grid, block, 0, s, a, r, pos_x, pos_y, t_env, reward_out, terminal_out, \
                     (const uint8_t*)nullptr, gv_scale, gv_x, gv_y, diag)
  if (auto_reset && gv) RIAB_TASK_LAUNCH(7);
  else if (auto_reset) RIAB_TASK_LAUNCH(3);
  else if (gv) RIAB_TASK_LAUNCH(5);
  else RIAB_TASK_LAUNCH(1);
#undef RIAB_TASK_LAUNCH
  return (int)hipGetLastError();
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
  const ResetArgs r = {};
  hipLaunchKernelGGL(task_kernel<1>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a, r, pos_x, pos_y,
                     t_env, reward_out, terminal_out, (const uint8_t*)nullptr, 0.0, (double*)nullptr, (double*)nullptr, diag);
  return (int)hipGetLastError();
}

extern "C" int riab_task_goal_vector(const RiabEnv* env, const RiabTask* task, double* task_state, const double* pos_x,
                                     const double* pos_y, int64_t B, double scale, double* out_x, double* out_y,
                                     riab_stream_t stream) {
  TaskArgs a;
  const int rc = fill_args(a, env, task, task_state, B);
  if (rc) return rc;
  if (!pos_x || !pos_y || !out_x || !out_y) return RIAB_EINVAL;
  const ResetArgs r = {};
  hipLaunchKernelGGL(task_kernel<4>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a, r, pos_x, pos_y,
                     0.0, (double*)nullptr, (uint8_t*)nullptr, (const uint8_t*)nullptr, scale, out_x, out_y,
                     (int32_t*)nullptr);
  return (int)hipGetLastError();
}

extern "C" int riab_task_reset(const RiabEnv* env, const RiabTask* task, double* task_state, const uint8_t* mask, int64_t B,
                               int64_t agent_id0, double t_env, int32_t n_select, int32_t ordered, uint64_t seed,
                               uint64_t counter, int32_t teleport, const double* new_x, const double* new_y,
                               double* pos_x, double* pos_y, float* hist_x, float* hist_y, double* ep_log,
                               int64_t ep_log_cap, int32_t* ep_count, int32_t* diag, riab_stream_t stream) {
  TaskArgs a;
  int rc = fill_args(a, env, task, task_state, B);
  if (rc) return rc;
  if (!diag) return RIAB_EINVAL;
  ResetArgs r;
  rc = fill_reset(r, env, agent_id0, n_select, ordered, seed, counter, teleport, new_x, new_y, pos_x, pos_y, hist_x, hist_y,
                  ep_log, ep_log_cap, ep_count);
  if (rc) return rc;
  // (positions are only read when teleporting writes them: any valid per-lane array does for the loads)
  const double* px = pos_x ? pos_x : task_state;
  const double* py = pos_y ? pos_y : task_state;
  hipLaunchKernelGGL(task_kernel<2>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a, r, px, py, t_env,
                     (double*)nullptr, (uint8_t*)nullptr, mask, 0.0, (double*)nullptr, (double*)nullptr, diag);
  return (int)hipGetLastError();
}
