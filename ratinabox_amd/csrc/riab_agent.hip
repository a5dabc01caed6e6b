// Agent.update on gfx950: the C ABI entry point of the motion kernel (riab_agent_kernel.h) and the step
// plan's fused motion + task launch.
#include "riab_agent_kernel.h"
#include "riab_traj4_kernel.h"
#include "riab_task_kernel.h"  // (after the motion kernel: this header turns fp contraction off for its own code)

namespace riab {

// One closed-loop step of a TaskEnvironment in ONE launch: Agent.update for this lane (a one-step,
// single-wave motion launch), then — the lane reads back the position it has just written — the task
// kernel's step + auto-reset + next scripted action.  Both are the device bodies of the stand-alone
// kernels; what is saved is one dependent dispatch per step (the closed loop is bound by dispatch
// latency, not by work).
template <int MODE>
__global__ __launch_bounds__(64) void motion_task_kernel(const AgentArgs ma, const TaskArgs a, const ResetArgs r,
                                                         const double* pos_x, const double* pos_y, double t_env,
                                                         double* reward_out, uint8_t* terminal_out, double gv_scale,
                                                         double* gv_x, double* gv_y, int32_t* diag) {
  agent_step_body<double, 0, false>(ma);
  task_body<MODE>(a, r, pos_x, pos_y, t_env, reward_out, terminal_out, nullptr, gv_scale, gv_x, gv_y, diag);
}

int launch_motion_task(const AgentArgs& ma, const RiabEnv* env, const RiabTask* task, double* task_state, double* pos_x,
                       double* pos_y, int64_t task_B, double t_env, double* reward_out, uint8_t* terminal_out,
                       int32_t* diag, bool auto_reset, int64_t agent_id0, int32_t n_select, int32_t ordered, uint64_t seed,
                       uint64_t counter, int32_t teleport, float* hist_x, float* hist_y, double* ep_log,
                       int64_t ep_log_cap, int32_t* ep_count, double gv_scale, double* gv_x, double* gv_y, hipStream_t s) {
  TaskArgs a;
  int rc = fill_args(a, env, task, task_state, task_B);
  if (rc) return rc;
  if (!pos_x || !pos_y || !reward_out || !terminal_out || !diag) return RIAB_EINVAL;
  ResetArgs r = {};
  if (auto_reset) {
    rc = fill_reset(r, env, agent_id0, n_select, ordered, seed, counter, teleport, nullptr, nullptr, pos_x, pos_y, hist_x,
                    hist_y, ep_log, ep_log_cap, ep_count);
    if (rc) return rc;
  }
  const dim3 grid((unsigned)((ma.B + 63) / 64)), block(64);  // the motion batch is the (padded) larger one
  const bool gv = gv_x != nullptr;
#define RIAB_MT_LAUNCH(MODE)                                                                                 \
  hipLaunchKernelGGL(motion_task_kernel<MODE>, grid, block, 0, s, ma, a, r, pos_x, pos_y, t_env, reward_out, \
                     terminal_out, gv_scale, gv_x, gv_y, diag)
  if (auto_reset && gv) RIAB_MT_LAUNCH(7);
  else if (auto_reset) RIAB_MT_LAUNCH(3);
  else if (gv) RIAB_MT_LAUNCH(5);
  else RIAB_MT_LAUNCH(1);
#undef RIAB_MT_LAUNCH
  return (int)hipGetLastError();
}

// The publishing variant of the trajectory kernel (riab_simulate_*): float64, four waves per 64 agents
// (riab_traj4_kernel.h), Philox noise or explicit normals (a.z_in); a.ctrl carries the control words.  Whole waves
// only (B % 64 == 0), any T.  RIAB_OPT_TRAJ_KERNEL = 2 (A/B comparisons): the two-wave kernel of round 1 (Philox only).
// `*state_published` (optional): true when the kernel launched makes its LAST publication after its state stores have
// been written through and acknowledged (the four-wave kernel): a consumer that has waited for that publication needs
// no other ordering against this launch.
int launch_agent_pub(const AgentArgs& a_in, hipStream_t s, bool* state_published) {
  if (!a_in.ctrl || !a_in.hist || a_in.forced || a_in.B % 4 != 0) return RIAB_EINVAL;
  AgentArgs a = a_in;
  a.pub_single = g_options[RIAB_OPT_PUB_SINGLE_ROWS];
  const dim3 grid((unsigned)((a.B + 63) / 64));
  const bool two_wave = g_options[RIAB_OPT_TRAJ_KERNEL] == 2 && !a.z_in && !a.z_out && a.B % 64 == 0;
  if (state_published) *state_published = !two_wave;
  if (two_wave) hipLaunchKernelGGL((agent_step_kernel<double, 0, true, true>), grid, dim3(128), 0, s, a);
  else if (a.z_in) hipLaunchKernelGGL((traj4_kernel<1, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((traj4_kernel<0, true>), grid, dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// registers per lane the publishing trajectory kernel holds (the larger of its two instantiations, as allocated: in
// units of 8), asked of the code object once: what the reserving shape of the rate kernel has to leave free on a SIMD
// next to its own six waves (riab_rates.hip launch_stream_cell)
int traj_kernel_regs() {
  static int regs = 0;
  if (!regs) {
    int worst = 0;
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, (const void*)traj4_kernel<0, true>) == hipSuccess && attr.numRegs > worst) worst = attr.numRegs;
    if (hipFuncGetAttributes(&attr, (const void*)traj4_kernel<1, true>) == hipSuccess && attr.numRegs > worst) worst = attr.numRegs;
    (void)hipGetLastError();
    regs = worst > 0 ? (worst + 7) / 8 * 8 : 512;   // (unknown: nothing can be promised)
  }
  return regs;
}

// the forced-position trajectory (Agent.import_trajectory / forced_next_position) of riab_simulate: the single-wave
// kernel in forced mode, plain stores (what follows it on the stream is ordered by the stream)
int launch_agent_plain(const AgentArgs& a, hipStream_t s);

int launch_agent_forced(const AgentArgs& a, hipStream_t s) {
  if (!a.forced || !a.hist) return RIAB_EINVAL;
  const dim3 grid((unsigned)((a.B + 63) / 64));
  hipLaunchKernelGGL((agent_step_kernel<double, 2, false>), grid, dim3(64), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace riab

using namespace riab;

extern "C" int riab_agent_step(const RiabEnv* env, const RiabMotion* motion, double* state, int64_t B,
                               int64_t agent_id0, const double* drift, const double* z_in, double* z_out,
                               const double* forced_pos, const double* resample_pos, uint64_t seed, uint64_t step0,
                               int32_t T, float* hist, int32_t* diag, riab_stream_t stream) {
  AgentArgs a;
  const int rc = fill_agent_args(a, env, motion, state, B, agent_id0, drift, z_in, z_out, forced_pos, seed, step0, T, hist,
                                 diag, resample_pos);
  if (rc) return rc;
  return riab::launch_agent_plain(a, (hipStream_t)stream);
}

// the trajectory kernel that fits the launch (no publication: what follows on the stream is ordered by the stream):
// riab_agent_step, and riab_simulate for an agent without populations
int riab::launch_agent_plain(const AgentArgs& a, hipStream_t s) {
  {
  const int64_t B = a.B;
  const int32_t T = a.T;
  const double* const z_out = a.z_out;
  const dim3 grid((unsigned)((B + 63) / 64));
  const int in = a.forced ? 2 : (a.z_in ? 1 : 0);
  // (A/B comparisons, riab_set_option(RIAB_OPT_TRAJ_KERNEL): 1 the single-wave kernel for every launch, 2 round 1's
  // two-wave kernel)
  const bool no_pc = g_options[RIAB_OPT_TRAJ_KERNEL] == 1, two_wave = g_options[RIAB_OPT_TRAJ_KERNEL] == 2;
  const bool pc = in == 0 && T >= 2 * RIAB_Z_BATCH && B % 64 == 0 && !z_out && !no_pc && two_wave;
  // multi-step launches: one agent's step over four specialised waves (riab_traj4_kernel.h)
  const bool t4 = in != 2 && T >= 8 && B % 4 == 0 && !no_pc && !two_wave;
  if (t4 && in == 0) hipLaunchKernelGGL((traj4_kernel<0, false>), grid, dim3(256), 0, s, a);
  else if (t4) hipLaunchKernelGGL((traj4_kernel<1, false>), grid, dim3(256), 0, s, a);
  else if (pc) hipLaunchKernelGGL((agent_step_kernel<double, 0, true>), grid, dim3(128), 0, s, a);
  else if (in == 0) hipLaunchKernelGGL((agent_step_kernel<double, 0, false>), grid, dim3(64), 0, s, a);
  else if (in == 1) hipLaunchKernelGGL((agent_step_kernel<double, 1, false>), grid, dim3(64), 0, s, a);
  else hipLaunchKernelGGL((agent_step_kernel<double, 2, false>), grid, dim3(64), 0, s, a);
  }
  return (int)hipGetLastError();
}
