// The open-loop path as one native call: the trajectory kernel and a persistent firing-rate kernel coupled by
// flags in device memory (include/riab_hip.h: riab_simulate_fused).  This file is host code only: the kernels live
// in riab_agent_kernel.h (PUB variant) and riab_rates.hip (rate_stream_kernel, stream_gate_kernel).
//
// Why not one launch: a launch has ONE register and LDS allocation.  The trajectory kernel needs 256 VGPRs and
// 58 KB of LDS per 64 agents; the rate kernel needs 40 VGPRs and no LDS and lives on occupancy.  Why not launches
// per chunk (Agent.simulate's two-stream pipeline): every chunk pays the fill of the first stage and a dependent
// launch boundary, and a 20-step run has nothing to overlap.  With flags the rate waves start 4 steps behind the
// trajectory and stay there.
#include <hip/hip_ext.h>

#include <new>

#include "riab_agent_kernel.h"

namespace riab {
int launch_agent_pub(const AgentArgs& a, hipStream_t s);
int stream_supported(const RiabEnv* env, const RiabPopulation* pop, int64_t B);
int launch_rate_stream(const RiabEnv* env, const RiabPopulation* pop, const float* hist, int64_t B, int32_t T, float dt,
                       uint64_t seed, uint64_t step0, int64_t agent_id0, uint32_t* ctrl, int max_wgs, int gpi,
                       uint32_t spin_limit, bool any_order, hipStream_t s);
int launch_stream_gate(uint32_t* ctrl, uint32_t target, uint32_t spin_limit, hipStream_t s);
}  // namespace riab

struct RiabStreamer {
  hipStream_t side;          // the rate kernel's stream (mode 0)
  hipEvent_t fork, join;     // main -> side, side -> main
  hipEvent_t t0, t1;         // timing of the rate kernel (created on first use)
  bool timed;
  uint32_t started_total;    // trajectory workgroups launched so far through this object (wraps like the device word)
  int cus;                   // compute units of the device the object was created on
};

extern "C" RiabStreamer* riab_streamer_create(void) {
  RiabStreamer* h = new (std::nothrow) RiabStreamer();
  if (!h) return nullptr;
  h->side = nullptr;
  h->fork = h->join = h->t0 = h->t1 = nullptr;
  h->timed = false;
  h->started_total = 0;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
      hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->join, hipEventDisableTiming) != hipSuccess) {
    delete h;
    return nullptr;
  }
  h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return h;
}

extern "C" void riab_streamer_destroy(RiabStreamer* h) {
  if (!h) return;
  if (h->t0) (void)hipEventDestroy(h->t0);
  if (h->t1) (void)hipEventDestroy(h->t1);
  if (h->fork) (void)hipEventDestroy(h->fork);
  if (h->join) (void)hipEventDestroy(h->join);
  if (h->side) (void)hipStreamDestroy(h->side);
  delete h;
}

extern "C" float riab_streamer_last_rate_ms(RiabStreamer* h) {
  if (!h || !h->timed) return -1.0f;
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, h->t0, h->t1) != hipSuccess) return -1.0f;
  return ms;
}

extern "C" int riab_simulate_fused(RiabStreamer* h, const RiabEnv* env, const RiabMotion* motion, double* state, int64_t B,
                                   int64_t agent_id0, const double* drift, uint64_t seed, uint64_t step0, int32_t T,
                                   float* hist, int32_t* diag, const RiabPopulation* pop, uint32_t* ctrl,
                                   int32_t wgs_per_cu, int32_t mode, int32_t timing, riab_stream_t stream) {
  if (!h || !pop || !ctrl || !hist) return RIAB_EINVAL;
  if (mode != 0 && mode != 1) return RIAB_EINVAL;
  int rc = riab::stream_supported(env, pop, B);
  if (rc) return rc;
  riab::AgentArgs a;
  rc = riab::fill_agent_args(a, env, motion, state, B, agent_id0, drift, nullptr, nullptr, nullptr, seed, step0, T, hist,
                             diag, 64);
  if (rc) return rc;
  a.ctrl = ctrl;
  hipStream_t main_s = (hipStream_t)stream;
  const bool any_order = mode == 1;
  hipStream_t rate_s = any_order ? main_s : h->side;
  // ~0.3 us per poll: a generous second or two before a wait gives up (a healthy wait is tens of microseconds)
  const uint32_t spin_limit = 4u << 20;
  if (wgs_per_cu <= 0) wgs_per_cu = 7;
  if (wgs_per_cu > 8) wgs_per_cu = 8;
  int gpi = 2;
  if (const char* e = getenv("RIAB_STREAM_GPI")) gpi = atoi(e);
  if (timing && !h->t0) {
    if (hipEventCreate(&h->t0) != hipSuccess || hipEventCreate(&h->t1) != hipSuccess) return RIAB_EINVAL;
  }
  h->timed = false;
  if (!any_order) {
    hipError_t e = hipEventRecord(h->fork, main_s);  // the side stream starts after everything queued on `stream`
    if (e == hipSuccess) e = hipStreamWaitEvent(h->side, h->fork, 0);
    if (e != hipSuccess) return (int)e;
  }
  rc = riab::launch_agent_pub(a, main_s);
  if (rc) return rc;
  const uint32_t n_traj = (uint32_t)(B / 64);
  h->started_total += n_traj;
  if (!any_order) {
    rc = riab::launch_stream_gate(ctrl, h->started_total, spin_limit, rate_s);
    if (rc) return rc;
  }
  if (timing) (void)hipEventRecord(h->t0, rate_s);
  rc = riab::launch_rate_stream(env, pop, hist, B, T, (float)motion->dt, seed, step0, agent_id0, ctrl, wgs_per_cu * h->cus,
                                gpi, spin_limit, any_order, rate_s);
  if (rc) return rc;
  if (timing) {
    (void)hipEventRecord(h->t1, rate_s);
    h->timed = true;
  }
  if (!any_order) {
    hipError_t e = hipEventRecord(h->join, h->side);
    if (e == hipSuccess) e = hipStreamWaitEvent(main_s, h->join, 0);
    if (e != hipSuccess) return (int)e;
  }
  return RIAB_OK;
}

extern "C" int64_t riab_abi_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(RiabEnv);
    case 1: return (int64_t)sizeof(RiabMotion);
    case 2: return (int64_t)sizeof(RiabRateIO);
    case 3: return (int64_t)sizeof(RiabPopulation);
    case 4: return (int64_t)sizeof(RiabTask);
    case 5: return (int64_t)sizeof(RiabFFInput);
    case 6: return (int64_t)RIAB_TS_ROWS;
    default: return -1;
  }
}
