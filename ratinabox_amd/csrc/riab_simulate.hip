// The open-loop path as one native call: the trajectory kernel and the firing-rate stage run CONCURRENTLY on two
// streams, coupled by flags in device memory (include/riab_hip.h: riab_simulate_fused).  This file is
// host code only: the kernels live in riab_agent_kernel.h (PUB variant: rows written through + published) and
// riab_rates.hip (rate_kernel_gated, stream_gate_kernel, rate_kernel_wide).
//
// Why not one launch: a launch has ONE register and LDS allocation.  The trajectory kernel needs 256 VGPRs and
// 58 KB of LDS per 64 agents; the rate kernel needs 40 VGPRs and no LDS and lives on occupancy.  Why not launches
// per chunk behind HIP events (Agent.simulate's two-stream pipeline): every chunk pays the fill of the first stage
// and a dependent launch boundary, and a 20-step run has nothing to overlap.  With flags the rate stage starts 4
// steps behind the trajectory and stays there.  Up to RIAB_STREAM_POLL_MAX (256) steps the rate stage is ONE
// ordinary (non-persistent) kernel over all rows whose waves each wait for the row they need; beyond that it is one
// rate_kernel_wide launch per chunk of rows behind a one-wave gate kernel that waits for the chunk's last row (a
// kernel with ten thousand waiting workgroups in front of the runnable ones starves them).  A PERSISTENT rate kernel
// was built first (three versions) and removed: waves that stay resident hold store credits and lose 9-12 % of the
// store bandwidth against freshly dispatched ones (tools/stream_bench.hip).
#include <hip/hip_ext.h>

#include <new>
#include <vector>

#include "riab_agent_kernel.h"

namespace riab {
int launch_agent_pub(const AgentArgs& a, hipStream_t s);
int stream_supported(const RiabEnv* env, const RiabPopulation* pop, int64_t B);
int launch_rate_stream(const RiabEnv* env, const RiabPopulation* pop, const float* hist, int64_t B, int32_t T, float dt,
                       uint64_t seed, uint64_t step0, int64_t agent_id0, uint32_t* ctrl, uint32_t spin_limit, hipStream_t s,
                       hipEvent_t ev_start, hipEvent_t ev_stop, bool dry_run);
int launch_stream_gate(uint32_t* ctrl, uint32_t started_target, uint32_t n_traj, uint32_t progress_target,
                       uint32_t spin_limit, bool sleep_long, hipStream_t s);
int launch_rate_rows(const RiabEnv* env, const RiabPopulation* pop, const float* hist, int64_t B, int32_t t0, int32_t tc,
                     float dt, uint64_t seed, uint64_t step0, int64_t agent_id0, hipStream_t s);
}  // namespace riab

struct RiabStreamer {
  hipStream_t side;          // the rate kernel's stream (mode 0)
  hipEvent_t join;           // side -> main
  hipEvent_t t0, t1;         // timing of the rate kernel (created on first use)
  std::vector<hipEvent_t> pairs;  // riab_simulate_pops: (start, stop) around every launch of the timed population
  int n_pairs;               // pairs recorded by the last call (0: t0 / t1 hold the measurement)
  bool timed;
  uint32_t started_total;    // trajectory workgroups launched so far through this object (wraps like the device word)
  int cus;                   // compute units of the device the object was created on
  int gate_mode;             // RIAB_STREAMER_OPT_GATE
  int poll_max;              // RIAB_STREAMER_OPT_POLL_MAX
};

extern "C" RiabStreamer* riab_streamer_create(void) {
  RiabStreamer* h = new (std::nothrow) RiabStreamer();
  if (!h) return nullptr;
  h->side = nullptr;
  h->join = h->t0 = h->t1 = nullptr;
  h->timed = false;
  h->n_pairs = 0;
  h->started_total = 0;
  h->gate_mode = RIAB_GATE_ALWAYS;
  h->poll_max = 256;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
      hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->join, hipEventDisableTiming) != hipSuccess) {
    delete h;
    return nullptr;
  }
  h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return h;
}

extern "C" int riab_streamer_configure(RiabStreamer* h, int32_t option, int32_t value) {
  if (!h) return RIAB_EINVAL;
  switch (option) {
    case RIAB_STREAMER_OPT_GATE:
      if (value != RIAB_GATE_ALWAYS && value != RIAB_GATE_WHEN_BUSY) return RIAB_EINVAL;
      h->gate_mode = value;
      return RIAB_OK;
    case RIAB_STREAMER_OPT_POLL_MAX:
      if (value < 0 || value > 65535) return RIAB_EINVAL;
      h->poll_max = value;
      return RIAB_OK;
    default: return RIAB_EINVAL;
  }
}

extern "C" void riab_streamer_destroy(RiabStreamer* h) {
  if (!h) return;
  if (h->t0) (void)hipEventDestroy(h->t0);
  if (h->t1) (void)hipEventDestroy(h->t1);
  for (hipEvent_t e : h->pairs) (void)hipEventDestroy(e);
  if (h->join) (void)hipEventDestroy(h->join);
  if (h->side) (void)hipStreamDestroy(h->side);
  delete h;
}

extern "C" float riab_streamer_last_rate_ms(RiabStreamer* h) {
  if (!h || !h->timed) return -1.0f;
  float ms = -1.0f;
  if (h->n_pairs > 0) {  // the sum over the timed population's launches of the last riab_simulate_pops call
    float total = 0.0f;
    for (int i = 0; i < h->n_pairs; ++i) {
      if (hipEventElapsedTime(&ms, h->pairs[2 * i], h->pairs[2 * i + 1]) != hipSuccess) return -1.0f;
      total += ms;
    }
    return total;
  }
  if (hipEventElapsedTime(&ms, h->t0, h->t1) != hipSuccess) return -1.0f;
  return ms;
}

extern "C" int riab_simulate_fused(RiabStreamer* h, const RiabEnv* env, const RiabMotion* motion, double* state, int64_t B,
                                   int64_t agent_id0, const double* drift, uint64_t seed, uint64_t step0, int32_t T,
                                   float* hist, int32_t* diag, const RiabPopulation* pop, uint32_t* ctrl,
                                   int32_t timing, riab_stream_t stream) {
  if (!h || !pop || !ctrl || !hist) return RIAB_EINVAL;
  int rc = riab::stream_supported(env, pop, B);
  if (rc) return rc;
  riab::AgentArgs a;
  rc = riab::fill_agent_args(a, env, motion, state, B, agent_id0, drift, nullptr, nullptr, nullptr, seed, step0, T, hist,
                             diag, 64);
  if (rc) return rc;
  a.ctrl = ctrl;
  hipStream_t main_s = (hipStream_t)stream;
  hipStream_t rate_s = h->side;
  // ~0.3 us per poll: a generous second or two before a wait gives up (a healthy wait is tens of microseconds)
  const uint32_t spin_limit = 1u << 20;
  const bool chunks = T > h->poll_max;
  // Every argument check of the rate stage runs BEFORE the trajectory kernel is launched (a dry run of the launch
  // path): a call that returns an argument error has launched nothing and advanced no state.
  if (!chunks) {
    rc = riab::launch_rate_stream(env, pop, hist, B, T, (float)motion->dt, seed, step0, agent_id0, ctrl, spin_limit, rate_s,
                                  nullptr, nullptr, /*dry_run=*/true);
    if (rc) return rc;
  } else if (T > 65535) {
    return RIAB_ETOOBIG;
  }
  if (timing && !h->t0) {
    if (hipEventCreate(&h->t0) != hipSuccess || hipEventCreate(&h->t1) != hipSuccess) return RIAB_EINVAL;
  }
  h->timed = false;
  h->n_pairs = 0;
  // Both kernels must be resident at once or the rate waves spin for nothing: the trajectory kernel is launched first,
  // and a one-wave gate on the side stream holds the rate stage back until every trajectory workgroup of THIS launch
  // has announced itself — the rate waves can then never occupy the slots the kernel they wait for still needs,
  // whatever else is queued on this stream or running on the device (another stream, another process: two ranks
  // sharing one GPU without the gate ran into the waits' time limit).  The gate costs < 1 us of a 20-step region
  // [MI355X: 119.0 vs 118.2 us]; RIAB_GATE_WHEN_BUSY drops it when the caller's stream is idle at the call (for
  // callers that own the device).  (The side stream needs no event to order it behind what is queued on `stream`:
  // the words it waits for are only written by this launch's trajectory kernel.)
  const bool gate = h->gate_mode == RIAB_GATE_ALWAYS || hipStreamQuery(main_s) != hipSuccess;
  rc = riab::launch_agent_pub(a, main_s);
  if (rc) return rc;
  const uint32_t n_traj = (uint32_t)(B / 64);
  h->started_total += n_traj;
  // From here on the state has advanced: a later failure still joins the side stream and is reported as
  // RIAB_EPARTIAL (the trajectory rows are complete, the rates of this call are not).
  // Two forms of the rate stage:
  //  * up to `poll_max` (256) steps: ONE rate kernel for all rows whose waves wait for their rows themselves
  //    (rate_kernel_gated): it follows the trajectory at a distance of one block of steps, which is what a short
  //    run needs; its per-wave poll costs ~15 % of the store bandwidth;
  //  * longer runs: the population's ordinary kernel per chunk of rows, each chunk behind a progress gate (one
  //    wave) on the same stream — full store bandwidth, at the price of a chunk of distance to the trajectory
  //    (the first chunks are short) and a gate + launch boundary (~5 us) per chunk.  (A progress gate also waits
  //    for the started count: no separate started gate.)
  int fail = RIAB_OK;
  if (!chunks) {
    // (the started gate is one sleeping wave; it may have to sit out whatever was queued on `stream`: ~1 min)
    if (gate) fail = riab::launch_stream_gate(ctrl, h->started_total, 0, 0, 1u << 24, true, rate_s);
    // (timed through the launch's own start / stop events: the kernel's duration as rocprofv3 reports it)
    if (!fail)
      fail = riab::launch_rate_stream(env, pop, hist, B, T, (float)motion->dt, seed, step0, agent_id0, ctrl, spin_limit,
                                      rate_s, timing ? h->t0 : nullptr, timing ? h->t1 : nullptr, false);
    if (!fail && timing) h->timed = true;
  } else {
    if (timing) (void)hipEventRecord(h->t0, rate_s);
    const bool box_room = !env->polygon && !env->hole_mask && !env->periodic && env->n_walls >= 4;
    int32_t t0 = 0, k = 0;
    while (t0 < T && !fail) {
      // A chunk should be finished by the trajectory when the stream gets to its gate.  In a solid rectangular room
      // the trajectory pulls away from the rate kernels (which need ~2.7 us per row) and a chunk may be ~1.5x its
      // predecessor; in other rooms ~1.25x is the most (a 16, 16, 32, 64, 128 ramp then spent 230 us of a 1024-step
      // run inside the gates [MI355X, rocprofv3 trace]).  Larger chunks are cheaper per row (3.6 us at 16 rows, 2.7 at
      // 128) and every chunk costs a gate (~5 us).
      static const int32_t ramp_fast[5] = {16, 28, 44, 64, 96};
      static const int32_t ramp_slow[10] = {16, 16, 20, 24, 32, 40, 48, 64, 80, 96};
      int32_t tc = box_room ? (k < 5 ? ramp_fast[k] : 128) : (k < 10 ? ramp_slow[k] : 128);
      if (tc > T - t0 || T - t0 - tc < 32) tc = T - t0;  // (no sliver at the end)
      // ~0.5 us per poll: seconds before a gate gives up (a healthy wait is one chunk of trajectory, < 1 ms); the
      // first gate may also have to sit out what is queued in front of the trajectory kernel
      fail = riab::launch_stream_gate(ctrl, h->started_total, n_traj, (uint32_t)step0 + (uint32_t)(t0 + tc),
                                      k == 0 ? 1u << 24 : 1u << 22, false, rate_s);
      if (!fail) fail = riab::launch_rate_rows(env, pop, hist, B, t0, tc, (float)motion->dt, seed, step0, agent_id0, rate_s);
      t0 += tc;
      ++k;
    }
    if (!fail && timing) {  // (the whole stage: gates and chunk kernels)
      (void)hipEventRecord(h->t1, rate_s);
      h->timed = true;
    }
  }
  hipError_t e = hipEventRecord(h->join, h->side);
  if (e == hipSuccess) e = hipStreamWaitEvent(main_s, h->join, 0);
  if (fail) return RIAB_EPARTIAL;
  if (e != hipSuccess) return (int)e;
  return RIAB_OK;
}


// ---- any set of populations: the chunked form of the rate stage for all of them, one native call ----------------
// rows [t0, t0 + tc) of population i from the trajectory rows of the same chunk (the T-row form of riab_plan.hip's
// launch_population: same entry points, same arguments as `tc` calls of Neurons.update on successive rows)
static int launch_pop_rows(const RiabEnv* env, const RiabPopulation* pops, int i, const float* hist, int64_t B, int32_t t0,
                           int32_t tc, float dt, uint64_t seed, uint64_t step0, int64_t agent_id0, hipStream_t s) {
  const RiabPopulation& q = pops[i];
  RiabRateIO io = q.io;
  const float* row = hist + (int64_t)t0 * RIAB_HIST_ROWS * B;
  io.pos_x = row + (int64_t)RIAB_H_POS_X * B;
  io.pos_y = row + (int64_t)RIAB_H_POS_Y * B;
  io.hd_x = row + (int64_t)RIAB_H_HD_X * B;
  io.hd_y = row + (int64_t)RIAB_H_HD_Y * B;
  io.pos_ld = (int64_t)RIAB_HIST_ROWS * B;
  io.T = tc;
  io.B = B;
  io.rates = q.rates_base + (int64_t)t0 * q.n * B;
  io.spikes = q.spikes_base ? q.spikes_base + (int64_t)t0 * q.n * B : nullptr;
  io.u_in = nullptr;
  io.dt = dt;
  io.seed = seed;
  io.step0 = step0 + 1 + (uint64_t)t0;  // Neurons.update after the (step0 + t + 1)-th Agent.update
  io.agent_id0 = agent_id0;
  const bool noisy = q.noise_state != nullptr;
  uint8_t* const spikes = io.spikes;
  if (noisy) io.spikes = nullptr;  // spikes are drawn on the final rates, after the noise pass
  int rc = RIAB_EUNSUPPORTED;
  switch (q.kind) {
    case RIAB_POP_PLACE:
      rc = riab_place_cells(env, &io, q.table, q.n, q.description, q.geometry, q.top_hat_width, s);
      break;
    case RIAB_POP_GRID: rc = riab_grid_cells(&io, q.table, q.n, q.description, q.f0, s); break;
    case RIAB_POP_HDC: rc = riab_head_direction_cells(&io, q.table, q.n, s); break;
    case RIAB_POP_SPEED:  // history["vel"]: the measured velocity rows
      io.hd_x = row + (int64_t)RIAB_H_VEL_X * B;
      io.hd_y = row + (int64_t)RIAB_H_VEL_Y * B;
      rc = riab_speed_cell(&io, q.one_sigma_speed, s);
      break;
    case RIAB_POP_RANDOM_SPATIAL:
      rc = riab_random_spatial_neurons(env, &io, q.table, q.n_anchors, q.targets, q.n, q.geometry, s);
      break;
    case RIAB_POP_BVC:
      rc = riab_boundary_vector_cells_windowed(env, &io, q.test_dirs, q.ray_rden, q.K, q.table, q.vm_table, q.inv_norm, q.n,
                                               q.egocentric, nullptr, q.cell_rows, q.windows, s);
      break;
    case RIAB_POP_OVC:
      rc = riab_object_vector_cells(env, &io, q.objects, q.object_types, q.n_objects, q.table, q.n, q.walls_occlude,
                                    q.egocentric, s);
      break;
    case RIAB_POP_FF: {
      RiabFFInput in[RIAB_FF_MAX_INPUTS];
      for (int l = 0; l < q.n_inputs; ++l) {
        const RiabPopulation& src = pops[q.input_index[l]];  // (an earlier population: its rows of this chunk exist)
        in[l].rates = src.rates_base + (int64_t)t0 * src.n * B;
        in[l].wt = q.input_wt[l];
        in[l].n_in = src.n;
      }
      rc = riab_feedforward(in, q.n_inputs, q.bias, q.n, tc, B, q.activation, q.act_params, io.rates, nullptr, s);
      if (rc == RIAB_OK && io.spikes) rc = riab_spikes(&io, q.n, s);
      break;
    }
    default: break;  // (velocity cells read the float64 state: they advance through a step plan)
  }
  if (rc) return rc;
  if (noisy) {
    rc = riab_neuron_noise(q.noise_state, io.rates, nullptr, q.n, B, tc, q.noise_theta_dt, q.noise_sigma_dt, seed,
                           step0 + 1 + (uint64_t)t0, q.io.pop_id, agent_id0, s);
    if (rc) return rc;
    if (spikes) {
      io.spikes = spikes;
      rc = riab_spikes(&io, q.n, s);
    }
  }
  return rc;
}

extern "C" int riab_simulate_pops(RiabStreamer* h, const RiabEnv* env, const RiabMotion* motion, double* state, int64_t B,
                                  int64_t agent_id0, const double* drift, uint64_t seed, uint64_t step0, int32_t T,
                                  float* hist, int32_t* diag, const RiabPopulation* pops, int32_t n_pops, uint32_t* ctrl,
                                  int32_t timed_pop, riab_stream_t stream) {
  if (!h || !env || !pops || !ctrl || !hist || n_pops <= 0 || T <= 0) return RIAB_EINVAL;
  if (B <= 0 || B % 64 != 0) return RIAB_EUNSUPPORTED;  // (the publishing trajectory kernel runs whole waves)
  for (int i = 0; i < n_pops; ++i) {
    const RiabPopulation& q = pops[i];
    if (q.n <= 0 || !q.rates_base || q.capacity_rows < T) return RIAB_EINVAL;
    switch (q.kind) {
      case RIAB_POP_PLACE: case RIAB_POP_GRID: case RIAB_POP_HDC: case RIAB_POP_SPEED: case RIAB_POP_RANDOM_SPATIAL:
      case RIAB_POP_BVC: case RIAB_POP_OVC: break;
      case RIAB_POP_FF:
        if (q.n_inputs <= 0 || q.n_inputs > RIAB_FF_MAX_INPUTS) return RIAB_EINVAL;
        for (int l = 0; l < q.n_inputs; ++l)
          if (q.input_index[l] < 0 || q.input_index[l] >= i || pops[q.input_index[l]].capacity_rows < T) return RIAB_EINVAL;
        break;
      default: return RIAB_EUNSUPPORTED;
    }
  }
  riab::AgentArgs a;
  int rc = riab::fill_agent_args(a, env, motion, state, B, agent_id0, drift, nullptr, nullptr, nullptr, seed, step0, T, hist,
                                 diag, 64);
  if (rc) return rc;
  a.ctrl = ctrl;
  hipStream_t main_s = (hipStream_t)stream, rate_s = h->side;
  // chunk schedule of riab_simulate_fused's long runs (the trajectory is the faster stage here too: every population
  // kernel of a chunk runs before the next chunk's gate is looked at)
  const bool box_room = !env->polygon && !env->hole_mask && !env->periodic && env->n_walls >= 4;
  static const int32_t ramp_fast[5] = {16, 28, 44, 64, 96};
  static const int32_t ramp_slow[10] = {16, 16, 20, 24, 32, 40, 48, 64, 80, 96};
  std::vector<int32_t> sched;
  for (int32_t t0 = 0, k = 0; t0 < T; ++k) {
    int32_t tc = box_room ? (k < 5 ? ramp_fast[k] : 128) : (k < 10 ? ramp_slow[k] : 128);
    if (tc > T - t0 || T - t0 - tc < 32) tc = T - t0;
    sched.push_back(tc);
    t0 += tc;
  }
  const bool timing = timed_pop >= 0 && timed_pop < n_pops;
  if (timing) {
    while (h->pairs.size() < 2 * sched.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return RIAB_EINVAL;
      h->pairs.push_back(e);
    }
  }
  h->timed = false;
  h->n_pairs = 0;
  rc = riab::launch_agent_pub(a, main_s);
  if (rc) return rc;
  const uint32_t n_traj = (uint32_t)(B / 64);
  h->started_total += n_traj;
  // (from here on the state has advanced: a later failure still joins the side stream and returns RIAB_EPARTIAL.
  // The first progress gate also waits until every trajectory workgroup of this launch is resident — it may have to
  // sit out what is queued in front of the trajectory kernel — so there is no separate started gate.)
  int fail = RIAB_OK;
  int32_t t0 = 0;
  for (size_t k = 0; k < sched.size() && !fail; ++k) {
    const int32_t tc = sched[k];
    fail = riab::launch_stream_gate(ctrl, h->started_total, n_traj, (uint32_t)step0 + (uint32_t)(t0 + tc),
                                    k == 0 ? 1u << 24 : 1u << 22, false, rate_s);
    for (int i = 0; i < n_pops && !fail; ++i) {
      if (timing && i == timed_pop) (void)hipEventRecord(h->pairs[2 * k], rate_s);
      fail = launch_pop_rows(env, pops, i, hist, B, t0, tc, (float)motion->dt, seed, step0, agent_id0, rate_s);
      if (timing && i == timed_pop) (void)hipEventRecord(h->pairs[2 * k + 1], rate_s);
    }
    t0 += tc;
  }
  if (!fail && timing) {
    h->n_pairs = (int)sched.size();
    h->timed = true;
  }
  hipError_t e = hipEventRecord(h->join, h->side);
  if (e == hipSuccess) e = hipStreamWaitEvent(main_s, h->join, 0);
  if (fail) return RIAB_EPARTIAL;
  if (e != hipSuccess) return (int)e;
  return RIAB_OK;
}

extern "C" int riab_host_wait_spin(int32_t on) {
  // hipDeviceScheduleSpin: the host thread spins on the completion signal in hipDeviceSynchronize / hipStreamSynchronize
  // instead of blocking in the kernel driver after ~100 us of active waiting (the default): a region of a few kernels
  // ends 10-20 us sooner from the host's point of view, at the price of a busy core while it waits.
  const hipError_t e = hipSetDeviceFlags(on ? hipDeviceScheduleSpin : hipDeviceScheduleAuto);
  return e == hipSuccess ? RIAB_OK : (int)e;
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
