// Firing-rate kernels for gfx950 (MI355X): PlaceCells, GridCells,
// HeadDirectionCells, the Poisson-spike epilogue, neuron noise and the
// streaming-store calibration kernel.
//
// Shape of the problem: out[t][c][b] = f(cell c, position of agent b at step t).
// No contraction (nothing for MFMA); the only large traffic is the write of
// `rates` (4 B per (cell, agent-step)) so the kernels are HBM-write bound.
// Mapping (wide kernel, B >= 1024 agents per time row):
//   * grid = (1024-agent segments, 4-cell groups, time rows), x fastest: consecutive
//     workgroups write consecutive addresses of out[t][c][b], so the chip-wide store stream
//     walks HBM in address order (measured on MI355X: 6.1 TB/s vs 4.6-5.0 TB/s for a
//     per-lane loop striding over many cell rows; tools/store_bench.hip);
//   * a lane owns FOUR consecutive agents (one float4 of x and of y, loaded once) and
//     CPB = 4 cells; every store is a 16-B float4 = 1 KiB contiguous per wave-instruction;
//   * the cell group's parameters (AoS table [n][NP]) are fetched by ONE coalesced load
//     (lane i takes the i-th float of the group) and broadcast with v_readlane_b32 into
//     SGPRs: no LDS, no per-lane table traffic;
// small batches (B < 1024, e.g. one agent over a long trajectory) use the generic kernel:
// lanes are quads of agents flattened over all time rows, each walking a chunk of cells.
#include <hip/hip_ext.h>

#include <type_traits>

#include "riab_device.h"
#include "riab_rate_cells.h"
// how the spike bytes of the open-loop kernels are stored (riab_device.h: store_stream)
#ifndef RIAB_SPIKE_POLICY_GATED
#define RIAB_SPIKE_POLICY_GATED RIAB_STORE_NT
#endif
#ifndef RIAB_SPIKE_POLICY_WIDE
#define RIAB_SPIKE_POLICY_WIDE RIAB_STORE_NT
#endif

namespace riab {

// (v4f / RateArgs / PosQuad / spike_store and the PlaceCell / GridCell / HDCell functors: riab_rate_cells.h)

// ---- drivers --------------------------------------------------------------------------------
// SPK: 0 none, 1 Philox uniforms, 2 explicit uniforms.
template <class Cell, int SPK, int CPB, bool NT>
__global__ __launch_bounds__(256) void rate_kernel_wide(const RateArgs a, Cell cell) {
  __shared__ double s_lds[Cell::LDS_DOUBLES];
  cell.stage(s_lds);
  constexpr int NP = Cell::NP;
  static_assert(NP * CPB <= 64, "a cell group's parameters must fit one wave");
  const int lane = threadIdx.x & 63;
  const int c0 = blockIdx.y * CPB;
  const uint32_t t = blockIdx.z;
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  const bool live = q < (uint32_t)a.qrow;
  const uint32_t qc = live ? q : 0u;
  // one coalesced load brings the whole group's parameters into the wave
  const int pi = c0 * NP + lane;
  const float mine = (lane < NP * CPB && pi < a.n * NP) ? cell.tab[pi] : 0.0f;
  const typename Cell::Pos P = cell.load(a, (int64_t)t * a.pos_ld + 4 * (int64_t)qc);
  int64_t off = ((int64_t)t * a.n + c0) * a.B + 4 * (int64_t)qc;
  const uint32_t step = a.step0 + t;
  const uint32_t group = a.group0 + qc;
#pragma unroll
  for (int j = 0; j < CPB; ++j) {
    if (c0 + j < a.n) {  // wave-uniform
      float p[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i)
        p[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), j * NP + i));
      v4f r = cell.eval(p, P);
      r = finish_rate(r * a.fr_scale + a.fr_min, P);  // [0,1] -> [min_fr, max_fr]
      if (live) {
        // (NT: a launch of ONE row — a population's update() inside a closed loop, kernels that need the result behind it:
        // streamed and written through, nothing left for the dispatch's closing release to flush, riab_device.h; the
        // many-row launches of an open-loop run keep ordinary stores: - 4 % at 1024 steps with nontemporal ones)
        if (NT) store_stream<RIAB_STORE_WT>(a.rates + off, r);
        else *reinterpret_cast<v4f*>(a.rates + off) = r;
        if (SPK == 1) spike_store<false, NT ? RIAB_STORE_WT : RIAB_SPIKE_POLICY_WIDE>(a, r, off, step, (uint32_t)(c0 + j), group);
        if (SPK == 2) spike_store<true, NT ? RIAB_STORE_WT : RIAB_SPIKE_POLICY_WIDE>(a, r, off, step, (uint32_t)(c0 + j), group);
      }
      off += a.B;
    }
  }
}

template <class Cell, int SPK>
__global__ __launch_bounds__(256) void rate_kernel_generic(const RateArgs a, Cell cell) {
  __shared__ double s_lds[Cell::LDS_DOUBLES];
  cell.stage(s_lds);
  constexpr int NP = Cell::NP;
  const uint32_t p4 = blockIdx.x * 256u + threadIdx.x;
  const bool live = p4 < a.nquads;
  const uint32_t pc = live ? p4 : 0u;
  const uint32_t t = pc / (uint32_t)a.qrow;
  const uint32_t q = pc - t * (uint32_t)a.qrow;
  const typename Cell::Pos P = cell.load(a, (int64_t)t * a.pos_ld + 4 * (int64_t)q);
  const int c0 = blockIdx.y * a.cells_per_block;
  const int c1 = min(a.n, c0 + a.cells_per_block);
  int64_t off = ((int64_t)t * a.n + c0) * a.B + 4 * (int64_t)q;
  const uint32_t step = a.step0 + t;
  const uint32_t group = a.group0 + q;
  const int lane = threadIdx.x & 63;
  for (int cb = c0; cb < c1; cb += 64) {
    // lane i holds the parameters of cell cb+i; the inner loop broadcasts cell j's with v_readlane
    float mine[NP];
    const int cm = min(cb + lane, a.n - 1);
#pragma unroll
    for (int i = 0; i < NP; ++i) mine[i] = cell.tab[cm * NP + i];
    const int cnt = min(64, c1 - cb);
    for (int j = 0; j < cnt; ++j) {
      float p[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i)
        p[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine[i]), j));
      v4f r = cell.eval(p, P);
      r = finish_rate(r * a.fr_scale + a.fr_min, P);
      if (live) {
        store_stream<RIAB_STORE_NT>(a.rates + off, r);
        if (SPK == 1) spike_store<false>(a, r, off, step, (uint32_t)(cb + j), group);
        if (SPK == 2) spike_store<true>(a, r, off, step, (uint32_t)(cb + j), group);
      }
      off += a.B;
    }
  }
}

// one_hot (Neurons.py:971-973): 1 for argmin_c |dist|, first minimum wins.  Each lane scans
// every cell for its four agents, then writes the one-hot columns.
template <int GX>
__global__ __launch_bounds__(256) void place_one_hot_kernel(const RateArgs a, PlaceCell<RIAB_PC_GAUSSIAN, GX> cell) {
  __shared__ double s_lds[PlaceCell<RIAB_PC_GAUSSIAN, GX>::LDS_DOUBLES];
  cell.stage(s_lds);
  const uint32_t p4 = blockIdx.x * 256u + threadIdx.x;
  if (p4 >= a.nquads) return;
  const uint32_t t = p4 / (uint32_t)a.qrow;
  const uint32_t q = p4 - t * (uint32_t)a.qrow;
  const PosQuad P = cell.load(a, (int64_t)t * a.pos_ld + 4 * (int64_t)q);
  float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
  int arg[4] = {0, 0, 0, 0};
  for (int c = 0; c < a.n; ++c) {
    const float cxs = cell.tab[3 * c], cys = cell.tab[3 * c + 1];
    const float d[4] = {cell.dist2(cxs, cys, P.x.x, P.y.x), cell.dist2(cxs, cys, P.x.y, P.y.y),
                        cell.dist2(cxs, cys, P.x.z, P.y.z), cell.dist2(cxs, cys, P.x.w, P.y.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (d[j] < best[j]) {
        best[j] = d[j];
        arg[j] = c;
      }
  }
  int64_t off = ((int64_t)t * a.n) * a.B + 4 * (int64_t)q;
  const uint32_t step = a.step0 + t;
  const uint32_t group = a.group0 + q;
  for (int c = 0; c < a.n; ++c) {
    v4f r = {arg[0] == c ? 1.0f : 0.0f, arg[1] == c ? 1.0f : 0.0f, arg[2] == c ? 1.0f : 0.0f,
             arg[3] == c ? 1.0f : 0.0f};
    r = r * a.fr_scale + a.fr_min;
    store_stream<RIAB_STORE_NT>(a.rates + off, r);
    if (a.spikes) {
      if (a.u_in) spike_store<true>(a, r, off, step, (uint32_t)c, group);
      else spike_store<false>(a, r, off, step, (uint32_t)c, group);
    }
    off += a.B;
  }
}

// ---- RandomSpatialNeurons (reference Neurons.py:2916-2960) ------------------------------------
// rate[c][p] = sum_m k(p, X_m) targets[m][c] / sum_m k(p, X_m), k = exp(-d_env(p, X_m)^2 / (2 l^2)):
// the "local average of targets" over the M anchor points X the smooth random functions were sampled
// on.  A lane owns four positions; blockIdx.y picks a chunk of RS_CH cells whose sums stay in
// registers; anchors and target rows are wave-uniform loads.
constexpr int RS_CH = 8;
template <int GX>
__global__ __launch_bounds__(256) void random_spatial_kernel(const RateArgs a, PlaceCell<RIAB_PC_GAUSSIAN, GX> cell,
                                                             const float* __restrict__ targets, int M) {
  __shared__ double s_lds[PlaceCell<RIAB_PC_GAUSSIAN, GX>::LDS_DOUBLES];
  cell.stage(s_lds);
  const uint32_t p4 = blockIdx.x * 256u + threadIdx.x;
  if (p4 >= a.nquads) return;
  const uint32_t t = p4 / (uint32_t)a.qrow;
  const uint32_t q = p4 - t * (uint32_t)a.qrow;
  const PosQuad P = cell.load(a, (int64_t)t * a.pos_ld + 4 * (int64_t)q);
  const int c0 = blockIdx.y * RS_CH;
  v4f den = {0.0f, 0.0f, 0.0f, 0.0f};
  v4f acc[RS_CH];
#pragma unroll
  for (int j = 0; j < RS_CH; ++j) acc[j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
  for (int m = 0; m < M; ++m) {
    const float cxs = cell.tab[3 * m], cys = cell.tab[3 * m + 1], k = cell.tab[3 * m + 2];
    const v4f kv = {__builtin_amdgcn_exp2f(cell.dist2(cxs, cys, P.x.x, P.y.x) * k),
                    __builtin_amdgcn_exp2f(cell.dist2(cxs, cys, P.x.y, P.y.y) * k),
                    __builtin_amdgcn_exp2f(cell.dist2(cxs, cys, P.x.z, P.y.z) * k),
                    __builtin_amdgcn_exp2f(cell.dist2(cxs, cys, P.x.w, P.y.w) * k)};
    den += kv;
    const float* tr = targets + (int64_t)m * a.n;
#pragma unroll
    for (int j = 0; j < RS_CH; ++j) {
      const int c = (c0 + j < a.n) ? c0 + j : a.n - 1;  // wave-uniform clamp; surplus sums are not stored
      acc[j] += kv * tr[c];
    }
  }
  int64_t off = ((int64_t)t * a.n + c0) * a.B + 4 * (int64_t)q;
  const uint32_t step = a.step0 + t;
  const uint32_t group = a.group0 + q;
#pragma unroll
  for (int j = 0; j < RS_CH; ++j) {
    if (c0 + j < a.n) {
      const v4f r = acc[j] / den;  // no position in range of any anchor: 0/0 = NaN, like the reference
      store_stream<RIAB_STORE_NT>(a.rates + off, r);
      if (a.spikes) {
        if (a.u_in) spike_store<true>(a, r, off, step, (uint32_t)(c0 + j), group);
        else spike_store<false>(a, r, off, step, (uint32_t)(c0 + j), group);
      }
      off += a.B;
    }
  }
}

// ---- standalone spikes on existing rates ----------------------------------------------------
template <bool EXPLICIT_U>
__global__ __launch_bounds__(256) void spikes_kernel(const RateArgs a) {
  const uint32_t p4 = blockIdx.x * 256u + threadIdx.x;
  if (p4 >= a.nquads) return;
  const uint32_t t = p4 / (uint32_t)a.qrow;
  const uint32_t q = p4 - t * (uint32_t)a.qrow;
  const int c0 = blockIdx.y * a.cells_per_block;
  const int c1 = min(a.n, c0 + a.cells_per_block);
  int64_t off = ((int64_t)t * a.n + c0) * a.B + 4 * (int64_t)q;
  for (int c = c0; c < c1; ++c) {
    const v4f r = ldv4(a.rates + off);
    spike_store<EXPLICIT_U>(a, r, off, a.step0 + t, (uint32_t)c, a.group0 + q);
    off += a.B;
  }
}

// ---- Neurons.update noise (reference Neurons.py:153-168) -----------------------------------
__global__ __launch_bounds__(256) void noise_kernel(float* noise, float* rates, const float* z_in, int n,
                                                    int64_t qrow, int T, float theta_dt, float sigma_dt, uint32_t k0,
                                                    uint32_t k1, uint32_t step0, uint32_t tag, uint32_t group0) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (q >= qrow) return;
  const int64_t off = ((int64_t)c * qrow + q) * 4;
  const int64_t row = (int64_t)n * qrow * 4;  // elements between consecutive time rows
  v4f x = ldv4(noise + off);
  // the OU recurrence is sequential in time: one lane walks the T rows of its (cell, 4 agents)
  for (int t = 0; t < T; ++t) {
    v4f z;
    if (z_in) {
      z = ldv4(z_in + t * row + off);
    } else {
      const u32x4 w = philox4x32_spikes(step0 + (uint32_t)t, (uint32_t)c, group0 + (uint32_t)q, tag, k0, k1);
      // two Box-Muller pairs in fp32
      const float u0 = ((float)(w.x >> 8) + 0.5f) * 0x1.0p-24f, u1 = (float)(w.y >> 8) * 0x1.0p-24f;
      const float u2 = ((float)(w.z >> 8) + 0.5f) * 0x1.0p-24f, u3 = (float)(w.w >> 8) * 0x1.0p-24f;
      const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
      z = v4f{r0 * __builtin_amdgcn_cosf(u1), r0 * __builtin_amdgcn_sinf(u1), r1 * __builtin_amdgcn_cosf(u3),
              r1 * __builtin_amdgcn_sinf(u3)};
    }
    // utils.ornstein_uhlenbeck with drift 0: dx = theta*(0 - x)*dt + sigma*(dt*z)
    x = x + (-theta_dt) * x + sigma_dt * z;
    const v4f r = ldv4(rates + t * row + off);
    *reinterpret_cast<v4f*>(rates + t * row + off) = r + x;
  }
  *reinterpret_cast<v4f*>(noise + off) = x;
}

// one float4 per thread, workgroups in address order: the store roofline of the chip (6.9 TB/s
// measured on MI355X) and the calibration kernel for the WRITE_SIZE counter
__global__ __launch_bounds__(256) void fill_kernel(float* dst, int64_t n4, float value) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) reinterpret_cast<v4f*>(dst)[i] = v4f{value, value, value, value};
}

// ---- consumer of a trajectory that is still being written (riab_simulate_fused) --------------------------------
// The trajectory kernel (riab_agent_kernel.h, PUB variant) publishes, per workgroup of 64 agents, how many steps
// of history rows it has written (write-through stores, then ctrl[RIAB_CTRL_PROGRESS + workgroup]).  This kernel
// is rate_kernel_wide launched ONCE for all T time rows — same grid (1024-agent segments, CPB-cell groups, time
// rows; x fastest), same address-ordered store stream — in which every WAVE first waits until the four
// trajectory workgroups of its 256 agents have published row t:
//   * workgroups are dispatched in grid order, so only the waves at the frontier ever wait, and what they wait for
//     is produced by a kernel that is already resident (the gate kernel below guarantees that): no deadlock;
//   * the wait is one relaxed agent-scope load of four words by four lanes + s_sleep with back-off; it is
//     bounded: a wave that gives up sets ctrl[RIAB_CTRL_ABORT] (every later wait returns at once) and counts
//     itself in ctrl[RIAB_CTRL_TIMEOUTS]; the host treats a non-zero count as an error;
//   * positions are then read with agent-scope (sc1) loads: the producer's stores are write-through (sc1), the
//     pairing MI355X_MICROARCH.md lists as valid without a cache-invalidating acquire.
// Why not a persistent kernel (a resident grid whose waves stride over the items; built and measured first, see
// tools/stream_bench.hip): a persistent wave holds its stores' credits — vmcnt retires in order, at most 63 in
// flight — whereas a wave of this kernel ENDS after its stores and the slot is refilled at once.  With the
// PlaceCells arithmetic in front of every store the persistent form reached 5.5-5.8 TB/s in every item shape
// (5.3 in the real kernel), the non-persistent one with a poll per wave 6.2-6.3 [MI355X].
typedef __attribute__((address_space(1))) unsigned long long gu64;
struct StreamArgs {
  uint32_t* ctrl;
  uint32_t step_base;     // (uint32) step0 of the launch: progress words are absolute step counts
  uint32_t spin_limit;
  uint32_t stamps;        // != 0: the waves of the first / last time row record the device clock in ctrl[RIAB_CTRL_STAMPS]
  uint32_t sleep_max;     // longest s_sleep between two polls (riab_set_option(RIAB_OPT_POLL_SLEEP))
  uint32_t serial_rows;   // >= 8: rows of the whole call — the grid's first wave counts the call in ctrl[RIAB_CTRL_SERIALISED]
                          // when it finds all of them published already AND the trajectory kernel ended less than
  uint32_t serial_gap;    // `serial_gap` ticks of the device's constant clock ago; serial_rows == 0: no check
};
// "The rate stage found every row published" has two causes: the two kernels shared a hardware queue — the rate stage then
// starts within a few microseconds of the trajectory kernel's end (barrier, [gate,] dispatch) — or the HOST was late
// with the second launch (a kernel's first launch in a process resolves its code object: tens of microseconds; a
// descheduled thread), in which case the start is anywhere after that end.  Only the first is what the counter is for.
__device__ __forceinline__ bool ended_just_now(const uint32_t* ctrl, uint32_t gap_ticks) {
  const unsigned long long end = __hip_atomic_load((gu64*)(uintptr_t)(ctrl + RIAB_CTRL_TRAJ_STAMPS + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();
  return now >= end && now - end < (unsigned long long)gap_ticks;
}
// rows of sub-segment q published so far, relative to this launch: one 16-byte read of the sub-segment's own line
__device__ __forceinline__ int stream_progress(const StreamArgs& s, uint32_t q, int lane) {
  int rel = 0x7fffffff;
  if (lane < 4) {
    const uint32_t v = __hip_atomic_load((gu32*)(uintptr_t)(s.ctrl + RIAB_CTRL_PROGRESS + 32 * q + lane), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    rel = (int)(v - s.step_base);      // stale words of earlier launches are <= step_base
    rel = rel < 0 ? 0 : rel;
  }
  const int r0 = __builtin_amdgcn_readlane(rel, 0), r1 = __builtin_amdgcn_readlane(rel, 1);
  const int r2 = __builtin_amdgcn_readlane(rel, 2), r3 = __builtin_amdgcn_readlane(rel, 3);
  return __builtin_amdgcn_readfirstlane(min(min(r0, r1), min(r2, r3)));
}
// wait until row t of sub-segment q is published
__device__ __forceinline__ void stream_wait(const StreamArgs& s, uint32_t q, int t, int lane) {
  int known = stream_progress(s, q, lane);
  for (uint32_t spins = 0; known <= t; ++spins) {
    const uint32_t ab = __hip_atomic_load((gu32*)(uintptr_t)(s.ctrl + RIAB_CTRL_ABORT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__builtin_amdgcn_readfirstlane((int)ab)) return;  // the pipeline was aborted: results are invalid anyway
    if (spins >= s.spin_limit) {
      if (lane == 0) {
        atomicAdd(s.ctrl + RIAB_CTRL_TIMEOUTS, 1u);
        __hip_atomic_store((gu32*)(uintptr_t)(s.ctrl + RIAB_CTRL_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    // back off: the waves at the frontier all poll the same few lines
    if (spins < 4) __builtin_amdgcn_s_sleep(4);
    else if (spins < 16 || s.sleep_max <= 16) __builtin_amdgcn_s_sleep(16);
    else if (s.sleep_max >= 48) __builtin_amdgcn_s_sleep(48);
    else __builtin_amdgcn_s_sleep(32);
    known = stream_progress(s, q, lane);
  }
}

// LONG changes nothing but the kernel's NAME: launches of more than 256 time rows are a different workload (a run, not
// a step of a closed loop) and get their own line in a profiler's per-kernel statistics.
// WAVES: 4, or 12 = the RESERVING shape (riab_hip.h "Residency"): a workgroup of twelve waves is three per SIMD, two
// such workgroups fill 6 of a SIMD's 8 wave slots and a third does not fit, so one slot per SIMD stays free on every
// compute unit for a trajectory workgroup whatever this kernel does.  Waves 4g .. 4g+3 of a workgroup take cell group
// 3 * blockIdx.y + g: the same (1024 agents) x (CPB cells) tile per four waves as in the four-wave shape.
template <class Cell, int SPK, int CPB, bool LONG, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rate_kernel_gated(const RateArgs a, Cell cell, const StreamArgs s) {
  __shared__ double s_lds[Cell::LDS_DOUBLES];
  cell.stage(s_lds);
  constexpr int NP = Cell::NP;
  static_assert(NP * CPB <= 64, "a cell group's parameters must fit one wave");
  static_assert(WAVES % 4 == 0, "four waves share a (1024 agents) x (CPB cells) tile");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c0 = (int)(blockIdx.y * (WAVES / 4) + (wv >> 2)) * CPB;
  const uint32_t t = blockIdx.z;
  const uint32_t q = blockIdx.x * 256u + (threadIdx.x & 255u);
  const uint32_t wq = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (uint32_t)(wv & 3)));
  if (wq * 64u >= (uint32_t)a.qrow || c0 >= a.n) return;  // (B is a multiple of 256: whole waves)
  // one coalesced load brings the whole group's parameters into the wave (independent of the trajectory)
  const int pi = c0 * NP + lane;
  const float mine = (lane < NP * CPB && pi < a.n * NP) ? cell.tab[pi] : 0.0f;
  // kernel duration without a host-side event, on the device's constant clock: the first workgroup of the grid (the
  // first one dispatched) leaves its start, the workgroups of the last cell group of the last time row (the last ones
  // dispatched) the latest end, after their stores have been acknowledged.  (A first version let every wave of the
  // first / last ROW take part: 4096 device-scope atomics on two words, 86 instead of 60 us per launch.)
  const bool grid_first = t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && wv == 0;
  if (s.stamps && grid_first && lane == 0)
    __hip_atomic_store((gu64*)(uintptr_t)(s.ctrl + RIAB_CTRL_STAMPS), (unsigned long long)__builtin_amdgcn_s_memrealtime(),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // The two kernels are meant to run side by side.  When the grid's first wave finds EVERY row of the call published
  // already, the trajectory kernel had finished before the rate stage began — both streams on one hardware queue
  // (DESIGN.md 7): counted, the host warns (Agent.diagnostics["pipeline_serialised"]).
  if (s.serial_rows >= 8u && grid_first) {
    if (stream_progress(s, wq, lane) >= (int)s.serial_rows && ended_just_now(s.ctrl, s.serial_gap) && lane == 0)
      atomicAdd(s.ctrl + RIAB_CTRL_SERIALISED, 1u);
  }
#ifdef RIAB_PIPE_PROFILE  // (tools/pipe_profile.py: per time row, on the device's constant clock, u64 words behind the ctrl block)
  gu64* const dbg = (gu64*)(uintptr_t)(s.ctrl + 2048);
  const bool first_wg = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  if (first_wg) dbg[1 * 64 + t] = (unsigned long long)__builtin_amdgcn_s_memrealtime();   // the row's first workgroup runs
#endif
  stream_wait(s, wq, (int)t, lane);
#ifdef RIAB_PIPE_PROFILE
  if (first_wg) dbg[2 * 64 + t] = (unsigned long long)__builtin_amdgcn_s_memrealtime();   // ... and has its row
#endif
  const int64_t po = (int64_t)t * a.pos_ld + 4 * (int64_t)q;
  const typename Cell::Pos P = cell.load_agent(a, po);  // agent-scope (sc1) loads: the producer's rows are written through
  int64_t off = ((int64_t)t * a.n + c0) * a.B + 4 * (int64_t)q;
  const uint32_t step = a.step0 + t;
  const uint32_t group = a.group0 + q;
#pragma unroll
  for (int j = 0; j < CPB; ++j) {
    if (c0 + j < a.n) {  // wave-uniform
      float p[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i)
        p[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), j * NP + i));
      v4f r = cell.eval(p, P);
      r = finish_rate(r * a.fr_scale + a.fr_min, P);
      // Nontemporal: the rows stream past the L2 (nothing of them is read again by this call) and leave no dirty lines to
      // write back when the kernel ends.  [MI355X] cfg 2 against ordinary stores: 20 steps: the kernel 57.9 -> 54.4 us,
      // the region 102-104 -> 98-100 us; 64 / 256 steps + 5 / + 6 %.  (The ungated rate_kernel_wide is the other way
      // round: - 4 % at 1024 steps, - 6 % at cfg 4 with nontemporal stores — RIAB_OPT_NT_STORES.)
      store_stream<RIAB_STORE_NT>(a.rates + off, r);
      if (SPK == 1) spike_store<false, RIAB_SPIKE_POLICY_GATED>(a, r, off, step, (uint32_t)(c0 + j), group);
      off += a.B;
    }
  }
#ifdef RIAB_PIPE_PROFILE
  if (c0 + CPB >= a.n && blockIdx.x + 1 == gridDim.x && (threadIdx.x & 255u) == 0) {  // the row's last workgroup is done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dbg[3 * 64 + t] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
  }
#endif
  if (s.stamps && t + 1 == gridDim.z && c0 + CPB >= a.n) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
      __hip_atomic_fetch_max((gu64*)(uintptr_t)(s.ctrl + RIAB_CTRL_STAMPS + 2), (unsigned long long)__builtin_amdgcn_s_memrealtime(),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The gates on the rate kernels' stream (one wave each):
//   started gate   returns once every trajectory workgroup of this launch is resident (ctrl[RIAB_CTRL_STARTED] has
//                  reached `started_target`): the rate waves that follow can then never occupy the slots the kernel
//                  they wait for still needs;
//   progress gate  (n_traj > 0) additionally returns only once all n_traj trajectory workgroups have published
//                  `progress_target` steps: what follows on the stream is a plain rate kernel for rows below that.
//   `final_target` != 0 (the FIRST gate of a chunk-form call of >= 8 rows): the progress every workgroup has reached
//                  when its last row is out; a gate that finds ALL of them there at its very first look counts the call
//                  in ctrl[RIAB_CTRL_SERIALISED] (the trajectory kernel had finished before the rate stage began).
__global__ __launch_bounds__(64) void stream_gate_kernel(uint32_t* ctrl, uint32_t started_target, uint32_t n_traj,
                                                         uint32_t progress_target, uint32_t spin_limit, uint32_t sleep_long,
                                                         uint32_t final_target, uint32_t serial_gap) {
  const int lane = threadIdx.x;
  for (uint32_t spins = 0;; ++spins) {
    const uint32_t v = __hip_atomic_load((gu32*)(uintptr_t)(ctrl + RIAB_CTRL_STARTED), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = (int32_t)(v - started_target) >= 0;
    bool all_done = ok && final_target != 0u && spins == 0u;
    if (ok) {
      for (uint32_t w = lane; w < n_traj; w += 64) {
        const uint32_t p = __hip_atomic_load((gu32*)(uintptr_t)(ctrl + RIAB_CTRL_PROGRESS_WORD(w)), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (int32_t)(p - progress_target) >= 0;
        all_done = all_done && (int32_t)(p - final_target) >= 0;
      }
    }
    if (spins == 0u && final_target != 0u && n_traj > 0u && __builtin_amdgcn_ballot_w64(!all_done) == 0 &&
        ended_just_now(ctrl, serial_gap) && lane == 0)
      atomicAdd(ctrl + RIAB_CTRL_SERIALISED, 1u);
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) return;
    if (__hip_atomic_load((gu32*)(uintptr_t)(ctrl + RIAB_CTRL_ABORT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (spins >= spin_limit) {
      if (lane == 0) {
        atomicAdd(ctrl + RIAB_CTRL_TIMEOUTS, 1u);
        __hip_atomic_store((gu32*)(uintptr_t)(ctrl + RIAB_CTRL_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    // the started gate may have to sit out whatever was queued in front of the trajectory kernel (long sleeps: 3.4 us),
    // but normally that kernel is a microsecond away from announcing itself: the first polls are 0.2 us apart;
    // a progress gate is on the critical path of its chunk
    if (sleep_long && spins >= 64) __builtin_amdgcn_s_sleep(127);
    else if (sleep_long) __builtin_amdgcn_s_sleep(8);
    else __builtin_amdgcn_s_sleep(16);
  }
}

// The opening kernel of a STRICT riab_simulate call (riab_hip.h "Two modes"): one wave on the caller's stream, in front
// of everything else the call enqueues.  It zeroes the announcement counter and sets every progress word of the call to
// `step_base`, so that nothing the call's gates and waiting waves compare depends on what an earlier call — or an
// earlier replay of the same captured call — left in the control block.
__global__ __launch_bounds__(64) void stream_open_kernel(uint32_t* ctrl, uint32_t n_traj, uint32_t step_base) {
  for (uint32_t w = threadIdx.x; w < n_traj; w += 64)
    __hip_atomic_store((gu32*)(uintptr_t)(ctrl + RIAB_CTRL_PROGRESS_WORD(w)), step_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0)
    __hip_atomic_store((gu32*)(uintptr_t)(ctrl + RIAB_CTRL_STARTED), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- host side ------------------------------------------------------------------------------
static int check_io(const RiabRateIO* io, int n, bool need_pos, bool need_hd) {
  if (!io || n <= 0 || io->T <= 0 || io->B <= 0 || !io->rates) return RIAB_EINVAL;
  if (need_pos && (!io->pos_x || !io->pos_y)) return RIAB_EINVAL;
  if (need_hd && (!io->hd_x || !io->hd_y)) return RIAB_EINVAL;
  if (io->B % 4 != 0 || io->pos_ld % 4 != 0 || io->agent_id0 % 4 != 0) return RIAB_EALIGN;
  const uintptr_t m = (uintptr_t)io->rates | (uintptr_t)io->pos_x | (uintptr_t)io->pos_y | (uintptr_t)io->hd_x |
                      (uintptr_t)io->hd_y | (uintptr_t)io->u_in;
  if (m & 15) return RIAB_EALIGN;
  if ((uintptr_t)io->spikes & 3) return RIAB_EALIGN;
  if (io->u_in && !io->spikes) return RIAB_EINVAL;
  if (io->T * (io->B / 4) >= ((int64_t)1 << 31)) return RIAB_ETOOBIG;  // 32-bit quad index in the kernels
  return RIAB_OK;
}

static RateArgs make_args(const RiabRateIO* io, int n, dim3* grid) {
  RateArgs a;
  a.pos_x = io->pos_x;
  a.pos_y = io->pos_y;
  a.hd_x = io->hd_x;
  a.hd_y = io->hd_y;
  a.pos_ld = io->pos_ld;
  a.qrow = io->B / 4;
  a.nquads = io->T * a.qrow;
  a.B = io->B;
  a.rates = io->rates;
  a.spikes = io->spikes;
  a.u_in = io->u_in;
  a.dt = io->dt;
  a.fr_scale = io->max_fr - io->min_fr;
  a.fr_min = io->min_fr;
  a.k0 = (uint32_t)io->seed;
  a.k1 = (uint32_t)(io->seed >> 32);
  a.step0 = (uint32_t)io->step0;
  a.tag = RIAB_TAG_SPIKES | ((uint32_t)io->pop_id & 0xFFu);
  a.group0 = (uint32_t)(io->agent_id0 / 4);
  a.n = n;
  const int64_t pblocks = (a.nquads + 255) / 256;
  // enough workgroups to fill 256 CUs x several waves, but keep >= 16 cells per lane so the
  // position load and index arithmetic amortise
  int chunks = (int)((4096 + pblocks - 1) / pblocks);
  if (chunks < 1) chunks = 1;
  int cpb = (n + chunks - 1) / chunks;
  cpb = ((cpb + 63) / 64) * 64;  // whole 64-cell groups (one table register per lane)
  a.cells_per_block = cpb;
  *grid = dim3((unsigned)pblocks, (unsigned)((n + cpb - 1) / cpb), 1);
  return a;
}

template <class Cell>
static int launch_rate(const RiabRateIO* io, int n, const Cell& cell, hipStream_t s) {
  constexpr int kCellsPerGroup = Cell::CPB;
  dim3 grid;
  const RateArgs a = make_args(io, n, &grid);
  if (a.qrow >= 256 && io->T <= 65535 && (n + kCellsPerGroup - 1) / kCellsPerGroup <= 65535) {
    // address-ordered wide kernel
    const dim3 g((unsigned)((a.qrow + 255) / 256), (unsigned)((n + kCellsPerGroup - 1) / kCellsPerGroup), (unsigned)io->T);
    if (g_options[RIAB_OPT_NT_STORES] || io->T == 1) {  // (streamed, written-through stores: one-row launches; the option: A/B at any length)
      if (!io->spikes) hipLaunchKernelGGL((rate_kernel_wide<Cell, 0, kCellsPerGroup, true>), g, dim3(256), 0, s, a, cell);
      else if (!io->u_in) hipLaunchKernelGGL((rate_kernel_wide<Cell, 1, kCellsPerGroup, true>), g, dim3(256), 0, s, a, cell);
      else hipLaunchKernelGGL((rate_kernel_wide<Cell, 2, kCellsPerGroup, true>), g, dim3(256), 0, s, a, cell);
    } else if (!io->spikes) hipLaunchKernelGGL((rate_kernel_wide<Cell, 0, kCellsPerGroup, false>), g, dim3(256), 0, s, a, cell);
    else if (!io->u_in) hipLaunchKernelGGL((rate_kernel_wide<Cell, 1, kCellsPerGroup, false>), g, dim3(256), 0, s, a, cell);
    else hipLaunchKernelGGL((rate_kernel_wide<Cell, 2, kCellsPerGroup, false>), g, dim3(256), 0, s, a, cell);
    return (int)hipGetLastError();
  }
  if (!io->spikes) hipLaunchKernelGGL((rate_kernel_generic<Cell, 0>), grid, dim3(256), 0, s, a, cell);
  else if (!io->u_in) hipLaunchKernelGGL((rate_kernel_generic<Cell, 1>), grid, dim3(256), 0, s, a, cell);
  else hipLaunchKernelGGL((rate_kernel_generic<Cell, 2>), grid, dim3(256), 0, s, a, cell);
  return (int)hipGetLastError();
}

template <int GX>
static int launch_place(const RiabRateIO* io, int n, int desc, const PlaceCell<RIAB_PC_GAUSSIAN, GX>& base,
                        hipStream_t s) {
  switch (desc) {
    case RIAB_PC_GAUSSIAN: return launch_rate(io, n, base, s);
    case RIAB_PC_GAUSSIAN_THRESHOLD: return launch_rate(io, n, base.template as<RIAB_PC_GAUSSIAN_THRESHOLD>(), s);
    case RIAB_PC_DIFF_OF_GAUSSIANS: return launch_rate(io, n, base.template as<RIAB_PC_DIFF_OF_GAUSSIANS>(), s);
    case RIAB_PC_TOP_HAT: return launch_rate(io, n, base.template as<RIAB_PC_TOP_HAT>(), s);
    case RIAB_PC_ONE_HOT: {
      dim3 grid;
      const RateArgs a = make_args(io, n, &grid);
      grid.y = 1;
      hipLaunchKernelGGL((place_one_hot_kernel<GX>), grid, dim3(256), 0, s, a, base);
      return (int)hipGetLastError();
    }
    default: return RIAB_EINVAL;
  }
}

template <int GX>
static int place_dispatch(const RiabEnv* env, const RiabRateIO* io, const float* cells, int n, int desc, float thw,
                          hipStream_t s) {
  PlaceCell<RIAB_PC_GAUSSIAN, GX> c;
  c.tab = cells;
  c.scale = (float)env->scale;
  c.half_scale = (float)(env->scale / 2);
  c.top_hat_w2 = thw * thw;
  c.walls = env->walls;
  c.n_internal = env->n_walls > 4 ? env->n_walls - 4 : 0;
  if (GX == 2 && c.n_internal > 1) c.n_internal = 1;
  c.e0 = env->extent[0]; c.e1 = env->extent[1]; c.e2 = env->extent[2]; c.e3 = env->extent[3];
  c.shape = make_env_shape(env);
  c.lds = nullptr;
  return launch_place<GX>(io, n, desc, c, s);
}

// ---- launches of the flag-coupled rate stage (called by riab_simulate_fused, riab_simulate.hip) ----------------
// start / stop events of the one rate_kernel_gated launch of a launch_rate_stream call (kernel-level timing through
// hipExtLaunchKernel; per thread, set and cleared by launch_rate_stream)
static thread_local hipEvent_t t_stream_ev0 = nullptr, t_stream_ev1 = nullptr;

// 25 us in ticks of the device's constant clock (s_memrealtime): how soon after the trajectory kernel's end a rate
// stage that shared its hardware queue begins (barrier + [one-wave gate + barrier] + dispatch: 4-12 us [MI355X])
static uint32_t serial_gap_ticks() {
  static uint32_t ticks = 0;
  if (!ticks) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
      khz = 100000;
    ticks = (uint32_t)((long long)khz * 25 / 1000);
  }
  return ticks;
}

int traj_kernel_regs();   // riab_agent.hip
static thread_local bool t_stream_reserve = false;  // launch the reserving (twelve-wave) shape: set by launch_rate_stream

template <class Cell>
static int launch_stream_cell(const RateArgs& a, const Cell& cell, const StreamArgs& st_in, int T, bool spikes, bool dry_run,
                              hipStream_t s) {
  const hipEvent_t ev0 = t_stream_ev0, ev1 = t_stream_ev1;
  // Twice the wide kernel's cells per wave where the group's parameters still fit one wave (PlaceCells: 8): a gated
  // wave pays two dependent round trips to memory (progress words, then the write-through positions) before its
  // first store, so it should bring more stores with it.  [MI355X] cfg 2: K = 20 +1.5 %, K = 128 +4 %; x4: the same.
  constexpr int CPB = (Cell::NP * 2 * Cell::CPB <= 64) ? 2 * Cell::CPB : Cell::CPB;
  const int64_t groups = (a.n + CPB - 1) / CPB;
  if (T > 65535 || groups > 65535) return RIAB_ETOOBIG;  // grid y / z limits: the caller splits longer runs
  const bool reserve = t_stream_reserve;
  const bool lng = T > 256;
  if (reserve) {
    // The reserving shape keeps its promise (riab_hip.h "Residency") only while wave slots are the ONLY resource it can
    // exhaust: two twelve-wave workgroups per compute unit are six waves per SIMD, and next to them a trajectory
    // workgroup's 224 registers per lane must still fit the SIMD's 512: at most 48 per lane here (the euclidean /
    // periodic place, grid and head-direction kernels hold 32-40; the line-of-sight and geodesic ones 88-96, where one
    // more register class would fit two workgroups and leave too little).  Asked of the code object once per kernel.
    static int regs[4] = {0, 0, 0, 0};
    int& r = regs[(spikes ? 2 : 0) + (lng ? 1 : 0)];
    if (r == 0) {
      hipFuncAttributes attr;
      const void* f = spikes ? (lng ? (const void*)rate_kernel_gated<Cell, 1, CPB, true, 12> : (const void*)rate_kernel_gated<Cell, 1, CPB, false, 12>)
                             : (lng ? (const void*)rate_kernel_gated<Cell, 0, CPB, true, 12> : (const void*)rate_kernel_gated<Cell, 0, CPB, false, 12>);
      r = (hipFuncGetAttributes(&attr, f) == hipSuccess && attr.numRegs > 0) ? attr.numRegs : 1 << 20;
      // (... and LDS: two of these workgroups next to a trajectory workgroup's 80 KB must fit the compute unit's 160 KB; the
      // kernels that are offered the shape declare 8 bytes — anything beyond a few KB is refused like too many registers)
      if (r < (1 << 20) && 2 * (int64_t)attr.sharedSizeBytes + 81920 > 160 * 1024) r = 1 << 20;
      (void)hipGetLastError();
    }
    // (six waves of this kernel and one of the trajectory kernel on a SIMD: 6 x r + its registers <= 512; with the
    // trajectory kernel's 224 that is r <= 48)
    if (6 * ((r + 7) / 8 * 8) + traj_kernel_regs() > 512) return RIAB_EUNSUPPORTED;  // (the caller falls back to the started gate)
  }
  if (dry_run) return RIAB_OK;  // (every argument check is above: nothing is launched)
  // (a kernel's FIRST launch in a process resolves its code object on the host — tens of microseconds in which a short
  // trajectory kernel finishes: that call says nothing about hardware queues, it is not examined)
  static bool launched[8] = {false, false, false, false, false, false, false, false};
  bool& seen = launched[(reserve ? 4 : 0) + (spikes ? 2 : 0) + (lng ? 1 : 0)];
  StreamArgs st = st_in;
  if (!seen) st.serial_rows = 0u;
  seen = true;
  const dim3 grid((unsigned)((a.qrow + 255) / 256), (unsigned)(reserve ? (groups + 2) / 3 : groups), (unsigned)T);
  const dim3 block(reserve ? 768 : 256);
  auto go = [&](auto kernel) {
    if (ev0 || ev1) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, ev0, ev1, 0u, a, cell, st);
    else hipLaunchKernelGGL(kernel, grid, block, 0, s, a, cell, st);
  };
  if (reserve) {  // (short calls of one population from an idle stream: no spikes-with-LONG zoo needed, but keep all four)
    if (spikes) {
      if (lng) go(rate_kernel_gated<Cell, 1, CPB, true, 12>);
      else go(rate_kernel_gated<Cell, 1, CPB, false, 12>);
    } else {
      if (lng) go(rate_kernel_gated<Cell, 0, CPB, true, 12>);
      else go(rate_kernel_gated<Cell, 0, CPB, false, 12>);
    }
  } else if (spikes) {
    if (lng) go(rate_kernel_gated<Cell, 1, CPB, true, 4>);
    else go(rate_kernel_gated<Cell, 1, CPB, false, 4>);
  } else {
    if (lng) go(rate_kernel_gated<Cell, 0, CPB, true, 4>);
    else go(rate_kernel_gated<Cell, 0, CPB, false, 4>);
  }
  return (int)hipGetLastError();
}

template <int GX>
static int launch_stream_place(const RiabEnv* env, const RiabPopulation* pop, const RateArgs& a, const StreamArgs& st, int T,
                               bool spikes, bool dry_run, hipStream_t s) {
  PlaceCell<RIAB_PC_GAUSSIAN, GX> c;
  c.tab = pop->table;
  c.scale = (float)env->scale;
  c.half_scale = (float)(env->scale / 2);
  c.top_hat_w2 = pop->top_hat_width * pop->top_hat_width;
  c.walls = env->walls;
  c.n_internal = env->n_walls > 4 ? env->n_walls - 4 : 0;
  if (GX == 2 && c.n_internal > 1) c.n_internal = 1;
  c.e0 = env->extent[0]; c.e1 = env->extent[1]; c.e2 = env->extent[2]; c.e3 = env->extent[3];
  c.shape = make_env_shape(env);
  c.lds = nullptr;
  switch (pop->description) {
    case RIAB_PC_GAUSSIAN: return launch_stream_cell(a, c, st, T, spikes, dry_run, s);
    case RIAB_PC_GAUSSIAN_THRESHOLD:
      return launch_stream_cell(a, c.template as<RIAB_PC_GAUSSIAN_THRESHOLD>(), st, T, spikes, dry_run, s);
    case RIAB_PC_DIFF_OF_GAUSSIANS:
      return launch_stream_cell(a, c.template as<RIAB_PC_DIFF_OF_GAUSSIANS>(), st, T, spikes, dry_run, s);
    case RIAB_PC_TOP_HAT: return launch_stream_cell(a, c.template as<RIAB_PC_TOP_HAT>(), st, T, spikes, dry_run, s);
    default: return RIAB_EUNSUPPORTED;  // one_hot scans every cell per position: not a streaming shape
  }
}

// 0 when the stream kernel covers this population, RIAB_EUNSUPPORTED otherwise (checked before anything is launched)
int stream_supported(const RiabEnv* env, const RiabPopulation* pop, int64_t B) {
  if (!env || !pop || pop->n <= 0 || B <= 0 || B % 256 != 0) return RIAB_EUNSUPPORTED;
  if (pop->noise_state) return RIAB_EUNSUPPORTED;  // the OU noise pass is sequential over the finished rows
  switch (pop->kind) {
    case RIAB_POP_PLACE:
      if (pop->description == RIAB_PC_ONE_HOT) return RIAB_EUNSUPPORTED;
      if (env->periodic && pop->geometry != RIAB_GEOM_EUCLIDEAN) return RIAB_EUNSUPPORTED;
      if (pop->geometry != RIAB_GEOM_EUCLIDEAN && env->n_walls - 4 > RIAB_MAX_WALLS) return RIAB_EUNSUPPORTED;
      if (pop->geometry == RIAB_GEOM_GEODESIC && env->n_walls > 5) return RIAB_EUNSUPPORTED;
      return pop->table ? RIAB_OK : RIAB_EINVAL;
    case RIAB_POP_GRID:
    case RIAB_POP_HDC: return pop->table ? RIAB_OK : RIAB_EINVAL;
    default: return RIAB_EUNSUPPORTED;
  }
}

int launch_rate_stream(const RiabEnv* env, const RiabPopulation* pop, const float* hist, int64_t B, int32_t T, float dt,
                       uint64_t seed, uint64_t step0, int64_t agent_id0, uint32_t* ctrl, uint32_t spin_limit, bool stamps,
                       hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, bool dry_run, bool reserve,
                       uint32_t serial_rows) {
  struct EventScope {
    EventScope(hipEvent_t a, hipEvent_t b, bool r) { t_stream_ev0 = a; t_stream_ev1 = b; t_stream_reserve = r; }
    ~EventScope() { t_stream_ev0 = t_stream_ev1 = nullptr; t_stream_reserve = false; }
  } scope(ev_start, ev_stop, reserve);
  int rc = stream_supported(env, pop, B);
  if (rc) return rc;
  if (!hist || !ctrl || !pop->rates_base || T <= 0 || pop->capacity_rows < T || agent_id0 % 4) return RIAB_EINVAL;
  if ((((uintptr_t)hist | (uintptr_t)pop->rates_base) & 15) || ((uintptr_t)pop->spikes_base & 3)) return RIAB_EALIGN;
  RateArgs a;
  a.pos_x = hist + (int64_t)RIAB_H_POS_X * B;
  a.pos_y = hist + (int64_t)RIAB_H_POS_Y * B;
  a.hd_x = hist + (int64_t)RIAB_H_HD_X * B;
  a.hd_y = hist + (int64_t)RIAB_H_HD_Y * B;
  a.pos_ld = (int64_t)RIAB_HIST_ROWS * B;
  a.qrow = B / 4;
  a.nquads = (int64_t)T * a.qrow;
  a.B = B;
  a.rates = pop->rates_base;
  a.spikes = pop->spikes_base;
  a.u_in = nullptr;
  a.dt = dt;
  a.fr_scale = pop->io.max_fr - pop->io.min_fr;
  a.fr_min = pop->io.min_fr;
  a.k0 = (uint32_t)seed;
  a.k1 = (uint32_t)(seed >> 32);
  a.step0 = (uint32_t)(step0 + 1);  // Neurons.update after the (step0 + t + 1)-th Agent.update
  a.tag = RIAB_TAG_SPIKES | ((uint32_t)pop->io.pop_id & 0xFFu);
  a.group0 = (uint32_t)(agent_id0 / 4);
  a.n = pop->n;
  a.cells_per_block = 0;
  StreamArgs st;
  st.ctrl = ctrl;
  st.step_base = (uint32_t)step0;
  st.spin_limit = spin_limit;
  st.stamps = stamps ? 1u : 0u;
  st.sleep_max = (uint32_t)g_options[RIAB_OPT_POLL_SLEEP];
  st.serial_rows = serial_rows;
  st.serial_gap = serial_gap_ticks();
  const bool spikes = pop->spikes_base != nullptr;
  switch (pop->kind) {
    case RIAB_POP_PLACE:
      if (env->periodic) return launch_stream_place<3>(env, pop, a, st, T, spikes, dry_run, s);
      switch (pop->geometry) {
        case RIAB_GEOM_EUCLIDEAN: return launch_stream_place<0>(env, pop, a, st, T, spikes, dry_run, s);
        case RIAB_GEOM_LINE_OF_SIGHT: return launch_stream_place<1>(env, pop, a, st, T, spikes, dry_run, s);
        case RIAB_GEOM_GEODESIC: return launch_stream_place<2>(env, pop, a, st, T, spikes, dry_run, s);
        default: return RIAB_EINVAL;
      }
    case RIAB_POP_GRID:
      if (pop->description == RIAB_GC_RECTIFIED) {
        GridCell<RIAB_GC_RECTIFIED> c{pop->table, pop->f0, 1.0f / (1.0f - pop->f0)};
        return launch_stream_cell(a, c, st, T, spikes, dry_run, s);
      }
      if (pop->description == RIAB_GC_SHIFTED) {
        GridCell<RIAB_GC_SHIFTED> c{pop->table, pop->f0, 1.0f};
        return launch_stream_cell(a, c, st, T, spikes, dry_run, s);
      }
      return RIAB_EINVAL;
    case RIAB_POP_HDC: {
      HDCell<0> c{pop->table, 0.0f, nullptr, nullptr};
      return launch_stream_cell(a, c, st, T, spikes, dry_run, s);
    }
    default: return RIAB_EUNSUPPORTED;
  }
}

int launch_stream_open(uint32_t* ctrl, uint32_t n_traj, uint32_t step_base, hipStream_t s) {
  hipLaunchKernelGGL(stream_open_kernel, dim3(1), dim3(64), 0, s, ctrl, n_traj, step_base);
  return (int)hipGetLastError();
}

int launch_stream_gate(uint32_t* ctrl, uint32_t started_target, uint32_t n_traj, uint32_t progress_target,
                       uint32_t spin_limit, bool sleep_long, uint32_t final_target, hipStream_t s) {
  static bool launched = false;   // (the gate's first launch in the process: see launch_stream_cell)
  if (!launched) final_target = 0u;
  launched = true;
  hipLaunchKernelGGL(stream_gate_kernel, dim3(1), dim3(64), 0, s, ctrl, started_target, n_traj, progress_target, spin_limit,
                     sleep_long ? 1u : 0u, final_target, serial_gap_ticks());
  return (int)hipGetLastError();
}

}  // namespace riab

using namespace riab;

extern "C" int riab_place_cells(const RiabEnv* env, const RiabRateIO* io, const float* cells, int32_t n,
                                int32_t description, int32_t geometry, float top_hat_width, riab_stream_t stream) {
  if (!env || !cells) return RIAB_EINVAL;
  const int rc = check_io(io, n, true, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (env->periodic) {
    if (geometry != RIAB_GEOM_EUCLIDEAN) return RIAB_EUNSUPPORTED;  // Neurons.py:908-921
    return place_dispatch<3>(env, io, cells, n, description, top_hat_width, s);
  }
  if (geometry != RIAB_GEOM_EUCLIDEAN) {
    if (env->n_walls > 4 && !env->walls) return RIAB_EINVAL;
    if (env->n_walls - 4 > RIAB_MAX_WALLS) return RIAB_ETOOBIG;
    if (geometry == RIAB_GEOM_GEODESIC && env->n_walls > 5) return RIAB_EUNSUPPORTED;  // Environment.py:736-739
  }
  switch (geometry) {
    case RIAB_GEOM_EUCLIDEAN: return place_dispatch<0>(env, io, cells, n, description, top_hat_width, s);
    case RIAB_GEOM_LINE_OF_SIGHT: return place_dispatch<1>(env, io, cells, n, description, top_hat_width, s);
    case RIAB_GEOM_GEODESIC: return place_dispatch<2>(env, io, cells, n, description, top_hat_width, s);
    default: return RIAB_EINVAL;
  }
}

template <int GX>
static int random_spatial_dispatch(const RiabEnv* env, const RiabRateIO* io, const float* anchors, int M,
                                   const float* targets, int n, hipStream_t s) {
  PlaceCell<RIAB_PC_GAUSSIAN, GX> c;
  c.tab = anchors;
  c.scale = (float)env->scale;
  c.half_scale = (float)(env->scale / 2);
  c.top_hat_w2 = 0.0f;
  c.walls = env->walls;
  c.n_internal = env->n_walls > 4 ? env->n_walls - 4 : 0;
  if (GX == 2 && c.n_internal > 1) c.n_internal = 1;
  c.e0 = env->extent[0]; c.e1 = env->extent[1]; c.e2 = env->extent[2]; c.e3 = env->extent[3];
  c.shape = make_env_shape(env);
  c.lds = nullptr;
  dim3 grid;
  RateArgs a = make_args(io, n, &grid);
  a.fr_scale = 1.0f;  // the targets already lie in [min_fr, max_fr] (Neurons.py:2911-2912)
  a.fr_min = 0.0f;
  grid.y = (unsigned)((n + RS_CH - 1) / RS_CH);
  hipLaunchKernelGGL((random_spatial_kernel<GX>), grid, dim3(256), 0, s, a, c, targets, M);
  return (int)hipGetLastError();
}

extern "C" int riab_random_spatial_neurons(const RiabEnv* env, const RiabRateIO* io, const float* anchors, int32_t M,
                                           const float* targets, int32_t n, int32_t geometry, riab_stream_t stream) {
  if (!env || !anchors || !targets || M <= 0) return RIAB_EINVAL;
  const int rc = check_io(io, n, true, false);
  if (rc) return rc;
  if ((n + RS_CH - 1) / RS_CH > 65535) return RIAB_ETOOBIG;
  hipStream_t s = (hipStream_t)stream;
  if (env->periodic) {
    if (geometry != RIAB_GEOM_EUCLIDEAN) return RIAB_EUNSUPPORTED;
    return random_spatial_dispatch<3>(env, io, anchors, M, targets, n, s);
  }
  if (geometry != RIAB_GEOM_EUCLIDEAN) {
    if (env->n_walls > 4 && !env->walls) return RIAB_EINVAL;
    if (env->n_walls - 4 > RIAB_MAX_WALLS) return RIAB_ETOOBIG;
    if (geometry == RIAB_GEOM_GEODESIC && env->n_walls > 5) return RIAB_EUNSUPPORTED;
  }
  switch (geometry) {
    case RIAB_GEOM_EUCLIDEAN: return random_spatial_dispatch<0>(env, io, anchors, M, targets, n, s);
    case RIAB_GEOM_LINE_OF_SIGHT: return random_spatial_dispatch<1>(env, io, anchors, M, targets, n, s);
    case RIAB_GEOM_GEODESIC: return random_spatial_dispatch<2>(env, io, anchors, M, targets, n, s);
    default: return RIAB_EINVAL;
  }
}

extern "C" int riab_grid_cells(const RiabRateIO* io, const float* table, int32_t n, int32_t description, float f0,
                               riab_stream_t stream) {
  if (!table) return RIAB_EINVAL;
  const int rc = check_io(io, n, true, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (description == RIAB_GC_RECTIFIED) {
    GridCell<RIAB_GC_RECTIFIED> c{table, f0, 1.0f / (1.0f - f0)};
    return launch_rate(io, n, c, s);
  }
  if (description == RIAB_GC_SHIFTED) {
    GridCell<RIAB_GC_SHIFTED> c{table, f0, 1.0f};
    return launch_rate(io, n, c, s);
  }
  return RIAB_EINVAL;
}

extern "C" int riab_head_direction_cells(const RiabRateIO* io, const float* table, int32_t n,
                                         riab_stream_t stream) {
  if (!table) return RIAB_EINVAL;
  const int rc = check_io(io, n, false, true);
  if (rc) return rc;
  HDCell<0> c{table, 0.0f, nullptr, nullptr};
  return launch_rate(io, n, c, (hipStream_t)stream);
}

extern "C" int riab_velocity_cells(const RiabRateIO* io, const float* table, int32_t n, float one_sigma_speed,
                                   const double* vel_x, const double* vel_y, riab_stream_t stream) {
  if (!table || !(one_sigma_speed > 0.0f) || (vel_x == nullptr) != (vel_y == nullptr)) return RIAB_EINVAL;
  const int rc = check_io(io, n, false, vel_x == nullptr);
  if (rc) return rc;
  if (vel_x && io->T != 1) return RIAB_EINVAL;  // the state holds the current step only
  if (((uintptr_t)vel_x | (uintptr_t)vel_y) & 31) return RIAB_EALIGN;
  HDCell<1> c{table, 1.0f / one_sigma_speed, vel_x, vel_y};
  return launch_rate(io, n, c, (hipStream_t)stream);
}

extern "C" int riab_speed_cell(const RiabRateIO* io, float one_sigma_speed, riab_stream_t stream) {
  if (!(one_sigma_speed > 0.0f)) return RIAB_EINVAL;
  const int rc = check_io(io, 1, false, true);
  if (rc) return rc;
  // the single cell has no parameters; the table pointer only has to be readable (three floats)
  HDCell<2> c{io->hd_x, 1.0f / one_sigma_speed, nullptr, nullptr};
  return launch_rate(io, 1, c, (hipStream_t)stream);
}

extern "C" int riab_spikes(const RiabRateIO* io, int32_t n, riab_stream_t stream) {
  const int rc = check_io(io, n, false, false);
  if (rc) return rc;
  if (!io->spikes) return RIAB_EINVAL;
  dim3 grid;
  const RateArgs a = make_args(io, n, &grid);
  if (io->u_in) hipLaunchKernelGGL((spikes_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((spikes_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int riab_neuron_noise(float* noise, float* rates, const float* z_in, int32_t n, int64_t B, int32_t T,
                                 float theta_dt, float sigma_dt, uint64_t seed, uint64_t step, int32_t pop_id,
                                 int64_t agent_id0, riab_stream_t stream) {
  if (!noise || !rates || n <= 0 || B <= 0 || T <= 0) return RIAB_EINVAL;
  if (B % 4 || agent_id0 % 4 || (((uintptr_t)noise | (uintptr_t)rates | (uintptr_t)z_in) & 15)) return RIAB_EALIGN;
  const int64_t qrow = B / 4;
  dim3 grid((unsigned)((qrow + 255) / 256), (unsigned)n, 1);
  hipLaunchKernelGGL(noise_kernel, grid, dim3(256), 0, (hipStream_t)stream, noise, rates, z_in, n, qrow, T, theta_dt,
                     sigma_dt, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step,
                     RIAB_TAG_NOISE | ((uint32_t)pop_id & 0xFFu), (uint32_t)(agent_id0 / 4));
  return (int)hipGetLastError();
}

extern "C" int riab_fill(void* dst, int64_t bytes, float value, riab_stream_t stream) {
  if (!dst || bytes <= 0) return RIAB_EINVAL;
  if ((bytes & 15) || ((uintptr_t)dst & 15)) return RIAB_EALIGN;
  const int64_t n4 = bytes / 16;
  const int64_t blocks = (n4 + 255) / 256;
  if (blocks > 0x7fffffffLL) return RIAB_ETOOBIG;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)dst, n4, value);
  return (int)hipGetLastError();
}

namespace riab {
int g_options[RIAB_OPT_COUNT] = {0, 1, 1, 0, 4, 48, 1, 22, 1};
}
extern "C" int riab_set_option(int32_t option, int32_t value) {
  static const int lo[RIAB_OPT_COUNT] = {0, 0, 0, 0, 0, 1, 0, 0, 0};
  static const int hi[RIAB_OPT_COUNT] = {2, 1, 1, 1, 64, 127, 2, 30, 1};
  if (option < 0 || option >= RIAB_OPT_COUNT || value < lo[option] || value > hi[option]) return RIAB_EINVAL;
  const int old = riab::g_options[option];
  riab::g_options[option] = value;
  return old;
}

extern "C" int riab_abi_version(void) { return RIAB_ABI_VERSION; }

extern "C" const char* riab_strerror(int code) {
  switch (code) {
    case RIAB_OK: return "ok";
    case RIAB_EINVAL: return "invalid argument (null pointer, non-positive size or unknown enum)";
    case RIAB_EALIGN: return "agent axis not a multiple of 4 or row pointer not 16-byte aligned";
    case RIAB_ETOOBIG: return "too many walls / test angles for the LDS staging";
    case RIAB_EUNSUPPORTED: return "combination not supported on device";
    case RIAB_EFULL: return "a step plan's history chunk is full: attach a new chunk";
    case RIAB_EPARTIAL: return "a launch failed after the trajectory kernel had been launched: the state has advanced, the rates of this call are incomplete";
    case RIAB_ECHANGED: return "a watched host array differs from its snapshot: the cached device tables are stale (nothing was launched)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown riab error";
  }
}
