// The open-loop path as ONE native call (include/riab_hip.h: riab_simulate): the trajectory kernel and the
// firing-rate stage run CONCURRENTLY on two streams, coupled by flags in device memory.  This file is host code only:
// the kernels live in riab_traj4_kernel.h (the publishing trajectory kernel: rows written through + progress words)
// and riab_rates.hip (rate_kernel_gated, stream_gate_kernel, and every population's ordinary kernel).
//
// Why not one launch: a launch has ONE register and LDS allocation.  The trajectory kernel needs 220 VGPRs and 80 KB
// of LDS per 64 agents; the rate kernels need ~40 VGPRs and no LDS and live on occupancy.  Why not launches per chunk
// behind HIP events: every chunk pays the fill of the first stage and a dependent launch boundary, and a 20-step run
// has nothing to overlap.  With flags the rate stage starts a step or two behind the trajectory and stays there.
//
// Which stream carries what.  The rate stage ends last, so IT runs on the caller's stream and the trajectory kernel on
// the streamer's own: the join (caller's stream waits for the trajectory kernel) is then a barrier whose dependency was
// met tens of microseconds before the rate stage finishes, and a host synchronisation returns as soon after the last
// kernel as after any single kernel (~10 us).  The other way round — rate stage on the second stream, round 2 — the
// join sat behind the LAST kernel: its signal had to travel second queue -> event -> barrier packet on the first
// queue -> host, 25 us from the end of the rate kernel to the return of hipDeviceSynchronize [MI355X, rocprofv3
// --hip-trace --kernel-trace of the driver's bench command].
//
// Two forms of the rate stage:
//  * one store-bound population (place / grid / head-direction cells without OU noise), up to `poll_max` steps
//    (default 65535, the grid's limit): ONE rate kernel for all rows whose waves wait for their rows themselves
//    (rate_kernel_gated, nontemporal stores).  Workgroups are dispatched in row order, so the resident ones are always
//    the oldest unfinished rows — the ones at or behind the trajectory's frontier.
//  * anything else: per chunk of rows (16, 28, 44, ... 128) a one-wave progress gate, then every population's ordinary
//    kernel over the chunk in list order, at the price of a chunk of distance to the trajectory and a gate + launch
//    boundary (~5 us) per chunk.
//    Until round 3 the first form was limited to 256 steps: with the single-wave trajectory kernel (2.6 us per step,
//    slower than most rate kernels) and ordinary stores the chunk form was ahead beyond that.  With the four-wave
//    trajectory kernel and nontemporal stores the one-kernel form is level or ahead at every length and population
//    size measured [MI355X, tools/form_probe.py, 4096 agents, 1024 / 4096 steps: n = 64: 5.0 / 5.5 G agent-steps/s in
//    both forms; n = 384: 3.47 vs 2.57 G; n = 1024: 1.36 vs 1.15 G at 1024 steps, level at 4096].
//    A PERSISTENT rate kernel was built first (three versions) and removed: waves that stay resident hold store credits
//    and lose 9-12 % of the store bandwidth against freshly dispatched ones (tools/stream_bench.hip).
// Forced (imported) trajectories have no recurrence to hide: their kernel and the populations' kernels simply follow
// each other on the caller's stream.
#include <hip/hip_ext.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "riab_agent_kernel.h"

namespace riab {
int launch_agent_pub(const AgentArgs& a, hipStream_t s, bool* state_published);
int launch_agent_forced(const AgentArgs& a, hipStream_t s);
int launch_agent_plain(const AgentArgs& a, hipStream_t s);
int stream_supported(const RiabEnv* env, const RiabPopulation* pop, int64_t B);
int launch_rate_stream(const RiabEnv* env, const RiabPopulation* pop, const float* hist, int64_t B, int32_t T, float dt,
                       uint64_t seed, uint64_t step0, int64_t agent_id0, uint32_t* ctrl, uint32_t spin_limit, bool stamps,
                       hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, bool dry_run, bool reserve,
                       uint32_t serial_rows);
int launch_stream_gate(uint32_t* ctrl, uint32_t started_target, uint32_t n_traj, uint32_t progress_target,
                       uint32_t spin_limit, bool sleep_long, uint32_t final_target, hipStream_t s);
int launch_stream_open(uint32_t* ctrl, uint32_t n_traj, uint32_t step_base, hipStream_t s);
}  // namespace riab

struct RiabStreamer {
  hipStream_t side;          // the trajectory kernel's stream (SIDE_STREAM 0): BORROWED from the process-wide pool of screened
                             // streams (side_stream_for), for the caller's stream `side_main`; `side_screened`: it has been timed
  hipStream_t side_main;
  bool side_screened;
  int dev;
  int side_pair_ns, side_ref_ns, side_rejected;   // what the screening measured (riab_streamer_info 4 / 5 / 6)
  hipStream_t side_normal;   // ... at the default priority (created when RIAB_STREAMER_OPT_SIDE_STREAM = 1 is first used)
  int side_mode;             // RIAB_STREAMER_OPT_SIDE_STREAM
  // form selection (populations form against chunk form): a trajectory step next to the rate stage / the lead's store rate
  int64_t step_ns_cfg, lead_mbps_cfg;    // configured (0: measure)
  int64_t step_ns_meas, lead_mbps_meas;  // measured from the stamps of an earlier call (0: not yet)
  bool calibrated;                       // the one blocking read of the stamps has been made (or given up)
  // what the last call left for that measurement: its ctrl block, rows, walls, the lead's bytes per row (0: no lead)
  uint32_t* prev_ctrl;
  int32_t prev_T, prev_walls;
  int64_t prev_lead_row_bytes;
  uint32_t progress_end;     // (uint32) step0 + T of the last call on prev_ctrl: a later call must start at or beyond it
  bool progress_valid;
  int last_launches;         // kernel launches of the last call
  int64_t late_calls;        // calls whose rate stage the HOST launched late (> 12 us after the trajectory kernel's launch had
                             // returned: a descheduled thread, a kernel's first launch): such a call can find every row
                             // published without any hardware queue being shared — the RIAB_CTRL_SERIALISED count is read
                             // against this one (riab_streamer_info 7)
  hipEvent_t fork, join;     // caller's stream -> side (only when the caller's stream is busy), side -> caller's stream
  hipEvent_t t0, t1;         // HIP-event timing of the rate kernel (created on first use)
  std::vector<hipEvent_t> pairs;  // chunk form: (start, stop) around every launch of the timed population
  int n_pairs;               // pairs recorded by the last call (0: t0 / t1 or the device stamps hold the measurement)
  int timed;                 // 0 nothing, 1 events, 2 device time stamps (ctrl[RIAB_CTRL_STAMPS])
  uint32_t* stamp_ctrl;      // control block of the last stamped call
  uint32_t started_total;    // trajectory workgroups launched so far through this object (wraps like the device word)
  int wall_khz;              // rate of the device's constant clock (s_memrealtime)
  int gate_mode;             // RIAB_STREAMER_OPT_GATE
  int poll_max;              // RIAB_STREAMER_OPT_POLL_MAX
  int head_rows;             // RIAB_STREAMER_OPT_HEAD_ROWS
  int last_form;             // riab_streamer_last_form
  // the strict mode (riab_hip.h "Two modes")
  int strict;                // RIAB_STREAMER_OPT_STRICT
  hipStream_t side_own;      // the trajectory kernel's stream in strict calls: the streamer's own, made by riab_streamer_warmup
  bool resync;               // a strict call has re-based the announcement counter: the next call of either mode opens with
                             // stream_open_kernel as well
  bool captured_once;        // ... and every later call does, once a strict call has been CAPTURED (it may be replayed any time)
  uint32_t spin_limit;       // RIAB_STREAMER_OPT_SPIN_LIMIT: polls before a waiting wave / gate gives up (0: the defaults)
  int last_strict;           // the last call ran in strict mode (riab_streamer_info 8)
};

// ---- the trajectory kernel's stream: screened, process-wide -----------------------------------------------------------
// Not every HIP stream is as good as the next.  The runtime multiplexes a process's streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES), and a stream that lands on the CALLER's queue costs every pair of launches on the two streams
// ~50 us: tools/queue_probe.hip — a tiny kernel on the null stream + one on the new stream + synchronise: 21 us on most
// streams, 52-75 us on every third or fourth one created; simulate(20) at the bench shape then takes 124 us instead
// of 80 although its kernels overlap on the device as usual (tools/slow_mode_probe.py) [MI355X, ROCm 7.2].  Which streams
// are slow depends on how many the process has created and kept, at either priority.  So the second stream is not
// simply created: candidates (highest priority: queues apart from the default-priority streams' in most processes,
// DESIGN.md 7) are TIMED against the caller's stream — the same pair of launches, best of six — and the first one
// within 20 us of two launches on the caller's stream alone is kept; rejected candidates stay allocated (destroying
// one hands its queue to the next candidate).  One screened stream per (device, caller's stream) for the whole process:
// every streamer of that caller borrows it (calls on one caller's stream are ordered anyway).  Screening needs the
// caller's stream idle (it synchronises it); a first call that finds it busy borrows an unscreened stream and the
// screening happens at the first idle call.
// The pool is BOUNDED: at most SIDE_POOL_CAP caller streams per device have an entry.  One more caller's stream takes over
// the least recently used entry with its stream (unscreened for the new pair; a streamer still holding that stream for
// its old caller keeps working: a HIP stream shared by two callers is still an ordered queue, the two only queue behind
// each other there), and once PARKED_CAP rejected candidates are parked they are tried again for other callers instead of
// creating more.  A process that makes a new torch stream per task therefore holds a few dozen HIP streams, not six per
// stream it ever used.
namespace {
constexpr size_t SIDE_POOL_CAP = 8, PARKED_CAP = 24;
struct SideEntry {
  int dev;
  hipStream_t main, side;
  bool screened;
  int pair_ns, ref_ns, rejected;
  uint64_t last_use;
};
uint64_t g_side_tick = 0;
std::mutex g_side_mu;
std::vector<SideEntry> g_side;
std::vector<hipStream_t> g_parked;  // rejected candidates: kept for the life of the process
uint32_t* g_scratch[64] = {nullptr};

double host_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// a one-wave kernel that returns at once (started_target 0 against zeroed words)
void tiny_launch(uint32_t* scratch, hipStream_t s) { (void)riab::launch_stream_gate(scratch, 0u, 0u, 0u, 1u, false, 0u, s); }

// best of `n`: [one launch on `a`, one on `b`, synchronise both], microseconds
double pair_us(uint32_t* scratch, hipStream_t a, hipStream_t b, int n) {
  double best = 1e30;
  for (int k = 0; k < n + 2; ++k) {
    const double t0 = host_us();
    tiny_launch(scratch, a);
    tiny_launch(scratch, b);
    (void)hipStreamSynchronize(b);
    (void)hipStreamSynchronize(a);
    const double t = host_us() - t0;
    if (k >= 2 && t < best) best = t;  // (two warm-up rounds: a stream's first launches set its queue up)
  }
  return best;
}

hipStream_t new_candidate() {
  if (g_parked.size() >= PARKED_CAP) {  // (set aside for another caller's stream: may do for this one)
    hipStream_t s = g_parked.front();
    g_parked.erase(g_parked.begin());
    return s;
  }
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) != hipSuccess) return nullptr;
  return s;
}

// the screened stream for (dev, main_s); `idle`: the caller's stream is idle, screening may synchronise it
bool side_stream_for(RiabStreamer* h, hipStream_t main_s, bool idle) {
  std::lock_guard<std::mutex> lock(g_side_mu);
  SideEntry* e = nullptr;
  for (SideEntry& x : g_side)
    if (x.dev == h->dev && x.main == main_s) e = &x;
  if (!e) {
    size_t on_dev = 0;
    SideEntry* lru = nullptr;
    for (SideEntry& x : g_side) {
      if (x.dev != h->dev) continue;
      on_dev += 1;
      if (!lru || x.last_use < lru->last_use) lru = &x;
    }
    if (on_dev >= SIDE_POOL_CAP && lru) {  // bounded pool: the least recently used caller's entry, stream included
      lru->main = main_s;
      lru->screened = false;
      lru->pair_ns = lru->ref_ns = lru->rejected = 0;
      e = lru;
    } else {
      g_side.push_back(SideEntry{h->dev, main_s, nullptr, false, 0, 0, 0, 0});
      e = &g_side.back();
    }
  }
  e->last_use = ++g_side_tick;
  int cur_dev = -1;
  (void)hipGetDevice(&cur_dev);   // (screening allocates and launches on the CURRENT device: only when that is the streamer's)
  if (!e->screened && idle && cur_dev == h->dev && h->dev >= 0 && h->dev < 64) {
    uint32_t*& scratch = g_scratch[h->dev];
    if (!scratch) {
      if (hipMalloc((void**)&scratch, 256) != hipSuccess || hipMemset(scratch, 0, 256) != hipSuccess) scratch = nullptr;
    }
    if (scratch) {
      const double ref = pair_us(scratch, main_s, main_s, 6);
      hipStream_t best_s = e->side;  // (an unscreened stream borrowed by an earlier, busy call is the first candidate)
      double best = 1e30;
      int rejected = 0;
      for (int c = 0; c < 6; ++c) {
        hipStream_t cand = (c == 0 && e->side) ? e->side : new_candidate();
        if (!cand) break;
        const double t = pair_us(scratch, main_s, cand, 6);
        if (t < best) {
          if (best_s && best_s != cand) g_parked.push_back(best_s);
          best = t;
          best_s = cand;
        } else {
          g_parked.push_back(cand);
        }
        if (t <= ref + 20.0) break;
        ++rejected;
      }
      if (best_s) {
        e->side = best_s;
        e->screened = true;
        e->pair_ns = (int)(best * 1e3);
        e->ref_ns = (int)(ref * 1e3);
        e->rejected = rejected;
      }
    }
  }
  if (!e->side) e->side = new_candidate();
  if (!e->side) return false;
  h->side = e->side;
  h->side_main = main_s;
  h->side_screened = e->screened;
  h->side_pair_ns = e->pair_ns;
  h->side_ref_ns = e->ref_ns;
  h->side_rejected = e->rejected;
  return true;
}
}  // namespace

extern "C" RiabStreamer* riab_streamer_create(void) {
  RiabStreamer* h = new (std::nothrow) RiabStreamer();
  if (!h) return nullptr;
  h->side = h->side_normal = nullptr;
  h->side_main = nullptr;
  h->side_screened = false;
  h->dev = 0;
  h->side_pair_ns = h->side_ref_ns = h->side_rejected = 0;
  h->side_mode = 0;
  h->step_ns_cfg = h->lead_mbps_cfg = h->step_ns_meas = h->lead_mbps_meas = 0;
  h->calibrated = false;
  h->prev_ctrl = nullptr;
  h->prev_T = h->prev_walls = 0;
  h->prev_lead_row_bytes = 0;
  h->progress_end = 0;
  h->progress_valid = false;
  h->last_launches = 0;
  h->late_calls = 0;
  h->fork = h->join = h->t0 = h->t1 = nullptr;
  h->timed = 0;
  h->n_pairs = 0;
  h->stamp_ctrl = nullptr;
  h->started_total = 0;
  h->gate_mode = RIAB_GATE_RESERVED;
  h->poll_max = 65535;
  h->head_rows = 256;
  h->last_form = RIAB_FORM_NONE;
  h->strict = 2;
  h->side_own = nullptr;
  h->resync = false;
  h->captured_once = false;
  h->spin_limit = 0u;
  h->last_strict = 0;
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipEventCreateWithFlags(&h->join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->fork, hipEventDisableTiming) != hipSuccess) {
    riab_streamer_destroy(h);
    return nullptr;
  }
  h->dev = dev;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  h->wall_khz = khz;
  return h;
}

extern "C" int riab_streamer_configure(RiabStreamer* h, int32_t option, int32_t value) {
  if (!h) return RIAB_EINVAL;
  switch (option) {
    case RIAB_STREAMER_OPT_GATE:
      if (value != RIAB_GATE_ALWAYS && value != RIAB_GATE_WHEN_BUSY && value != RIAB_GATE_RESERVED) return RIAB_EINVAL;
      h->gate_mode = value;
      return RIAB_OK;
    case RIAB_STREAMER_OPT_SIDE_STREAM:
      if (value < 0 || value > 2) return RIAB_EINVAL;
      if (value == 1 && !h->side_normal && hipStreamCreateWithFlags(&h->side_normal, hipStreamNonBlocking) != hipSuccess)
        return RIAB_EINVAL;
      h->side_mode = value;
      return RIAB_OK;
    case RIAB_STREAMER_OPT_STEP_NS:
      if (value < 0) return RIAB_EINVAL;
      h->step_ns_cfg = value;
      return RIAB_OK;
    case RIAB_STREAMER_OPT_LEAD_MBPS:
      if (value < 0) return RIAB_EINVAL;
      h->lead_mbps_cfg = value;
      return RIAB_OK;
    case RIAB_STREAMER_OPT_POLL_MAX:
      if (value < 0 || value > 65535) return RIAB_EINVAL;
      h->poll_max = value;
      return RIAB_OK;
    case RIAB_STREAMER_OPT_HEAD_ROWS:
      if (value < 1 || value > 65535) return RIAB_EINVAL;
      h->head_rows = value;
      return RIAB_OK;
    case RIAB_STREAMER_OPT_STRICT:
      if (value < 0 || value > 2) return RIAB_EINVAL;
      h->strict = value;
      return RIAB_OK;
    case RIAB_STREAMER_OPT_SPIN_LIMIT:
      if (value < 0) return RIAB_EINVAL;
      h->spin_limit = (uint32_t)value;
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
  if (h->fork) (void)hipEventDestroy(h->fork);
  // (h->side belongs to the process-wide pool: side_stream_for)
  if (h->side_normal) (void)hipStreamDestroy(h->side_normal);
  if (h->side_own) (void)hipStreamDestroy(h->side_own);
  delete h;
}

extern "C" int riab_streamer_last_form(RiabStreamer* h) { return h ? h->last_form : RIAB_FORM_NONE; }

// ---- what the choice "populations form or chunk form" compares ----------------------------------------------------------
// The lead's stores of a row must take at least 1.5 x as long as a trajectory step next to it, or the row-following
// kernel sits waiting for rows while nothing else runs.  Built-in figures [MI355X]: a trajectory step takes 0.9 us in
// an open room + 0.25 us per further wall (cfg 2 / cfg 5 next to their rate stage: 0.85-1.1 us, cfg 3's nine walls
// 2.05); a store-bound population writes 6.5 TB/s.  They are the fallback: the streamer MEASURES both on the chip it
// runs on from the device-clock stamps every call leaves in its control block (trajectory workgroup 0's start / last
// publication; the row-following kernel's first-wave start / last-wave end) — read once, by the first call with
// several populations that finds the caller's stream idle (the earlier call is then known to have finished), and kept
// as a factor on the built-in step time (rooms differ in walls, chips in clocks).
static double builtin_step_ns(int n_walls) { return 900.0 + 250.0 * (n_walls > 4 ? n_walls - 4 : 0); }

static void calibrate_from_stamps(RiabStreamer* h) {
  if (!h->prev_ctrl || h->prev_T < 8) return;  // (nothing to read yet: try again at the next call)
  h->calibrated = true;                         // one blocking copy per streamer, not one per call
  unsigned long long st[4] = {0, 0, 0, 0};      // STAMPS (2 x u64) and TRAJ_STAMPS (2 x u64): words 8 .. 15
  static_assert(RIAB_CTRL_TRAJ_STAMPS == RIAB_CTRL_STAMPS + 4, "the two stamp pairs are read with one copy");
  if (hipMemcpy(st, h->prev_ctrl + RIAB_CTRL_STAMPS, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return;
  const double tick_ns = 1e6 / (double)h->wall_khz;
  if (st[3] > st[2]) {
    const double step = (double)(st[3] - st[2]) * tick_ns / (double)h->prev_T;
    if (step > 50.0 && step < 1e6) h->step_ns_meas = (int64_t)(step * 1000.0 / builtin_step_ns(h->prev_walls));  // per mille
  }
  if (h->prev_lead_row_bytes > 0 && st[1] > st[0]) {
    const double ns = (double)(st[1] - st[0]) * tick_ns;
    const double mbps = (double)h->prev_lead_row_bytes * (double)h->prev_T / ns * 1e3;  // bytes / ns = GB/s; x 1e3 = MB/s
    if (mbps > 1e4 && mbps < 2e7) h->lead_mbps_meas = (int64_t)mbps;
  }
}

static double step_ns_now(const RiabStreamer* h, int n_walls) {
  if (h->step_ns_cfg) return (double)h->step_ns_cfg;
  const double b = builtin_step_ns(n_walls);
  return h->step_ns_meas ? b * (double)h->step_ns_meas / 1000.0 : b;
}
static double lead_mbps_now(const RiabStreamer* h) {
  return h->lead_mbps_cfg ? (double)h->lead_mbps_cfg : (h->lead_mbps_meas ? (double)h->lead_mbps_meas : 6.5e6);
}

extern "C" int64_t riab_streamer_info(RiabStreamer* h, int32_t which) {
  if (!h) return -1;
  switch (which) {
    case 0: return (int64_t)step_ns_now(h, 4);
    case 1: return (int64_t)lead_mbps_now(h);
    case 2: return (h->step_ns_cfg == 0 && h->step_ns_meas != 0) ? 1 : 0;
    case 3: return h->last_launches;
    case 4: return h->side_screened ? h->side_pair_ns : -1;
    case 5: return h->side_rejected;
    case 6: return h->side_ref_ns;
    case 7: return h->late_calls;
    case 8: return h->last_strict;
    case 9: return h->side_own ? 1 : 0;
    case 10: {  // HIP streams the process-wide pool holds on this streamer's device (entries' + parked)
      std::lock_guard<std::mutex> lock(g_side_mu);
      int64_t n = (int64_t)g_parked.size();
      for (const SideEntry& x : g_side) n += (x.dev == h->dev && x.side) ? 1 : 0;
      return n;
    }
    default: return -1;
  }
}

extern "C" float riab_streamer_last_rate_ms(RiabStreamer* h) {
  if (!h || !h->timed) return -1.0f;
  float ms = -1.0f;
  if (h->timed == 2) {  // first wave's start / last wave's end of the gated rate kernel, on the device's constant clock
    unsigned long long st[2] = {0, 0};
    if (hipMemcpy(st, h->stamp_ctrl + RIAB_CTRL_STAMPS, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return -1.0f;
    if (st[1] <= st[0]) return -1.0f;
    return (float)((double)(st[1] - st[0]) / (double)h->wall_khz);
  }
  if (h->n_pairs > 0) {  // the sum over the timed population's launches of the last chunk-form call
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

static int check_populations(const RiabPopulation* pops, int32_t n_pops, int32_t T) {
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
      default: return RIAB_EUNSUPPORTED;  // (velocity cells read the float64 state: they advance through a step plan)
    }
  }
  return RIAB_OK;
}

// chunks of rows of the chunk form: a chunk should be finished by the trajectory when the stream gets to its gate.  In a
// solid rectangular room the trajectory pulls away from the rate kernels (which need ~2.7 us per row at cfg 2) and a
// chunk may be ~1.5x its predecessor; in other rooms ~1.25x is the most (a 16, 16, 32, 64, 128 ramp then spent 230 us
// of a 1024-step run inside the gates [MI355X, rocprofv3 trace]).  Larger chunks are cheaper per row (3.6 us at 16
// rows, 2.7 at 128) and every chunk costs a gate (~5 us).
static std::vector<int32_t> chunk_schedule(const RiabEnv* env, int32_t T) {
  const bool box_room = !env->polygon && !env->hole_mask && !env->periodic && env->n_walls >= 4;
  static const int32_t ramp_fast[5] = {16, 28, 44, 64, 96};
  static const int32_t ramp_slow[10] = {16, 16, 20, 24, 32, 40, 48, 64, 80, 96};
  std::vector<int32_t> sched;
  for (int32_t t0 = 0, k = 0; t0 < T; ++k) {
    int32_t tc = box_room ? (k < 5 ? ramp_fast[k] : 128) : (k < 10 ? ramp_slow[k] : 128);
    if (tc > T - t0 || T - t0 - tc < 32) tc = T - t0;  // (no sliver at the end)
    sched.push_back(tc);
    t0 += tc;
  }
  return sched;
}

extern "C" int riab_watch_compare(const RiabWatch* watch, int32_t n) {
  if (n < 0 || (n > 0 && !watch)) return RIAB_EINVAL;
  for (int32_t i = 0; i < n; ++i) {
    const RiabWatch& w = watch[i];
    if (w.bytes < 0 || (w.bytes > 0 && (!w.live || !w.snapshot))) return RIAB_EINVAL;
    if (w.bytes > 0 && memcmp(w.live, w.snapshot, (size_t)w.bytes) != 0) return RIAB_ECHANGED;
  }
  return RIAB_OK;
}

// Everything riab_simulate may have to set up, done ahead of it (riab_hip.h "Two modes"): the strict mode's own second
// stream, the screening of the default mode's second stream for `stream` (which synchronises it), the events a timed
// call records.  Allocates and synchronises — that is its purpose; calls after it do neither.
extern "C" int riab_streamer_warmup(RiabStreamer* h, riab_stream_t stream) {
  if (!h) return RIAB_EINVAL;
  hipStream_t main_s = (hipStream_t)stream;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(main_s, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) return RIAB_EUNSUPPORTED;
  if (!h->side_own) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&h->side_own, hipStreamNonBlocking, greatest) != hipSuccess) {
      h->side_own = nullptr;
      return RIAB_EINVAL;
    }
  }
  if (!h->t0 && (hipEventCreate(&h->t0) != hipSuccess || hipEventCreate(&h->t1) != hipSuccess)) return RIAB_EINVAL;
  while (h->pairs.size() < 64) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return RIAB_EINVAL;
    h->pairs.push_back(e);
  }
  if (hipStreamSynchronize(main_s) != hipSuccess) return RIAB_EINVAL;
  if (h->side_mode == 0 && !side_stream_for(h, main_s, true)) return RIAB_EINVAL;
  return RIAB_OK;
}

extern "C" int riab_simulate(RiabStreamer* h, const RiabSimulate* q, riab_stream_t stream) {
  if (!h || !q || !q->env || !q->motion || !q->ctrl || !q->hist || q->n_pops < 0 || (q->n_pops > 0 && !q->pops) || q->T <= 0)
    return RIAB_EINVAL;
  const RiabEnv* env = q->env;
  const RiabPopulation* pops = q->pops;
  const int32_t T = q->T, n_pops = q->n_pops;
  const int64_t B = q->B;
  if (B <= 0 || B % 4 != 0) return RIAB_EALIGN;
  if (q->forced_pos && (q->noise || q->drift)) return RIAB_EINVAL;
  // the caller's cached tables against the host arrays they were built from (RiabSimulate.watch): before anything else
  int rc = riab_watch_compare(q->watch, q->n_watch);
  if (rc) return rc;
  rc = check_populations(pops, n_pops, T);
  if (rc) return rc;
  riab::AgentArgs a;
  rc = riab::fill_agent_args(a, env, q->motion, q->state, B, q->agent_id0, q->drift, q->noise, nullptr, q->forced_pos, q->seed,
                             q->step0, T, q->hist, q->diag, q->resample_pos);
  if (rc) return rc;
  a.ctrl = q->ctrl;
  hipStream_t main_s = (hipStream_t)stream;
  const float dt = (float)q->motion->dt;
  // Two modes (riab_hip.h).  STRICT — asked for (RIAB_STREAMER_OPT_STRICT) or imposed by a stream that is being captured —
  // allocates nothing, synchronises nothing, queries nothing and touches nothing outside this streamer: its own second
  // stream (riab_streamer_warmup), an opening kernel that re-bases the control words, the two streams joined by the
  // streamer's events at both ends, the started gate always.  (The check comes first: a capturing stream must not be
  // queried by anything below.)
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(main_s, &cap_status) == hipSuccess && cap_status != hipStreamCaptureStatusNone;
  // (RIAB_STREAMER_OPT_STRICT = 2, the default: strict for runs of more than 256 steps — where its two extra launches are
  // 0.2 % of the call [MI355X, cfg 2, 1024 steps: 1.423 against 1.426 G agent-steps/s] — once riab_streamer_warmup has made
  // the streamer's own stream; the mode tuned for one short call per synchronisation otherwise: 846 against 1016 M at the
  // 20 steps of the driver's command)
  // (a call with several populations chooses its form from a step time the DEFAULT mode measures once per streamer —
  // calibrate_from_stamps —: until that has happened such a call stays in the default mode)
  const bool strict = h->strict == 1 || capturing ||
                      (h->strict == 2 && T > 256 && h->side_own != nullptr && (n_pops <= 1 || h->calibrated || h->step_ns_cfg != 0));
  if (strict && !h->side_own) return RIAB_EUNSUPPORTED;  // (riab_streamer_warmup makes it; nothing was launched)
  bool timing = q->timed_pop >= 0 && q->timed_pop < n_pops && !capturing;  // (no timing events inside a capture)
  h->timed = 0;
  h->n_pairs = 0;
  h->last_strict = strict ? 1 : 0;

  h->last_form = RIAB_FORM_NONE;
  h->last_launches = 0;
  // ---- an agent without populations: the trajectory alone, on the caller's stream -----------------------------------
  if (n_pops == 0) {
    h->last_launches = 1;
    return riab::launch_agent_plain(a, main_s);
  }

  // ---- forced positions: no recurrence, nothing to overlap: one stream, kernel after kernel --------------------------
  if (q->forced_pos) {
    rc = riab::launch_agent_forced(a, main_s);
    if (rc) return rc;
    h->last_form = RIAB_FORM_SERIAL;
    int fail = RIAB_OK;
    for (int32_t t0 = 0; t0 < T && !fail; t0 += 4096) {  // (time rows are a grid axis of the rate kernels)
      const int32_t tc = T - t0 < 4096 ? T - t0 : 4096;
      for (int i = 0; i < n_pops && !fail; ++i)
        fail = launch_pop_rows(env, pops, i, q->hist, B, t0, tc, dt, q->seed, q->step0, q->agent_id0, main_s);
    }
    return fail ? RIAB_EPARTIAL : RIAB_OK;
  }

  // ---- which form of the rate stage; every argument check before the first launch ------------------------------------
  // ~0.3 us per poll: a generous second or two before a wait gives up (a healthy wait is tens of microseconds)
  const uint32_t spin_limit = h->spin_limit ? h->spin_limit : 1u << 20;
  auto gate_limit = [&](uint32_t dflt) { return h->spin_limit ? h->spin_limit : dflt; };
  std::vector<int32_t> sched;
  // `lead`: the population whose kernel follows the trajectory row by row (rate_kernel_gated: store-bound cells without
  // OU noise, whole 256-agent groups).  Alone it is the one-kernel form.  With other populations beside it, it runs
  // first — when it ends every row has been published — and the others follow as their ordinary kernels over the whole
  // run, in list order (a layer's inputs are earlier entries, or the lead): no gate and launch boundary per chunk, no
  // short first chunks (bvc_kernel: 22 us per row in a 16-row chunk, 16.7 in a 104-row one [MI355X, cfg 3]).  That
  // needs the lead's stores of a row (at 6.5 TB/s) to take at least as long as a step of the trajectory kernel next to
  // it (0.9 us in an open room, + 0.25 us per further wall, x 1.5 next to a kernel that saturates HBM [MI355X: cfg 2 /
  // cfg 5 0.85-1.1 us per step, cfg 3's nine walls 2.05]); otherwise its waves would sit waiting for rows while nothing
  // else runs, and the chunk form keeps the trajectory hidden behind ALL the populations' work.  [MI355X, 1024 steps:
  // cfg 5 158-161 -> 167-168 M agent-steps/s, bvc_kernel 38.4 -> 33.8 ms; cfg 3 (16.8 MB of GridCells per row against
  // nine walls) would lose 3 % and keeps the chunks.]
  int lead = -1;
  if (T <= h->poll_max) {
    if (n_pops == 1) {
      if (riab::stream_supported(env, &pops[0], B) == RIAB_OK) lead = 0;
    } else {
      int64_t best = 0;
      for (int i = 0; i < n_pops; ++i) {
        const int64_t bytes = (int64_t)pops[i].n * B * (pops[i].spikes_base ? 5 : 4);
        if (pops[i].kind != RIAB_POP_FF && bytes > best && riab::stream_supported(env, &pops[i], B) == RIAB_OK) {
          best = bytes;
          lead = i;
        }
      }
      // (the lead's stores of a row against 1.5 trajectory steps: measured on this chip once an earlier call's stamps can
      // be read without waiting — the caller's stream is idle —, the MI355X constants until then: builtin_step_ns)
      if (lead >= 0) {
        // (never in strict mode; and only when the earlier call ran on THIS stream: its idleness then says that call is over)
        if (!strict && !h->calibrated && h->step_ns_cfg == 0 && h->side_main == main_s && hipStreamQuery(main_s) == hipSuccess)
          calibrate_from_stamps(h);
        const double row_ns = (double)best / lead_mbps_now(h) * 1e3;  // bytes / (MB/s) = us; x 1e3 = ns
        // x 1.5: next to a kernel that saturates HBM the trajectory kernel's clock and its write-throughs slow down by that
        // much (cfg 2 / cfg 5: 0.85-1.1 us per step measured next to the stores against the formula's 1.35; cfg 3: 2.05
        // against 3.2).  x 2 for a MEASURED step: it was taken next to whatever the earlier call ran — in the chunk form
        // the arithmetic-bound boundary-vector kernel, which slows the trajectory less than a store stream does — and
        // includes a short call's start-up [MI355X: cfg 3's 32-step warm-up gives 1.6 us per step, not 2.05].
        const double margin = (h->step_ns_cfg == 0 && h->step_ns_meas != 0) ? 2.0 : 1.5;
        if (row_ns < margin * step_ns_now(h, env->n_walls)) lead = -1;
      }
    }
  }
  // Very long runs (more than 2048 rows): the row-following kernel serves the first `head_rows` rows — by then the
  // trajectory kernel (0.8-1.1 us per step) is far ahead of the rate stage (2.5 us per row at cfg 2) — and the rest of
  // the lead's rows go through its ordinary kernel, `tail_piece` rows per launch behind a progress gate that the
  // trajectory has long passed: no poll and no agent-scope position loads per workgroup.  [MI355X, cfg 2, all rows by
  // the row-following kernel vs head + pieces: 1024 steps 1.38-1.40 vs 1.37 G agent-steps/s, 2048 level, 3072 1.39-1.41
  // vs 1.45, 4096 1.34-1.38 vs 1.43-1.46, 8192 1.38-1.44 vs 1.49; cfg 4 at 1024 steps 0.37-0.39 vs 0.36 G.]
  const int32_t tail_piece = 512;
  const int32_t head = lead < 0 ? 0 : ((T <= 2048 || T <= h->head_rows) ? T : h->head_rows);
  const bool pure = lead >= 0 && n_pops == 1 && head == T;  // exactly one kernel: device stamps / launch events time it
  h->last_form = lead < 0 ? RIAB_FORM_CHUNKS
                          : (n_pops > 1 ? RIAB_FORM_POPULATIONS : (pure ? RIAB_FORM_ONE_KERNEL : RIAB_FORM_HEAD_AND_PIECES));
  const int32_t rest_piece = 1024;  // rows per launch of another population (their kernels' grids, the noise pass's loop)
  bool reservable = false;
  if (lead >= 0) {
    rc = riab::launch_rate_stream(env, &pops[lead], q->hist, B, head, dt, q->seed, q->step0, q->agent_id0, q->ctrl, spin_limit,
                                  false, main_s, nullptr, nullptr, /*dry_run=*/true, false, 0u);
    if (rc) return rc;
    if (pure && h->gate_mode == RIAB_GATE_RESERVED)
      reservable = riab::launch_rate_stream(env, &pops[lead], q->hist, B, head, dt, q->seed, q->step0, q->agent_id0, q->ctrl,
                                            spin_limit, false, main_s, nullptr, nullptr, /*dry_run=*/true, true, 0u) == RIAB_OK;
    // (a strict call creates nothing: it is timed with the events riab_streamer_warmup made, or not at all)
    if (pure && q->timing_mode == RIAB_TIMING_EVENTS && timing && !h->t0) {
      if (strict) timing = false;
      else if (hipEventCreate(&h->t0) != hipSuccess || hipEventCreate(&h->t1) != hipSuccess) return RIAB_EINVAL;
    }
    if (timing && !pure) {
      const size_t need = 2 * (size_t)(1 + (T + rest_piece - 1) / rest_piece);
      if (strict && h->pairs.size() < need) timing = false;
      while (timing && h->pairs.size() < need) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return RIAB_EINVAL;
        h->pairs.push_back(e);
      }
    }
  } else {
    sched = chunk_schedule(env, T);
    if (timing) {
      if (strict && h->pairs.size() < 2 * sched.size()) timing = false;
      while (timing && h->pairs.size() < 2 * sched.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return RIAB_EINVAL;
        h->pairs.push_back(e);
      }
    }
  }

  // ---- the trajectory kernel, on the streamer's own stream -----------------------------------------------------------
  // It has to start after what the caller's stream holds (earlier calls' kernels wrote the state it reads).  When that
  // stream is idle — the case that matters for latency: one short call per synchronisation — there is nothing to wait
  // for and the launch goes out with no API call in front of it but the query; otherwise an event carries the order.
  const bool idle = strict ? false : hipStreamQuery(main_s) == hipSuccess;
  // (RIAB_STREAMER_OPT_SIDE_STREAM: 1 a stream at the default priority, 2 the caller's own stream — the two kernels then
  // run one after the other, which is what the RIAB_CTRL_SERIALISED diagnostic is tested with)
  if (!strict && h->side_mode == 0 && (h->side_main != main_s || !h->side || (!h->side_screened && idle))) {
    if (!side_stream_for(h, main_s, idle)) return RIAB_EINVAL;
  }
  const hipStream_t side_s = strict ? h->side_own
                                    : (h->side_mode == 2 ? main_s : (h->side_mode == 1 && h->side_normal ? h->side_normal : h->side));
  const uint32_t n_traj = (uint32_t)((B + 63) / 64);
  // The opening kernel: a strict call re-bases the announcement counter and this call's progress words on the device, so
  // that a replay of the captured call finds what the first run found; once that has happened, the host's mirror of the
  // counter means nothing any more and the next call of either mode re-bases as well.
  const bool open = strict || h->resync;
  if (open) {
    rc = riab::launch_stream_open(q->ctrl, n_traj, (uint32_t)q->step0, main_s);
    if (rc) return rc;
    h->started_total = 0;
    if (capturing) h->captured_once = true;
    h->resync = strict || h->captured_once;   // (a default-mode call has re-based: its successors count on from here)
  }
  if ((!idle || open) && side_s != main_s) {
    hipError_t e = hipEventRecord(h->fork, main_s);
    if (e == hipSuccess) e = hipStreamWaitEvent(side_s, h->fork, 0);
    if (e != hipSuccess) return (int)e;
  }
  // Progress words hold absolute step counts and every trajectory workgroup resets its own to step0 before it
  // announces itself.  A call that starts at or beyond the end of the last call on this block can never meet a word
  // above its own rows; any other call (the same block replayed, another ctrl block) must not let a consumer look at
  // the words before every workgroup has announced itself: it takes the started gate.
  const bool monotonic = h->progress_valid && h->prev_ctrl == q->ctrl && (int32_t)((uint32_t)q->step0 - h->progress_end) >= 0;
  bool state_published = false;
  rc = riab::launch_agent_pub(a, side_s, &state_published);
  if (rc) return rc;
  const double t_traj_launched = host_us();
  bool late_noted = false;
  auto note_first_rate_launch = [&]() {   // (after the first launch of the rate stage has returned)
    // (only calls that CAN be counted as serialised — eight rows or more — are counted as late: the two counts are
    // subtracted from each other)
    if (!late_noted && T >= 8 && host_us() - t_traj_launched > 12.0) ++h->late_calls;
    late_noted = true;
  };
  h->last_launches = open ? 2 : 1;
  h->started_total += n_traj;
  h->prev_ctrl = q->ctrl;
  h->prev_T = T;
  h->prev_walls = env->n_walls;
  h->prev_lead_row_bytes = 0;
  h->progress_end = (uint32_t)q->step0 + (uint32_t)T;
  h->progress_valid = true;
  const uint32_t serial_rows = T >= 8 ? (uint32_t)T : 0u;
  // From here on the state has advanced: a later failure still joins the two streams and is reported as RIAB_EPARTIAL
  // (the trajectory rows are complete, the rates of this call are not).
  int fail = RIAB_OK;
  if (lead >= 0) {
    // Both kernels must be resident at once or the rate waves spin for nothing (riab_hip.h "Residency"): either the
    // row-following kernel is launched in its reserving shape — twelve-wave workgroups, one wave slot per SIMD always
    // free for a trajectory workgroup: no gate — or a one-wave gate in front of it returns only once every trajectory
    // workgroup of THIS launch has announced itself: the rate waves can then never occupy the slots the kernel they wait
    // for still needs, whatever else runs on the device (another stream, another process: two ranks sharing one GPU ran
    // into the waits' time limit with neither).  RIAB_GATE_WHEN_BUSY drops both when the caller's stream was idle.
    const bool timed_lead = timing && q->timed_pop == lead;
    const bool events = pure && timed_lead && q->timing_mode == RIAB_TIMING_EVENTS;
    // (device-clock stamps of the row-following kernel: one store by the grid's first wave, one maximum by the sixteen
    // waves of the last row's last cell group — always on unless HIP events time the launch; they need no reset: the
    // clock only moves forward)
    const bool stamps = !events;
    // residency (riab_hip.h): the reserving shape of the row-following kernel needs no gate; everything else takes one —
    // unless the caller has said that it owns the device (RIAB_GATE_WHEN_BUSY) and its stream is idle
    const bool quiet = idle && !open && pure && monotonic && side_s != main_s;
    // (the reserving shape is only offered by kernels whose registers leave a trajectory workgroup room: launch_stream_cell)
    const bool reserve = quiet && h->gate_mode == RIAB_GATE_RESERVED && reservable;
    const bool gate = !(reserve || (quiet && h->gate_mode == RIAB_GATE_WHEN_BUSY)) && side_s != main_s;
    // (the started gate is one sleeping wave; it may have to sit out whatever runs in front of the trajectory kernel)
    if (gate) {
      fail = riab::launch_stream_gate(q->ctrl, h->started_total, 0, 0, gate_limit(1u << 24), true, 0u, main_s);
      ++h->last_launches;
    }
    int n_timed = 0;
    if (timed_lead && !pure) (void)hipEventRecord(h->pairs[0], main_s);
    if (!fail) {
      fail = riab::launch_rate_stream(env, &pops[lead], q->hist, B, head, dt, q->seed, q->step0, q->agent_id0, q->ctrl,
                                      spin_limit, stamps, main_s, events ? h->t0 : nullptr, events ? h->t1 : nullptr, false,
                                      reserve, serial_rows);
      ++h->last_launches;
      note_first_rate_launch();
      if (head == T) h->prev_lead_row_bytes = (int64_t)pops[lead].n * B * (pops[lead].spikes_base ? 5 : 4);
    }
    for (int32_t t0 = head; t0 < T && !fail; t0 += tail_piece) {
      const int32_t tc = T - t0 < tail_piece ? T - t0 : tail_piece;
      fail = riab::launch_stream_gate(q->ctrl, h->started_total, n_traj, (uint32_t)q->step0 + (uint32_t)(t0 + tc),
                                      gate_limit(1u << 22), false, 0u, main_s);
      if (!fail) fail = launch_pop_rows(env, pops, lead, q->hist, B, t0, tc, dt, q->seed, q->step0, q->agent_id0, main_s);
      h->last_launches += 2;
    }
    if (timed_lead && !pure) {
      (void)hipEventRecord(h->pairs[1], main_s);
      n_timed = 1;
    }
    // (every workgroup of the row-following kernel has waited for its row, the last gate of a tail for row T: when the
    // lead's last kernel ends, all T rows are in memory, and the kernel boundary makes them visible to ordinary loads)
    for (int i = 0; i < n_pops && !fail; ++i) {
      if (i == lead) continue;
      for (int32_t t0 = 0; t0 < T && !fail; t0 += rest_piece) {
        const int32_t tc = T - t0 < rest_piece ? T - t0 : rest_piece;
        const bool timed = timing && i == q->timed_pop;
        if (timed) (void)hipEventRecord(h->pairs[2 * n_timed], main_s);
        fail = launch_pop_rows(env, pops, i, q->hist, B, t0, tc, dt, q->seed, q->step0, q->agent_id0, main_s);
        if (timed) {
          (void)hipEventRecord(h->pairs[2 * n_timed + 1], main_s);
          ++n_timed;
        }
      }
    }
    if (!fail && timing) {
      if (pure) {
        h->timed = events ? 1 : 2;
        h->stamp_ctrl = q->ctrl;
      } else {
        h->n_pairs = n_timed;
        h->timed = n_timed > 0 ? 1 : 0;
      }
    }
  } else {
    // (the first progress gate also waits until every trajectory workgroup of this launch is resident — it may have to
    // sit out what is queued in front of the trajectory kernel — so there is no separate started gate)
    int32_t t0 = 0;
    for (size_t k = 0; k < sched.size() && !fail; ++k) {
      const int32_t tc = sched[k];
      // ~0.5 us per poll: seconds before a gate gives up (a healthy wait is one chunk of trajectory, < 1 ms)
      fail = riab::launch_stream_gate(q->ctrl, h->started_total, n_traj, (uint32_t)q->step0 + (uint32_t)(t0 + tc),
                                      gate_limit(k == 0 ? 1u << 24 : 1u << 22), false,
                                      (k == 0 && serial_rows) ? (uint32_t)q->step0 + (uint32_t)T : 0u, main_s);
      note_first_rate_launch();
      h->last_launches += 1 + n_pops;
      for (int i = 0; i < n_pops && !fail; ++i) {
        if (timing && i == q->timed_pop) (void)hipEventRecord(h->pairs[2 * k], main_s);
        fail = launch_pop_rows(env, pops, i, q->hist, B, t0, tc, dt, q->seed, q->step0, q->agent_id0, main_s);
        if (timing && i == q->timed_pop) (void)hipEventRecord(h->pairs[2 * k + 1], main_s);
      }
      t0 += tc;
    }
    if (!fail && timing) {
      h->n_pairs = (int)sched.size();
      h->timed = 1;
    }
  }
  // The caller's stream continues after the trajectory kernel as well (its state rows, its last history rows).  Every
  // form above ends with a kernel on the caller's stream that has waited for the trajectory kernel's LAST publication,
  // and the four-wave kernel makes that publication after its state stores have been written through (t4_store_state):
  // the stream order of the caller's stream already carries the dependency.  An event between the streams is only
  // needed for the two-wave kernel (RIAB_OPT_TRAJ_KERNEL = 2) — and after a failure, when nothing may have waited.
  // [MI355X, cfg 2, 20 steps: the event's record + wait were 4.7 us of host time and its barrier packet sat between the
  // rate kernel's end and the host's wake-up: 97 -> 89 us per call without them.]
  // When the caller's stream was BUSY at the call, host time is not what the call is short of: the event is recorded as
  // well — should a wait of the rate stage give up (abort flag), whatever follows on the caller's stream is then still
  // ordered behind the trajectory kernel.  (After an abort on an idle stream the host layer synchronises the device
  // before it clears the flags: Agent._check_pipeline.)
  hipError_t e = hipSuccess;
  if ((!state_published || fail || !idle || open) && side_s != main_s) {
    e = hipEventRecord(h->join, side_s);
    if (e == hipSuccess) e = hipStreamWaitEvent(main_s, h->join, 0);
  }
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
    case 7: return (int64_t)sizeof(RiabSimulate);
    case 8: return (int64_t)sizeof(RiabWatch);
    default: return -1;
  }
}
