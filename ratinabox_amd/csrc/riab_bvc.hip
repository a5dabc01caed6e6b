// BoundaryVectorCells.get_state for gfx950 (reference Neurons.py:1617-1778).
//
// rate[c][p] = (1/norm_c) * sum_k exp(-(d_k(p) - mu_c)^2 / 2 sigma_c^2) * exp(kappa_c (cos(theta_k - phi_c) - 1))
// where d_k(p) is the distance from position p to the first wall along test direction k.
// n*K exponentials per position (46 080 at n = 256, K = 180): the kernel is bound by
// transcendental issue, not HBM, and has no contraction for MFMA (the summand depends on
// all of c, k and p).
//
// Workgroup = 512 threads = 8 waves sharing one tile of 64 positions (lane = position):
//   stage A  all 8 waves cast the K rays of the tile against the walls in float64 (the
//            nearest-wall decision is discrete; walls and the 1/denominator table are scalar loads),
//            a ray and its opposite from one pair of wall intercepts where the table holds opposites,
//            and leave d[k][lane] (fp32) in LDS;
//   stage B  wave w owns the 4-cell groups g = w (mod 8); per group the K-loop reads d[k][lane]
//            (conflict-free ds_read_b32) and the wave-uniform angular table entry (scalar
//            load), one fused exponent per term: exp2(-(a d - a mu)^2 + T[c][k]).
#include <type_traits>
#include "riab_device.h"

#ifndef RIAB_BVC_XCH_KB
#define RIAB_BVC_XCH_KB 1  // pairs of rays per pass over the walls in a workgroup that exchanges its rays (see cast_rays)
#endif
namespace riab {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct BvcArgs {
  const float* pos_x;
  const float* pos_y;
  const float* hd_x;
  const float* hd_y;
  int64_t pos_ld;
  int64_t P;     // T * B positions
  int64_t B;
  float* rates;  // [T][n][B]
  uint8_t* spikes;
  const float* u_in;
  float dt, fr_scale, fr_min;
  uint32_t k0, k1, step0, tag;
  int64_t agent_id0;
  int n, K, Kp, n_walls;  // Kp = K rounded up to a multiple of 4 (table row stride, d-tile rows)
  const double* walls;      // [n_walls][4]
  const double* test_dirs;  // [K][2]
  const double* ray_rden;   // [K][n_walls] = 1 / (u_k x s_w)
  const float* cells;       // [4][n]
  const float* vm;          // [n][K] or [2][n][K]
  const float* inv_norm;    // [n]
  float* ray_out;           // [T][K][B] or null
  // direction windows (allocentric, K % 4 == 0): the table rows are given in an order that groups cells with
  // overlapping angular support; rows[i] = output row (cell index) of table row i, win[g] = (first direction,
  // number of directions) — both multiples of 4, wrapping modulo K — outside which all four cells of group g
  // have a von Mises weight below the caller's threshold.  NULL: identity order, every direction.
  const int* rows;
  const int* win;
  double e0, e1, e2, e3;  // extent
  int rect_room;          // solid rectangular boundary, no holes: the first four walls may be the room's own edges
  // ray exchange between the workgroups of a tile (gridDim.y > 1; a step plan's one-row launches): null = everybody casts
  // every ray.  xch: float [tiles][Kp][64]; xch_count: uint32 [tiles], counts arrivals over all launches; xch_target:
  // the count at which this launch's gridDim.y workgroups of a tile have all published their share
  float* xch;
  uint32_t* xch_count;
  uint32_t xch_target;
};

typedef const __attribute__((address_space(4))) double* const_f64_ptr;

// word `qi` of the Philox block held by lane J of this lane's aligned quad (DPP quad_perm broadcast; every lane of the
// wave must execute it)
template <int J>
__device__ __forceinline__ uint32_t quad_word(const u32x4& blk, int qi) {
  const uint32_t wx = (uint32_t)__builtin_amdgcn_mov_dpp((int)blk.x, J * 0x55, 0xF, 0xF, true);
  const uint32_t wy = (uint32_t)__builtin_amdgcn_mov_dpp((int)blk.y, J * 0x55, 0xF, 0xF, true);
  const uint32_t wz = (uint32_t)__builtin_amdgcn_mov_dpp((int)blk.z, J * 0x55, 0xF, 0xF, true);
  const uint32_t ww = (uint32_t)__builtin_amdgcn_mov_dpp((int)blk.w, J * 0x55, 0xF, 0xF, true);
  return qi == 0 ? wx : (qi == 1 ? wy : (qi == 2 ? wz : ww));
}

// LDS (dynamic): float d[Kp][64] — the tile's first-wall distances.  Walls, test directions and the
// 1/denominator table are wave-uniform: they are read from global memory with scalar loads.
template <bool EGO>
__global__ __launch_bounds__(512) void bvc_kernel(const BvcArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* s_d = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: loop counters and table indices stay scalar
  const int nw = a.n_walls, K = a.K;
  const const_f64_ptr walls = (const_f64_ptr)(const void*)a.walls;
  const const_f64_ptr dirs = (const_f64_ptr)(const void*)a.test_dirs;
  const const_f64_ptr rden = (const_f64_ptr)(const void*)a.ray_rden;

  // (the tile's workgroups have ids `tiles` apart — on one XCD when the tiles are a multiple of eight; giving them
  // neighbouring ids instead, so that they are dispatched together, measured slower: 41.4 -> 42.8 us per row at cfg 3)
  const int tile = (int)blockIdx.x, part = (int)blockIdx.y, nparts = (int)gridDim.y;
  const int64_t p = (int64_t)tile * 64 + lane;
  const bool live = p < a.P;
  const int64_t pc = live ? p : 0;
  const int64_t t = pc / a.B;
  const int64_t b = pc - t * a.B;
  const float pxf = a.pos_x[t * a.pos_ld + b], pyf = a.pos_y[t * a.pos_ld + b];
  const double px = pxf, py = pyf;

  // ---- stage A: first-wall distance along each test direction (Neurons.py:1655-1684, 1746-1778)
  // utils.vector_intercepts (utils.py:96-97) with sa = unit ray, sb = wall:
  //   l_a = (d0 . sb_p) / (sa . sb_p),  l_b = (-d0 . sa_p) / (sb . sa_p),  sb . sa_p = -(sa . sb_p)
  // Four test directions per pass over the walls: the wall-only quantities (d0, its cross product
  // with the wall) are computed once per wall and pass, leaving 1 + 3 multiply-adds per (ray, wall).
  // Box fast path.  When the first four walls are the edges of the rectangular room (checked here on the wall table
  // itself: axis-aligned, on the extent, spanning it) and every position of the tile is strictly inside it, a ray
  // leaves the room through the edge whose LINE it crosses first: the nearest positive l_a over the four edges is
  // the hit, and l_b — whether the crossing lies on the segment, half of the per-(ray, wall) arithmetic — need not
  // be computed for them.  Same l_a expression, same value; interior walls keep the full test after the edges.
  bool box4 = false;
  if (a.rect_room && nw >= 4) {
    bool ok = true;
    int nh = 0, nv = 0;
    const double tol = 1e-9 * ((a.e1 - a.e0) + (a.e3 - a.e2));
    for (int w = 0; w < 4; ++w) {
      const double ax = walls[4 * w], ay = walls[4 * w + 1], bx = walls[4 * w + 2], by = walls[4 * w + 3];
      if (ay == by && ax != bx) {
        ++nh;
        ok = ok && (ay == a.e2 || ay == a.e3) && fmin(ax, bx) <= a.e0 + tol && fmax(ax, bx) >= a.e1 - tol;
      } else if (ax == bx && ay != by) {
        ++nv;
        ok = ok && (ax == a.e0 || ax == a.e1) && fmin(ay, by) <= a.e2 + tol && fmax(ay, by) >= a.e3 - tol;
      } else {
        ok = false;
      }
    }
    ok = ok && nh == 2 && nv == 2;
    const bool inside = px > a.e0 && px < a.e1 && py > a.e2 && py < a.e3;
    box4 = ok && __builtin_amdgcn_ballot_w64(live && !inside) == 0;  // (wave-uniform; the same in every wave of the tile)
  }
  const int w_full = box4 ? 4 : 0;  // walls from here on take the full (l_a, l_b) test
  // Which rays THIS workgroup casts: all of them — or, where the tile's gridDim.y workgroups exchange their rays (a.xch),
  // every gridDim.y-th batch: `vw` of `nvw` virtual waves.  (The whole of stage A is the lambda `cast_rays`; a workgroup
  // whose partners do not show up casts the rest itself.)
  // (KB_: test directions / pairs of them per pass over the walls — four where a wave has many passes; a wave of an
  // exchanging workgroup has one or two pairs in all, and a pass of four computes two or three it then throws away:
  // [MI355X] the row with four / two / one per pass: cfg 3 (9 walls) 41.4 / 41.3 / 39.5 us, 64 walls 80.2 / 62.8 / 58.5)
  auto cast_rays = [&](auto KB_, const int vw, const int nvw, const bool publish) {
  constexpr int KB = decltype(KB_)::value;
  float* const xrow = publish ? a.xch + (int64_t)tile * a.Kp * 64 + lane : nullptr;
  auto put = [&](int k, float d) {
    s_d[k * 64 + lane] = d;
    if (publish) __hip_atomic_store((__attribute__((address_space(1))) float*)(uintptr_t)(xrow + k * 64), d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // rays without a partner (see below), `count` of them: table index 0 for t = 0, t + off otherwise
  auto cast_single = [&](int count, int off) {
    for (int t0 = vw; t0 < count; t0 += nvw * KB) {
      double ux[KB], uy[KB], best[KB], fallback[KB];
      int kk[KB];
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const int t = min(t0 + nvw * i, count - 1);  // (wave-uniform; a clamped duplicate is computed but not stored)
        kk[i] = t == 0 ? 0 : t + off;
        ux[i] = dirs[2 * kk[i]];
        uy[i] = dirs[2 * kk[i] + 1];
        best[i] = INFINITY;  // smallest valid l_a == largest preference 1/l_a; first index wins ties
        fallback[i] = 0.0;
      }
      for (int w = 0; w < w_full; ++w) {  // the room's own edges (box fast path): the first line crossed is the hit
        const double ax = walls[4 * w], ay = walls[4 * w + 1];
        const double sx = walls[4 * w + 2] - ax, sy = walls[4 * w + 3] - ay;
        const double d0x = ax - px, d0y = ay - py;
        const double num_a = d0x * (-sy) + d0y * sx;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const double la = num_a * rden[kk[i] * nw + w];
          if (la > 0.0 && la < best[i]) best[i] = la;
          if (w == 0) fallback[i] = la;
        }
      }
      for (int w = w_full; w < nw; ++w) {
        const double ax = walls[4 * w], ay = walls[4 * w + 1];
        const double sx = walls[4 * w + 2] - ax, sy = walls[4 * w + 3] - ay;
        const double d0x = ax - px, d0y = ay - py;
        const double num_a = d0x * (-sy) + d0y * sx;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const double rd = rden[kk[i] * nw + w];
          const double la = num_a * rd;
          const double lb = ((-d0x) * (-uy[i]) + (-d0y) * ux[i]) * (-rd);
          const bool valid = (la > 0.0) && !(lb < 0.0) && !(lb > 1.0);
          if (valid && la < best[i]) best[i] = la;
          if (w == 0) fallback[i] = la;  // argmax over all -1 preferences picks wall 0
        }
      }
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        if (t0 + nvw * i < count) {
          const int k = kk[i];
          const float d = (float)((best[i] < INFINITY) ? best[i] : fallback[i]);
          put(k, d);
          if (a.ray_out && live && (publish || part == 0)) a.ray_out[(t * K + k) * a.B + b] = d;
        }
      }
    }
  };
  // Opposite rays share their wall intercepts.  With u' = -u the line through the position is the same: l_b (the
  // place on the wall) is unchanged and l_a changes sign, so ONE (l_a, l_b) pair per wall serves the ray (hits with
  // l_a > 0) and its opposite (hits with l_a < 0, at distance -l_a): 14 instead of 24 float64 instructions per pair of
  // rays and wall.  The reference's table of K test directions (0, 0, d, 2d, ... 360 - 2d degrees, Neurons.py:1584-1596)
  // holds an opposite for the indices 1 .. K/2 - 1 when K is even; whether THIS table does is checked here, on the
  // table itself (every wave for itself: a few loads per lane, once per tile), and any other table takes the
  // one-ray-at-a-time path.  The opposite direction in the table is -u up to its own rounding (1e-16).
  const int m = K >> 1, np = K - 1 - m;  // pairs (j, j + m) for j = 1 .. np
  bool paired = false;
  if ((K & 1) == 0 && np >= 8) {
    bool ok = true;
    for (int jq = 1 + lane; jq <= np; jq += 64) {
      const double e0 = a.test_dirs[2 * jq] + a.test_dirs[2 * (jq + m)];
      const double e1 = a.test_dirs[2 * jq + 1] + a.test_dirs[2 * (jq + m) + 1];
      ok = ok && fabs(e0) <= 1e-12 && fabs(e1) <= 1e-12;
    }
    paired = __builtin_amdgcn_ballot_w64(!ok) == 0;  // (wave-uniform, and the same in every wave)
  }
  if (paired) {
    for (int q0 = vw; q0 < np; q0 += nvw * KB) {
      double ux[KB], uy[KB], bpos[KB], bneg[KB], fallback[KB];
      int jj[KB];
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        jj[i] = 1 + min(q0 + nvw * i, np - 1);
        ux[i] = dirs[2 * jj[i]];
        uy[i] = dirs[2 * jj[i] + 1];
        bpos[i] = INFINITY;  // nearest hit along +u ...
        bneg[i] = INFINITY;  // ... and along -u
        fallback[i] = 0.0;
      }
      for (int w = 0; w < w_full; ++w) {  // the room's own edges (box fast path)
        const double ax = walls[4 * w], ay = walls[4 * w + 1];
        const double sx = walls[4 * w + 2] - ax, sy = walls[4 * w + 3] - ay;
        const double d0x = ax - px, d0y = ay - py;
        const double num_a = d0x * (-sy) + d0y * sx;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const double la = num_a * rden[jj[i] * nw + w];
          const double nla = -la;
          if (la > 0.0 && la < bpos[i]) bpos[i] = la;
          if (nla > 0.0 && nla < bneg[i]) bneg[i] = nla;
          if (w == 0) fallback[i] = la;
        }
      }
      for (int w = w_full; w < nw; ++w) {
        const double ax = walls[4 * w], ay = walls[4 * w + 1];
        const double sx = walls[4 * w + 2] - ax, sy = walls[4 * w + 3] - ay;
        const double d0x = ax - px, d0y = ay - py;
        const double num_a = d0x * (-sy) + d0y * sx;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const double rd = rden[jj[i] * nw + w];
          const double la = num_a * rd;
          const double lb = ((-d0x) * (-uy[i]) + (-d0y) * ux[i]) * (-rd);
          const bool on_wall = !(lb < 0.0) && !(lb > 1.0);
          const double nla = -la;
          if (on_wall && la > 0.0 && la < bpos[i]) bpos[i] = la;
          if (on_wall && nla > 0.0 && nla < bneg[i]) bneg[i] = nla;
          if (w == 0) fallback[i] = la;
        }
      }
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        if (q0 + nvw * i < np) {
          const int k = jj[i];
          const float dp = (float)((bpos[i] < INFINITY) ? bpos[i] : fallback[i]);
          const float dn = (float)((bneg[i] < INFINITY) ? bneg[i] : -fallback[i]);
          put(k, dp);
          put(k + m, dn);
          if (a.ray_out && live && (publish || part == 0)) {
            a.ray_out[(t * K + k) * a.B + b] = dp;
            a.ray_out[(t * K + k + m) * a.B + b] = dn;
          }
        }
      }
    }
    cast_single(1 + (m - np), np);  // table index 0 (the duplicated first direction) and np + 1 .. m
  } else {
    cast_single(K, 0);
  }
  };  // cast_rays
  const bool exchange = a.xch != nullptr && nparts > 1;
  if (!exchange) {
    cast_rays(std::integral_constant<int, 4>{}, wave, 8, false);
  } else {
    // ---- a tile's workgroups share stage A: each casts every gridDim.y-th batch of rays, publishes them write-through,
    // announces itself on the tile's counter, and waits — a few tens of microseconds at most — until all have: then the
    // whole tile comes from the exchange rows (past L1).  Partners that do not show up in time (a device that does not
    // hold the whole grid at once): this workgroup casts every ray itself — the same values either way.
    cast_rays(std::integral_constant<int, RIAB_BVC_XCH_KB>{}, wave + 8 * part, 8 * nparts, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's rows have been acknowledged)
    __syncthreads();
    __shared__ int s_all_here;
    if (tid == 0) {
      typedef __attribute__((address_space(1))) uint32_t gu32;
      gu32* const cnt = (gu32*)(uintptr_t)(a.xch_count + tile);
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int ok = 0;
      for (int spins = 0; spins < 256; ++spins) {  // (~0.3 us per poll)
        if ((int32_t)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.xch_target) >= 0) {
          ok = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      s_all_here = ok;
    }
    __syncthreads();
    if (s_all_here) {
      const float* const x = a.xch + (int64_t)tile * a.Kp * 64;
      for (int i = tid; i < K * 16; i += 512) {  // 16 bytes per thread and pass
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        gu64* const g = (gu64*)(uintptr_t)(x + 4 * i);
        const unsigned long long lo = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<unsigned long long*>(s_d + 4 * i) = lo;
        *reinterpret_cast<unsigned long long*>(s_d + 4 * i + 2) = hi;
      }
    } else {
      cast_rays(std::integral_constant<int, 4>{}, wave, 8, false);
    }
  }
  // pad rows: an infinite distance makes the term exp2(-inf) = 0 for any (finite or -inf) table entry
  for (int k = K + wave; k < a.Kp; k += 8) s_d[k * 64 + lane] = INFINITY;
  __syncthreads();

  // ---- stage B ------------------------------------------------------------------------------
  // wave w owns the 4-cell groups g = w (mod 8).  Per group the k-loop walks 4 test directions at
  // a time: 4 ds_read_b32 of d (shared by the 4 cells), 4 s_load_dwordx4 of the angular table
  // (wave-uniform -> SGPR operands), 16 fused exponents on 4 independent accumulators.
  float ch = 1.0f, sh = 0.0f;
  if (EGO) {
    // cos / sin of utils.get_angle(head_direction) = atan2(hy, hx + 1e-6)
    const float hx = a.hd_x[t * a.pos_ld + b] + 1e-6f, hy = a.hd_y[t * a.pos_ld + b];
    const float inv = 1.0f / sqrtf(hx * hx + hy * hy);
    ch = hx * inv;
    sh = hy * inv;
  }
  const int n = a.n, Kp = a.Kp;
  const int n_groups = (n + 3) >> 2;
  // (gridDim.y > 1: a launch of few tiles — one row of a closed loop — deals its cell groups to several workgroups per
  // tile, each of which has cast the tile's rays for itself: see the launch)
  for (int g = wave + 8 * part; g < n_groups; g += 8 * nparts) {
    float aa[4], nmu[4], kap[4];
    const_f32_ptr tc[4];
    const_f32_ptr ts[4];
    const const_f32_ptr cells = as_const_table(a.cells);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = min(4 * g + j, n - 1);
      nmu[j] = -cells[c];
      aa[j] = cells[n + c];
      kap[j] = cells[2 * n + c];
      tc[j] = as_const_table(a.vm + (int64_t)c * Kp);
      ts[j] = as_const_table(a.vm + ((int64_t)n + c) * Kp);
    }
    v2f acc2[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
    // register double-buffering: the d values and table entries of step k+4 are requested before
    // the 16 exponentials of step k are issued, so LDS / scalar-cache latency hides under them
    // (table rows are read four directions at a time: one s_load_dwordx4 per cell and iteration)
    typedef const __attribute__((address_space(4))) v4f* const_v4f_ptr;
    float d[4];
    v4f vc[4], vs[4];
    // the group's direction window [k, k + wlen) modulo Kw (the whole circle without a window table)
    typedef const __attribute__((address_space(4))) int* const_i32_ptr;
    int k = 0, wlen = Kp;
    const int Kw = Kp;
    if (!EGO && a.win) {
      const const_i32_ptr win = (const_i32_ptr)(const void*)a.win;
      k = win[2 * g];
      wlen = win[2 * g + 1];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = s_d[(k + i) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vc[j] = *(const_v4f_ptr)(tc[j] + k);
      if (EGO) vs[j] = *(const_v4f_ptr)(ts[j] + k);
    }
    for (int kk = 0; kk < wlen; kk += 4) {
      // next four directions of the window, wrapping round the circle (or, on the last pass, these again)
      const int kn = (kk + 4 < wlen) ? ((k + 4 < Kw) ? k + 4 : k + 4 - Kw) : k;
      float dn[4];
      v4f vcn[4], vsn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) dn[i] = s_d[(kn + i) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vcn[j] = *(const_v4f_ptr)(tc[j] + kn);
        if (EGO) vsn[j] = *(const_v4f_ptr)(ts[j] + kn);
      }
      // the three non-transcendental operations of a term are issued as packed fp32 (v_pk_fma_f32,
      // v_pk_add_f32: two terms per instruction); only the exp2 itself stays one per term
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const v2f dd = {d[i], d[i + 1]};
          const v2f tt = __builtin_elementwise_fma(dd, v2f{aa[j], aa[j]}, v2f{nmu[j], nmu[j]});
          v2f v;
          if (EGO) {
            const v2f c2 = {vc[j][i], vc[j][i + 1]}, s2 = {vs[j][i], vs[j][i + 1]};
            const v2f rot = __builtin_elementwise_fma(c2, v2f{ch, ch}, s2 * v2f{sh, sh});
            v = v2f{kap[j], kap[j]} * (rot - v2f{1.0f, 1.0f});
          } else {
            v = v2f{vc[j][i], vc[j][i + 1]};
          }
          const v2f e = __builtin_elementwise_fma(-tt, tt, v);
          acc2[j] += v2f{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = dn[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vc[j] = vcn[j];
        if (EGO) vs[j] = vsn[j];
      }
      k = kn;
    }
    // Spikes: one Philox block serves four consecutive agents of one cell (words x, y, z, w: riab_rates.hip
    // spike_store).  Here a lane is ONE agent and holds the group's four cells: lane i of each aligned quad computes the
    // block of cell 4g + i once and the quad exchanges words (DPP quad_perm) — agent i of the quad takes word i of every
    // cell's block — instead of every lane computing all four blocks for one word each.  (Quads are whole: B, agent_id0
    // and the tile base are multiples of four, so the four lanes share the time row and gid >> 2.)
    u32x4 blk = {0u, 0u, 0u, 0u};
    const int qi = lane & 3;
    if (a.spikes && !a.u_in) {
      const int ci = min(4 * g + qi, n - 1);
      const int c = a.rows ? a.rows[ci] : ci;
      const uint64_t gid = (uint64_t)(a.agent_id0 + b);
      blk = philox4x32_spikes(a.step0 + (uint32_t)t, (uint32_t)c, (uint32_t)(gid >> 2), a.tag, a.k0, a.k1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = 4 * g + j;  // table row
      // word `qi` of the block that lane j of the quad holds (all lanes take part in the exchange)
      const uint32_t word = j == 0 ? quad_word<0>(blk, qi) : (j == 1 ? quad_word<1>(blk, qi) : (j == 2 ? quad_word<2>(blk, qi) : quad_word<3>(blk, qi)));
      if (ci < n && live) {
        const int c = a.rows ? a.rows[ci] : ci;  // the cell it belongs to
        float r = (acc2[j].x + acc2[j].y) * a.inv_norm[ci];
        r = r * a.fr_scale + a.fr_min;
        const int64_t off = (t * n + c) * a.B + b;
        // (a launch of one row — the closed loop's — writes its values through: riab_device.h, store_stream)
        if (a.P == a.B) asm volatile("global_store_dword %0, %1, off " RIAB_WT_BITS ::"v"(a.rates + off), "v"(r) : "memory");
        else a.rates[off] = r;
        if (a.spikes) {
          const float u = a.u_in ? a.u_in[off] : u01_24(word);
          a.spikes[off] = (u < a.dt * r) ? 1 : 0;
        }
      }
    }
  }
}

int launch_bvc(const RiabEnv* env, const RiabRateIO* io, const double* test_dirs, const double* ray_rden, int32_t K,
               const float* cells, const float* vm_table, const float* inv_norm, int32_t n, int32_t egocentric,
               float* ray_out, const int32_t* cell_rows, const int32_t* windows, float* xch, uint32_t* xch_count,
               uint32_t* xch_arrivals, int n_cus, hipStream_t stream);

}  // namespace riab

using namespace riab;

extern "C" int riab_boundary_vector_cells(const RiabEnv* env, const RiabRateIO* io, const double* test_dirs,
                                          const double* ray_rden, int32_t K, const float* cells, const float* vm_table, const float* inv_norm,
                                          int32_t n, int32_t egocentric, float* ray_out, riab_stream_t stream) {
  return riab_boundary_vector_cells_windowed(env, io, test_dirs, ray_rden, K, cells, vm_table, inv_norm, n, egocentric,
                                             ray_out, nullptr, nullptr, stream);
}

extern "C" int riab_boundary_vector_cells_windowed(const RiabEnv* env, const RiabRateIO* io, const double* test_dirs,
                                                   const double* ray_rden, int32_t K, const float* cells,
                                                   const float* vm_table, const float* inv_norm, int32_t n,
                                                   int32_t egocentric, float* ray_out, const int32_t* cell_rows,
                                                   const int32_t* windows, riab_stream_t stream) {
  return riab::launch_bvc(env, io, test_dirs, ray_rden, K, cells, vm_table, inv_norm, n, egocentric, ray_out, cell_rows, windows,
                          nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
}

// ... + the ray exchange of a step plan's one-row launches (RiabPopulation.bvc_xch): `xch_launches` counts the plan's
// launches on `xch_count` (incremented here when the exchange is used), `n_cus` the compute units the plan counts on
int riab::launch_bvc(const RiabEnv* env, const RiabRateIO* io, const double* test_dirs, const double* ray_rden, int32_t K,
                     const float* cells, const float* vm_table, const float* inv_norm, int32_t n, int32_t egocentric,
                     float* ray_out, const int32_t* cell_rows, const int32_t* windows, float* xch, uint32_t* xch_count,
                     uint32_t* xch_arrivals, int n_cus, hipStream_t stream) {
  if ((cell_rows == nullptr) != (windows == nullptr)) return RIAB_EINVAL;
  if (windows && (egocentric || K % 4 != 0)) return RIAB_EUNSUPPORTED;
  if (!env || !io || !test_dirs || !ray_rden || !cells || !vm_table || !inv_norm || n <= 0 || K <= 0) return RIAB_EINVAL;
  if (io->T <= 0 || io->B <= 0 || !io->rates || !io->pos_x || !io->pos_y) return RIAB_EINVAL;
  if (egocentric && (!io->hd_x || !io->hd_y)) return RIAB_EINVAL;
  if (env->n_walls <= 0 || !env->walls) return RIAB_EINVAL;  // BVCs need solid boundaries (Neurons.py:1580-1582)
  if (env->n_walls > RIAB_MAX_WALLS || K > RIAB_MAX_TEST_ANGLES) return RIAB_ETOOBIG;
  if (io->u_in && !io->spikes) return RIAB_EINVAL;
  // (in-kernel spike draws share one Philox block between the four agents of an aligned quad, like the rate kernels)
  if (io->spikes && !io->u_in && (io->B % 4 != 0 || io->agent_id0 % 4 != 0)) return RIAB_EALIGN;
  BvcArgs a;
  a.pos_x = io->pos_x;
  a.pos_y = io->pos_y;
  a.hd_x = io->hd_x;
  a.hd_y = io->hd_y;
  a.pos_ld = io->pos_ld;
  a.P = io->T * io->B;
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
  a.agent_id0 = io->agent_id0;
  a.n = n;
  a.K = K;
  a.Kp = (K + 3) / 4 * 4;
  a.n_walls = env->n_walls;
  a.walls = env->walls;
  a.test_dirs = test_dirs;
  a.ray_rden = ray_rden;
  a.cells = cells;
  a.vm = vm_table;
  a.inv_norm = inv_norm;
  a.ray_out = ray_out;
  a.rows = cell_rows;
  a.win = windows;
  a.e0 = env->extent[0]; a.e1 = env->extent[1]; a.e2 = env->extent[2]; a.e3 = env->extent[3];
  a.rect_room = (!env->polygon && !env->hole_mask && !env->periodic && g_options[RIAB_OPT_BVC_BOX]) ? 1 : 0;
  const size_t lds = sizeof(float) * (size_t)((K + 3) / 4 * 4) * 64;
  if (lds > 160 * 1024) return RIAB_ETOOBIG;
  // One workgroup per tile of 64 positions is the throughput shape (many rows per launch: every compute unit has tiles
  // to spare).  ONE row of a closed loop is 64 tiles at 4096 agents — a quarter of the chip, each workgroup walking all
  // of its cell groups alone: such a launch deals the cell groups of a tile to up to eight workgroups (grid.y), every one
  // of which casts the tile's rays for itself (stage A: ~5-15 % of a tile's work) — what counts there is the row's
  // latency.  [MI355X] cfg 3's closed-loop step 110 -> see DESIGN.md 3.2.
  const int64_t tiles = (a.P + 63) / 64;
  const int n_groups = (n + 3) / 4;
  int split = 1;
  while (split < 8 && tiles * split * 2 <= 512 && split * 2 * 8 <= n_groups) split *= 2;
  const dim3 grid((unsigned)tiles, (unsigned)split);
  // The exchange: the tile's workgroups each cast every split-th batch of rays and read the others' — in a room of many
  // walls stage A is most of a one-row launch and was cast `split` times over.
  // Only where the whole grid is resident at once (three of these workgroups fit a compute unit; a workgroup whose partners
  // do not show up within ~50 us casts everything itself, so a wrong guess costs time, not correctness).
  a.xch = nullptr;
  a.xch_count = nullptr;
  a.xch_target = 0u;
  // (not inside a stream capture: a replayed launch would meet a counter that has already passed its target)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = xch && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  // ... and only in rooms with interior walls: in an open box stage A is the four edges' fast path and the exchange costs more
  // than it saves ([MI355X] closed-loop step, with / without: cfg 5 (4 walls) 88.6 / 83.4 us, cfg 3 (9 walls) 55.9 / 58.0,
  // cfg3_64w (64 walls) 95 / 184; with one pair per pass in the exchanging workgroups 74 — and still nothing in the open box: 78.2 / 77.7)
  if (xch && xch_count && xch_arrivals && !capturing && split > 1 && io->T == 1 && n_cus > 0 && env->n_walls >= 8 &&
      tiles * split <= 2 * (int64_t)n_cus) {
    *xch_arrivals += (uint32_t)split;
    a.xch = xch;
    a.xch_count = xch_count;
    a.xch_target = *xch_arrivals;
  }
  hipStream_t s = (hipStream_t)stream;
  if (egocentric) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)bvc_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bvc_kernel<true>, grid, dim3(512), lds, s, a);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)bvc_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bvc_kernel<false>, grid, dim3(512), lds, s, a);
  }
  return (int)hipGetLastError();
}
