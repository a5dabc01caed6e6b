// Environment geometry queries for gfx950 (MI355X): the stand-alone forms of the helpers the
// motion and rate kernels inline (pairwise vectors / distances with the wall geometries, shortest
// vectors from walls, wall-collision tests, boundary conditions).  All float64, like the
// reference's own functions; lanes run over positions (SoA rows), walls sit in LDS and are walked
// wave-uniformly.  These are convenience entry points for callers of the Environment API, not
// part of the per-step pipeline (riab_agent_step / riab_place_cells do this work in-register).
#include "riab_device.h"

namespace riab {

typedef __attribute__((address_space(3))) const double* lds_walls_ptr;

struct EnvQuery {
  double e0, e1, e2, e3;  // extent
  double scale;
  int periodic;
  int n_walls;
  const double* walls;  // [n_walls][4]
  EnvShape shape;       // boundary polygon / holes
};

static EnvQuery make_query(const RiabEnv* env) {
  EnvQuery q;
  q.e0 = env->extent[0]; q.e1 = env->extent[1]; q.e2 = env->extent[2]; q.e3 = env->extent[3];
  q.scale = env->scale;
  q.periodic = env->periodic;
  q.n_walls = env->n_walls;
  q.walls = env->walls;
  q.shape = make_env_shape(env);
  return q;
}

// walls[first:] -> LDS (4 doubles per wall)
__device__ __forceinline__ void stage_walls(double* s, const EnvQuery& q, int first) {
  const int n = 4 * (q.n_walls - first);
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = q.walls[4 * first + i];
  __syncthreads();
}

// ---- get_vectors_between / get_distances_between (Environment.py:657-779) --------------------
// GEOM as RIAB_GEOM_*.  One workgroup row per pos1 element, lanes over pos2.
template <int GEOM>
__global__ __launch_bounds__(256) void pairwise_kernel(const EnvQuery q, const double* __restrict__ x1,
                                                       const double* __restrict__ y1, int64_t N1,
                                                       const double* __restrict__ x2, const double* __restrict__ y2,
                                                       int64_t N2, double* __restrict__ dist,
                                                       double* __restrict__ vec_x, double* __restrict__ vec_y) {
#pragma clang fp contract(off)
  __shared__ double s_w[4 * RIAB_MAX_WALLS];
  const int n_int = (GEOM == RIAB_GEOM_EUCLIDEAN || q.n_walls <= 4) ? 0 : q.n_walls - 4;  // Environment.py:715-717
  if (GEOM != RIAB_GEOM_EUCLIDEAN && n_int > 0) stage_walls(s_w, q, 4);
  const lds_walls_ptr w = (lds_walls_ptr)s_w;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= N2) return;
  const double bx = x2[j], by = y2[j];
  const double hs = q.scale / 2;
  for (int64_t i = blockIdx.y; i < N1; i += gridDim.y) {
    const double ax = x1[i], ay = y1[i];
    double dx = ax - bx, dy = ay - by;
    if (q.periodic) {  // Environment.py:670-674
      if (fabs(dx) > hs) dx = -copysign(q.scale - fabs(dx), dx);
      if (fabs(dy) > hs) dy = -copysign(q.scale - fabs(dy), dy);
    }
    const int64_t o = i * N2 + j;
    if (vec_x) vec_x[o] = dx;
    if (vec_y) vec_y[o] = dy;
    if (!dist) continue;
    double d = sqrt(dx * dx + dy * dy);
    if (GEOM == RIAB_GEOM_LINE_OF_SIGHT) {
      bool blocked = false;
      for (int k = 0; k < n_int; ++k)
        blocked |= seg_hit(ax, ay, bx, by, w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
      if (blocked) d = 1000.0;  // Environment.py:730
    }
    if (GEOM == RIAB_GEOM_GEODESIC && n_int > 0) {
      if (seg_hit(ax, ay, bx, by, w[0], w[1], w[2], w[3])) {
        // Environment.py:744-774: the shortest route round a wall endpoint strictly inside the environment
        double best = INFINITY;
        bool any = false;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const double ex = w[2 * e], ey = w[2 * e + 1];
          if (env_contains(q.shape, ex, ey, [&](int k, double& ax, double& ay, double& bx, double& by) {
                ax = q.walls[4 * k]; ay = q.walls[4 * k + 1]; bx = q.walls[4 * k + 2]; by = q.walls[4 * k + 3];
              })) {
            const double d1 = sqrt((ax - ex) * (ax - ex) + (ay - ey) * (ay - ey));
            const double d2 = sqrt((ex - bx) * (ex - bx) + (ey - by) * (ey - by));
            best = fmin(best, d1 + d2);
            any = true;
          }
        }
        if (any) d = best;
      }
    }
    dist[o] = d;
  }
}

// ---- vectors_from_walls (Environment.py:843-853, utils.py:121-184) ---------------------------
__global__ __launch_bounds__(256) void vectors_from_walls_kernel(const EnvQuery q, const double* __restrict__ px,
                                                                 const double* __restrict__ py, int64_t P,
                                                                 double* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double s_w[4 * RIAB_MAX_WALLS];
  stage_walls(s_w, q, 0);
  const lds_walls_ptr w = (lds_walls_ptr)s_w;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const double x = px[p], y = py[p];
  for (int k = 0; k < q.n_walls; ++k) {
    const double ax = w[4 * k], ay = w[4 * k + 1];
    const double sx = w[4 * k + 2] - ax, sy = w[4 * k + 3] - ay;
    double l = ((x - ax) * sx + (y - ay) * sy) / (sx * sx + sy * sy);
    l = (l > 1.0) ? 1.0 : l;  // np.where order of utils.py:166-167: NaN stays NaN
    l = (l < 0.0) ? 0.0 : l;
    out[(2 * (int64_t)k) * P + p] = x - (ax + l * sx);
    out[(2 * (int64_t)k + 1) * P + p] = y - (ay + l * sy);
  }
}

// ---- check_wall_collisions (Environment.py:820-841, utils.py:74-106) -------------------------
__global__ __launch_bounds__(256) void wall_collisions_kernel(const EnvQuery q, const double* __restrict__ x0,
                                                              const double* __restrict__ y0,
                                                              const double* __restrict__ x1,
                                                              const double* __restrict__ y1, int64_t P,
                                                              uint8_t* __restrict__ out) {
  __shared__ double s_w[4 * RIAB_MAX_WALLS];
  stage_walls(s_w, q, 0);
  const lds_walls_ptr w = (lds_walls_ptr)s_w;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const double ax = x0[p], ay = y0[p], bx = x1[p], by = y1[p];
  for (int k = 0; k < q.n_walls; ++k)
    out[(int64_t)k * P + p] = seg_hit(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3], ax, ay, bx, by) ? 1 : 0;
}

// ---- check_if_position_is_in_environment + apply_boundary_conditions (Environment.py:781-894) --
__global__ __launch_bounds__(256) void boundary_conditions_kernel(const EnvQuery q, double* __restrict__ px,
                                                                  double* __restrict__ py, int64_t P,
                                                                  uint8_t* __restrict__ inside_out, int apply) {
#pragma clang fp contract(off)
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  double x = px[p], y = py[p];
  const bool in_box = x > q.e0 && x < q.e1 && y > q.e2 && y < q.e3;  // strict interior (shapely `contains`)
  bool inside = in_box;
  if (q.shape.boundary_mask || q.shape.hole_mask)
    inside = env_contains(q.shape, x, y, [&](int k, double& ax, double& ay, double& bx, double& by) {
      ax = q.walls[4 * k]; ay = q.walls[4 * k + 1]; bx = q.walls[4 * k + 2]; by = q.walls[4 * k + 3];
    });
  // 2: outside, and what apply_boundary_conditions does about it is a random draw (in a hole / outside a polygon,
  // Environment.py:886-893): left to the caller
  const bool resample = !inside && (q.shape.boundary_mask || in_box);
  if (inside_out) inside_out[p] = inside ? 1 : (resample ? 2 : 0);
  if (inside || !apply || resample) return;
  if (q.periodic) {
    x = x - q.e1 * floor(x / q.e1);  // np.mod(pos, extent)
    y = y - q.e3 * floor(y / q.e3);
  } else {
    const double lo_x = q.e0 + 0.01, hi_x = q.e1 - 0.01, lo_y = q.e2 + 0.01, hi_y = q.e3 - 0.01;
    x = (lo_x > x) ? lo_x : x;  // python max(pos, lo): NaN stays NaN
    x = (hi_x < x) ? hi_x : x;
    y = (lo_y > y) ? lo_y : y;
    y = (hi_y < y) ? hi_y : y;
  }
  px[p] = x;
  py[p] = y;
}

static int check_env(const RiabEnv* env, bool need_walls) {
  if (!env || env->n_walls < 0) return RIAB_EINVAL;
  if (check_env_shape(env)) return RIAB_EINVAL;
  if ((env->polygon || env->hole_mask) && !env->walls) return RIAB_EINVAL;
  if (env->n_walls > RIAB_MAX_WALLS) return RIAB_ETOOBIG;
  if (need_walls && env->n_walls > 0 && !env->walls) return RIAB_EINVAL;
  return RIAB_OK;
}

}  // namespace riab

using namespace riab;

extern "C" int riab_env_pairwise(const RiabEnv* env, const double* x1, const double* y1, int64_t N1, const double* x2,
                                 const double* y2, int64_t N2, int32_t geometry, double* dist, double* vec_x,
                                 double* vec_y, riab_stream_t stream) {
  if (!x1 || !y1 || !x2 || !y2 || N1 <= 0 || N2 <= 0 || (!dist && !vec_x && !vec_y)) return RIAB_EINVAL;
  const int rc = check_env(env, geometry != RIAB_GEOM_EUCLIDEAN);
  if (rc) return rc;
  if (geometry != RIAB_GEOM_EUCLIDEAN) {
    if (env->periodic) return RIAB_EUNSUPPORTED;                                          // Neurons.py:908-921
    if (geometry == RIAB_GEOM_GEODESIC && env->n_walls > 5) return RIAB_EUNSUPPORTED;     // Environment.py:736-739
  }
  const int64_t bx = (N2 + 255) / 256;
  if (bx >= ((int64_t)1 << 31)) return RIAB_ETOOBIG;
  const dim3 grid((unsigned)bx, (unsigned)(N1 < 65535 ? N1 : 65535));
  const EnvQuery q = make_query(env);
  hipStream_t s = (hipStream_t)stream;
  switch (geometry) {
    case RIAB_GEOM_EUCLIDEAN:
      hipLaunchKernelGGL((pairwise_kernel<RIAB_GEOM_EUCLIDEAN>), grid, dim3(256), 0, s, q, x1, y1, N1, x2, y2, N2, dist,
                         vec_x, vec_y);
      break;
    case RIAB_GEOM_LINE_OF_SIGHT:
      hipLaunchKernelGGL((pairwise_kernel<RIAB_GEOM_LINE_OF_SIGHT>), grid, dim3(256), 0, s, q, x1, y1, N1, x2, y2, N2,
                         dist, vec_x, vec_y);
      break;
    case RIAB_GEOM_GEODESIC:
      hipLaunchKernelGGL((pairwise_kernel<RIAB_GEOM_GEODESIC>), grid, dim3(256), 0, s, q, x1, y1, N1, x2, y2, N2, dist,
                         vec_x, vec_y);
      break;
    default: return RIAB_EINVAL;
  }
  return (int)hipGetLastError();
}

extern "C" int riab_env_vectors_from_walls(const RiabEnv* env, const double* pos_x, const double* pos_y, int64_t P,
                                           double* out, riab_stream_t stream) {
  if (!pos_x || !pos_y || !out || P <= 0) return RIAB_EINVAL;
  const int rc = check_env(env, true);
  if (rc) return rc;
  if (env->n_walls == 0) return RIAB_OK;
  hipLaunchKernelGGL(vectors_from_walls_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     make_query(env), pos_x, pos_y, P, out);
  return (int)hipGetLastError();
}

extern "C" int riab_env_check_wall_collisions(const RiabEnv* env, const double* x0, const double* y0, const double* x1,
                                              const double* y1, int64_t P, uint8_t* out, riab_stream_t stream) {
  if (!x0 || !y0 || !x1 || !y1 || !out || P <= 0) return RIAB_EINVAL;
  const int rc = check_env(env, true);
  if (rc) return rc;
  if (env->n_walls == 0) return RIAB_OK;
  hipLaunchKernelGGL(wall_collisions_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     make_query(env), x0, y0, x1, y1, P, out);
  return (int)hipGetLastError();
}

extern "C" int riab_env_boundary_conditions(const RiabEnv* env, double* pos_x, double* pos_y, int64_t P,
                                            uint8_t* inside_out, int32_t apply, riab_stream_t stream) {
  if (!pos_x || !pos_y || P <= 0 || (!inside_out && !apply)) return RIAB_EINVAL;
  const int rc = check_env(env, false);
  if (rc) return rc;
  hipLaunchKernelGGL(boundary_conditions_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     make_query(env), pos_x, pos_y, P, inside_out, (int)apply);
  return (int)hipGetLastError();
}
