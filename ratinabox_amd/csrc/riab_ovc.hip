// ObjectVectorCells.get_state for gfx950 (reference Neurons.py:1991-2116; FieldOfViewOVCs
// Neurons.py:2119-2150 is the same kernel on a radial manifold):
//
//   rate[c][p] = sum over objects m of type(c):  exp(-(d_pm - mu_c)^2 / 2 sigma_c^2)
//                                              * exp(kappa_c (cos(beta_pm - hb_p - phi_c) - 1))
//
// d_pm / beta_pm = distance / bearing from position p to object m (Environment.
// get_distances_between___accounting_for_environment, Environment.py:677-730: periodic wrap;
// line_of_sight sets d = 1000 when an internal wall blocks the view), hb_p = head bearing when
// egocentric.  Workgroup = 4 waves on one tile of 64 positions (lane = position): stage A puts
// (d, cos, sin of the bearing) per (object, position) in LDS — the occlusion predicate in float64
// sign logic — stage B: wave w owns cells c = w (mod 4) and, per cell, walks the objects of its
// type with one fused exponent per term.  No trigonometry in the kernel: cos(beta - hb - phi) is
// assembled from the unit vectors.
#include "riab_device.h"

namespace riab {

struct OvcArgs {
  const float* pos_x;
  const float* pos_y;
  const float* hd_x;
  const float* hd_y;
  int64_t pos_ld, P, B;
  float* rates;
  uint8_t* spikes;
  const float* u_in;
  float dt, fr_scale, fr_min;
  uint32_t k0, k1, step0, tag;
  int64_t agent_id0;
  int n, M, n_internal, periodic, occlude, ego;
  float scale, half_scale;
  const double* walls;    // [n_walls][4]; internal walls = walls[4:]
  const float* objects;   // [M][2]
  const int* types;       // [M]
  const float* cells;     // [n][6] = a*mu_d, a, cos(phi), sin(phi), kappa*log2(e), type
  // AgentVectorCells (Neurons.py:2204-2320): the one "object" is another agent, a different point for
  // every lane: rows like pos_x / pos_y; objects / types are unused (every cell responds to it)
  const float* other_x;
  const float* other_y;
  int64_t other_ld;
};

__global__ __launch_bounds__(256) void ovc_kernel(const OvcArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* s_d = reinterpret_cast<float*>(smem);  // [M][64]
  float* s_c = s_d + (size_t)a.M * 64;           // [M][64]
  float* s_s = s_c + (size_t)a.M * 64;           // [M][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t p = (int64_t)blockIdx.x * 64 + lane;
  const bool live = p < a.P;
  const int64_t pc = live ? p : 0;
  const int64_t t = pc / a.B, b = pc - t * a.B;
  const float px = a.pos_x[t * a.pos_ld + b], py = a.pos_y[t * a.pos_ld + b];
  float ch = 1.0f, sh = 0.0f;
  if (a.ego) {  // utils.get_angle(head_direction) = atan2(hy, hx + 1e-6)
    const float hx = a.hd_x[t * a.pos_ld + b] + 1e-6f, hy = a.hd_y[t * a.pos_ld + b];
    const float inv = 1.0f / sqrtf(hx * hx + hy * hy);
    ch = hx * inv;
    sh = hy * inv;
  }
  const const_f32_ptr objs = as_const_table(a.objects);
  typedef const __attribute__((address_space(4))) double* cf64;
  const cf64 walls = (cf64)(const void*)a.walls;
  // ---- stage A ---------------------------------------------------------------------------------
  for (int m = wave; m < a.M; m += 4) {
    const float ox = a.other_x ? a.other_x[t * a.other_ld + b] : objs[2 * m];
    const float oy = a.other_x ? a.other_y[t * a.other_ld + b] : objs[2 * m + 1];
    float vx = ox - px, vy = oy - py;  // object - position (Neurons.py:2038-2040)
    if (a.periodic) {                  // Environment.py:670-674
      if (fabsf(vx) > a.half_scale) vx = -copysignf(a.scale - fabsf(vx), vx);
      if (fabsf(vy) > a.half_scale) vy = -copysignf(a.scale - fabsf(vy), vy);
    }
    float d = sqrtf(fmaf(vy, vy, vx * vx));
    if (a.occlude) {
      bool blocked = false;
      for (int w = 0; w < a.n_internal; ++w)
        blocked |= seg_hit(px, py, ox, oy, walls[16 + 4 * w], walls[16 + 4 * w + 1], walls[16 + 4 * w + 2],
                           walls[16 + 4 * w + 3]);
      if (blocked) d = 1000.0f;  // Environment.py:730
    }
    // bearing = utils.get_angle(vector) = atan2(vy, vx + 1e-6): its cosine / sine
    const float bx = vx + 1e-6f;
    const float inv = 1.0f / sqrtf(fmaf(vy, vy, bx * bx));
    const float cb = bx * inv, sb = vy * inv;
    // subtract the head bearing (egocentric): rotate by -hb
    s_d[m * 64 + lane] = d;
    s_c[m * 64 + lane] = fmaf(cb, ch, sb * sh);
    s_s[m * 64 + lane] = fmaf(sb, ch, -cb * sh);
  }
  __syncthreads();
  // ---- stage B ---------------------------------------------------------------------------------
  const const_f32_ptr cells = as_const_table(a.cells);
  typedef const __attribute__((address_space(4))) int* ci32;
  const ci32 types = (ci32)(const void*)a.types;
  for (int c = wave; c < a.n; c += 4) {
    const float amu = cells[6 * c], aa = cells[6 * c + 1], cphi = cells[6 * c + 2], sphi = cells[6 * c + 3],
                kap = cells[6 * c + 4];
    const int ctype = (int)cells[6 * c + 5];
    float acc = 0.0f;
    for (int m = 0; m < a.M; ++m) {
      if (!a.other_x && types[m] != ctype) continue;  // wave-uniform
      const float tt = fmaf(s_d[m * 64 + lane], aa, -amu);
      const float cosd = fmaf(s_c[m * 64 + lane], cphi, s_s[m * 64 + lane] * sphi);  // cos(bearing - phi)
      acc += __builtin_amdgcn_exp2f(fmaf(-tt, tt, kap * (cosd - 1.0f)));
    }
    if (live) {
      const float r = acc * a.fr_scale + a.fr_min;
      const int64_t off = (t * a.n + c) * a.B + b;
      a.rates[off] = r;
      if (a.spikes) {
        float u;
        if (a.u_in) {
          u = a.u_in[off];
        } else {
          const uint64_t gid = (uint64_t)(a.agent_id0 + b);
          const u32x4 w4 = philox4x32_spikes(a.step0 + (uint32_t)t, (uint32_t)c, (uint32_t)(gid >> 2), a.tag, a.k0, a.k1);
          const uint32_t jj = (uint32_t)gid & 3u;
          u = u01_24(jj == 0 ? w4.x : (jj == 1 ? w4.y : (jj == 2 ? w4.z : w4.w)));
        }
        a.spikes[off] = (u < a.dt * r) ? 1 : 0;
      }
    }
  }
}

}  // namespace riab

using namespace riab;

static int launch_ovc(const RiabEnv* env, const RiabRateIO* io, const float* objects, const int32_t* object_types,
                      int32_t n_objects, const float* other_x, const float* other_y, int64_t other_ld,
                      const float* cells, int32_t n, int32_t walls_occlude, int32_t egocentric, hipStream_t stream) {
  if (io->T <= 0 || io->B <= 0 || !io->rates || !io->pos_x || !io->pos_y) return RIAB_EINVAL;
  if (egocentric && (!io->hd_x || !io->hd_y)) return RIAB_EINVAL;
  if (io->u_in && !io->spikes) return RIAB_EINVAL;
  if (walls_occlude && env->periodic) return RIAB_EUNSUPPORTED;  // Environment.py:711-713
  if (walls_occlude && env->n_walls > 4 && !env->walls) return RIAB_EINVAL;
  if (n_objects > 256) return RIAB_ETOOBIG;
  OvcArgs a;
  a.pos_x = io->pos_x; a.pos_y = io->pos_y; a.hd_x = io->hd_x; a.hd_y = io->hd_y;
  a.pos_ld = io->pos_ld; a.P = io->T * io->B; a.B = io->B;
  a.rates = io->rates; a.spikes = io->spikes; a.u_in = io->u_in;
  a.dt = io->dt; a.fr_scale = io->max_fr - io->min_fr; a.fr_min = io->min_fr;
  a.k0 = (uint32_t)io->seed; a.k1 = (uint32_t)(io->seed >> 32); a.step0 = (uint32_t)io->step0;
  a.tag = RIAB_TAG_SPIKES | ((uint32_t)io->pop_id & 0xFFu);
  a.agent_id0 = io->agent_id0;
  a.n = n; a.M = n_objects;
  a.n_internal = (walls_occlude && env->n_walls > 4) ? env->n_walls - 4 : 0;
  a.periodic = env->periodic; a.occlude = walls_occlude ? 1 : 0; a.ego = egocentric ? 1 : 0;
  a.scale = (float)env->scale; a.half_scale = (float)(env->scale / 2);
  a.walls = env->walls; a.objects = objects; a.types = object_types; a.cells = cells;
  a.other_x = other_x; a.other_y = other_y; a.other_ld = other_ld;
  const size_t lds = sizeof(float) * 3 * 64 * (size_t)n_objects;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)ovc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ovc_kernel, dim3((unsigned)((a.P + 63) / 64)), dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

extern "C" int riab_object_vector_cells(const RiabEnv* env, const RiabRateIO* io, const float* objects,
                                        const int32_t* object_types, int32_t n_objects, const float* cells, int32_t n,
                                        int32_t walls_occlude, int32_t egocentric, riab_stream_t stream) {
  if (!env || !io || !objects || !object_types || !cells || n <= 0 || n_objects <= 0) return RIAB_EINVAL;
  return launch_ovc(env, io, objects, object_types, n_objects, nullptr, nullptr, 0, cells, n, walls_occlude, egocentric,
                    (hipStream_t)stream);
}

extern "C" int riab_agent_vector_cells(const RiabEnv* env, const RiabRateIO* io, const float* other_x,
                                       const float* other_y, int64_t other_ld, const float* cells, int32_t n,
                                       int32_t walls_occlude, int32_t egocentric, riab_stream_t stream) {
  if (!env || !io || !other_x || !other_y || !cells || n <= 0 || other_ld < 0) return RIAB_EINVAL;
  return launch_ovc(env, io, nullptr, nullptr, 1, other_x, other_y, other_ld, cells, n, walls_occlude, egocentric,
                    (hipStream_t)stream);
}
