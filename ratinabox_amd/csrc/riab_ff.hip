// FeedForwardLayer.get_state for gfx950 (reference Neurons.py:2797-2847, utils.activate
// utils.py:919-1026):
//
//     out[t][m][b] = act( sum_layers sum_k W_l[m][k] * I_l[t][k][b] + bias[m] )
//
// The one dense contraction on the path: a GEMM with M = n_out, K = n_in (summed over the
// input layers) and N = positions (agents x time rows, contiguous in memory for both the
// input rates and the output).  It runs on the fp32-input matrix cores,
// v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain), 64 FLOP/clk/SIMD.
//
// Workgroup = 4 waves; block tile = up to 128 outputs x 128 positions; wave w owns positions
// [32w, 32w+32) and all (<= 4) 32-row output tiles, so a B fragment read from LDS feeds 4 MFMAs.
// K is walked in slabs of 32: the slab of W^T ([k][m], prepared on the host so that both
// operands are read with unit stride) and of the input rates are staged in LDS by coalesced
// 16-B loads.  Fragment layout (lane l): A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// C/D: column = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// Epilogue: bias + activation (+ optional derivative) fused, 128-B row segments stored.
#include "riab_device.h"

namespace riab {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int FF_MAX_LAYERS = 8;
constexpr int FF_KT = 32;    // K slab
constexpr int FF_NB = 128;   // positions per block
constexpr int FF_MB = 128;   // outputs per block (4 MFMA row tiles)

struct FFArgs {
  const float* rates[FF_MAX_LAYERS];  // [T][n_in][B]
  const float* wt[FF_MAX_LAYERS];     // [n_in][Mp]  (W transposed, Mp = n_out padded to 32)
  int n_in[FF_MAX_LAYERS];
  int n_layers;
  const float* bias;  // [n_out]
  int n_out, Mp;
  int64_t B;
  int T;
  int act;
  float p0, p1, p2, p3;  // activation parameters
  float* out;            // [T][n_out][B]
  float* out_prime;      // or null
};

// utils.activate (reference utils.py:919-1026); returns f(x), writes df/dx.
__device__ __forceinline__ float activate(int act, float x, float p0, float p1, float p2, float p3, float* d) {
  switch (act) {
    case RIAB_ACT_LINEAR: *d = 1.0f; return x;
    case RIAB_ACT_SIGMOID: {  // p0 max_fr, p1 min_fr, p2 mid_x, p3 beta = ln(19)/(width_x/2)
      const float s = 1.0f / (1.0f + expf(-p3 * (x - p2)));
      const float f = (p0 - p1) * s + p1;
      *d = p3 * (f - p1) * (1.0f - (f - p1) / (p0 - p1));
      return f;
    }
    case RIAB_ACT_RELU: {  // p0 gain, p1 threshold
      *d = p0 * ((x - p1) > 0.0f ? 1.0f : 0.0f);
      return p0 * fmaxf(0.0f, x - p1);
    }
    case RIAB_ACT_TANH: {  // the reference's derivative ignores the threshold (utils.py:998)
      const float th = tanhf(x);
      *d = p0 * (1.0f - th * th);
      return p0 * tanhf(x - p1);
    }
    case RIAB_ACT_RETANH: {
      const float th = tanhf(x);
      *d = p0 * (1.0f - th * th) * ((x - p1) > 0.0f ? 1.0f : 0.0f);
      return p0 * fmaxf(0.0f, tanhf(x - p1));
    }
    case RIAB_ACT_SOFTMAX: {  // "softmax" in the reference is softplus: gain*log(1+exp(x-thr))
      const float z = x - p1;
      *d = p0 / (1.0f + expf(-z));
      return p0 * (z > 20.0f ? z : log1pf(expf(z)));
    }
  }
  *d = 0.0f;
  return 0.0f;
}

__global__ __launch_bounds__(256) void ff_kernel(const FFArgs a) {
  __shared__ __align__(16) float s_a[FF_KT][FF_MB];  // W^T slab  [k][m]
  __shared__ __align__(16) float s_b[FF_KT][FF_NB];  // rates slab [k][position]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b0 = (int64_t)blockIdx.x * FF_NB;  // first agent of the block's position tile
  const int m0 = blockIdx.y * FF_MB;               // first output of the block
  const int t = blockIdx.z;
  const int mt_count = min(4, (a.Mp - m0) >> 5);   // 32-row output tiles in this block
  v16f acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  for (int l = 0; l < a.n_layers; ++l) {
    const int K = a.n_in[l];
    const float* rates = a.rates[l] + (int64_t)t * K * a.B;
    const float* wt = a.wt[l];
    for (int k0 = 0; k0 < K; k0 += FF_KT) {
      __syncthreads();
      // stage: 32 x 128 floats each = 1024 float4; 256 threads x 4 passes
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int idx = pass * 256 + tid;  // float4 index
        const int kk = idx >> 5, c4 = (idx & 31) * 4;
        const int k = k0 + kk;
        v4f va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
        if (k < K) {
          if (m0 + c4 < a.Mp) va = *reinterpret_cast<const v4f*>(wt + (int64_t)k * a.Mp + m0 + c4);
          const int64_t b = b0 + c4;
          if (b + 3 < a.B) vb = *reinterpret_cast<const v4f*>(rates + (int64_t)k * a.B + b);
          else if (b < a.B) {
            vb.x = rates[(int64_t)k * a.B + b];
            if (b + 1 < a.B) vb.y = rates[(int64_t)k * a.B + b + 1];
            if (b + 2 < a.B) vb.z = rates[(int64_t)k * a.B + b + 2];
          }
        }
        *reinterpret_cast<v4f*>(&s_a[kk][c4]) = va;
        *reinterpret_cast<v4f*>(&s_b[kk][c4]) = vb;
      }
      __syncthreads();
      const int kh = lane >> 5, j = lane & 31;
#pragma unroll 4
      for (int kk = 0; kk < FF_KT; kk += 2) {
        const float bf = s_b[kk + kh][wave * 32 + j];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (mt < mt_count) {  // wave-uniform
            const float af = s_a[kk + kh][mt * 32 + j];
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[mt], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- epilogue: bias + activation, C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int64_t b = b0 + wave * 32 + (lane & 31);
  if (b < a.B) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt < mt_count) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < a.n_out) {
            float d;
            const float f = activate(a.act, acc[mt][r] + a.bias[m], a.p0, a.p1, a.p2, a.p3, &d);
            const int64_t off = ((int64_t)t * a.n_out + m) * a.B + b;
            a.out[off] = f;
            if (a.out_prime) a.out_prime[off] = d;
          }
        }
      }
    }
  }
}

}  // namespace riab

using namespace riab;

extern "C" int riab_feedforward(const RiabFFInput* inputs, int32_t n_inputs, const float* bias, int32_t n_out,
                                int64_t T, int64_t B, int32_t activation, const float* act_params, float* out,
                                float* out_prime, riab_stream_t stream) {
  if (!inputs || n_inputs <= 0 || !bias || n_out <= 0 || T <= 0 || B <= 0 || !out || !act_params) return RIAB_EINVAL;
  if (n_inputs > FF_MAX_LAYERS) return RIAB_ETOOBIG;
  if (activation < RIAB_ACT_LINEAR || activation > RIAB_ACT_SOFTMAX) return RIAB_EINVAL;
  if (B % 4 != 0 || T > 65535) return B % 4 ? RIAB_EALIGN : RIAB_ETOOBIG;
  FFArgs a;
  a.Mp = (n_out + 31) / 32 * 32;
  for (int l = 0; l < n_inputs; ++l) {
    if (!inputs[l].rates || !inputs[l].wt || inputs[l].n_in <= 0) return RIAB_EINVAL;
    if ((((uintptr_t)inputs[l].rates | (uintptr_t)inputs[l].wt) & 15)) return RIAB_EALIGN;
    a.rates[l] = inputs[l].rates;
    a.wt[l] = inputs[l].wt;
    a.n_in[l] = inputs[l].n_in;
  }
  a.n_layers = n_inputs;
  a.bias = bias;
  a.n_out = n_out;
  a.B = B;
  a.T = (int)T;
  a.act = activation;
  a.p0 = act_params[0];
  a.p1 = act_params[1];
  a.p2 = act_params[2];
  a.p3 = act_params[3];
  a.out = out;
  a.out_prime = out_prime;
  const dim3 grid((unsigned)((B + FF_NB - 1) / FF_NB), (unsigned)((a.Mp + FF_MB - 1) / FF_MB), (unsigned)T);
  hipLaunchKernelGGL(ff_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
