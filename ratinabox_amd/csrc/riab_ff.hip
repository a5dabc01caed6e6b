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
// Workgroup = 4 waves; block tile = 32*MT outputs x 128 positions (MT = 4, 2 or 1 by layer width);
// wave w owns positions [32w, 32w+32) and all MT 32-row output tiles, so a B fragment read from
// LDS feeds MT MFMAs.
// K is walked in double-buffered slabs of 16: the slab of W^T ([k][m], prepared on the host so that both
// operands are read with unit stride) and of the input rates are staged in LDS by coalesced
// 16-B loads.  Fragment layout (lane l): A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// C/D: column = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// Epilogue: bias + activation (+ optional derivative) fused, 128-B row segments stored.
#include <type_traits>

#include "riab_device.h"

namespace riab {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int FF_MAX_LAYERS = 8;
constexpr int FF_KT = 16;    // K slab
constexpr int FF_NB = 128;   // positions per block

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

// utils.activate (reference utils.py:919-1026); returns f(x), writes df/dx.  ACT is a template
// parameter so that the 64-element epilogue of a lane is straight-line code.
template <int ACT>
__device__ __forceinline__ float activate(float x, float p0, float p1, float p2, float p3, float* d) {
  if constexpr (ACT == RIAB_ACT_LINEAR) {
    *d = 1.0f;
    return x;
  } else if constexpr (ACT == RIAB_ACT_SIGMOID) {  // p0 max_fr, p1 min_fr, p2 mid_x, p3 beta = ln(19)/(width_x/2)
    const float s = 1.0f / (1.0f + expf(-p3 * (x - p2)));
    const float f = (p0 - p1) * s + p1;
    *d = p3 * (f - p1) * (1.0f - (f - p1) / (p0 - p1));
    return f;
  } else if constexpr (ACT == RIAB_ACT_RELU) {  // p0 gain, p1 threshold
    *d = p0 * ((x - p1) > 0.0f ? 1.0f : 0.0f);
    return p0 * fmaxf(0.0f, x - p1);
  } else if constexpr (ACT == RIAB_ACT_TANH) {  // the reference's derivative ignores the threshold (utils.py:998)
    const float th = tanhf(x);
    *d = p0 * (1.0f - th * th);
    return p0 * tanhf(x - p1);
  } else if constexpr (ACT == RIAB_ACT_RETANH) {
    const float th = tanhf(x);
    *d = p0 * (1.0f - th * th) * ((x - p1) > 0.0f ? 1.0f : 0.0f);
    return p0 * fmaxf(0.0f, tanhf(x - p1));
  } else {  // RIAB_ACT_SOFTMAX: "softmax" in the reference is softplus, gain*log(1+exp(x-thr))
    const float z = x - p1;
    *d = p0 / (1.0f + expf(-z));
    return p0 * (z > 20.0f ? z : log1pf(expf(z)));
  }
}

// One thread's share of a K slab: waves 0-1 fetch W^T, waves 2-3 the rates; a thread takes rows
// k0 + 8g + 2s + kh (s = 0..3) x 4 consecutive columns.  Out-of-range rows / columns read a clamped
// address (zeroed by the stash), so the fetch is branch-free (B and Mp are multiples of 4: a
// float4 is either wholly inside or outside).
struct FFSlab {
  v4f v[4];
};

__device__ __forceinline__ FFSlab ff_fetch(const float* src, int64_t ld, int64_t col, int K, int kbase) {
  FFSlab f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = kbase + 2 * s;
    const int kc = k < K ? k : K - 1;
    f.v[s] = *reinterpret_cast<const v4f*>(src + (int64_t)kc * ld + col);
  }
  return f;
}

// LDS slab layout [g][kh][column][s] (k = 8g + 2s + kh): the four k-steps a lane needs for one
// column are one 16-B read, so a group of 4*MT MFMAs is fed by 1 + MT ds_read_b128.  Slabs are
// double-buffered: the global loads of slab i+1 are in flight while slab i feeds the MFMAs.
// 32 KB of LDS and < 170 registers per wave keep 3 workgroups (12 waves) on a CU, which is what hides
// the stash / barrier of one workgroup and the store epilogue of another behind the MFMAs of the third.
template <int MT>
__global__ __launch_bounds__(256) void ff_kernel(const FFArgs a) {
  __shared__ __align__(16) float s_a[2][FF_KT / 4][MT * 32][4];  // W^T slab, index [2g + kh]
  __shared__ __align__(16) float s_b[2][FF_KT / 4][FF_NB][4];    // rates slab
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b0 = (int64_t)blockIdx.x * FF_NB;  // first agent of the block's position tile
  const int m0 = blockIdx.y * (MT * 32);           // first output of the block
  const int t = blockIdx.z;
  // bias rows of this block, read back through LDS in the epilogue: a global load there would make
  // every s_waitcnt vmcnt also wait for the stores issued before it (one write round trip per row)
  __shared__ float s_bias[MT * 32];
  if (tid < MT * 32) s_bias[tid] = m0 + tid < a.n_out ? a.bias[m0 + tid] : 0.0f;
  v16f acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  // fetch role of this thread
  const bool is_b = wave >= 2;                       // wave-uniform
  const int c4 = (tid & 31) * 4, rg = (tid >> 5) & 3;  // rg = 2g + kh
  const int krow = (rg >> 1) * 8 + (rg & 1);
  const bool col_ok = is_b ? (b0 + c4 < a.B) : (c4 < MT * 32 && m0 + c4 < a.Mp);
  const int64_t col = col_ok ? (is_b ? b0 + c4 : (int64_t)m0 + c4) : 0;
  const int64_t ld = is_b ? a.B : (int64_t)a.Mp;
  // columns are XOR-swizzled in LDS (physical = col ^ ((col >> 3) & 7)): the transposed 16-B writes of
  // lanes 4 columns apart and the fragment reads of consecutive columns are both bank-conflict free
  const int sw = (c4 >> 3) & 7;
  float* const s_dst = is_b ? &s_b[0][rg][0][0] : &s_a[0][rg][0][0];
  const int buf_stride = is_b ? (FF_KT / 4) * FF_NB * 4 : (FF_KT / 4) * MT * 32 * 4;
  // 4x4 register transpose [s][column] -> [column][s]; rows >= K and columns outside the problem become 0
  auto stash = [&](const FFSlab& f, int buf, int K, int k0) {
    if (!is_b && c4 >= MT * 32) return;
    v4f v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (col_ok && k0 + krow + 2 * s < K) ? f.v[s] : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<v4f*>(s_dst + buf * buf_stride + ((c4 + i) ^ sw) * 4) = v4f{v[0][i], v[1][i], v[2][i], v[3][i]};
  };
  const int kh = lane >> 5, j = lane & 31;
  const int jb = (wave * 32 + j) ^ (((wave * 32 + j) >> 3) & 7);
  int ja[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ja[mt] = (mt * 32 + j) ^ (((mt * 32 + j) >> 3) & 7);
  int buf = 0;
  for (int l = 0; l < a.n_layers; ++l) {
    const int K = a.n_in[l];
    const float* src = is_b ? a.rates[l] + (int64_t)t * K * a.B : a.wt[l];
    __syncthreads();  // the previous layer's last slab has been consumed
    stash(ff_fetch(src, ld, col, K, krow), buf, K, 0);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += FF_KT) {
      const bool more = k0 + FF_KT < K;  // block-uniform
      FFSlab nxt;
      if (more) nxt = ff_fetch(src, ld, col, K, k0 + FF_KT + krow);
#pragma unroll
      for (int g = 0; g < FF_KT / 8; ++g) {
        const v4f bq = *reinterpret_cast<const v4f*>(&s_b[buf][2 * g + kh][jb][0]);
        v4f aq[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aq[mt] = *reinterpret_cast<const v4f*>(&s_a[buf][2 * g + kh][ja[mt]][0]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[mt][s], bq[s], acc[mt], 0, 0, 0);
      }
      if (more) {
        stash(nxt, buf ^ 1, K, k0 + FF_KT);  // the other buffer was last read one iteration ago (barrier below)
        __syncthreads();
        buf ^= 1;
      }
    }
  }
  // ---- epilogue: bias + activation, C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int64_t b = b0 + wave * 32 + (lane & 31);
  if (b < a.B) {
    const int row0 = 4 * (lane >> 5);
    const int64_t off0 = ((int64_t)t * a.n_out + m0 + row0) * a.B + b;
    const int rows_left = a.n_out - m0 - row0;  // rows ml (relative to row0) < rows_left are real outputs
    auto store_tiles = [&](auto act_tag, auto prime_tag) {
      constexpr int ACT = decltype(act_tag)::value;
      constexpr bool PRIME = decltype(prime_tag)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = mt * 32 + (r & 3) + 8 * (r >> 2);  // compile-time
          if (ml < rows_left) {
            float d;
            const float f = activate<ACT>(acc[mt][r] + s_bias[ml + row0], a.p0, a.p1, a.p2, a.p3, &d);
            a.out[off0 + (int64_t)ml * a.B] = f;
            if (PRIME) a.out_prime[off0 + (int64_t)ml * a.B] = d;
          }
        }
      }
    };
    auto with_act = [&](auto act_tag) {
      if (a.out_prime) store_tiles(act_tag, std::true_type{});
      else store_tiles(act_tag, std::false_type{});
    };
    switch (a.act) {  // block-uniform
      case RIAB_ACT_LINEAR: with_act(std::integral_constant<int, RIAB_ACT_LINEAR>{}); break;
      case RIAB_ACT_SIGMOID: with_act(std::integral_constant<int, RIAB_ACT_SIGMOID>{}); break;
      case RIAB_ACT_RELU: with_act(std::integral_constant<int, RIAB_ACT_RELU>{}); break;
      case RIAB_ACT_TANH: with_act(std::integral_constant<int, RIAB_ACT_TANH>{}); break;
      case RIAB_ACT_RETANH: with_act(std::integral_constant<int, RIAB_ACT_RETANH>{}); break;
      default: with_act(std::integral_constant<int, RIAB_ACT_SOFTMAX>{}); break;
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
  // outputs per block: 128 (4 MFMA row tiles), 64 or 32 for narrow layers
  const int mt = a.Mp >= 128 ? 4 : (a.Mp >= 64 ? 2 : 1);
  const dim3 grid((unsigned)((B + FF_NB - 1) / FF_NB), (unsigned)((a.Mp + mt * 32 - 1) / (mt * 32)), (unsigned)T);
  if (mt == 4) hipLaunchKernelGGL(ff_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (mt == 2) hipLaunchKernelGGL(ff_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(ff_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
