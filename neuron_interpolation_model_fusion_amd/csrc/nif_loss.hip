// nif_loss.hip -- fused cluster-weighted squared-error loss of the ALF level fit (alf.py:414-422,
// 434-435: mean over all elements of w[channel] * (pred - true)^2, or plain MSE when w is null).
//
// Both operands are [outer][C][inner] contiguous (MLP: inner = 1; conv: inner = H*W; transformer:
// inner = seq), i.e. the level network's own output layout, so no flatten/permute copy is made.
// HBM-bound: forward reads 8 B/element, backward reads 8 and writes 4.  The reduction is
// deterministic: per-thread f64 partials, fixed-shape block tree, per-block partials summed in
// block order by the last pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

constexpr int LOSS_NT = 256;
constexpr int LOSS_MAX_BLOCKS = 2048;

struct ChannelOf {
  int64_t inner;
  int C;
  // channel of flat element i in [outer][C][inner]
  __device__ __forceinline__ int operator()(int64_t i) const {
    if (inner == 1) return (int)(i % C);
    return (int)((i / inner) % C);
  }
};

__device__ __forceinline__ double block_sum_f64(double x, double* sh) {
  x = wave_sum_f64(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < LOSS_NT / 64; ++w) s += sh[w];
  return s;  // valid on thread 0
}

// VEC: 4 consecutive elements per thread per trip (numel % 4 == 0, 16-byte aligned pointers).
template <bool VEC>
__global__ __launch_bounds__(LOSS_NT) void wmse_partial_kernel(const float* __restrict__ pred,
                                                               const float* __restrict__ truth,
                                                               const float* __restrict__ w, int64_t numel,
                                                               ChannelOf ch, double* __restrict__ partial) {
  __shared__ double sh[LOSS_NT / 64];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * LOSS_NT;
  if (VEC) {
    const int64_t nv = numel >> 2;
    for (int64_t v = (int64_t)blockIdx.x * LOSS_NT + threadIdx.x; v < nv; v += stride) {
      const float4 p = reinterpret_cast<const float4*>(pred)[v];
      const float4 t = reinterpret_cast<const float4*>(truth)[v];
      const float d[4] = {p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ww = w ? w[ch(4 * v + e)] : 1.0f;
        acc += (double)(d[e] * d[e] * ww);
      }
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * LOSS_NT + threadIdx.x; i < numel; i += stride) {
      const float d = pred[i] - truth[i];
      const float ww = w ? w[ch(i)] : 1.0f;
      acc += (double)(d * d * ww);
    }
  }
  const double s = block_sum_f64(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void wmse_final_kernel(const double* __restrict__ partial, int nblocks, int64_t numel,
                                                        float* __restrict__ loss) {
  // one wave; lane l sums partial[l], partial[l + 64], ... in order, then a fixed xor tree
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[b];
  s = wave_sum_f64(s);
  if (threadIdx.x == 0) loss[0] = (float)(s / (double)numel);
}

template <bool VEC>
__global__ __launch_bounds__(LOSS_NT) void wmse_backward_kernel(const float* __restrict__ pred,
                                                                const float* __restrict__ truth,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ grad_out, int64_t numel,
                                                                ChannelOf ch, float* __restrict__ grad_pred) {
  // d/dpred mean(w (p - t)^2) = 2 w (p - t) / numel, times the incoming scalar gradient
  const float scale = grad_out[0] * (float)(2.0 / (double)numel);
  const int64_t stride = (int64_t)gridDim.x * LOSS_NT;
  if (VEC) {
    const int64_t nv = numel >> 2;
    for (int64_t v = (int64_t)blockIdx.x * LOSS_NT + threadIdx.x; v < nv; v += stride) {
      const float4 p = reinterpret_cast<const float4*>(pred)[v];
      const float4 t = reinterpret_cast<const float4*>(truth)[v];
      float4 g;
      g.x = (p.x - t.x) * (w ? w[ch(4 * v + 0)] : 1.0f) * scale;
      g.y = (p.y - t.y) * (w ? w[ch(4 * v + 1)] : 1.0f) * scale;
      g.z = (p.z - t.z) * (w ? w[ch(4 * v + 2)] : 1.0f) * scale;
      g.w = (p.w - t.w) * (w ? w[ch(4 * v + 3)] : 1.0f) * scale;
      reinterpret_cast<float4*>(grad_pred)[v] = g;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * LOSS_NT + threadIdx.x; i < numel; i += stride)
      grad_pred[i] = (pred[i] - truth[i]) * (w ? w[ch(i)] : 1.0f) * scale;
  }
}

static inline int loss_blocks(int64_t numel, bool vec) {
  const int64_t per_block = (int64_t)LOSS_NT * (vec ? 4 : 1) * 4;  // ~4 trips per thread at least
  int64_t b = (numel + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;
  return (int)b;
}

static inline bool loss_vec_ok(const void* a, const void* b, const void* c, int64_t numel) {
  return (numel % 4 == 0) && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) % 16 == 0);
}

}  // namespace nif

extern "C" size_t nif_wmse_workspace_bytes(void) { return (size_t)nif::LOSS_MAX_BLOCKS * sizeof(double); }

extern "C" int nif_wmse_forward(const float* pred, const float* truth, const float* w, int64_t outer, int C,
                                int64_t inner, float* loss, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_wmse_forward");
  using namespace nif;
  if (!pred || !truth || !loss || outer <= 0 || C <= 0 || inner <= 0) return NIF_ERR_INVALID_ARG;
  if (!ws || ws_bytes < nif_wmse_workspace_bytes()) return NIF_ERR_WORKSPACE;
  const int64_t numel = outer * (int64_t)C * inner;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  const bool vec = loss_vec_ok(pred, truth, pred, numel);
  const int nb = loss_blocks(numel, vec);
  ChannelOf ch{inner, C};
  clear_err();
  if (vec)
    hipLaunchKernelGGL(wmse_partial_kernel<true>, dim3(nb), dim3(LOSS_NT), 0, st, pred, truth, w, numel, ch, partial);
  else
    hipLaunchKernelGGL(wmse_partial_kernel<false>, dim3(nb), dim3(LOSS_NT), 0, st, pred, truth, w, numel, ch, partial);
  hipLaunchKernelGGL(wmse_final_kernel, dim3(1), dim3(64), 0, st, partial, nb, numel, loss);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_wmse_backward(const float* pred, const float* truth, const float* w, const float* grad_out,
                                 int64_t outer, int C, int64_t inner, float* grad_pred, void* stream) {
  nif::Range nif_range_("nif_wmse_backward");
  using namespace nif;
  if (!pred || !truth || !grad_out || !grad_pred || outer <= 0 || C <= 0 || inner <= 0) return NIF_ERR_INVALID_ARG;
  const int64_t numel = outer * (int64_t)C * inner;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = loss_vec_ok(pred, truth, grad_pred, numel);
  const int nb = loss_blocks(numel, vec);
  ChannelOf ch{inner, C};
  clear_err();
  if (vec)
    hipLaunchKernelGGL(wmse_backward_kernel<true>, dim3(nb), dim3(LOSS_NT), 0, st, pred, truth, w, grad_out, numel, ch,
                       grad_pred);
  else
    hipLaunchKernelGGL(wmse_backward_kernel<false>, dim3(nb), dim3(LOSS_NT), 0, st, pred, truth, w, grad_out, numel,
                       ch, grad_pred);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
