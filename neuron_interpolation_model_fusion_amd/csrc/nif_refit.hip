// nif_refit.hip -- weight write-back under a matching: a row/column gather of the aligned
// model's weights fused with the (importance-weighted) average.  HBM-bound: 8 bytes read and
// 4 written per output element, coalesced along the input axis.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

__global__ void refit_gather_kernel(const float* __restrict__ W0, const float* __restrict__ W1,
                                    const int32_t* __restrict__ rowsrc, const int32_t* __restrict__ colsrc,
                                    const float* __restrict__ alpha_v, const float* __restrict__ beta_v, float alpha,
                                    float beta, int n_out, int n_in, int sp, float* __restrict__ out) {
  const int i = blockIdx.y;
  const int64_t width = (int64_t)n_in * sp;
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out || x >= width) return;
  const int c = (int)(x / sp);
  const int e = (int)(x - (int64_t)c * sp);
  const int ri = rowsrc ? rowsrc[i] : i;
  const int ci = colsrc ? colsrc[c] : c;
  const float a = alpha_v ? alpha_v[i] : alpha;
  const float b = beta_v ? beta_v[i] : beta;
  // -1 = a unit the assignment left unmatched (non-finite costs): contributes nothing instead of reading out of bounds
  const float w1 = (ri < 0 || ci < 0) ? 0.f : W1[(int64_t)ri * width + (int64_t)ci * sp + e];
  const float w0 = W0 ? W0[(int64_t)i * width + x] : 0.f;
  out[(int64_t)i * width + x] = a * w0 + b * w1;
}

// HF pair interpolation: the LS solutions are the COLUMNS of W (d x N, f64); the fused layer wants
// one ROW per neuron.  32 x 32 tiles go through LDS so both the reads (along N) and the writes
// (along d) are coalesced.
__global__ void refit_pairs_kernel(const double* __restrict__ W, int64_t ldw, const int32_t* __restrict__ col,
                                   const float* __restrict__ s0, const float* __restrict__ s1, int n0, int d,
                                   float* __restrict__ out) {
  __shared__ float t0[32][33];
  __shared__ float t1[32][33];
  const int i0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int rr = ty; rr < 32; rr += 8) {
    const int r = r0 + rr, i = i0 + tx;
    float a = 0.f, b = 0.f;
    if (r < d && i < n0) {
      a = (float)W[(int64_t)r * ldw + i];                 // coalesced along i
      const int ci = col[i];
      b = ci < 0 ? 0.f : (float)W[(int64_t)r * ldw + n0 + ci];  // gathered partner column (-1: unmatched)
    }
    t0[rr][tx] = a;
    t1[rr][tx] = b;
  }
  __syncthreads();
  for (int ii = ty; ii < 32; ii += 8) {
    const int i = i0 + ii, r = r0 + tx;
    if (i < n0 && r < d) {
      const int ci = col[i];
      const float si = s0[i], sj = ci < 0 ? 0.f : s1[ci];
      const float a = si / (si + sj), b = sj / (si + sj);
      out[(int64_t)i * d + r] = a * t0[tx][ii] + b * t1[tx][ii];
    }
  }
}

}  // namespace nif

extern "C" int nif_refit_pairs(const double* W, int64_t ldw, const int32_t* col, const float* s0, const float* s1,
                               int n0, int d, float* out, void* stream) {
  nif::Range nif_range_("nif_refit_pairs");
  if (!W || !col || !s0 || !s1 || !out || n0 <= 0 || d <= 0 || ldw < n0) return NIF_ERR_INVALID_ARG;
  nif::clear_err();
  dim3 grid((n0 + 31) / 32, (d + 31) / 32);
  hipLaunchKernelGGL(nif::refit_pairs_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, ldw, col,
                     s0, s1, n0, d, out);
  return nif::launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_refit_gather(const float* W0, const float* W1, const int32_t* rowsrc, const int32_t* colsrc,
                                const float* alpha_v, const float* beta_v, float alpha, float beta, int n_out, int n_in,
                                int sp, float* out, void* stream) {
  nif::Range nif_range_("nif_refit_gather");
  if (!W1 || !out || n_out <= 0 || n_in <= 0 || sp <= 0) return NIF_ERR_INVALID_ARG;
  const int64_t width = (int64_t)n_in * sp;
  dim3 grid((unsigned)((width + 255) / 256), n_out);
  nif::clear_err();
  hipLaunchKernelGGL(nif::refit_gather_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W0, W1,
                     rowsrc, colsrc, alpha_v, beta_v, alpha, beta, n_out, n_in, sp, out);
  return nif::launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
