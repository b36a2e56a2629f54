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
  const float w1 = W1[(int64_t)ri * width + (int64_t)ci * sp + e];
  const float w0 = W0 ? W0[(int64_t)i * width + x] : 0.f;
  out[(int64_t)i * width + x] = a * w0 + b * w1;
}

}  // namespace nif

extern "C" int nif_refit_gather(const float* W0, const float* W1, const int32_t* rowsrc, const int32_t* colsrc,
                                const float* alpha_v, const float* beta_v, float alpha, float beta, int n_out, int n_in,
                                int sp, float* out, void* stream) {
  if (!W1 || !out || n_out <= 0 || n_in <= 0 || sp <= 0) return NIF_ERR_INVALID_ARG;
  const int64_t width = (int64_t)n_in * sp;
  dim3 grid((unsigned)((width + 255) / 256), n_out);
  nif::clear_err();
  hipLaunchKernelGGL(nif::refit_gather_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W0, W1,
                     rowsrc, colsrc, alpha_v, beta_v, alpha, beta, n_out, n_in, sp, out);
  return nif::launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
