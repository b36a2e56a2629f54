// nif_cost.hip -- O(1)-per-pair cost-matrix epilogues on top of the Gram quantities.
// HBM-bound elementwise kernels: 8 bytes read + 4 written per (i, j), coalesced along j.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

// Euclidean distance of activation columns from the Gram form, optionally scaled by the
// harmonic importance weight 2 s0 s1 / (s0 + s1) (git_rebasin.py:70-77 in closed form:
// avg - a0 = b (a1 - a0), avg - a1 = a (a0 - a1), a = s0/(s0+s1), b = 1 - a, so
// s0 ||avg - a0|| + s1 ||avg - a1|| = (s0 b + s1 a) ||a0 - a1||) and by a per-column factor.
__global__ void cost_dist_kernel(const double* __restrict__ G, int64_t ldg, const double* __restrict__ na,
                                 const double* __restrict__ nb, const float* __restrict__ s0,
                                 const float* __restrict__ s1, const float* __restrict__ colscale, int n0, int n1,
                                 float* __restrict__ C, int64_t ldc) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n1 || i >= n0) return;
  double d2 = na[i] + nb[j] - 2.0 * G[(int64_t)i * ldg + j];
  if (!(d2 > 0.0)) d2 = 0.0;
  double c = sqrt(d2);
  if (s0 != nullptr) {
    const double a = (double)s0[i], b = (double)s1[j];
    c *= 2.0 * a * b / (a + b);
  }
  if (colscale != nullptr) c *= (double)colscale[j];
  C[(int64_t)i * ldc + j] = (float)c;
}

// hf_kf.py:109-119.  pred - a0_i = b (p1_j - p0_i) - (a0_i - p0_i): the first term lies in
// col(X), the second is the LS residual, orthogonal to col(X), so the squared norm splits.
__global__ void cost_hf_kernel(const double* __restrict__ G01, int64_t ldg, const double* __restrict__ pp0,
                               const double* __restrict__ pp1, const double* __restrict__ r0,
                               const double* __restrict__ r1, const float* __restrict__ s0,
                               const float* __restrict__ s1, int n0, int n1, float* __restrict__ C, int64_t ldc) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n1 || i >= n0) return;
  double d2 = pp0[i] + pp1[j] - 2.0 * G01[(int64_t)i * ldg + j];
  if (!(d2 > 0.0)) d2 = 0.0;
  const double si = (double)s0[i], sj = (double)s1[j];
  const double a = si / (si + sj), b = sj / (si + sj);
  double e0 = b * b * d2 + r0[i];
  double e1 = a * a * d2 + r1[j];
  if (!(e0 > 0.0)) e0 = 0.0;
  if (!(e1 > 0.0)) e1 = 0.0;
  C[(int64_t)i * ldc + j] = (float)(si * sqrt(e0) + sj * sqrt(e1));
}

}  // namespace nif

extern "C" int nif_cost_dist(const double* G, int64_t ldg, const double* na, const double* nb, const float* s0,
                             const float* s1, const float* colscale, int n0, int n1, float* C, int64_t ldc,
                             void* stream) {
  nif::Range nif_range_("nif_cost_dist");
  if (!G || !na || !nb || !C || n0 <= 0 || n1 <= 0 || ldg < n1 || ldc < n1) return NIF_ERR_INVALID_ARG;
  if ((s0 == nullptr) != (s1 == nullptr)) return NIF_ERR_INVALID_ARG;
  dim3 grid((n1 + 255) / 256, n0);
  nif::clear_err();
  hipLaunchKernelGGL(nif::cost_dist_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G, ldg, na, nb,
                     s0, s1, colscale, n0, n1, C, ldc);
  return nif::launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_cost_hf(const double* G01, int64_t ldg, const double* pp0, const double* pp1, const double* r0,
                           const double* r1, const float* s0, const float* s1, int n0, int n1, float* C, int64_t ldc,
                           void* stream) {
  nif::Range nif_range_("nif_cost_hf");
  if (!G01 || !pp0 || !pp1 || !r0 || !r1 || !s0 || !s1 || !C || n0 <= 0 || n1 <= 0 || ldg < n1 || ldc < n1)
    return NIF_ERR_INVALID_ARG;
  dim3 grid((n1 + 255) / 256, n0);
  nif::clear_err();
  hipLaunchKernelGGL(nif::cost_hf_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G01, ldg, pp0,
                     pp1, r0, r1, s0, s1, n0, n1, C, ldc);
  return nif::launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
