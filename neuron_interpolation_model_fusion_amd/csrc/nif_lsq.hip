// nif_lsq.hip -- the least-squares refit W = pinv(X) A of HF / KF (hf_kf.py:213,233-234,377,397) as one entry point.
//
// torch.linalg.pinv on f32 data drops the singular values <= max(m, d) * eps_f32 * sigma_max; the refit reproduces
// that on the normal equations, in f64:
//   m >= d (primal):  G = X^T X, R = X^T A;  G = V L V^T;  W = V L_kept^-1 V^T R
//   m <  d (dual):    K = X X^T;  K = U L U^T;  W = X^T (U L_kept^-1 U^T A)
// with "kept" = eigenvalues above (max(m, d) eps_f32)^2 * lambda_max.  Every product runs on the f64 matrix cores
// (nif_xty), the eigen-decomposition is nif_eigh_kept_f64.  Columns of X that are exactly zero (dead ReLU channels of
// the previous layer times their kernel taps) are eigenvectors with eigenvalue 0: they are taken out before the
// factorisation and their rows of W are zero, as pinv leaves them.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

static inline size_t lq256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline unsigned grid_rows(int64_t rows) { return (unsigned)(rows < 1 ? 1 : (rows > 32768 ? 32768 : rows)); }  // grid y extent

// idx[0 .. count) = indices i with G[i][i] > 0, ascending; count[0] = their number.  One workgroup.
__global__ __launch_bounds__(1024) void lsq_live_kernel(const double* __restrict__ G, int64_t ldg, int d, int* __restrict__ idx,
                                                        int* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < d; i0 += 1024) {
    const int i = i0 + tid;
    const bool live = i < d && G[(size_t)i * ldg + i] > 0.0;
    const unsigned long long m = __ballot(live);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (live) idx[off + before] = i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wsum[w];
      base += t;
    }
    __syncthreads();
  }
  if (tid == 0) count[0] = base;
}

__global__ void lsq_gather_sym_kernel(const double* __restrict__ G, int64_t ldg, const int* __restrict__ idx, int nl,
                                      double* __restrict__ Gc, int64_t ldc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < nl) Gc[(size_t)r * ldc + c] = G[(size_t)idx[r] * ldg + idx[c]];
}

// out[r][c] = in[idx[r]][c] (gather) or out[idx[r]][c] = in[r][c] (scatter), N columns
__global__ void lsq_rows_kernel(const double* __restrict__ in, int64_t ldi, const int* __restrict__ idx, int rows, int N,
                                double* __restrict__ out, int64_t ldo, int scatter) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= N || r >= rows) return;
  if (scatter) out[(size_t)idx[r] * ldo + c] = in[(size_t)r * ldi + c];
  else out[(size_t)r * ldo + c] = in[(size_t)idx[r] * ldi + c];
}

// Z[k][c] /= lam[k]
__global__ void lsq_scale_rows_kernel(double* __restrict__ Z, int64_t ldz, const double* __restrict__ lam, int K, int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (c < N && k < K) Z[(size_t)k * ldz + c] /= lam[k];
}

// column scaling for norm_acts (fusion_utils.py:38-40): R[:, c] /= sqrt(na[c]); na[c] = 1 afterwards (second kernel)
// (rows walk the grid's y extent with a stride: the dual form puts d, which nothing bounds, on this axis)
__global__ void lsq_scale_cols_kernel(double* __restrict__ R, int64_t ldr, const double* __restrict__ na, int rows, int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const double nrm = sqrt(na[c]);
  for (int r = blockIdx.y; r < rows; r += gridDim.y) R[(size_t)r * ldr + c] /= nrm;
}
__global__ void lsq_ones_kernel(double* __restrict__ v, int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < N) v[c] = 1.0;
}

// f32 -> f64 copy of an m x n matrix, optionally dividing column c by sqrt(na[c])
__global__ void lsq_widen_kernel(const float* __restrict__ in, int64_t ldi, int64_t m, int n, const double* __restrict__ na,
                                 double* __restrict__ out, int64_t ldo) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = blockIdx.y;
  if (c >= n || r >= m) return;
  double v = (double)in[r * ldi + c];
  if (na) v /= sqrt(na[c]);
  out[r * ldo + c] = v;
}

__global__ void lsq_zero_kernel(double* __restrict__ W, int64_t ldw, int rows, int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) W[(size_t)r * ldw + c] = 0.0;
}

struct LsqSolveWs {
  size_t off_idx, off_cnt, off_Gc, off_Rc, off_lam, off_V, off_Z, off_Wc, off_xty, off_eigh, total;
  size_t xty_bytes, eigh_bytes;
  int64_t ldc;
};

// Upper bound of nif_xty's workspace on the f64 path for outputs up to n0 x n1, whatever the contraction length: the
// split-K plan aims at no more than 512 workgroups, so it holds at most max(512, tiles) slabs of 128 x 128 doubles.
static size_t xty_f64_bound(int n0 = 128, int n1 = 128) {
  const size_t p0 = ((size_t)n0 + 127) / 128 * 128, p1 = ((size_t)n1 + 127) / 128 * 128;
  const size_t whole = p0 * p1 * 8 + ((size_t)4 << 20);
  const size_t split = (size_t)72 << 20;
  return whole > split ? whole : split;
}

static LsqSolveWs lsq_solve_ws(int d, int N) {
  LsqSolveWs w;
  w.ldc = (d + 1) & ~(int64_t)1;
  size_t o = 0;
  auto take = [&o](size_t bytes) { const size_t at = o; o += lq256(bytes); return at; };
  w.off_idx = take((size_t)d * 4);
  w.off_cnt = take(64);
  w.off_Gc = take((size_t)d * w.ldc * 8);
  w.off_Rc = take((size_t)d * N * 8);
  w.off_lam = take((size_t)d * 8);
  w.off_V = take((size_t)d * d * 8);
  w.off_Z = take((size_t)d * N * 8);
  w.off_Wc = take((size_t)d * N * 8);
  w.xty_bytes = xty_f64_bound(d, N > d ? N : d);
  w.off_xty = take(w.xty_bytes);
  w.eigh_bytes = nif_eigh_kept_workspace_bytes(d);
  w.off_eigh = take(w.eigh_bytes);
  w.total = o;
  return w;
}

}  // namespace nif

extern "C" size_t nif_lsq_solve_workspace_bytes(int d, int N) {
  if (d <= 0 || N <= 0 || d > NIF_EIGH_MAX_N) return 0;
  return nif::lsq_solve_ws(d, N).total;
}

extern "C" int nif_lsq_solve_f64(const double* G, int64_t ldg, const double* R, int64_t ldr, int d, int N, double cutoff_rel,
                                 double* W, int64_t ldw, int32_t* rank_host, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_lsq_solve_f64");
  using namespace nif;
  if (!G || !R || !W || !rank_host || d <= 0 || N <= 0 || ldg < d || ldr < N || ldw < N || !(cutoff_rel >= 0.0))
    return NIF_ERR_INVALID_ARG;
  if (d > NIF_EIGH_MAX_N) return NIF_ERR_UNSUPPORTED;
  const LsqSolveWs w = lsq_solve_ws(d, N);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  int* idx = reinterpret_cast<int*>(b + w.off_idx);
  int* cnt = reinterpret_cast<int*>(b + w.off_cnt);
  double* Gc = reinterpret_cast<double*>(b + w.off_Gc);
  double* Rc = reinterpret_cast<double*>(b + w.off_Rc);
  double* lam = reinterpret_cast<double*>(b + w.off_lam);
  double* V = reinterpret_cast<double*>(b + w.off_V);
  double* Z = reinterpret_cast<double*>(b + w.off_Z);
  double* Wc = reinterpret_cast<double*>(b + w.off_Wc);
  *rank_host = 0;

  // live columns (G[i][i] > 0); the count sizes everything below, so the host waits for it
  hipLaunchKernelGGL(lsq_live_kernel, dim3(1), dim3(1024), 0, st, G, ldg, d, idx, cnt);
  int nl = 0;
  if (hipMemcpyAsync(&nl, cnt, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return NIF_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return NIF_ERR_LAUNCH;
  if (nl < 0 || nl > d) return NIF_ERR_LAUNCH;
  const dim3 tb(256);
  if (nl < d) hipLaunchKernelGGL(lsq_zero_kernel, dim3((N + 255) / 256, grid_rows(d)), tb, 0, st, W, ldw, d, N);
  if (nl == 0) return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;   // X = 0: pinv(X) = 0
  const double* Ge = G;
  int64_t lde = ldg;
  const double* Re = R;
  int64_t ldre = ldr;
  if (nl < d) {
    hipLaunchKernelGGL(lsq_gather_sym_kernel, dim3((nl + 255) / 256, nl), tb, 0, st, G, ldg, idx, nl, Gc, w.ldc);
    hipLaunchKernelGGL(lsq_rows_kernel, dim3((N + 255) / 256, nl), tb, 0, st, R, ldr, idx, nl, N, Rc, (int64_t)N, 0);
    Ge = Gc; lde = w.ldc; Re = Rc; ldre = N;
  }
  if (!launch_ok()) return NIF_ERR_LAUNCH;
  int32_t K = 0;
  int rc = nif_eigh_kept_f64(Ge, lde, nl, cutoff_rel, lam, V, nl, &K, b + w.off_eigh, w.eigh_bytes, stream);
  if (rc != NIF_OK) return rc;
  *rank_host = K;
  double* Wout = nl < d ? Wc : W;
  const int64_t ldwo = nl < d ? (int64_t)N : ldw;
  if (K == 0) {
    hipLaunchKernelGGL(lsq_zero_kernel, dim3((N + 255) / 256, grid_rows(d)), tb, 0, st, W, ldw, d, N);
    return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
  }
  // Z = V^T R (K x N), Z /= lam, W = V Z
  rc = nif_xty(V, nl, Re, ldre, nl, K, N, NIF_F64, NIF_F64, 0, Z, N, nullptr, nullptr, b + w.off_xty, w.xty_bytes, stream);
  if (rc != NIF_OK) return rc;
  hipLaunchKernelGGL(lsq_scale_rows_kernel, dim3((N + 255) / 256, K), tb, 0, st, Z, (int64_t)N, lam, K, N);
  rc = nif_xty(V, nl, Z, N, K, nl, N, NIF_F64, NIF_F64, NIF_TRANS_A, Wout, ldwo, nullptr, nullptr, b + w.off_xty, w.xty_bytes,
               stream);
  if (rc != NIF_OK) return rc;
  if (nl < d) hipLaunchKernelGGL(lsq_rows_kernel, dim3((N + 255) / 256, nl), tb, 0, st, Wc, (int64_t)N, idx, nl, N, W, ldw, 1);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

namespace nif {
struct LsqRefitWs {
  size_t off_G, off_K, off_lam, off_U, off_A64, off_X64, off_Z, off_Y, off_xty, off_sub, total;
  size_t xty_bytes, sub_bytes;
  bool dual;
};

static LsqRefitWs lsq_refit_ws(int64_t m, int d, int N, int flags) {
  LsqRefitWs w{};
  w.dual = m < d && !(flags & NIF_LSQ_PRIMAL);
  size_t o = 0;
  auto take = [&o](size_t bytes) { const size_t at = o; o += lq256(bytes); return at; };
  w.xty_bytes = 0;
  // Gram builds: the f32-in / f64-accumulate plan for the actual shapes
  size_t x1 = nif_xty_workspace_bytes(m, d, N, NIF_F64, 0);
  if (w.dual) {
    const size_t x2 = nif_xty_workspace_bytes(d, (int)m, (int)m, NIF_F64, NIF_SYMMETRIC | NIF_TRANS_A | NIF_TRANS_B);
    if (x2 > x1) x1 = x2;
    const size_t x3 = nif_xty_workspace_bytes(m, d, N, NIF_F64, 0);
    if (x3 > x1) x1 = x3;
  } else {
    const size_t x2 = nif_xty_workspace_bytes(m, d, d, NIF_F64, NIF_SYMMETRIC);
    if (x2 > x1) x1 = x2;
  }
  const int big = (int)(w.dual ? (m > d ? m : d) : d);
  if (xty_f64_bound(big, N > big ? N : big) > x1) x1 = xty_f64_bound(big, N > big ? N : big);
  w.xty_bytes = x1;
  w.off_xty = take(w.xty_bytes);
  if (w.dual) {
    const int mm = (int)m;
    w.off_K = take((size_t)mm * mm * 8);
    w.off_lam = take((size_t)mm * 8);
    w.off_U = take((size_t)mm * mm * 8);
    w.off_A64 = take((size_t)mm * N * 8);
    w.off_X64 = take((size_t)mm * d * 8);
    w.off_Z = take((size_t)mm * N * 8);
    w.off_Y = take((size_t)mm * N * 8);
    w.sub_bytes = nif_eigh_kept_workspace_bytes(mm);
  } else {
    w.off_G = take((size_t)d * d * 8);
    w.sub_bytes = nif_lsq_solve_workspace_bytes(d, N);
  }
  w.off_sub = take(w.sub_bytes);
  w.total = o;
  return w;
}
}  // namespace nif

extern "C" size_t nif_lsq_refit_workspace_bytes(int64_t m, int d, int N, int flags) {
  if (m <= 0 || d <= 0 || N <= 0) return 0;
  const bool dual = m < d && !(flags & NIF_LSQ_PRIMAL);
  if ((dual ? m : (int64_t)d) > NIF_EIGH_MAX_N) return 0;
  return nif::lsq_refit_ws(m, d, N, flags).total;
}

extern "C" int nif_lsq_refit_f32(const float* X, int64_t ldx, int64_t m, int d, const float* A, int64_t lda, int N, int flags,
                                 double* W, int64_t ldw, double* R, int64_t ldr, double* na, int32_t* rank_host, void* ws,
                                 size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_lsq_refit_f32");
  using namespace nif;
  if (!X || !A || !W || !R || !na || !rank_host || m <= 0 || d <= 0 || N <= 0 || ldx < d || lda < N || ldw < N || ldr < N)
    return NIF_ERR_INVALID_ARG;
  const bool dual = m < d && !(flags & NIF_LSQ_PRIMAL);
  if ((dual ? m : (int64_t)d) > NIF_EIGH_MAX_N) return NIF_ERR_UNSUPPORTED;
  const LsqRefitWs w = lsq_refit_ws(m, d, N, flags);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  void* xws = b + w.off_xty;
  const dim3 tb(256);
  const double eps32 = 1.1920928955078125e-07;
  const double rtol = (double)(m > d ? m : (int64_t)d) * eps32;   // torch.linalg.pinv's default for f32 inputs
  const double cutoff = rtol * rtol;
  *rank_host = 0;

  // R = X^T A and the squared column norms of A (the HF cost needs both: hf_kf.py:109-119 in closed form)
  int rc = nif_xty(X, ldx, A, lda, m, d, N, NIF_F32, NIF_F64, 0, R, ldr, nullptr, na, xws, w.xty_bytes, stream);
  if (rc != NIF_OK) return rc;
  const bool normalize = (flags & NIF_LSQ_NORMALIZE) != 0;
  if (normalize && !dual) {
    hipLaunchKernelGGL(lsq_scale_cols_kernel, dim3((N + 255) / 256, grid_rows(d)), tb, 0, st, R, ldr, na, d, N);
    hipLaunchKernelGGL(lsq_ones_kernel, dim3((N + 255) / 256), tb, 0, st, na, N);
  }
  if (!dual) {
    double* G = reinterpret_cast<double*>(b + w.off_G);
    rc = nif_xty(X, ldx, X, ldx, m, d, d, NIF_F32, NIF_F64, NIF_SYMMETRIC, G, d, nullptr, nullptr, xws, w.xty_bytes, stream);
    if (rc != NIF_OK) return rc;
    return nif_lsq_solve_f64(G, d, R, ldr, d, N, cutoff, W, ldw, rank_host, b + w.off_sub, w.sub_bytes, stream);
  }
  // dual form: every row of X is needed on this device
  const int mm = (int)m;
  double* Km = reinterpret_cast<double*>(b + w.off_K);
  double* lam = reinterpret_cast<double*>(b + w.off_lam);
  double* U = reinterpret_cast<double*>(b + w.off_U);
  double* A64 = reinterpret_cast<double*>(b + w.off_A64);
  double* X64 = reinterpret_cast<double*>(b + w.off_X64);
  double* Z = reinterpret_cast<double*>(b + w.off_Z);
  double* Y = reinterpret_cast<double*>(b + w.off_Y);
  rc = nif_xty(X, ldx, X, ldx, d, mm, mm, NIF_F32, NIF_F64, NIF_SYMMETRIC | NIF_TRANS_A | NIF_TRANS_B, Km, mm, nullptr,
               nullptr, xws, w.xty_bytes, stream);
  if (rc != NIF_OK) return rc;
  hipLaunchKernelGGL(lsq_widen_kernel, dim3((N + 255) / 256, (unsigned)m), tb, 0, st, A, lda, m, N, normalize ? na : nullptr,
                     A64, (int64_t)N);
  hipLaunchKernelGGL(lsq_widen_kernel, dim3((d + 255) / 256, (unsigned)m), tb, 0, st, X, ldx, m, d, (const double*)nullptr,
                     X64, (int64_t)d);
  if (normalize) {  // after A64 took the norms: R and na in their normalised form
    hipLaunchKernelGGL(lsq_scale_cols_kernel, dim3((N + 255) / 256, grid_rows(d)), tb, 0, st, R, ldr, na, d, N);
    hipLaunchKernelGGL(lsq_ones_kernel, dim3((N + 255) / 256), tb, 0, st, na, N);
  }
  if (!launch_ok()) return NIF_ERR_LAUNCH;
  int32_t K = 0;
  rc = nif_eigh_kept_f64(Km, mm, mm, cutoff, lam, U, mm, &K, b + w.off_sub, w.sub_bytes, stream);
  if (rc != NIF_OK) return rc;
  *rank_host = K;
  if (K == 0) {
    hipLaunchKernelGGL(lsq_zero_kernel, dim3((N + 255) / 256, grid_rows(d)), tb, 0, st, W, ldw, d, N);
    return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
  }
  rc = nif_xty(U, mm, A64, N, mm, K, N, NIF_F64, NIF_F64, 0, Z, N, nullptr, nullptr, xws, w.xty_bytes, stream);  // U^T A
  if (rc != NIF_OK) return rc;
  hipLaunchKernelGGL(lsq_scale_rows_kernel, dim3((N + 255) / 256, K), tb, 0, st, Z, (int64_t)N, lam, K, N);
  rc = nif_xty(U, mm, Z, N, K, mm, N, NIF_F64, NIF_F64, NIF_TRANS_A, Y, N, nullptr, nullptr, xws, w.xty_bytes, stream);  // U Z
  if (rc != NIF_OK) return rc;
  rc = nif_xty(X64, d, Y, N, mm, d, N, NIF_F64, NIF_F64, 0, W, ldw, nullptr, nullptr, xws, w.xty_bytes, stream);  // X^T Y
  if (rc != NIF_OK) return rc;
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
