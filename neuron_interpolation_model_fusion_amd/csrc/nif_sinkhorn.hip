// nif_sinkhorn.hip -- entropic optimal transport (Sinkhorn-Knopp) for dense n0 x n1 costs: an ADDITIONAL,
// documented-approximate solver for OTFusion's plan.  The reference has no such line -- it calls the exact
// `ot.emd` (otfusion.py:18-32, :309) -- so this is not a parity path: it restates the published
// algorithm of POT's `ot.sinkhorn(a, b, M, reg)` (method 'sinkhorn', POT 0.9.x `sinkhorn_knopp`):
//     K = exp(-M / reg); u = 1/n0, v = 1/n1
//     repeat: v = b / (K^T u); u = a / (K v); every 10th iteration err = || v * (K^T u) - b ||_2, stop below stopThr
//     T = diag(u) K diag(v)
// ("parity unpinned": POT is not installed here; tests check it against a torch f64 restatement and by its fixed-point
// properties.)
//
// HBM-bound: K (f32, n0 x n1) is streamed twice per iteration -- a column pass (K^T u) and a row pass (K v) -- 8 n0 n1
// bytes per iteration.  The scalings u, v and every sum are f64.  The convergence test costs no pass of its
// own: the column sums K^T u of iteration t + 1 are exactly what the marginal error of the state after iteration t
// needs, so the column pass's finishing kernel evaluates the error BEFORE it overwrites v, and freezes the state when
// it is below the threshold.  "Frozen" is a device flag every kernel reads first, so the host can enqueue iterations
// in batches without a synchronisation per check.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

constexpr int SK_COLS = 256;      // columns per column-pass workgroup (64 lanes x 4)
constexpr int SK_ROWCHUNKS = 64;  // row chunks of the column pass at most (partials [chunk][n1])

struct SkWs {
  size_t off_part, off_ctl, off_errp, total;
  int chunks, rows_per_chunk;
};
static SkWs sk_ws(int n0, int n1) {
  SkWs w;
  int chunks = (n0 + 63) / 64;
  if (chunks > SK_ROWCHUNKS) chunks = SK_ROWCHUNKS;
  if (chunks < 1) chunks = 1;
  w.rows_per_chunk = (n0 + chunks - 1) / chunks;
  w.chunks = (n0 + w.rows_per_chunk - 1) / w.rows_per_chunk;
  size_t o = 0;
  auto take = [&](size_t b) {
    const size_t at = o;
    o += (b + 255) & ~(size_t)255;
    return at;
  };
  w.off_part = take((size_t)w.chunks * n1 * sizeof(double));
  w.off_ctl = take(64);                                          // int ctl[4]: state, iterations done, checks done
  w.off_errp = take((size_t)((n1 + 63) / 64) * sizeof(double));  // per-block partial of the squared error
  w.total = o;
  return w;
}

// Iteration q (0-based, passed by the host) reads the u, v buffers of index q & 1 and writes the other pair.
// ctl[0]: 0 running, 1 converged, 2 numerical breakdown (a zero / non-finite column sum or scaling); when the state is
// frozen, ctl[1] = iterations completed and ctl[2] = index of the u, v buffers that hold the result; ctl[3]: set by a
// row pass that produced a zero / non-finite scaling (the next kernel to look freezes the state one iteration back).

__global__ void sk_init_kernel(const float* __restrict__ C, int64_t ldc, int n0, int n1, double reg,
                               float* __restrict__ K, int64_t ldk, double* __restrict__ u, double* __restrict__ v,
                               int* __restrict__ ctl, double* __restrict__ err) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n0 * n1;
  if (idx < total) {
    const int i = (int)(idx / n1), j = (int)(idx - (int64_t)i * n1);
    K[(int64_t)i * ldk + j] = (float)exp(-(double)C[(int64_t)i * ldc + j] / reg);
  }
  if (idx < n0) u[idx] = 1.0 / n0;
  if (idx < n1) v[idx] = 1.0 / n1;
  if (idx == 0) {
    ctl[0] = 0;
    ctl[1] = 0;
    ctl[2] = 0;
    ctl[3] = 0;
    *err = INFINITY;
  }
}

// column pass: part[chunk][j] = sum over the chunk's rows of K[i][j] u[i]; a lane owns 4 consecutive columns
__global__ __launch_bounds__(256) void sk_colpass_kernel(const float* __restrict__ K, int64_t ldk, int n0, int n1,
                                                         const double* __restrict__ ubuf, int rows_per_chunk, int q,
                                                         const int* __restrict__ ctl, double* __restrict__ part) {
  if (ctl[0] != 0 || ctl[3] != 0) return;
  const double* __restrict__ u = ubuf + (size_t)(q & 1) * n0;
  __shared__ double red[4][SK_COLS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = blockIdx.x * SK_COLS + lane * 4;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(n0, r0 + rows_per_chunk);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const bool vec = j0 + 3 < n1 && (ldk % 4 == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0);
  int i = r0 + wv;  // the 4 waves of the block take rows i, i + 1, i + 2, i + 3
  for (; i < r1; i += 4) {
    const double ui = u[i];
    const float* __restrict__ row = K + (int64_t)i * ldk;
    if (vec) {
      const float4 k4 = *reinterpret_cast<const float4*>(row + j0);
      acc[0] += ui * (double)k4.x;
      acc[1] += ui * (double)k4.y;
      acc[2] += ui * (double)k4.z;
      acc[3] += ui * (double)k4.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j0 + q < n1) acc[q] += ui * (double)row[j0 + q];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[wv][lane * 4 + q] = acc[q];
  __syncthreads();
  const int c = threadIdx.x;  // 256 threads, 256 columns
  const int j = blockIdx.x * SK_COLS + c;
  if (j < n1) part[(size_t)blockIdx.y * n1 + j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// finish of the column pass: s_j = sum of the partials in chunk order.  Every `check_every`-th iteration first the
// marginal error of the CURRENT state, err^2 = sum_j (v_j s_j - b_j)^2 (per-block partials, summed in block order by
// the last block to finish); below tol the state is frozen.  Otherwise v_new = b / s goes into the other buffer; a
// zero or non-finite sum freezes the state as a breakdown.
__global__ __launch_bounds__(256) void sk_colfinish_kernel(const double* __restrict__ part, int chunks, int n1,
                                                           const double* __restrict__ b, double* __restrict__ vbuf,
                                                           int check_every, double tol, int q, int* __restrict__ ctl,
                                                           double* __restrict__ errp, double* __restrict__ err,
                                                           unsigned* __restrict__ ticket) {
  __shared__ double red[4];
  __shared__ int sbad;
  __shared__ unsigned order;
  if (ctl[0] != 0) return;
  if (ctl[3] != 0) {  // the row pass of iteration q - 1 broke down: its inputs are the result (POT keeps uprev, vprev)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ctl[1] = q - 1;
      ctl[2] = (q - 1) & 1;
      ctl[0] = 2;
    }
    return;
  }
  const int cur = q & 1;
  const double* __restrict__ v = vbuf + (size_t)cur * n1;
  double* __restrict__ vn = vbuf + (size_t)(cur ^ 1) * n1;
  const int it = q;
  // POT tests after the iterations with ii % 10 == 0, i.e. after 1, 11, 21, ... completed iterations
  const bool check = it > 0 && ((it - 1) % check_every == 0);
  if (threadIdx.x == 0) sbad = 0;
  __syncthreads();
  // 64 columns per block; wave g sums the chunks g, g + 4, ... of its column (coalesced 512-byte reads), the four
  // partial sums are combined in a fixed order
  __shared__ double csum[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  double ps = 0.0;
  if (j < n1)
    for (int c = grp; c < chunks; c += 4) ps += part[(size_t)c * n1 + j];
  csum[grp][lane] = ps;
  __syncthreads();
  double e2 = 0.0;
  if (grp == 0 && j < n1) {
    const double s = (csum[0][lane] + csum[1][lane]) + (csum[2][lane] + csum[3][lane]);
    if (check) {
      const double d = v[j] * s - b[j];
      e2 = d * d;
    }
    const double nv = b[j] / s;
    if (s == 0.0 || !isfinite(nv)) atomicOr(&sbad, 1);
    vn[j] = nv;
  }
  e2 = wave_sum_f64(e2);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e2;
  __syncthreads();
  if (threadIdx.x == 0) {
    errp[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (sbad) atomicOr(reinterpret_cast<int*>(ticket + 1), 1);
    __threadfence();
    order = atomicAdd(ticket, 1u);
  }
  __syncthreads();
  if (order == gridDim.x - 1 && threadIdx.x == 0) {
    __threadfence();
    const int anybad = atomicOr(reinterpret_cast<int*>(ticket + 1), 0);
    ticket[0] = 0u;
    ticket[1] = 0u;
    int state = 0;
    if (check) {
      double tot = 0.0;
      for (unsigned q = 0; q < gridDim.x; ++q) tot += __hip_atomic_load(errp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double e = sqrt(tot);
      *err = e;
      if (e < tol) state = 1;
    }
    if (state == 0 && anybad) state = 2;
    if (state != 0) {  // frozen: the result is the state this iteration started from
      ctl[1] = q;
      ctl[2] = cur;
      __threadfence();
      ctl[0] = state;
    }
  }
}

// row pass: u_new[i] = a_i / sum_j K[i][j] v_new[j] (no per-block bookkeeping: 2048 blocks taking a ticket from one
// address cost more than the pass itself).
__global__ __launch_bounds__(256) void sk_rowpass_kernel(const float* __restrict__ K, int64_t ldk, int n0, int n1,
                                                         const double* __restrict__ a, const double* __restrict__ vbuf,
                                                         double* __restrict__ ubuf, int q, int* __restrict__ ctl) {
  if (ctl[0] != 0) return;
  const int cur = q & 1;
  const double* __restrict__ v = vbuf + (size_t)(cur ^ 1) * n1;  // the v just written
  double* __restrict__ un = ubuf + (size_t)(cur ^ 1) * n0;
  // A block takes 4 consecutive rows and walks them TOGETHER, its 256 lanes side by side along the row (4 KB of each
  // row per step, four independent loads in flight per lane): neighbouring blocks stream neighbouring rows, so the
  // chip reads a few thousand long contiguous runs instead of one short run per wave.
  __shared__ double rsum[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 4;
  int mybad = 0;
  {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const float* __restrict__ rows[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rows[q] = K + (int64_t)min(i0 + q, n0 - 1) * ldk;  // clamped rows are dropped below
    const bool vec = (n1 % 4 == 0) && (ldk % 4 == 0) && ((reinterpret_cast<uintptr_t>(K) & 15) == 0);
    if (vec) {
      for (int j = threadIdx.x * 4; j < n1; j += 1024) {
        float4 k4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) k4[q] = *reinterpret_cast<const float4*>(rows[q] + j);
        const double2 v01 = *reinterpret_cast<const double2*>(v + j);
        const double2 v23 = *reinterpret_cast<const double2*>(v + j + 2);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[q] += ((double)k4[q].x * v01.x + (double)k4[q].y * v01.y) + ((double)k4[q].z * v23.x + (double)k4[q].w * v23.y);
      }
    } else {
      for (int j = threadIdx.x; j < n1; j += 256) {
        const double vj = v[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += (double)rows[q][j] * vj;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double t = wave_sum_f64(acc[q]);
      if (lane == 0) rsum[q][wv] = t;
    }
    __syncthreads();
    if (threadIdx.x < 4 && i0 + (int)threadIdx.x < n0) {
      const int q = threadIdx.x;
      const double tot = (rsum[q][0] + rsum[q][1]) + (rsum[q][2] + rsum[q][3]);
      const double nu = a[i0 + q] / tot;
      un[i0 + q] = nu;
      if (tot == 0.0 || !isfinite(nu)) mybad = 1;
    }
  }
  if (mybad) atomicOr(&ctl[3], 1);  // rare; the next kernel that looks freezes the state one iteration back
}

// T[i][j] = u_i K[i][j] v_j with the current buffers; also copies u, v out
__global__ void sk_plan_kernel(const float* __restrict__ K, int64_t ldk, int n0, int n1,
                               const double* __restrict__ ubuf, const double* __restrict__ vbuf,
                               const int* __restrict__ ctl, int n_it, float* __restrict__ T, int64_t ldt,
                               double* __restrict__ u_out, double* __restrict__ v_out, int* __restrict__ info) {
  // not frozen: all n_it enqueued iterations ran (or the last row pass broke down: one back)
  const int state = ctl[0] != 0 ? ctl[0] : (ctl[3] != 0 ? 2 : 0);
  const int completed = ctl[0] != 0 ? ctl[1] : (ctl[3] != 0 ? n_it - 1 : n_it);
  const int cur = ctl[0] != 0 ? ctl[2] : (completed & 1);
  const double* __restrict__ u = ubuf + (size_t)cur * n0;
  const double* __restrict__ v = vbuf + (size_t)cur * n1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n0 * n1;
  if (idx < total) {
    const int i = (int)(idx / n1), j = (int)(idx - (int64_t)i * n1);
    T[(int64_t)i * ldt + j] = (float)(u[i] * (double)K[(int64_t)i * ldk + j] * v[j]);
  }
  if (u_out && idx < n0) u_out[idx] = u[idx];
  if (v_out && idx < n1) v_out[idx] = v[idx];
  if (idx == 0 && info) {
    info[0] = state;
    info[1] = completed;
  }
}

}  // namespace nif

extern "C" size_t nif_sinkhorn_workspace_bytes(int n0, int n1) {
  if (n0 <= 0 || n1 <= 0) return 0;
  const nif::SkWs w = nif::sk_ws(n0, n1);
  // + K (n0 x ld4(n1) f32), u and v double-buffered, ticket words
  const size_t ldk = ((size_t)n1 + 3) & ~(size_t)3;
  return w.total + (((size_t)n0 * ldk * sizeof(float) + 255) & ~(size_t)255) +
         (((size_t)2 * n0 * sizeof(double) + 255) & ~(size_t)255) + (((size_t)2 * n1 * sizeof(double) + 255) & ~(size_t)255) + 256;
}

extern "C" int nif_sinkhorn_f32(const float* C, int64_t ldc, int n0, int n1, const double* a, const double* b,
                                double reg, int max_iter, double tol, float* T, int64_t ldt, double* u_out,
                                double* v_out, int32_t* info, double* err, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_sinkhorn_f32");
  using namespace nif;
  if (!C || !a || !b || !T || !info || !err || n0 <= 0 || n1 <= 0 || ldc < n1 || ldt < n1 || !(reg > 0.0) ||
      max_iter < 0 || !(tol >= 0.0))
    return NIF_ERR_INVALID_ARG;
  if (!ws || ws_bytes < nif_sinkhorn_workspace_bytes(n0, n1)) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  const SkWs w = sk_ws(n0, n1);
  unsigned char* base = reinterpret_cast<unsigned char*>(ws);
  double* part = reinterpret_cast<double*>(base + w.off_part);
  int* ctl = reinterpret_cast<int*>(base + w.off_ctl);
  double* errp = reinterpret_cast<double*>(base + w.off_errp);
  size_t o = w.total;
  const int64_t ldk = ((int64_t)n1 + 3) & ~(int64_t)3;
  float* K = reinterpret_cast<float*>(base + o);
  o += ((size_t)n0 * ldk * sizeof(float) + 255) & ~(size_t)255;
  double* ubuf = reinterpret_cast<double*>(base + o);
  o += ((size_t)2 * n0 * sizeof(double) + 255) & ~(size_t)255;
  double* vbuf = reinterpret_cast<double*>(base + o);
  o += ((size_t)2 * n1 * sizeof(double) + 255) & ~(size_t)255;
  unsigned* ticket = reinterpret_cast<unsigned*>(base + o);
  if (hipMemsetAsync(ticket, 0, 256, st) != hipSuccess) return NIF_ERR_LAUNCH;
  const int64_t total = (int64_t)n0 * n1;
  const int64_t cover = total > n0 ? (total > n1 ? total : n1) : (n0 > n1 ? n0 : n1);
  const unsigned eb = (unsigned)((cover + 255) / 256);
  hipLaunchKernelGGL(sk_init_kernel, dim3(eb), dim3(256), 0, st, C, ldc, n0, n1, reg, K, ldk, ubuf, vbuf, ctl, err);
  const dim3 cgrid((n1 + SK_COLS - 1) / SK_COLS, w.chunks);
  const unsigned rgrid = (unsigned)((n0 + 3) / 4);
  // iterations are enqueued in batches; between batches the host reads the state word (the only synchronisations)
  const int batch = 50;
  int done = 0, n_it = 0;
  prof_begin(NIF_PROF_SINKHORN, st);  // one span per call
  while (n_it < max_iter && !done) {
    const int n = (max_iter - n_it) < batch ? (max_iter - n_it) : batch;
    for (int e = 0; e < n; ++e, ++n_it) {
      hipLaunchKernelGGL(sk_colpass_kernel, cgrid, dim3(256), 0, st, K, ldk, n0, n1, ubuf, w.rows_per_chunk, n_it, ctl, part);
      hipLaunchKernelGGL(sk_colfinish_kernel, dim3((n1 + 63) / 64), dim3(256), 0, st, part, w.chunks, n1, b, vbuf, 10, tol,
                         n_it, ctl, errp, err, ticket);
      hipLaunchKernelGGL(sk_rowpass_kernel, dim3(rgrid), dim3(256), 0, st, K, ldk, n0, n1, a, vbuf, ubuf, n_it, ctl);
    }
    if (n_it < max_iter) {
      int state[4] = {0, 0, 0, 0};
      if (hipMemcpyAsync(state, ctl, sizeof(state), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) {
        prof_end(NIF_PROF_SINKHORN, st);  // the span is closed on the error path too
        return NIF_ERR_LAUNCH;
      }
      done = state[0] != 0 || state[3] != 0;
    }
  }
  prof_end(NIF_PROF_SINKHORN, st);
  hipLaunchKernelGGL(sk_plan_kernel, dim3((unsigned)((cover + 255) / 256)), dim3(256), 0, st, K, ldk, n0, n1, ubuf, vbuf,
                     ctl, n_it, T, ldt, u_out, v_out, info);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
