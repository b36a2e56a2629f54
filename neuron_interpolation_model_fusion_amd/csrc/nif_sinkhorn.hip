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

// ---- small problems: every iteration inside ONE workgroup ----------------------------------------------------------
// Three launches per iteration are 13-15 us whatever the size, and below n ~ 1024 that is all an iteration costs.  Up to
// n ~ 192 one workgroup of 1024 lanes is quicker than that and runs the whole loop (measured, tools/sinkhorn_bench.py,
// us per iteration, one workgroup / three launches: n = 64 5.7 / 13.1, 128 7.6 / 13.3, 192 10.6 / 13.2; from n = 256 on
// one CU's loads are the slower side -- 13.5 / 13.2, 384: 25 / 14.5 -- and the launches stay): column pass (a lane owns four columns and every nsl-th row, partial sums combined
// in slice order), error test / v = b / s, row pass (a wave owns blocks of four rows), u = a / t, with u and v in LDS and
// the same test schedule, freeze and breakdown rules as the three kernels above; no host look between batches.  The
// final state goes where sk_plan_kernel expects it.
constexpr int SK_SMALL_MAX_ELEMS = 192 * 192;

__global__ __launch_bounds__(1024) void sk_small_kernel(const float* __restrict__ K, int64_t ldk, int n0, int n1,
                                                        const double* __restrict__ a, const double* __restrict__ b,
                                                        int max_iter, int check_every, double tol,
                                                        double* __restrict__ ubuf, double* __restrict__ vbuf,
                                                        int* __restrict__ ctl, double* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_sm[];
  const int Q = (n1 + 3) / 4;          // column quads
  const int nsl = 1024 / Q;            // row slices of the column pass (>= 4: n1 <= 1024)
  const int n1p = 4 * Q;
  double* su = reinterpret_cast<double*>(sk_sm);  // [2][n0]
  double* sv = su + 2 * n0;                        // [2][n1p]
  double* part = sv + 2 * n1p;                     // [nsl][n1p]
  double* red = part + (size_t)nsl * n1p;          // [16] + [1]
  int* flags = reinterpret_cast<int*>(red + 20);   // 0: bad column sum, 1: bad row sum
  double* sa = red + 22;                           // [n0] the marginals, read every iteration
  double* sb = sa + n0;                            // [n1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int i = tid; i < n0; i += 1024) {
    su[i] = ubuf[i];
    sa[i] = a[i];
  }
  for (int j = tid; j < n1p; j += 1024) sv[j] = j < n1 ? vbuf[j] : 0.0;
  for (int j = tid; j < n1; j += 1024) sb[j] = b[j];
  if (tid < 2) flags[tid] = 0;
  __syncthreads();

  const int cq = tid % Q, slice = tid / Q;
  int state = 0, completed = max_iter, cur = 0;
  double last_err = INFINITY;
  for (int q = 0; q < max_iter; ++q) {
    cur = q & 1;
    const double* __restrict__ u = su + (size_t)cur * n0;
    const double* __restrict__ v = sv + (size_t)cur * n1p;
    double* __restrict__ un = su + (size_t)(cur ^ 1) * n0;
    double* __restrict__ vn = sv + (size_t)(cur ^ 1) * n1p;
    // column pass
    if (slice < nsl) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      const float* __restrict__ col = K + 4 * cq;
      int i = slice;
      for (; i + 7 * nsl < n0; i += 8 * nsl) {  // eight rows in flight
        float4 k4[8];
        double ui[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          k4[r] = *reinterpret_cast<const float4*>(col + (int64_t)(i + r * nsl) * ldk);
          ui[r] = u[i + r * nsl];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          acc[0] += ui[r] * (double)k4[r].x;
          acc[1] += ui[r] * (double)k4[r].y;
          acc[2] += ui[r] * (double)k4[r].z;
          acc[3] += ui[r] * (double)k4[r].w;
        }
      }
      for (; i < n0; i += nsl) {
        const float4 k4 = *reinterpret_cast<const float4*>(col + (int64_t)i * ldk);
        const double ui = u[i];
        acc[0] += ui * (double)k4.x;
        acc[1] += ui * (double)k4.y;
        acc[2] += ui * (double)k4.z;
        acc[3] += ui * (double)k4.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) part[(size_t)slice * n1p + 4 * cq + e] = acc[e];
    }
    __syncthreads();
    // POT tests after the iterations with ii % 10 == 0, i.e. after 1, 11, 21, ... completed iterations
    const bool check = q > 0 && ((q - 1) % check_every == 0);
    double e2 = 0.0;
    for (int j = tid; j < n1; j += 1024) {
      double sj = 0.0;
      for (int sl = 0; sl < nsl; ++sl) sj += part[(size_t)sl * n1p + j];
      if (check) {
        const double d = v[j] * sj - sb[j];
        e2 += d * d;
      }
      const double nv = sb[j] / sj;
      if (sj == 0.0 || !isfinite(nv)) flags[0] = 1;
      vn[j] = nv;
    }
    if (check) {  // uniform
      e2 = wave_sum_f64(e2);
      if (lane == 0) red[wave] = e2;
    }
    __syncthreads();
    if (check) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < 16; ++w) tot += red[w];
      last_err = sqrt(tot);
      if (last_err < tol) state = 1;
    }
    if (state == 0 && flags[0]) state = 2;
    if (state != 0) {  // frozen: the result is the state this iteration started from
      completed = q;
      break;
    }
    // row pass: blocks of four rows per wave
    for (int g = wave; 4 * g < n0; g += 16) {
      const int i0 = 4 * g;
      const float* __restrict__ rows[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rows[r] = K + (int64_t)min(i0 + r, n0 - 1) * ldk;  // clamped rows are dropped below
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int j4 = lane; j4 < Q; j4 += 64) {
        float4 k4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) k4[r] = *reinterpret_cast<const float4*>(rows[r] + 4 * j4);
        const double2 v01 = *reinterpret_cast<const double2*>(vn + 4 * j4);
        const double2 v23 = *reinterpret_cast<const double2*>(vn + 4 * j4 + 2);
        // (columns beyond n1 inside the last quad: K's pad may hold anything, so they are masked, not multiplied by 0)
        const bool f1 = 4 * j4 + 1 < n1, f2 = 4 * j4 + 2 < n1, f3 = 4 * j4 + 3 < n1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[r] += ((double)k4[r].x * v01.x + (f1 ? (double)k4[r].y * v01.y : 0.0)) +
                    ((f2 ? (double)k4[r].z * v23.x : 0.0) + (f3 ? (double)k4[r].w * v23.y : 0.0));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double t = wave_sum_f64(acc[r]);
        if (lane == 0 && i0 + r < n0) {
          const double nu = sa[i0 + r] / t;
          un[i0 + r] = nu;
          if (t == 0.0 || !isfinite(nu)) flags[1] = 1;
        }
      }
    }
    __syncthreads();
    if (flags[1]) {  // the row pass broke down: its inputs are the result (POT keeps uprev, vprev)
      state = 2;
      completed = q;
      break;
    }
  }
  if (state == 0) cur = max_iter & 1;  // all iterations ran: the buffers written last
  __syncthreads();
  for (int i = tid; i < n0; i += 1024) ubuf[(size_t)cur * n0 + i] = su[(size_t)cur * n0 + i];
  for (int j = tid; j < n1; j += 1024) vbuf[(size_t)cur * n1 + j] = sv[(size_t)cur * n1p + j];
  if (tid == 0) {
    ctl[0] = state;
    ctl[1] = completed;
    ctl[2] = cur;
    ctl[3] = 0;
    if (last_err < INFINITY) *err = last_err;
  }
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
  // small problems: the whole loop in one workgroup (NIF_SINKHORN_LAUNCHES=1 keeps the three launches per iteration)
  static const int sk_launches = env_int("NIF_SINKHORN_LAUNCHES");
  if (!sk_launches && n1 <= 1024 && n0 <= 4096 && (int64_t)n0 * n1 <= SK_SMALL_MAX_ELEMS && max_iter > 0) {
    const int Qh = (n1 + 3) / 4, nslh = 1024 / Qh, n1ph = 4 * Qh;
    const size_t lds = ((size_t)3 * n0 + 2 * n1ph + (size_t)nslh * n1ph + 22 + n1) * sizeof(double);
    static PerDeviceOnce attr_once;
    attr_once.run([&] {
      (void)hipFuncSetAttribute((const void*)sk_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    if (lds <= 128 * 1024) {
      prof_begin(NIF_PROF_SINKHORN, st);
      hipLaunchKernelGGL(sk_small_kernel, dim3(1), dim3(1024), lds, st, K, ldk, n0, n1, a, b, max_iter, 10, tol, ubuf, vbuf,
                         ctl, err);
      prof_end(NIF_PROF_SINKHORN, st);
      hipLaunchKernelGGL(sk_plan_kernel, dim3((unsigned)((cover + 255) / 256)), dim3(256), 0, st, K, ldk, n0, n1, ubuf,
                         vbuf, ctl, max_iter, T, ldt, u_out, v_out, info);
      return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
    }
  }
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
