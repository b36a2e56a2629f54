// nif_eigh.hip -- symmetric eigendecomposition with a relative cutoff, hand-written for gfx950: the factor step of
// the least-squares refit (torch.linalg.pinv at hf_kf.py:213,377 drops singular values <= rtol * sigma_max, so the
// normal equations need the eigenpairs above rtol^2 * lambda_max, not a Cholesky factor).
//
//   1. Householder tridiagonalisation, one launch per column: the rank-2 update of step j-1 and the matrix-vector
//      product of step j are fused into ONE pass over the trailing matrix (every element is read once and written
//      once per column), row blocks spread over the whole chip; the O(n) vector work of a step (reflector, w) is
//      recomputed by every workgroup from L2-resident vectors, so a step needs no grid-wide synchronisation beyond
//      the launch boundary.
//   2. Eigenvalues of the tridiagonal matrix by Sturm-count multisection: one wavefront per eigenvalue, its 64 lanes
//      evaluate the count at 64 points of the bracket per pass (9 passes instead of 53 bisection steps).
//   3. Eigenvectors of the KEPT eigenvalues by inverse iteration (LU with partial pivoting of T - lambda I, one lane
//      per eigenvector, lanes in lock step over the rows, factors stored [row][vector] for coalescing), distinct
//      pseudo-random starts, modified Gram-Schmidt inside groups of eigenvalues closer than 1e-9 |T|.
//   4. Back-transformation by blocks of 64 reflectors: (I - V T V^T) Z = Z - (V T) (V^T Z) with
//      T = (diag(1/tau) + striu(V^T V))^-1.  V, T and V T of every block are prepared by four batched launches; the
//      chain over the blocks is then one product with the long contraction on the f64 matrix cores (nif_xty) and
//      one rank-64 update per block.
//
// Everything is f64.  The entry point synchronises the stream ONCE (after step 2) to learn how many eigenpairs are
// kept: that count is an output and sizes the launches of steps 3-4.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

constexpr int TRD_T = 512;      // threads of a tridiagonalisation workgroup (8 waves)
constexpr int TRD_ROWS = 4;     // rows a wave carries through the trailing pass together
constexpr int EIG_NB = 64;      // reflectors per back-transformation block

// deterministic block-wide sum (same value in every workgroup that runs the same code on the same data)
template <int T, bool FRESH = false>
__device__ __forceinline__ double block_sum(double x, double* red /* T/64 + 1 doubles of LDS */) {
  x = wave_sum_f64(x);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (!FRESH) __syncthreads();  // red may still be read from a previous call (FRESH: slots nobody has touched)
  if (lane == 0) red[wv] = x;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) s += red[w];
  return s;
}

struct TrdArgs {
  double* A;       // n x ld, symmetric, both triangles kept current; ld even, pad column zero
  int64_t ld;
  int n;
  int j;           // column being eliminated
  double* vbuf;    // 3 x ld: reflector of step j in slot j % 3 (rows j+1 .. n-1, v[j+1] = 1)
  double* wbuf;    // 2 x ld: w_{j-1} saved by a step that does not store the matrix (MODE 1), slot (j - 1) & 1
  double* ybuf;    // 2 x ld: y_j = (updated trailing matrix) v_j
  double* d;       // diagonal of T
  double* e;       // sub-diagonal of T
  double* tau;     // reflector scales
};

// One column of the reduction.  LDS: sv1 = v_{j-1}, sw = w_{j-1}, svj = v_j, each ld doubles, indexed by column.
// A step is latency-bound below n ~ 2500 (launch gap + a chain of dependent L2 round trips), so: every global load
// the vector phase needs is issued at the top; the two block reductions are the only barriers before the trailing
// pass; a workgroup takes one group of TRD_ROWS rows at a time and its 8 waves split the COLUMNS of the group
// (chunk t of 128 columns goes to wave t % 8), so all of a group's data is in flight at once -- CPW chunks per wave
// -- and the first group's loads are issued before the vector phase starts.
constexpr int TRD_EPT_MAX = (NIF_EIGH_MAX_N + TRD_T - 1) / TRD_T;  // vector elements a thread owns at most
constexpr int TRD_W = TRD_T / 64;

// TWO: a second row group per workgroup is fetched before the vector phase as well (its own register set): for
// trailing sizes where a workgroup owns about two groups, all of its share of the matrix is then in flight while the
// latency-bound vector phase runs, instead of the second half being requested after it.
//
// MODE: the stores of the trailing pass -- and the write-back of 8 m^2 dirty bytes that the launch boundary waits
// for -- are 18-40 % of a step above a trailing size of 1024.  So every other step leaves the matrix alone:
//   MODE 1  one update pending (v_{j-1}, w_{j-1}): applied on the fly for y = A v_j, NOT stored; w_{j-1} is saved;
//   MODE 2  two updates pending (j-2 from the saved vectors, j-1 as always): both applied, the result stored;
//   MODE 0  the classic step (one pending, stored): first and last column, and wherever MODE 2 does not fit.
// The older pair's vectors never enter LDS: a lane's columns are the same in every row group, so their entries sit
// in registers for the whole step, and the four row entries of a group are fetched with the group.
template <int CPW, bool TWO, int MODE>
__global__ __launch_bounds__(TRD_T) void trd_step_kernel(TrdArgs a) {
  constexpr bool KP2 = MODE == 2, STORE = MODE != 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = a.n, j = a.j, tid = threadIdx.x;
  const int64_t ld = a.ld;
  double* sv1 = reinterpret_cast<double*>(smem_raw);
  double* sw = sv1 + ld;
  double* svj = sw + ld;
  double* red = svj + ld;            // 8 reduction slots + 2 broadcast slots (+ 6 spare), 8 more for the second sum
  double* rowpart = red + 32;        // [2][TRD_W][TRD_ROWS] partial row sums, double-buffered by group parity
  const double* vprev = a.vbuf + (size_t)((j + 2) % 3) * ld;  // slot of j - 1
  const double* yprev = a.ybuf + (size_t)((j + 1) & 1) * ld;
  double* vcur = a.vbuf + (size_t)(j % 3) * ld;
  double* ycur = a.ybuf + (size_t)(j & 1) * ld;
  const double* v2g = a.vbuf + (size_t)((j + 1) % 3) * ld;   // KP2: v_{j-2}
  const double* w2g = a.wbuf + (size_t)(j & 1) * ld;         // KP2: w_{j-2}, saved by step j - 1
  double* wsave = a.wbuf + (size_t)((j + 1) & 1) * ld;       // MODE 1: w_{j-1} goes here
  const int c0 = (j + 1) & ~1;               // first column of the 16-byte aligned trailing pass (j or j + 1)
  const int cend = (n + 1) & ~1;             // one pad column when n is odd
  const bool pending = j > 0;
  const int lane = tid & 63, wv = tid >> 6;
  // vector elements a thread owns: the rows j .. n-1 are at most 1024 CPW + 1 (the host picks CPW from the trailing
  // width), so the loops below are CPW-sized instead of sized for the largest matrix
  constexpr int TRD_EPT = 2 * CPW + 1 < TRD_EPT_MAX ? 2 * CPW + 1 : TRD_EPT_MAX;

  // ---- loads of the vector phase first: they head the step's chain of dependent latencies (loads -> reduction ->
  // reflector -> pass), the matrix rows requested right after them are not needed before the pass
  double tau1 = 0.0, pj = 0.0, vjm = 0.0;
  if (pending) {
    tau1 = a.tau[j - 1];
    pj = yprev[j];
    vjm = vprev[j];  // = 1
  }
  double vv[TRD_EPT], pp[TRD_EPT], xx[TRD_EPT];
  double v2r[KP2 ? TRD_EPT : 1], w2r[KP2 ? TRD_EPT : 1];
  double v2j = 0.0, w2j = 0.0;
  if constexpr (KP2) { v2j = v2g[j]; w2j = w2g[j]; }
  const double* arow = a.A + (size_t)j * ld;
#pragma unroll
  for (int it = 0; it < TRD_EPT; ++it) {
    const int r = j + tid + it * TRD_T;
    vv[it] = 0.0; pp[it] = 0.0; xx[it] = 0.0;
    if constexpr (KP2) { v2r[it] = 0.0; w2r[it] = 0.0; }
    if (r < n) {
      xx[it] = arow[r];
      if (pending) { vv[it] = vprev[r]; pp[it] = yprev[r]; }
      if constexpr (KP2) { v2r[it] = v2g[r]; w2r[it] = w2g[r]; }
    }
  }

  // ---- this workgroup's share of the trailing pass: row groups of TRD_ROWS consecutive rows (absolute numbering);
  // group g always goes to a workgroup with blockIdx % 8 == g % 8, i.e. (workgroups are dealt round-robin over the
  // XCDs) to the same XCD at every step, so an XCD's share of the matrix is re-read from its own L2 once it fits.
  const int xcd = blockIdx.x & 7, sub = blockIdx.x >> 3, nsub = gridDim.x >> 3;
  const int gfirst = (j + 1) / TRD_ROWS;
  int g = gfirst + ((xcd - (gfirst & 7) + 8) & 7) + 8 * sub;
  const int cw = c0 + 2 * lane + 128 * wv;   // column of this lane in its wave's first chunk; next chunks + 1024
  double* row[TRD_ROWS];
  bool live[TRD_ROWS];
  int rr[TRD_ROWS];
  double2 x[CPW][TRD_ROWS];
  // the group after that (TWO only)
  double* rowB[TRD_ROWS];
  bool liveB[TRD_ROWS];
  int rrB[TRD_ROWS];
  double2 xB[TWO ? CPW : 1][TRD_ROWS];
  // KP2: the older pair at this lane's columns (fixed for the step) and at the rows of the group(s) in flight
  double2 c2v[KP2 ? CPW : 1], c2w[KP2 ? CPW : 1];
  double r2v[KP2 ? TRD_ROWS : 1], r2w[KP2 ? TRD_ROWS : 1], r2vB[KP2 && TWO ? TRD_ROWS : 1], r2wB[KP2 && TWO ? TRD_ROWS : 1];
  if constexpr (KP2) {
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int c = cw + 1024 * k;
      const int cc = c < cend ? c : cend - 2;
      c2v[k] = *reinterpret_cast<const double2*>(v2g + cc);
      c2w[k] = *reinterpret_cast<const double2*>(w2g + cc);
    }
  }
  auto fetch2 = [&](const int (&rr_)[TRD_ROWS], double* rv_, double* rw_) {
    if constexpr (KP2) {
#pragma unroll
      for (int q = 0; q < TRD_ROWS; ++q) { rv_[q] = v2g[rr_[q]]; rw_[q] = w2g[rr_[q]]; }
    }
  };
  auto setup = [&](int grp, double* (&row_)[TRD_ROWS], bool (&live_)[TRD_ROWS], int (&rr_)[TRD_ROWS]) {
#pragma unroll
    for (int q = 0; q < TRD_ROWS; ++q) {
      int r = grp * TRD_ROWS + q;
      live_[q] = r >= j + 1 && r < n;
      r = r < j + 1 ? j + 1 : (r < n ? r : n - 1);  // clamped duplicates are loaded and discarded
      rr_[q] = r;
      row_[q] = a.A + (size_t)r * ld;
    }
  };
  auto fetch = [&](double* const (&row_)[TRD_ROWS], double2 (*x_)[TRD_ROWS]) {
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int c = cw + 1024 * k;
      if (c < cend) {
#pragma unroll
        for (int q = 0; q < TRD_ROWS; ++q) x_[k][q] = *reinterpret_cast<const double2*>(row_[q] + c);
      }
    }
  };
  bool has = g * TRD_ROWS < n;
  if (has) {
    setup(g, row, live, rr);
    fetch(row, x);
    fetch2(rr, r2v, r2w);
  }
  int gB = g + 8 * nsub;
  bool hasB = false;
  if constexpr (TWO) {
    hasB = has && gB * TRD_ROWS < n;
    if (hasB) {
      setup(gB, rowB, liveB, rrB);
      fetch(rowB, xB);
      fetch2(rrB, r2vB, r2wB);
    }
  }

  // ---- vector phase, identical in every workgroup (its loads were issued at the top)
  pj *= tau1;
  double part = 0.0;
#pragma unroll
  for (int it = 0; it < TRD_EPT; ++it) {
    pp[it] *= tau1;
    part += pp[it] * vv[it];
  }
  // (a) w_{j-1} = p + alpha v with p = tau y, alpha = -tau/2 (p . v): rows j .. n-1 (v_{j-1}[j] = 1)
  const double dot = pending ? block_sum<TRD_T, true>(part, red) : 0.0;
  const double alpha = -0.5 * tau1 * dot;
  const double wjm = pj + alpha * vjm;
  // (b) row j of the updated matrix (= its column j): x_c = A[j][c] - v[j] w[c] - w[j] v[c]
  double part2 = 0.0;
#pragma unroll
  for (int it = 0; it < TRD_EPT; ++it) {
    const int r = j + tid + it * TRD_T;
    if (r < n) {
      const double w = pp[it] + alpha * vv[it];
      double x0 = xx[it];
      if constexpr (KP2) x0 -= v2j * w2r[it] + w2j * v2r[it];  // the older pending update, as a stored step applies it
      const double xr = x0 - vjm * w - wjm * vv[it];
      xx[it] = xr;
      if constexpr (MODE == 1) { if (blockIdx.x == 0) wsave[r] = w; }
      // column j is finished: it is masked out of the trailing pass (which starts at column j when j + 1 is odd)
      sv1[r] = r == j ? 0.0 : vv[it];
      sw[r] = r == j ? 0.0 : w;
      if (r >= j + 2) part2 += xr * xr;
      if (r == j) red[8] = xr;
      if (r == j + 1) red[9] = xr;
    }
  }
  if (tid == 0 && cend > n) { sv1[n] = 0.0; sw[n] = 0.0; svj[n] = 0.0; }
  const double xnorm2 = block_sum<TRD_T, true>(part2, red + 16);
  const double djj = red[8], alpha0 = red[9];
  double tau, beta, scale;
  if (xnorm2 == 0.0) {
    tau = 0.0; beta = alpha0; scale = 0.0;
  } else {
    beta = -copysign(sqrt(alpha0 * alpha0 + xnorm2), alpha0);
    tau = (beta - alpha0) / beta;
    scale = 1.0 / (alpha0 - beta);
  }
#pragma unroll
  for (int it = 0; it < TRD_EPT; ++it) {
    const int r = j + tid + it * TRD_T;
    if (r < n) {
      const double v = r >= j + 2 ? xx[it] * scale : (r == j + 1 ? 1.0 : 0.0);
      svj[r] = v;
      if (blockIdx.x == 0 && r >= j + 1) vcur[r] = v;
    }
  }
  if (blockIdx.x == 0 && tid == 0) { a.d[j] = djj; a.e[j] = beta; a.tau[j] = tau; }
  __syncthreads();

  // ---- (c) trailing pass: apply the pending rank-2 update, store, and accumulate y = A v_j
  int parity = 0;
  while (has) {   // uniform over the workgroup
    double vr[TRD_ROWS], wr[TRD_ROWS], acc[TRD_ROWS];
#pragma unroll
    for (int q = 0; q < TRD_ROWS; ++q) {
      vr[q] = sv1[rr[q]];
      wr[q] = sw[rr[q]];
      acc[q] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int c = cw + 1024 * k;
      if (c < cend) {
        const double2 w2 = *reinterpret_cast<const double2*>(sw + c);
        const double2 v2 = *reinterpret_cast<const double2*>(sv1 + c);
        const double2 u2 = *reinterpret_cast<const double2*>(svj + c);
#pragma unroll
        for (int q = 0; q < TRD_ROWS; ++q) {
          double2 y = x[k][q];
          if constexpr (KP2) {
            y.x -= r2v[q] * c2w[k].x + r2w[q] * c2v[k].x;
            y.y -= r2v[q] * c2w[k].y + r2w[q] * c2v[k].y;
          }
          y.x -= vr[q] * w2.x + wr[q] * v2.x;
          y.y -= vr[q] * w2.y + wr[q] * v2.y;
          acc[q] += y.x * u2.x + y.y * u2.y;
          if (STORE && pending && live[q]) *reinterpret_cast<double2*>(row[q] + c) = y;
        }
      }
    }
    // the finished group's identity, then the next group's loads go out before the reductions
    int frr[TRD_ROWS];
    bool flive[TRD_ROWS];
    double* frow[TRD_ROWS];
#pragma unroll
    for (int q = 0; q < TRD_ROWS; ++q) { frr[q] = rr[q]; flive[q] = live[q]; frow[q] = row[q]; }
    if constexpr (TWO) {
      // the group fetched earlier moves up, the one after it is requested
      has = hasB;
      if (has) {
#pragma unroll
        for (int q = 0; q < TRD_ROWS; ++q) { row[q] = rowB[q]; live[q] = liveB[q]; rr[q] = rrB[q]; }
        if constexpr (KP2 && TWO) {
#pragma unroll
          for (int q = 0; q < TRD_ROWS; ++q) { r2v[q] = r2vB[q]; r2w[q] = r2wB[q]; }
        }
#pragma unroll
        for (int k = 0; k < CPW; ++k)
#pragma unroll
          for (int q = 0; q < TRD_ROWS; ++q) x[k][q] = xB[TWO ? k : 0][q];
        gB += 8 * nsub;
        hasB = gB * TRD_ROWS < n;
        if (hasB) {
          setup(gB, rowB, liveB, rrB);
          fetch(rowB, xB);
          fetch2(rrB, r2vB, r2wB);
        }
      }
    } else {
      g += 8 * nsub;
      has = g * TRD_ROWS < n;
      if (has) {
        setup(g, row, live, rr);
        fetch(row, x);
        fetch2(rr, r2v, r2w);
      }
    }
    double* part_ = rowpart + parity * (TRD_W * TRD_ROWS);
#pragma unroll
    for (int q = 0; q < TRD_ROWS; ++q) {
      const double sq = wave_sum_f64(acc[q]);
      if (lane == 0) part_[wv * TRD_ROWS + q] = sq;
    }
    __syncthreads();
    if (tid < TRD_ROWS) {
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < TRD_W; ++w) sum += part_[w * TRD_ROWS + tid];
      // select this thread's row without dynamic register indexing
      int r = frr[0]; bool lv = flive[0]; double* rp = frow[0];
#pragma unroll
      for (int q = 1; q < TRD_ROWS; ++q) if (tid == q) { r = frr[q]; lv = flive[q]; rp = frow[q]; }
      if (lv) {
        ycur[r] = sum;
        if (r >= j + 2) rp[j] = svj[r];  // reflector j, stored below the sub-diagonal of column j
      }
    }
    parity ^= 1;
  }
}

// After the last step (j = n - 3): apply the pending update to the trailing 2 x 2 block.  n >= 3.
__global__ void trd_finish_kernel(TrdArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = a.n, jl = n - 3;
  const int64_t ld = a.ld;
  const double* v = a.vbuf + (size_t)(jl % 3) * ld;
  const double* y = a.ybuf + (size_t)(jl & 1) * ld;
  const double tau1 = a.tau[jl];
  const int r0 = n - 2, r1 = n - 1;
  const double p0 = tau1 * y[r0], p1 = tau1 * y[r1];
  const double alpha = -0.5 * tau1 * (p0 * v[r0] + p1 * v[r1]);
  const double w0 = p0 + alpha * v[r0], w1 = p1 + alpha * v[r1];
  a.d[r0] = a.A[(size_t)r0 * ld + r0] - 2.0 * v[r0] * w0;
  a.e[r0] = a.A[(size_t)r1 * ld + r0] - v[r1] * w0 - w1 * v[r0];
  a.d[r1] = a.A[(size_t)r1 * ld + r1] - 2.0 * v[r1] * w1;
  a.tau[r0] = 0.0;
}

// n <= 2: nothing to reduce
__global__ void trd_small_kernel(TrdArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  a.d[0] = a.A[0];
  if (a.n == 2) {
    a.e[0] = a.A[a.ld];
    a.d[1] = a.A[a.ld + 1];
    a.tau[0] = 0.0;
  }
}

// ---- eigenvalues of the tridiagonal matrix ---------------------------------------------------------------------
// params[0] = tnorm (max row sum of |T|), [1] = pivmin, [2] = lower Gershgorin bound, [3] = upper bound
__global__ void tri_bounds_kernel(const double* __restrict__ d, const double* __restrict__ e, int n, double* params) {
  const int tid = threadIdx.x;
  double tn = 0.0, lo = INFINITY, hi = -INFINITY, e2m = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double el = i > 0 ? fabs(e[i - 1]) : 0.0, er = i + 1 < n ? fabs(e[i]) : 0.0;
    tn = fmax(tn, fabs(d[i]) + el + er);
    lo = fmin(lo, d[i] - el - er);
    hi = fmax(hi, d[i] + el + er);
    e2m = fmax(e2m, er * er);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    tn = fmax(tn, shfl_xor_f64(tn, o));
    lo = fmin(lo, shfl_xor_f64(lo, o));
    hi = fmax(hi, shfl_xor_f64(hi, o));
    e2m = fmax(e2m, shfl_xor_f64(e2m, o));
  }
  __shared__ double r_tn[4], r_lo[4], r_hi[4], r_e2[4];
  if ((tid & 63) == 0) { r_tn[tid >> 6] = tn; r_lo[tid >> 6] = lo; r_hi[tid >> 6] = hi; r_e2[tid >> 6] = e2m; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      tn = fmax(tn, r_tn[w]); lo = fmin(lo, r_lo[w]); hi = fmax(hi, r_hi[w]); e2m = fmax(e2m, r_e2[w]);
    }
    if (!(tn > 0.0)) tn = DBL_MIN;
    const double pivmin = DBL_MIN * fmax(1.0, e2m);
    const double slack = 2.0 * tn * DBL_EPSILON * n + 2.0 * pivmin;
    params[0] = tn;
    params[1] = pivmin;
    params[2] = lo - slack;
    params[3] = hi + slack;
  }
}

constexpr int BIS_T = 512;   // threads of a workgroup: its waves share one LDS copy of T
constexpr int BIS_P = 16;    // points per eigenvalue and pass: a wave carries 64 / BIS_P eigenvalues

// lam[k] = k-th smallest eigenvalue.  A group of BIS_P lanes brackets one eigenvalue: every pass evaluates the Sturm
// count at BIS_P interior points and keeps the sub-interval that contains it (x BIS_P + 1 per pass, ~13 passes).
// Sixteen points per eigenvalue rather than 64: the kernel is bound by the f64 divide chains in flight per SIMD, and
// the narrower fan does a third of the total work for a third more passes.
__global__ __launch_bounds__(BIS_T) void tri_eigenvalues_kernel(const double* __restrict__ d, const double* __restrict__ e,
                                                                int n, const double* __restrict__ params,
                                                                double* __restrict__ lam) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sd = reinterpret_cast<double*>(smem_raw);
  double* se2 = sd + n;
  for (int i = threadIdx.x; i < n; i += BIS_T) {
    sd[i] = d[i];
    const double x = i + 1 < n ? e[i] : 0.0;
    se2[i] = x * x;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, grp = lane / BIS_P, pt = lane % BIS_P;
  const int k = (blockIdx.x * (BIS_T / 64) + (threadIdx.x >> 6)) * (64 / BIS_P) + grp;
  const double tnorm = params[0], pivmin = params[1];
  double lo = params[2], hi = params[3];
  const double tol = DBL_EPSILON * tnorm;
  const bool mine = k < n;
  for (int pass = 0; pass < 20; ++pass) {
    const bool active = mine && hi - lo > tol;
    if (!__any(active)) break;
    const double x = lo + (hi - lo) * ((double)(pt + 1) / (double)(BIS_P + 1));
    double q = sd[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    int cnt = q < 0.0;
    for (int i = 1; i < n; ++i) {
      // e^2 / q through the hardware reciprocal (about 24 bits) and two Newton steps: the full IEEE divide expansion
      // is what bounds this kernel, and |q| >= pivmin here so no special cases arise
      double r = __builtin_amdgcn_rcp(q);
      r = fma(fma(-q, r, 1.0), r, r);
      r = fma(fma(-q, r, 1.0), r, r);
      q = sd[i] - x - se2[i - 1] * r;
      if (fabs(q) < pivmin) q = -pivmin;
      cnt += q < 0.0;
    }
    // lambda_k < x  <=>  at least k + 1 eigenvalues lie below x
    const unsigned long long below = __ballot(cnt > k);
    const unsigned field = (unsigned)(below >> (grp * BIS_P)) & ((1u << BIS_P) - 1u);
    const int first = field ? __ffs((int)field) - 1 : BIS_P;
    if (active) {
      const double xl = lo, w = hi - lo;
      if (first < BIS_P) hi = xl + w * ((double)(first + 1) / (double)(BIS_P + 1));
      if (first > 0) lo = xl + w * ((double)first / (double)(BIS_P + 1));
    }
  }
  if (mine && pt == 0) lam[k] = 0.5 * (lo + hi);
}

// kept eigenvalues = those above cutoff_rel * lambda_max; shifts for the inverse iteration: the rule
// s_j = max(lam_j, s_{j-1} + pert), pert = 10 eps |T| (coincident eigenvalues get distinct factorizations), which
// unrolls to s_j = max_t (lam_{j-t} + t pert): every thread scans back over the eigenvalues within n * pert of its own
// (a handful at most unless the spectrum is numerically multiple).  kept[0] = K.
__global__ void tri_select_kernel(const double* __restrict__ lam, int n, double cutoff_rel, const double* params,
                                  double* __restrict__ shifts, int* __restrict__ kept) {
  const double lmax = lam[n - 1];
  const double thr = cutoff_rel * lmax;
  int lo = 0, hi = n;  // first index with lam > thr (lam ascending)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (lam[mid] > thr) hi = mid; else lo = mid + 1;
  }
  const int K = n - lo;
  if (blockIdx.x == 0 && threadIdx.x == 0) kept[0] = K;
  const double pert = 10.0 * DBL_EPSILON * params[0];
  const double reach = (double)n * pert;
  for (int jj = blockIdx.x * blockDim.x + threadIdx.x; jj < K; jj += gridDim.x * blockDim.x) {
    const double mine = lam[lo + jj];
    double best = mine;
    for (int t = 1; t <= jj; ++t) {
      const double other = lam[lo + jj - t];
      if (mine - other > reach) break;
      const double cand = other + (double)t * pert;
      if (cand > best) best = cand;
    }
    shifts[jj] = best;
  }
}

__device__ __forceinline__ double hash_unit(unsigned i, unsigned jj) {
  unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)jj * 0xC2B2AE3D27D4EB4Full + 0x165667B19E3779F9ull);
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
  return (double)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

struct InvitArgs {
  const double* d; const double* e; int n;
  const double* shifts; int K; const double* params;
  double* inv; double* du; double* du2; double* dl; unsigned char* pv;  // LU factors, [row][K]
  double* Z; int64_t ldz;                                                 // n x K eigenvectors (columns)
  int iters;
};

// One lane per kept eigenvector; the lanes of a wave walk the rows together (uniform loads of d, e; coalesced
// stores of the factors).
constexpr int INV_VPW = 16;  // eigenvectors per wave: the solves are latency chains and only K / 64 full waves would
                             // exist chip-wide, so a wave carries 16 vectors (128-byte row pieces) and 4x as many run
__global__ __launch_bounds__(64) void tri_invit_kernel(InvitArgs a) {
  const int jj = blockIdx.x * INV_VPW + threadIdx.x;
  if (threadIdx.x >= INV_VPW || jj >= a.K) return;
  const int n = a.n;
  // the factors of a wave's INV_VPW vectors lie together, row after row (128-byte pieces 128 bytes apart): a wave
  // streams its own contiguous region instead of touching one line per row per array scattered over n x K
  const size_t fb = (size_t)blockIdx.x * (size_t)n * INV_VPW + threadIdx.x;
  auto fi = [fb](int i) { return fb + (size_t)i * INV_VPW; };
  const double lam = a.shifts[jj];
  const double tiny = DBL_EPSILON * a.params[0];
  auto guard = [tiny](double p) { return fabs(p) < tiny ? (p < 0.0 ? -tiny : tiny) : p; };
  // factor T - lam I = P L U.  Branch-free (one divide per row whichever way the pivot goes), rows in chunks of 8 so
  // that the uniform loads of d and e are issued together instead of one scalar round trip per row.
  double ca = a.d[0] - lam, cb = n > 1 ? a.e[0] : 0.0;
  auto factor_row = [&](int i, double c, double nd, double ne) {
    const bool swap = fabs(c) > fabs(ca);
    const double piv = swap ? c : guard(ca);
    const double r = 1.0 / piv;
    const double l = (swap ? ca : c) * r;
    a.inv[fi(i)] = r;
    a.du[fi(i)] = swap ? nd : cb;
    a.du2[fi(i)] = swap ? ne : 0.0;
    a.dl[fi(i)] = l;
    a.pv[fi(i)] = swap ? 1 : 0;
    const double na_ = swap ? cb - l * nd : nd - l * cb;
    const double nb_ = swap ? -l * ne : ne;
    ca = na_; cb = nb_;
  };
  {
    int i = 0;
    for (; i + 8 <= n - 1; i += 8) {
      double dd[8], ee[9];
#pragma unroll
      for (int k = 0; k < 8; ++k) { dd[k] = a.d[i + 1 + k]; ee[k] = a.e[i + k]; }
      ee[8] = a.e[i + 8];  // index <= n - 1: inside the array; unused (zero) when row i + 7 is the last one
#pragma unroll
      for (int k = 0; k < 8; ++k) factor_row(i + k, ee[k], dd[k] - lam, i + k + 2 < n ? ee[k + 1] : 0.0);
    }
    for (; i + 1 < n; ++i) factor_row(i, a.e[i], a.d[i + 1] - lam, i + 2 < n ? a.e[i + 1] : 0.0);
  }
  a.inv[fi(n - 1)] = 1.0 / guard(ca);
  // start vector
  double* z = a.Z + jj;
  const int64_t ldz = a.ldz;
  for (int i = 0; i < n; ++i) z[i * ldz] = hash_unit((unsigned)i, (unsigned)jj);
  double nrm2 = 0.0;
  // The solves are chains of dependent FMAs fed by strided loads; only a few dozen waves exist chip-wide, so nothing
  // hides a load's round trip except the wave itself: operands travel in chunks of CH rows, and the next chunk's
  // loads are always issued before the current chunk's chain runs (two register sets, loop unrolled by two).
  // The main loops hold FULL chunks only and no branch: with a guard per row inside them the compiler can no longer
  // count the loads in flight and waits for all of them (vmcnt(0)) before every chunk, which serialises the prefetch
  // with the chain -- that version spent 190 ns per row, one memory round trip per 8 rows.  The ragged end runs
  // through a guarded copy once per pass.
  constexpr int CH = 8;
  struct Fwd { double nxt[CH], l[CH]; unsigned char sw[CH]; };
  struct Bwd { double y[CH], u1[CH], u2[CH], r[CH]; };
  auto load_f = [&](Fwd& f, int i0) {   // rows clamped into the matrix: callable past the end
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      int i = i0 + k;
      i = i < n - 1 ? i : n - 2;
      i = i < 0 ? 0 : i;
      f.nxt[k] = z[(int64_t)(i + 1 < n ? i + 1 : n - 1) * ldz];
      f.l[k] = a.dl[fi(i)];
      f.sw[k] = a.pv[fi(i)];
    }
  };
  auto load_b = [&](Bwd& f, int i0) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int i = i0 - k >= 0 ? i0 - k : 0;
      f.y[k] = z[(int64_t)i * ldz];
      f.u1[k] = a.du[fi(i)];
      f.u2[k] = a.du2[fi(i)];
      f.r[k] = a.inv[fi(i)];
    }
  };
  for (int it = 0; it < a.iters; ++it) {
    // the previous solve's result is normalised as it is read (three unscaled solves can reach 1e48, more would
    // overflow): no separate scaling pass between the iterations
    const double sc = it == 0 ? 1.0 : 1.0 / sqrt(nrm2);
    // forward: y = L^-1 P x
    double cur = z[0] * sc;
    auto step_f = [&](const Fwd& f, int k, int row) {
      const bool swap = f.sw[k] != 0;
      const double nx = f.nxt[k] * sc;
      const double top = swap ? nx : cur, bot = swap ? cur : nx;
      z[(int64_t)row * ldz] = top;
      cur = bot - f.l[k] * top;
    };
    if (n > 1) {
      const int nf = n - 1;
      Fwd fa, fb2;
      load_f(fa, 0);
      int i0 = 0;
      for (; i0 + 2 * CH <= nf; i0 += 2 * CH) {
        load_f(fb2, i0 + CH);
#pragma unroll
        for (int k = 0; k < CH; ++k) step_f(fa, k, i0 + k);
        load_f(fa, i0 + 2 * CH);
#pragma unroll
        for (int k = 0; k < CH; ++k) step_f(fb2, k, i0 + CH + k);
      }
      // fewer than 2 CH rows left; fa holds the first CH of them
      if (i0 + CH < nf) load_f(fb2, i0 + CH);
#pragma unroll
      for (int k = 0; k < CH; ++k)
        if (i0 + k < nf) step_f(fa, k, i0 + k);
      if (i0 + CH < nf) {
#pragma unroll
        for (int k = 0; k < CH; ++k)
          if (i0 + CH + k < nf) step_f(fb2, k, i0 + CH + k);
      }
    }
    // backward: U x = y
    double x1 = cur * a.inv[fi(n - 1)], x2 = 0.0;
    z[(int64_t)(n - 1) * ldz] = x1;
    nrm2 = x1 * x1;
    auto step_b = [&](const Bwd& f, int k, int row) {
      const double xi = (f.y[k] - f.u1[k] * x1 - f.u2[k] * x2) * f.r[k];
      z[(int64_t)row * ldz] = xi;
      nrm2 += xi * xi;
      x2 = x1; x1 = xi;
    };
    if (n > 1) {
      Bwd ba, bb;
      load_b(ba, n - 2);
      int i0 = n - 2;
      for (; i0 - 2 * CH + 1 >= 0; i0 -= 2 * CH) {
        load_b(bb, i0 - CH);
#pragma unroll
        for (int k = 0; k < CH; ++k) step_b(ba, k, i0 - k);
        load_b(ba, i0 - 2 * CH);
#pragma unroll
        for (int k = 0; k < CH; ++k) step_b(bb, k, i0 - CH - k);
      }
      if (i0 - CH >= 0) load_b(bb, i0 - CH);
#pragma unroll
      for (int k = 0; k < CH; ++k)
        if (i0 - k >= 0) step_b(ba, k, i0 - k);
      if (i0 - CH >= 0) {
#pragma unroll
        for (int k = 0; k < CH; ++k)
          if (i0 - CH - k >= 0) step_b(bb, k, i0 - CH - k);
      }
    }
  }
  // final normalisation: 32 independent rows in flight
  const double s = 1.0 / sqrt(nrm2);
  {
    int i = 0;
    for (; i + 32 <= n; i += 32) {
      double t[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) t[k] = z[(int64_t)(i + k) * ldz];
#pragma unroll
      for (int k = 0; k < 32; ++k) z[(int64_t)(i + k) * ldz] = t[k] * s;
    }
    for (; i < n; ++i) z[(int64_t)i * ldz] *= s;
  }
}

// Re-orthogonalisation inside every maximal run of kept eigenvalues whose neighbours are closer than
// ortol_rel * |T| (inverse iteration leaves an angle error ~ eps |T| / gap between such neighbours): classical
// Gram-Schmidt applied twice per vector ("twice is enough"), one workgroup per run, started by the run's first
// member; the other workgroups leave at once.  Phase 1 forms all inner products with the earlier members at once
// (threads along the members, so the row-major Z is read in contiguous pieces), phase 2 subtracts them row by row.
constexpr int ORT_T = 256;
__global__ __launch_bounds__(ORT_T) void tri_ortho_kernel(const double* __restrict__ lam_kept, int K, int n,
                                                          const double* __restrict__ params, double ortol_rel,
                                                          double* Z, int64_t ldz) {
  __shared__ double coef[NIF_EIGH_MAX_N];
  __shared__ double part[ORT_T / 64][64];
  __shared__ double red[ORT_T / 64 + 1];
  const int j0 = blockIdx.x, tid = threadIdx.x;
  if (j0 >= K) return;
  const double ortol = ortol_rel * params[0];
  if (j0 > 0 && lam_kept[j0] - lam_kept[j0 - 1] <= ortol) return;    // not the first of its run
  int j1 = j0;
  while (j1 + 1 < K && lam_kept[j1 + 1] - lam_kept[j1] <= ortol) ++j1;
  if (j1 == j0) return;
  const int tx = tid & 63, ty = tid >> 6;
  for (int av = j0 + 1; av <= j1; ++av) {
    const int nbv = av - j0;
    for (int rep = 0; rep < 2; ++rep) {
      if (nbv <= 16) {
        // short runs (the usual case: two to a handful of close eigenvalues): all threads along the rows, four
        // members at a time.  With the threads along the members, as below, a run of two kept four lanes busy, each
        // walking n / 4 rows one dependent load after the other: 0.6 ms per inner product at n = 4609.
        for (int b0 = 0; b0 < nbv; b0 += 4) {
          double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
          for (int i = tid; i < n; i += ORT_T) {
            const double* zr = Z + (int64_t)i * ldz;
            const double za = zr[av];
#pragma unroll
            for (int m = 0; m < 4; ++m) a4[m] += (b0 + m < nbv ? zr[j0 + b0 + m] : 0.0) * za;
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) a4[m] = wave_sum_f64(a4[m]);
          if (tx == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) part[ty][m] = a4[m];
          }
          __syncthreads();
          if (tid < 4 && b0 + tid < nbv) {
            double sum = 0.0;
#pragma unroll
            for (int q = 0; q < ORT_T / 64; ++q) sum += part[q][tid];
            coef[b0 + tid] = sum;
          }
          __syncthreads();
        }
      } else {
      for (int b0 = 0; b0 < nbv; b0 += 64) {
        const int bb = b0 + tx;
        double acc = 0.0;
        if (bb < nbv) {
#pragma unroll 8
          for (int i = ty; i < n; i += ORT_T / 64) acc += Z[(int64_t)i * ldz + j0 + bb] * Z[(int64_t)i * ldz + av];
        }
        part[ty][tx] = acc;
        __syncthreads();
        if (ty == 0 && bb < nbv) {
          double sum = 0.0;
#pragma unroll
          for (int q = 0; q < ORT_T / 64; ++q) sum += part[q][tx];
          coef[bb] = sum;
        }
        __syncthreads();
      }
      }
      for (int i = tid; i < n; i += ORT_T) {
        const double* zr = Z + (int64_t)i * ldz + j0;
        double acc = 0.0;
        for (int bb = 0; bb < nbv; ++bb) acc += coef[bb] * zr[bb];
        Z[(int64_t)i * ldz + av] -= acc;
      }
      __syncthreads();
    }
    double nn = 0.0;
    for (int i = tid; i < n; i += ORT_T) { const double v = Z[(int64_t)i * ldz + av]; nn += v * v; }
    nn = block_sum<ORT_T>(nn, red);
    const double sc = 1.0 / sqrt(nn);
    for (int i = tid; i < n; i += ORT_T) Z[(int64_t)i * ldz + av] *= sc;
    __syncthreads();
  }
}

// ---- back-transformation Z <- H_0 H_1 ... H_{n-3} Z -------------------------------------------------------------
// Block bi holds the reflectors j0 = 64 bi ... and touches the rows j0+1 .. n-1 (nr = n - 1 - j0 of them).  Everything
// about a block that does not depend on Z -- its reflectors gathered into a dense nr x 64 panel V, the triangular factor
// T of I - V T V^T, and the product V T -- is prepared for ALL blocks by four launches before the chain over the blocks
// starts, so the chain itself is two steps per block: M = V^T Z (matrix cores) and Z -= (V T) M.  (The first version
// ran seven dependent launches per block, two of them single-wave triangular kernels of 34 and 42 us.)
// The panels of the blocks lie one after the other: block bi starts at row bt_rows_before(n, bi).
__host__ __device__ inline size_t bt_rows_before(int n, int bi) {
  return (size_t)bi * (size_t)(n - 1) - (size_t)EIG_NB * (size_t)bi * (size_t)(bi > 0 ? bi - 1 : 0) / 2;
}

// Vall: panel bi, [(r - (j0 + 1))][l] = reflector j0 + l at row r (zero beyond the nbk reflectors of the last block)
__global__ void bt_extract_kernel(const double* __restrict__ A, int64_t ld, int n, int nref, double* __restrict__ Vall) {
  const int bi = blockIdx.y, j0 = bi * EIG_NB;
  const int nbk = nref - j0 < EIG_NB ? nref - j0 : EIG_NB;
  const int l = threadIdx.x & (EIG_NB - 1);
  const int rr = blockIdx.x * (256 / EIG_NB) + (threadIdx.x / EIG_NB);
  const int nr = n - (j0 + 1);
  if (rr >= nr) return;
  const int r = j0 + 1 + rr, j = j0 + l;
  double v = 0.0;
  if (l < nbk) {
    if (r == j + 1) v = 1.0;
    else if (r >= j + 2) v = A[(size_t)r * ld + j];
  }
  Vall[(bt_rows_before(n, bi) + rr) * EIG_NB + l] = v;
}

// Spart[bi][s] = (rows s * BT_RC ... of panel bi)^T (the same rows): 64 x 64, 256 threads, 4 x 4 per thread
constexpr int BT_RC = 512;
__global__ __launch_bounds__(256) void bt_gram_kernel(const double* __restrict__ Vall, int n, int splits,
                                                      double* __restrict__ Spart) {
  __shared__ double sV[32][EIG_NB + 2];
  const int bi = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x;
  const int nr = n - 1 - bi * EIG_NB;
  const int r0 = sp * BT_RC;
  if (r0 >= nr) return;  // uniform; the factor kernel only adds the splits that exist
  const int rend = nr < r0 + BT_RC ? nr : r0 + BT_RC;
  const double* __restrict__ Vb = Vall + bt_rows_before(n, bi) * EIG_NB;
  const int ti = (tid / 16) * 4, tj = (tid % 16) * 4;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) acc[i][k2] = 0.0;
  for (int rc = r0; rc < rend; rc += 32) {
    __syncthreads();
#pragma unroll
    for (int q = tid; q < 32 * EIG_NB; q += 256) {
      const int rr = rc + q / EIG_NB;
      const double x = Vb[(size_t)(rr < rend ? rr : rend - 1) * EIG_NB + q % EIG_NB];  // clamped, see bt_tile
      sV[q / EIG_NB][q % EIG_NB] = rr < rend ? x : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      double x[4], y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = sV[r][ti + i];
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) y[k2] = sV[r][tj + k2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) acc[i][k2] += x[i] * y[k2];
    }
  }
  double* __restrict__ out = Spart + ((size_t)bi * splits + sp) * (EIG_NB * EIG_NB);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) out[(ti + i) * EIG_NB + tj + k2] = acc[i][k2];
}

// T = (diag(1/tau) + striu(S))^-1, the triangular factor of the block reflector I - V T V^T; one wave per block of
// reflectors.  S is the sum of the block's partial Grams, added in split order.  Thread c owns column c
// of T in registers (fully unrolled back substitution, entries below the diagonal stay zero), every thread walks the
// same (i, l) sequence so S[i][l] is one LDS broadcast.  tau = 0 (identity reflector) gives a zero row and column.
__global__ __launch_bounds__(64) void bt_tfactor_kernel(const double* __restrict__ Spart, int splits, int n, int nref,
                                                        const double* __restrict__ tau_all, double* __restrict__ Tall) {
  __shared__ __attribute__((aligned(16))) double sS[EIG_NB][EIG_NB];
  __shared__ double st[EIG_NB];
  const int bi = blockIdx.x, j0 = bi * EIG_NB;
  const int nbk = nref - j0 < EIG_NB ? nref - j0 : EIG_NB;
  const int nsp = (n - 1 - j0 + BT_RC - 1) / BT_RC;
  const double* __restrict__ S = Spart + (size_t)bi * splits * (EIG_NB * EIG_NB);
  double* __restrict__ T = Tall + (size_t)bi * (EIG_NB * EIG_NB);
  const int c = threadIdx.x;
  for (int q = c; q < EIG_NB * EIG_NB; q += 64) {
    const int i = q / EIG_NB, l = q % EIG_NB;
    double v = 0.0;
    if (i < nbk && l < nbk)
      for (int sp = 0; sp < nsp; ++sp) v += S[(size_t)sp * (EIG_NB * EIG_NB) + q];
    sS[i][l] = v;
  }
  st[c] = c < nbk ? tau_all[j0 + c] : 0.0;
  __syncthreads();
  double t[EIG_NB];
#pragma unroll
  for (int l = 0; l < EIG_NB; ++l) t[l] = l == c ? st[c] : 0.0;
#pragma unroll
  for (int i = EIG_NB - 2; i >= 0; --i) {
    double sum = 0.0;
#pragma unroll
    for (int l = i + 1; l < EIG_NB; ++l) sum += sS[i][l] * t[l];
    const double v = -st[i] * sum;
    t[i] = i < c ? v : t[i];
  }
#pragma unroll
  for (int l = 0; l < EIG_NB; ++l) T[(size_t)l * EIG_NB + c] = t[l];
}

// acc = (rows r0 .. r0+63 of Vb) (columns cb .. cb+63 of Y): 64 x 64 tile, 256 threads, 4 x 4 per thread, l in two
// halves of 32
__device__ __forceinline__ void bt_tile(const double* __restrict__ Vb, const double* __restrict__ Y, int64_t ldy, int nr,
                                        int nbk, int K, int r0, int cb, double (&acc)[4][4]) {
  constexpr int LH = 32;
  __shared__ double sV[64][LH + 1];
  __shared__ double sY[LH][64 + 2];
  const int tid = threadIdx.x;
  const int tr = (tid / 16) * 4, tc = (tid % 16) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) acc[i][k2] = 0.0;
  for (int l0 = 0; l0 < nbk; l0 += LH) {
    __syncthreads();
    // clamped addresses and a select, not guarded loads: a guard per element makes the compiler wait for each load
    // before it issues the next (s_waitcnt vmcnt(0)), and a tile's sixteen loads per thread arrive one by one
#pragma unroll
    for (int q = tid; q < 64 * LH; q += 256) {
      const int rr = q / LH, l = l0 + q % LH;
      const bool in = r0 + rr < nr && l < nbk;
      const double x = Vb[(size_t)(r0 + rr < nr ? r0 + rr : nr - 1) * EIG_NB + (l < nbk ? l : nbk - 1)];
      sV[rr][q % LH] = in ? x : 0.0;
    }
#pragma unroll
    for (int q = tid; q < LH * 64; q += 256) {
      const int l = l0 + q / 64, cc = q % 64;
      const bool in = cb + cc < K && l < nbk;
      const double y = Y[(size_t)(l < nbk ? l : nbk - 1) * ldy + (cb + cc < K ? cb + cc : K - 1)];
      sY[q / 64][cc] = in ? y : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int l = 0; l < LH; ++l) {
      double v[4], y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = sV[tr + i][l];
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) y[k2] = sY[l][tc + k2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) acc[i][k2] += v[i] * y[k2];
    }
  }
}

// Z[j0+1+rr][c] -= sum_l VT[rr][l] M[l][c]
__global__ __launch_bounds__(256) void bt_update_kernel(const double* __restrict__ VT, const double* __restrict__ M,
                                                        int64_t ldm, int nr, int nbk, int K, double* __restrict__ Zs,
                                                        int64_t ldz) {
  const int r0 = blockIdx.y * 64, cb = blockIdx.x * 64;
  const int tr = (threadIdx.x / 16) * 4, tc = (threadIdx.x % 16) * 4;
  double acc[4][4];
  bt_tile(VT, M, ldm, nr, nbk, K, r0, cb, acc);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + tr + i;
    if (r >= nr) continue;
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
      const int c = cb + tc + k2;
      if (c < K) Zs[(size_t)r * ldz + c] -= acc[i][k2];
    }
  }
}

// VTall panel bi = (Vall panel bi) T_bi, every block of reflectors in one launch
__global__ __launch_bounds__(256) void bt_vt_kernel(const double* __restrict__ Vall, const double* __restrict__ Tall, int n,
                                                    double* __restrict__ VTall) {
  const int bi = blockIdx.y;
  const int nr = n - 1 - bi * EIG_NB;
  const int r0 = blockIdx.x * 64;
  if (r0 >= nr) return;  // uniform over the workgroup
  const size_t at = bt_rows_before(n, bi) * EIG_NB;
  const int tr = (threadIdx.x / 16) * 4, tc = (threadIdx.x % 16) * 4;
  double acc[4][4];
  bt_tile(Vall + at, Tall + (size_t)bi * (EIG_NB * EIG_NB), EIG_NB, nr, EIG_NB, EIG_NB, r0, 0, acc);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + tr + i;
    if (r >= nr) continue;
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) VTall[at + (size_t)r * EIG_NB + tc + k2] = acc[i][k2];
  }
}

// copy a symmetric matrix into the padded working layout (ld even, pad column zero)
__global__ void eigh_copy_in_kernel(const double* __restrict__ G, int64_t ldg, int n, double* __restrict__ A, int64_t ld) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= ld) return;
  A[(size_t)r * ld + c] = c < n ? G[(size_t)r * ldg + c] : 0.0;
}

__global__ void eigh_copy_lam_kernel(const double* __restrict__ lam, int n, int K, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K) out[i] = lam[n - K + i];
}

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

struct EighWs {
  size_t off_A, off_vbuf, off_wbuf, off_ybuf, off_d, off_e, off_tau, off_lam, off_shift, off_params, off_kept;
  size_t off_fac, off_pv, off_Vall, off_VTall, off_Spart, off_Tall, off_M, off_xty, total;
  int bt_blocks, bt_splits;
  int64_t ld, ldz;
  size_t xty_bytes;
};

static EighWs eigh_ws(int n) {
  EighWs w;
  w.ld = (n + 1) & ~(int64_t)1;
  w.ldz = (n + 3) & ~(int64_t)3;
  size_t o = 0;
  auto take = [&o](size_t bytes) { const size_t at = o; o += up256(bytes); return at; };
  w.off_A = take((size_t)n * w.ld * 8);
  w.off_vbuf = take(3 * (size_t)w.ld * 8);
  w.off_wbuf = take(2 * (size_t)w.ld * 8);
  w.off_ybuf = take(2 * (size_t)w.ld * 8);
  w.off_d = take((size_t)n * 8);
  w.off_e = take((size_t)n * 8);
  w.off_tau = take((size_t)n * 8);
  w.off_lam = take((size_t)n * 8);
  w.off_shift = take((size_t)n * 8);
  w.off_params = take(8 * 8);
  w.off_kept = take(64);
  const size_t nk = (size_t)((n + 15) / 16) * 16 * (size_t)n;   // n vectors rounded up to whole waves of 16
  w.off_fac = take(4 * nk * 8);   // inv, du, du2, dl for up to n kept vectors
  w.off_pv = take(nk);
  w.bt_blocks = n > 2 ? (n - 2 + EIG_NB - 1) / EIG_NB : 0;
  w.bt_splits = (n + BT_RC - 1) / BT_RC;
  const size_t panel_bytes = bt_rows_before(n, w.bt_blocks) * EIG_NB * 8;   // ~ n^2 / 2 doubles
  w.off_Vall = take(panel_bytes);
  w.off_VTall = take(panel_bytes);
  w.off_Spart = take((size_t)w.bt_blocks * w.bt_splits * EIG_NB * EIG_NB * 8);
  w.off_Tall = take((size_t)w.bt_blocks * EIG_NB * EIG_NB * 8);
  w.off_M = take((size_t)EIG_NB * w.ldz * 8);
  // nif_xty workspace of the two products of a block (f64 path, 128-wide tiles): the split-K plan aims at no more
  // than 512 workgroups, so tiles x splits <= 512 slabs of 128 x 128 doubles (64 MiB) plus the norm partials
  w.xty_bytes = (size_t)72 << 20;
  w.off_xty = take(w.xty_bytes + 4096);
  w.total = o;
  return w;
}

// launches of every trd_step_kernel<CPW, TWO, MODE> since the process started: [mode][class][two]
static unsigned long long g_trd_launches[3][5][2];

}  // namespace nif

extern "C" int nif_eigh_trd_launch_counts(uint64_t* counts30) {
  if (!counts30) return NIF_ERR_INVALID_ARG;
  for (int m = 0; m < 3; ++m)
    for (int c = 0; c < 5; ++c)
      for (int w = 0; w < 2; ++w)
        counts30[(m * 5 + c) * 2 + w] = __atomic_load_n(&nif::g_trd_launches[m][c][w], __ATOMIC_RELAXED);
  return NIF_OK;
}

extern "C" size_t nif_eigh_kept_workspace_bytes(int n) {
  if (n <= 0 || n > NIF_EIGH_MAX_N) return 0;
  return nif::eigh_ws(n).total;
}

extern "C" int nif_eigh_kept_f64(const double* G, int64_t ldg, int n, double cutoff_rel, double* lam, double* V,
                                 int64_t ldv, int32_t* kept_host, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_eigh_kept_f64");
  using namespace nif;
  if (!G || !lam || !V || !kept_host || n <= 0 || ldg < n || !(cutoff_rel >= 0.0)) return NIF_ERR_INVALID_ARG;
  if (n > NIF_EIGH_MAX_N) return NIF_ERR_UNSUPPORTED;
  const EighWs w = eigh_ws(n);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  if (ldv < n) return NIF_ERR_INVALID_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  TrdArgs t;
  t.A = reinterpret_cast<double*>(b + w.off_A);
  t.ld = w.ld;
  t.n = n;
  t.j = 0;
  t.vbuf = reinterpret_cast<double*>(b + w.off_vbuf);
  t.wbuf = reinterpret_cast<double*>(b + w.off_wbuf);
  t.ybuf = reinterpret_cast<double*>(b + w.off_ybuf);
  t.d = reinterpret_cast<double*>(b + w.off_d);
  t.e = reinterpret_cast<double*>(b + w.off_e);
  t.tau = reinterpret_cast<double*>(b + w.off_tau);
  double* lam_all = reinterpret_cast<double*>(b + w.off_lam);
  double* shifts = reinterpret_cast<double*>(b + w.off_shift);
  double* params = reinterpret_cast<double*>(b + w.off_params);
  int* kept_dev = reinterpret_cast<int*>(b + w.off_kept);

  // 1. tridiagonalisation
  hipLaunchKernelGGL(eigh_copy_in_kernel, dim3((unsigned)((w.ld + 255) / 256), n), dim3(256), 0, st, G, ldg, n, t.A, w.ld);
  if (n <= 2) {
    hipLaunchKernelGGL(trd_small_kernel, dim3(1), dim3(64), 0, st, t);
  } else {
    const size_t lds = (3 * (size_t)w.ld + 32 + 2 * TRD_W * TRD_ROWS) * sizeof(double);
    if (lds > 160 * 1024) return NIF_ERR_UNSUPPORTED;
    // kernel table: [mode][class][two]; classes 1, 2, 3, 4, 6 columns-chunks per wave
    typedef void (*trd_fn)(TrdArgs);
    static const trd_fn table[3][5][2] = {
        {{trd_step_kernel<1, false, 0>, nullptr}, {trd_step_kernel<2, false, 0>, trd_step_kernel<2, true, 0>},
         {trd_step_kernel<3, false, 0>, trd_step_kernel<3, true, 0>}, {trd_step_kernel<4, false, 0>, trd_step_kernel<4, true, 0>},
         {trd_step_kernel<6, false, 0>, nullptr}},
        {{nullptr, nullptr}, {trd_step_kernel<2, false, 1>, trd_step_kernel<2, true, 1>},
         {trd_step_kernel<3, false, 1>, trd_step_kernel<3, true, 1>}, {trd_step_kernel<4, false, 1>, trd_step_kernel<4, true, 1>},
         {nullptr, nullptr}},
        {{nullptr, nullptr}, {trd_step_kernel<2, false, 2>, trd_step_kernel<2, true, 2>},
         {trd_step_kernel<3, false, 2>, trd_step_kernel<3, true, 2>}, {trd_step_kernel<4, false, 2>, nullptr},
         {nullptr, nullptr}}};
    static PerDeviceOnce attr_once;  // the attribute is per device: a second GPU of the process needs it too
    attr_once.run([&] {
      for (int m = 0; m < 3; ++m)
        for (int c = 0; c < 5; ++c)
          for (int w2 = 0; w2 < 2; ++w2)
            if (table[m][c][w2])
              (void)hipFuncSetAttribute((const void*)table[m][c][w2], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    // tuning aids, read per call (the parity tests switch them inside one process to reach every kernel of the table)
    const int trd_one = env_int("NIF_EIGH_TRD_ONE_GROUP");  // the single-prefetch kernels everywhere
    const int trd_kp = env_int("NIF_EIGH_TRD_KP");          // 1 = every step stores (MODE 0 only)
    prof_begin(NIF_PROF_EIGH_TRD, st);
    double trd_bytes = 0.0;
    int pend = 0;  // rank-2 updates not yet stored into the matrix when step j starts
    for (int j = 0; j + 2 < n; ++j) {
      t.j = j;
      const int groups = (n - (j + 1) / TRD_ROWS * TRD_ROWS + TRD_ROWS - 1) / TRD_ROWS;  // row groups still alive
      int b8 = (groups + 7) / 8;   // one group of rows per workgroup when they all fit
      if (b8 > 32) b8 = 32;
      if (b8 < 1) b8 = 1;
      const int c0 = (j + 1) & ~1, cend = (n + 1) & ~1;
      const int chunks = (cend - c0 + 127) / 128;
      const int cpw = (chunks + TRD_W - 1) / TRD_W;   // chunks of 128 columns a wave holds in flight per group
      const dim3 grid(8 * b8), block(TRD_T);
      // every other step leaves the matrix unstored (MODE 1) and the next one applies both updates (MODE 2); the
      // first step has nothing pending, the last one must store for trd_finish_kernel, above 4096 trailing
      // columns the two-update kernel does not fit the register file, and up to 1024 the stores are too small a part
      // of a step to pay for the second update (measured: 5.3 + 6.0 us per pair of steps against 2 x 5.3)
      int mode;
      if (pend == 2) mode = 2;
      else if (pend == 1 && trd_kp != 1 && cpw >= 2 && cpw <= 4 && j + 3 < n) mode = 1;
      else mode = 0;
      pend = mode == 1 ? 2 : 1;
      // algorithmic bytes of the step: the trailing matrix read once, and written once unless the step leaves it alone
      trd_bytes += (mode == 1 || j == 0 ? 8.0 : 16.0) * (double)(n - j - 1) * (double)(n - j - 1);
      const int cls = cpw <= 1 ? 0 : (cpw == 2 ? 1 : (cpw == 3 ? 2 : (cpw == 4 ? 3 : 4)));
      // a second group in flight from the start wherever a workgroup owns more than one and the registers allow it
      int two = groups > 8 * b8 && !trd_one ? 1 : 0;
      if (!table[mode][cls][two]) two = 0;
      // (a MODE 1 step runs at an odd column j and its MODE 2 partner at j + 1, which has the same first trailing
      // column pair c0 and hence the same class: classes without a MODE 1 kernel need no MODE 2 kernel either)
      if (!table[mode][cls][two]) return NIF_ERR_UNSUPPORTED;
      hipLaunchKernelGGL(table[mode][cls][two], grid, block, lds, st, t);
      __atomic_fetch_add(&g_trd_launches[mode][cls][two], 1ull, __ATOMIC_RELAXED);
    }
    hipLaunchKernelGGL(trd_finish_kernel, dim3(1), dim3(64), 0, st, t);
    prof_end(NIF_PROF_EIGH_TRD, st);
    prof_add_work(NIF_PROF_EIGH_TRD, trd_bytes);
  }
  if (!launch_ok()) return NIF_ERR_LAUNCH;

  // 2. eigenvalues, cutoff, shifts
  hipLaunchKernelGGL(tri_bounds_kernel, dim3(1), dim3(256), 0, st, t.d, t.e, n, params);
  {
    const size_t lds = 2 * (size_t)n * sizeof(double);
    static PerDeviceOnce attr_once;
    attr_once.run([&] {
      (void)hipFuncSetAttribute((const void*)tri_eigenvalues_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(tri_eigenvalues_kernel, dim3((n + BIS_T / BIS_P - 1) / (BIS_T / BIS_P)), dim3(BIS_T), lds, st, t.d, t.e, n,
                       params, lam_all);
  }
  hipLaunchKernelGGL(tri_select_kernel, dim3((n + 255) / 256), dim3(256), 0, st, lam_all, n, cutoff_rel, params, shifts, kept_dev);
  if (!launch_ok()) return NIF_ERR_LAUNCH;
  int K = 0;
  if (hipMemcpyAsync(&K, kept_dev, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return NIF_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return NIF_ERR_LAUNCH;
  *kept_host = K;
  if (K <= 0 || K > n) return K == 0 ? NIF_OK : NIF_ERR_LAUNCH;
  hipLaunchKernelGGL(eigh_copy_lam_kernel, dim3((K + 255) / 256), dim3(256), 0, st, lam_all, n, K, lam);

  // 3. eigenvectors of T (columns of V, n x K, ld ldv)
  InvitArgs ia;
  ia.d = t.d; ia.e = t.e; ia.n = n; ia.shifts = shifts; ia.K = K; ia.params = params;
  double* fac = reinterpret_cast<double*>(b + w.off_fac);
  const size_t fstride = (size_t)((K + INV_VPW - 1) / INV_VPW) * INV_VPW * (size_t)n;   // K rounded up to whole waves
  ia.inv = fac;
  ia.du = fac + fstride;
  ia.du2 = fac + 2 * fstride;
  ia.dl = fac + 3 * fstride;
  ia.pv = b + w.off_pv;
  ia.Z = V;
  ia.ldz = ldv;
  ia.iters = 3;
  hipLaunchKernelGGL(tri_invit_kernel, dim3((K + INV_VPW - 1) / INV_VPW), dim3(64), 0, st, ia);
  hipLaunchKernelGGL(tri_ortho_kernel, dim3(K), dim3(ORT_T), 0, st, lam, K, n, params, 1e-9, V, ldv);
  if (!launch_ok()) return NIF_ERR_LAUNCH;

  // 4. back-transformation, last block of reflectors first
  if (n > 2) {
    double* Vall = reinterpret_cast<double*>(b + w.off_Vall);
    double* VTall = reinterpret_cast<double*>(b + w.off_VTall);
    double* Spart = reinterpret_cast<double*>(b + w.off_Spart);
    double* Tall = reinterpret_cast<double*>(b + w.off_Tall);
    double* M = reinterpret_cast<double*>(b + w.off_M);
    void* xws = b + w.off_xty;
    const int nref = n - 2;
    const int nblk = w.bt_blocks;
    // everything that does not depend on Z, for all blocks at once
    hipLaunchKernelGGL(bt_extract_kernel, dim3((n + 3) / 4, nblk), dim3(256), 0, st, t.A, w.ld, n, nref, Vall);
    hipLaunchKernelGGL(bt_gram_kernel, dim3(w.bt_splits, nblk), dim3(256), 0, st, Vall, n, w.bt_splits, Spart);
    hipLaunchKernelGGL(bt_tfactor_kernel, dim3(nblk), dim3(64), 0, st, Spart, w.bt_splits, n, nref, t.tau, Tall);
    hipLaunchKernelGGL(bt_vt_kernel, dim3((n + 63) / 64, nblk), dim3(256), 0, st, Vall, Tall, n, VTall);
    if (!launch_ok()) return NIF_ERR_LAUNCH;
    for (int bi = nblk - 1; bi >= 0; --bi) {
      const int j0 = bi * EIG_NB;
      const int nbk = nref - j0 < EIG_NB ? nref - j0 : EIG_NB;
      const int nr = n - (j0 + 1);
      const size_t at = bt_rows_before(n, bi) * EIG_NB;
      double* Zs = V + (size_t)(j0 + 1) * ldv;
      const int rc = nif_xty(Vall + at, EIG_NB, Zs, ldv, nr, nbk, K, NIF_F64, NIF_F64, 0, M, w.ldz, nullptr, nullptr, xws,
                             w.xty_bytes, stream);
      if (rc != NIF_OK) return rc;
      hipLaunchKernelGGL(bt_update_kernel, dim3((K + 63) / 64, (nr + 63) / 64), dim3(256), 0, st, VTall + at, M, w.ldz, nr,
                         nbk, K, Zs, ldv);
    }
  }
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
