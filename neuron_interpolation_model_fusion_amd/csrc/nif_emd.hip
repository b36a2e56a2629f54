// nif_emd.hip -- exact optimal transport (earth mover's distance) for dense n0 x n1 costs with
// arbitrary marginals: the general case of OTFusion's plan (otfusion.py:18-32 `ot.emd`, call site
// :309), i.e. non-uniform neuron importances or models of different widths.  The uniform square
// case is an assignment problem and goes through nif_lap.hip instead.
//
// Algorithm: successive shortest augmenting paths on the transportation problem (the
// Jonker-Volgenant scheme of nif_lap.hip with capacities).  Duals (u, v) keep every reduced cost
// c_ij - u_i - v_j >= 0 and = 0 on the support of the flow; for each source row with supply left a
// Dijkstra over the columns runs in the residual graph (forward arcs: any (i, j); backward arcs:
// support edges), ends at the nearest column with demand left, and the bottleneck amount is pushed
// along the path.  One workgroup: columns live in registers (CPT per lane), one cost row is
// streamed per expanded row (coalesced), the argmin is a wave-shuffle + LDS reduction.  The support
// is a forest-like sparse structure (<= n0 + n1 - 1 edges when non-degenerate) kept as per-column
// linked lists in the workspace; lane 0 edits it during an augmentation.
//
// Latency-bound like the LAP kernel: 4*n1 bytes per expanded row.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

struct EmdWs {
  size_t off_head, off_next, off_erow, off_eflow, off_dem, off_scanrow, off_scandr, off_u, off_v, total;
  int cap;
};

static EmdWs emd_ws(int n0, int n1) {
  EmdWs w;
  w.cap = 4 * (n0 + n1) + 64;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 255) & ~(size_t)255;
    return at;
  };
  w.off_head = take((size_t)n1 * sizeof(int));
  w.off_next = take((size_t)w.cap * sizeof(int));
  w.off_erow = take((size_t)w.cap * sizeof(int));
  w.off_eflow = take((size_t)w.cap * sizeof(double));
  w.off_dem = take((size_t)n1 * sizeof(double));
  w.off_scanrow = take((size_t)n0 * sizeof(int));
  w.off_scandr = take((size_t)n0 * sizeof(double));
  w.off_u = take((size_t)n0 * sizeof(double));
  w.off_v = take((size_t)n1 * sizeof(double));
  w.total = o;
  return w;
}

// v_j = min_i C_ij, u_i = 0: feasible duals to start from (all reduced costs >= 0).
__global__ void emd_init_kernel(const float* __restrict__ C, int64_t ldc, int n0, int n1, double* __restrict__ u,
                                double* __restrict__ v) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n1) {
    float m = INFINITY;
    for (int i = 0; i < n0; ++i) m = fminf(m, C[(int64_t)i * ldc + j]);
    v[j] = (double)m;
  }
  for (int i = j; i < n0; i += gridDim.x * blockDim.x) u[i] = 0.0;
}

__device__ __forceinline__ bool emd_better(double v, int f, int j, double bv, int bf, int bj) {
  if (v < bv) return true;
  if (v > bv) return false;
  if (f != bf) return f > bf;  // among equal distances a column that still has demand ends the search
  return j < bj;
}

// status[0]: number of support edges (>= 0), or NIF_ERR_* (< 0).
template <int CPT>
__global__ __launch_bounds__(1024) void emd_kernel(const float* __restrict__ C, int64_t ldc, int n0, int n1,
                                                   const double* __restrict__ a, const double* __restrict__ b,
                                                   double tol, double* __restrict__ u, double* __restrict__ vg,
                                                   int* __restrict__ head, int* __restrict__ nxt,
                                                   int* __restrict__ erow, double* __restrict__ eflow,
                                                   double* __restrict__ dem, int* __restrict__ scanrow,
                                                   double* __restrict__ scandr, int cap, int out_cap,
                                                   int* __restrict__ colptr, int* __restrict__ rowidx,
                                                   double* __restrict__ val, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) int emd_sm[];
  int* pred_row = emd_sm;       // [n1] row that set the column's tentative distance
  int* pred_col = emd_sm + n1;  // [n0] column a row was reached from; -1 root, -2 not reached
  double* redv = reinterpret_cast<double*>(emd_sm + ((n0 + n1 + 3) & ~3));  // [2][16]
  int* redj = reinterpret_cast<int*>(redv + 32);                            // [2][16]
  int* redf = redj + 32;                                                    // [2][16]
  int* ctl = redf + 32;  // 0: rows reached, 1: free-list head, 2: pool high-water mark, 3: error
  double* dctl = reinterpret_cast<double*>(ctl + 8);  // 0: amount pushed by the last augmentation

  const int T = blockDim.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nw = T >> 6;

  double v[CPT], dist[CPT];
  unsigned settled = 0, hasdem = 0;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    v[c] = 0.0;
    if (j < n1) {
      v[c] = vg[j];
      head[j] = -1;
      const double bj = b[j];
      dem[j] = bj;
      if (bj > tol) hasdem |= 1u << c;
    }
  }
  if (tid == 0) {
    ctl[1] = -1;
    ctl[2] = 0;
    ctl[3] = 0;
  }
  __syncthreads();

  int step = 0;
  long budget = 64L * (n0 + n1) + 1024;  // augmentations; far above the n0 + n1 a non-degenerate problem needs
  for (int i0 = 0; i0 < n0; ++i0) {
    double r = a[i0];
    while (r > tol) {
      if (--budget < 0 || ctl[3] != 0) break;
      settled = 0;
#pragma unroll
      for (int c = 0; c < CPT; ++c) dist[c] = INFINITY;
      for (int k = tid; k < n0; k += T) pred_col[k] = -2;
      __syncthreads();
      if (tid == 0) {
        pred_col[i0] = -1;
        scanrow[0] = i0;
        scandr[0] = 0.0;
        ctl[0] = 1;
      }
      auto relax = [&](int i, double base) {
        const double ui = u[i];
        const float* __restrict__ crow = C + (int64_t)i * ldc;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const int j = tid + c * T;
          if (j < n1 && !((settled >> c) & 1u)) {
            const double rr = base + (double)crow[j] - ui - v[c];
            if (rr < dist[c]) {
              dist[c] = rr;
              pred_row[j] = i;
            }
          }
        }
      };
      relax(i0, 0.0);

      int sink = -1;
      double D = 0.0;
      for (int it = 0; it <= n1; ++it) {  // every pass settles one more column
        double bv = INFINITY;
        int bj = INT_MAX, bf = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const int j = tid + c * T;
          if (j < n1 && !((settled >> c) & 1u)) {
            const int f = (hasdem >> c) & 1u;
            if (emd_better(dist[c], f, j, bv, bf, bj)) {
              bv = dist[c];
              bf = f;
              bj = j;
            }
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const double ov = shfl_xor_f64(bv, o);
          const int oj = __shfl_xor(bj, o, 64);
          const int of = __shfl_xor(bf, o, 64);
          if (oj != INT_MAX && emd_better(ov, of, oj, bv, bf, bj)) {
            bv = ov;
            bf = of;
            bj = oj;
          }
        }
        const int pb = (step & 1) * 16;
        ++step;
        if (lane == 0) {
          redv[pb + wave] = bv;
          redj[pb + wave] = bj;
          redf[pb + wave] = bf;
        }
        __syncthreads();
        bv = redv[pb];
        bj = redj[pb];
        bf = redf[pb];
        for (int w = 1; w < nw; ++w) {
          const double ov = redv[pb + w];
          const int oj = redj[pb + w], of = redf[pb + w];
          if (oj != INT_MAX && emd_better(ov, of, oj, bv, bf, bj)) {
            bv = ov;
            bf = of;
            bj = oj;
          }
        }
        if (bj == INT_MAX || !(bv < INFINITY)) break;  // nothing reachable is left
        D = bv;
        if (bf) {
          sink = bj;
          break;
        }
        if (tid == bj % T) settled |= 1u << (bj / T);
        // the settled column is saturated: continue through every row that ships to it
        for (int e = head[bj]; e != -1; e = nxt[e]) {
          const int row = erow[e];
          const bool fresh = pred_col[row] == -2;
          __syncthreads();
          if (fresh) {
            if (tid == 0) {
              pred_col[row] = bj;
              const int k = ctl[0];
              scanrow[k] = row;
              scandr[k] = D;
              ctl[0] = k + 1;
            }
            relax(row, D);
          }
        }
      }
      if (sink < 0) {  // rounding leftovers with every demand met: this source is done
        r = 0.0;
        break;
      }

      if (tid == 0) {
        double delta = fmin(r, dem[sink]);
        for (int j = sink;;) {  // bottleneck over the backward arcs of the path
          const int i = pred_row[j];
          const int jp = pred_col[i];
          if (jp < 0) break;
          int e = head[jp];
          while (erow[e] != i) e = nxt[e];
          delta = fmin(delta, eflow[e]);
          j = jp;
        }
        for (int j = sink;;) {
          const int i = pred_row[j];
          int e = head[j];
          while (e != -1 && erow[e] != i) e = nxt[e];
          if (e != -1) {
            eflow[e] += delta;
          } else {
            e = ctl[1];
            if (e != -1) {
              ctl[1] = nxt[e];
            } else if (ctl[2] < cap) {
              e = ctl[2]++;
            } else {
              ctl[3] = NIF_ERR_WORKSPACE;
              break;
            }
            erow[e] = i;
            eflow[e] = delta;
            nxt[e] = head[j];
            head[j] = e;
          }
          const int jp = pred_col[i];
          if (jp < 0) break;
          int prev = -1;
          e = head[jp];
          while (erow[e] != i) {
            prev = e;
            e = nxt[e];
          }
          eflow[e] -= delta;
          if (eflow[e] <= tol) {  // the arc leaves the support
            if (prev == -1) head[jp] = nxt[e];
            else nxt[prev] = nxt[e];
            nxt[e] = ctl[1];
            ctl[1] = e;
          }
          j = jp;
        }
        double left = dem[sink] - delta;
        if (left <= tol) left = 0.0;
        dem[sink] = left;
        dctl[0] = delta;
      }
      __syncthreads();
      r -= dctl[0];
      if (tid == sink % T && !(dem[sink] > tol)) hasdem &= ~(1u << (sink / T));
      // duals: settled columns move down, reached rows move up, so the new support stays tight
#pragma unroll
      for (int c = 0; c < CPT; ++c)
        if ((settled >> c) & 1u) v[c] += dist[c] - D;
      const int reached = ctl[0];
      for (int k = tid; k < reached; k += T) u[scanrow[k]] += D - scandr[k];
      __syncthreads();
    }
    if (budget < 0 || ctl[3] != 0) break;
  }

  // ---- emit the plan column by column (CSC) and the duals --------------------------------------------
  __syncthreads();
  int err = ctl[3];
  if (budget < 0 && err == 0) err = NIF_ERR_UNSUPPORTED;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    if (j < n1) {
      vg[j] = v[c];
      int cnt = 0;
      for (int e = head[j]; e != -1; e = nxt[e]) ++cnt;
      pred_row[j] = cnt;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int j = 0; j < n1; ++j) {
      const int cnt = pred_row[j];
      pred_row[j] = run;
      colptr[j] = run;
      run += cnt;
    }
    colptr[n1] = run;
    if (err == 0 && run > out_cap) err = NIF_ERR_WORKSPACE;
    status[0] = err != 0 ? err : run;
    ctl[3] = err;
  }
  __syncthreads();
  if (ctl[3] != 0) return;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    if (j < n1) {
      int at = pred_row[j];
      for (int e = head[j]; e != -1; e = nxt[e]) {
        rowidx[at] = erow[e];
        val[at] = eflow[e];
        ++at;
      }
    }
  }
}

template <int CPT>
static void launch_emd(int threads, size_t lds, hipStream_t st, const float* C, int64_t ldc, int n0, int n1,
                       const double* a, const double* b, double tol, double* u, double* v, unsigned char* base,
                       const EmdWs& w, int out_cap, int* colptr, int* rowidx, double* val, int* status) {
  auto kfn = emd_kernel<CPT>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(1), dim3(threads), lds, st, C, ldc, n0, n1, a, b, tol, u, v,
                     reinterpret_cast<int*>(base + w.off_head), reinterpret_cast<int*>(base + w.off_next),
                     reinterpret_cast<int*>(base + w.off_erow), reinterpret_cast<double*>(base + w.off_eflow),
                     reinterpret_cast<double*>(base + w.off_dem), reinterpret_cast<int*>(base + w.off_scanrow),
                     reinterpret_cast<double*>(base + w.off_scandr), w.cap, out_cap, colptr, rowidx, val, status);
}

// out[j][q*sp + s] = beta * W0[j][q*sp + s]
//                  + alpha * sum_{e in column j of R} rv[e] * sum_{f in column q of Cc} cv[f] * W1[ri[e]][ci[f]*sp + s]
// R / Cc are CSC plans; a null colptr stands for the identity.
__global__ void refit_sparse_kernel(const float* __restrict__ W0, const float* __restrict__ W1, int64_t ldw1,
                                    const int* __restrict__ rptr, const int* __restrict__ ri,
                                    const float* __restrict__ rv, const int* __restrict__ cptr,
                                    const int* __restrict__ ci, const float* __restrict__ cv, float alpha, float beta,
                                    int n_in, int sp, float* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  const int width = n_in * sp;
  if (col >= width) return;
  const int q = col / sp, s = col - q * sp;
  const int r0 = rptr ? rptr[j] : j, r1 = rptr ? rptr[j + 1] : j + 1;
  const int c0 = cptr ? cptr[q] : q, c1 = cptr ? cptr[q + 1] : q + 1;
  float acc = 0.f;
  for (int e = r0; e < r1; ++e) {
    const int i = rptr ? ri[e] : j;
    const float we = rptr ? rv[e] : 1.f;
    const float* __restrict__ wrow = W1 + (int64_t)i * ldw1;
    float inner = 0.f;
    for (int f = c0; f < c1; ++f) {
      const int p = cptr ? ci[f] : q;
      inner += (cptr ? cv[f] : 1.f) * wrow[(int64_t)p * sp + s];
    }
    acc += we * inner;
  }
  const int64_t o = (int64_t)j * width + col;
  out[o] = (W0 ? beta * W0[o] : 0.f) + alpha * acc;
}

}  // namespace nif

extern "C" size_t nif_emd_workspace_bytes(int n0, int n1) {
  if (n0 <= 0 || n1 <= 0 || n0 > NIF_EMD_MAX_N || n1 > NIF_EMD_MAX_N) return 0;
  return nif::emd_ws(n0, n1).total;
}

extern "C" int nif_emd_solve_f32(const float* C, int64_t ldc, int n0, int n1, const double* a, const double* b,
                                 double tol, int32_t* colptr, int32_t* rowidx, double* val, int32_t out_cap,
                                 int32_t* status, double* u, double* v, void* ws, size_t ws_bytes, void* stream) {
  using namespace nif;
  if (!C || !a || !b || !colptr || !rowidx || !val || !status || n0 <= 0 || n1 <= 0 || ldc < n1 || out_cap <= 0 ||
      !(tol >= 0.0))
    return NIF_ERR_INVALID_ARG;
  if (n0 > NIF_EMD_MAX_N || n1 > NIF_EMD_MAX_N) return NIF_ERR_UNSUPPORTED;
  const EmdWs w = emd_ws(n0, n1);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned char* base = reinterpret_cast<unsigned char*>(ws);
  double* uu = u ? u : reinterpret_cast<double*>(base + w.off_u);
  double* vv = v ? v : reinterpret_cast<double*>(base + w.off_v);
  clear_err();
  const int nmax = n0 > n1 ? n0 : n1;
  hipLaunchKernelGGL(emd_init_kernel, dim3((nmax + 255) / 256), dim3(256), 0, st, C, ldc, n0, n1, uu, vv);
  int cpt = 1;
  while (cpt < 8 && (n1 + cpt - 1) / cpt > 1024) cpt *= 2;
  const int threads = ((n1 + cpt - 1) / cpt + 63) / 64 * 64;
  if (threads > 1024) return NIF_ERR_UNSUPPORTED;
  const size_t lds = (size_t)((n0 + n1 + 3) & ~3) * sizeof(int) + 32 * sizeof(double) + (64 + 8) * sizeof(int) +
                     2 * sizeof(double);
  switch (cpt) {
    case 1: launch_emd<1>(threads, lds, st, C, ldc, n0, n1, a, b, tol, uu, vv, base, w, out_cap, colptr, rowidx, val, status); break;
    case 2: launch_emd<2>(threads, lds, st, C, ldc, n0, n1, a, b, tol, uu, vv, base, w, out_cap, colptr, rowidx, val, status); break;
    case 4: launch_emd<4>(threads, lds, st, C, ldc, n0, n1, a, b, tol, uu, vv, base, w, out_cap, colptr, rowidx, val, status); break;
    default: launch_emd<8>(threads, lds, st, C, ldc, n0, n1, a, b, tol, uu, vv, base, w, out_cap, colptr, rowidx, val, status); break;
  }
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_refit_sparse(const float* W0, const float* W1, int64_t ldw1, const int32_t* rptr, const int32_t* ri,
                                const float* rv, const int32_t* cptr, const int32_t* ci, const float* cv, float alpha,
                                float beta, int n_out, int n_in, int sp, float* out, void* stream) {
  using namespace nif;
  if (!W1 || !out || n_out <= 0 || n_in <= 0 || sp <= 0) return NIF_ERR_INVALID_ARG;
  if ((rptr != nullptr) != (ri != nullptr) || (rptr != nullptr) != (rv != nullptr)) return NIF_ERR_INVALID_ARG;
  if ((cptr != nullptr) != (ci != nullptr) || (cptr != nullptr) != (cv != nullptr)) return NIF_ERR_INVALID_ARG;
  const int64_t width = (int64_t)n_in * sp;
  if (width > INT_MAX || n_out > 65535) return NIF_ERR_UNSUPPORTED;
  clear_err();
  dim3 grid((unsigned)((width + 255) / 256), (unsigned)n_out);
  hipLaunchKernelGGL(refit_sparse_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W0, W1, ldw1, rptr,
                     ri, rv, cptr, ci, cv, alpha, beta, n_in, sp, out);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
