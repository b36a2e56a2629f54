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
// is a forest-like sparse structure (<= n0 + n1 - 1 arcs when non-degenerate) kept as a
// pool of (row, column) records with their flows; lane 0 edits it during an augmentation.
// Start: column minima then row minima as duals, and a greedy shipment along each row's tight arc.
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
  size_t off_arc, off_eflow, off_dem, off_sup, off_jbest, off_rowd, off_newrows, off_u, off_v, off_r4c, off_lap, total;
  size_t lap_bytes;
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
  w.off_arc = take((size_t)w.cap * sizeof(int2));  // (row, column); a free record has row -1 and chains through .y
  w.off_eflow = take((size_t)w.cap * sizeof(double));
  w.off_dem = take((size_t)n1 * sizeof(double));
  w.off_sup = take((size_t)n0 * sizeof(double));
  w.off_jbest = take((size_t)n0 * sizeof(int));
  w.off_rowd = take((size_t)n0 * sizeof(double));
  w.off_newrows = take((size_t)n0 * sizeof(int));
  w.off_u = take((size_t)n0 * sizeof(double));
  w.off_v = take((size_t)n1 * sizeof(double));
  // square problems start from the assignment solution (see nif_emd_solve_f32): its workspace and row4col
  w.lap_bytes = n0 == n1 ? nif_lap_workspace_bytes(n0) : 0;
  w.off_r4c = take(w.lap_bytes ? (size_t)n0 * sizeof(int) : 0);
  w.off_lap = take(w.lap_bytes);
  w.total = o;
  return w;
}

// LDS plan of the solver (word offsets).  The "big" pieces (arc records, per-row distances, the list of
// newly reached rows) live in LDS when they fit and in the workspace otherwise.
struct EmdLds {
  int off_bits;      // cset, dset bitsets
  int off_arcs;      // int2[cap_lds]            (LDS variant)
  int off_rowd;      // double[n0]               (LDS variant)
  int off_newrows;   // int[n0]                  (LDS variant)
  int off_red;       // reduction scratch + control words
  int cap_lds;
  size_t bytes;
};

static EmdLds emd_lds(int n0, int n1, bool big_in_lds) {
  EmdLds L;
  const int bw = (n1 + 31) / 32;
  // pred_row, pred_arc_col [n1]; pred_arc_row, rstamp [n0]
  L.off_bits = 2 * n1 + 2 * n0;
  int o = (L.off_bits + 2 * bw + 3) & ~3;
  L.cap_lds = n0 + n1 + 64;
  L.off_arcs = L.off_rowd = L.off_newrows = o;
  if (big_in_lds) {
    L.off_arcs = o;
    o += (2 * L.cap_lds + 3) & ~3;
    L.off_rowd = o;
    o += (2 * n0 + 3) & ~3;
    L.off_newrows = o;
    o += (n0 + 3) & ~3;
  }
  L.off_red = o;
  L.bytes = (size_t)o * sizeof(int) + 32 * sizeof(double) + (64 + 16) * sizeof(int) + 2 * sizeof(double);
  return L;
}

// Starting duals (all reduced costs >= 0, one tight arc per row): v_j = min_i C_ij, then
// u_i = min_j (C_ij - v_j) with jbest_i its argmin (lowest j among equals).
__global__ void emd_colmin_kernel(const float* __restrict__ C, int64_t ldc, int n0, int n1, double* __restrict__ v) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n1) return;
  float m = INFINITY;
  for (int i = 0; i < n0; ++i) m = fminf(m, C[(int64_t)i * ldc + j]);
  v[j] = (double)m;
}

__global__ __launch_bounds__(256) void emd_rowmin_kernel(const float* __restrict__ C, int64_t ldc, int n0, int n1,
                                                         const double* __restrict__ v, double* __restrict__ u,
                                                         int* __restrict__ jbest) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per row
  if (i >= n0) return;
  double bv = INFINITY;
  int bj = INT_MAX;
  for (int j = lane; j < n1; j += 64) {
    const double r = (double)C[(int64_t)i * ldc + j] - v[j];
    if (r < bv) {
      bv = r;
      bj = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = shfl_xor_f64(bv, o);
    const int oj = __shfl_xor(bj, o, 64);
    if (ov < bv || (ov == bv && oj < bj)) {
      bv = ov;
      bj = oj;
    }
  }
  if (lane == 0) {
    u[i] = bv;
    jbest[i] = bj;
  }
}

__device__ __forceinline__ bool emd_better(double v, int f, int j, double bv, int bf, int bj) {
  if (v < bv) return true;
  if (v > bv) return false;
  if (f != bf) return f > bf;  // among equal distances a column that still has demand ends the search
  return j < bj;
}

// status[0]: number of support arcs (>= 0), or NIF_ERR_* (< 0); [1..3] work counters.
//
// One search = Dijkstra over the columns from a row with supply left.  Support arcs are tight, so when
// the nearest unsettled column (distance D) is saturated, the whole component of the support forest
// hanging off it lies at distance D as well.  That component is swept breadth-first with one lane per
// arc (column -> rows that ship to it -> the other columns those rows ship to -> ...), stopping at the
// first level that holds a column with demand left; then all lanes relax the rows it reached, four
// cost rows in flight, and one reduction finds the next column.
template <int CPT, bool BIG_LDS>
__global__ __launch_bounds__(1024) void emd_kernel(const float* __restrict__ C, int64_t ldc, int n0, int n1,
                                                   const double* __restrict__ a, const double* __restrict__ b,
                                                   double tol, double* __restrict__ u, double* __restrict__ vg,
                                                   const int* __restrict__ jbest, int2* __restrict__ arc_g,
                                                   double* __restrict__ eflow, double* __restrict__ dem,
                                                   double* __restrict__ sup, double* __restrict__ rowd_g,
                                                   int* __restrict__ newrows_g, int cap, EmdLds L, int out_cap,
                                                   int* __restrict__ colptr, int* __restrict__ rowidx,
                                                   double* __restrict__ val, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) int emd_sm[];
  const int bw = (n1 + 31) / 32;
  int* pred_row = emd_sm;                   // [n1] row through which the column was reached
  int* pred_arc_col = emd_sm + n1;          // [n1] support arc the sweep reached the column through; -1 = by relaxation
  int* pred_arc_row = emd_sm + 2 * n1;      // [n0] support arc the row was reached through; -1 root, -2 not reached
  int* rstamp = emd_sm + 2 * n1 + n0;       // [n0] sweep level at which the row was reached
  unsigned* cset = reinterpret_cast<unsigned*>(emd_sm + L.off_bits);  // columns settled in this search
  unsigned* dset = cset + bw;                                         // columns with demand left
  int2* A = BIG_LDS ? reinterpret_cast<int2*>(emd_sm + L.off_arcs) : arc_g;
  double* rowd = BIG_LDS ? reinterpret_cast<double*>(emd_sm + L.off_rowd) : rowd_g;  // distance of a reached row
  int* newrows = BIG_LDS ? emd_sm + L.off_newrows : newrows_g;
  double* redv = reinterpret_cast<double*>(emd_sm + L.off_red);  // [2][16]
  int* redj = reinterpret_cast<int*>(redv + 32);                 // [2][16]
  int* redf = redj + 32;                                         // [2][16]
  // 1: free-list head, 2: pool high-water mark, 3: error, 4: sink of the sweep, 5/6: new columns flag (by level
  // parity), 7: number of new rows
  int* ctl = redf + 32;
  double* dctl = reinterpret_cast<double*>(ctl + 16);  // 0: amount pushed by the last augmentation

  const int T = blockDim.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nw = T >> 6;

  double v[CPT], dist[CPT];
  unsigned settled = 0;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    v[c] = 0.0;
    if (j < n1) {
      v[c] = vg[j];
      dem[j] = b[j];
    }
  }
  for (int k = tid; k < n0; k += T) {
    pred_arc_row[k] = -2;
    rstamp[k] = 0;
  }
  for (int k = tid; k < 2 * bw; k += T) cset[k] = 0u;
  __syncthreads();

  // arc pool (lane 0 only)
  auto arc_new = [&](int i, int j, double x) -> bool {
    int e = ctl[1];
    if (e != -1) {
      ctl[1] = A[e].y;
    } else if (ctl[2] < cap) {
      e = ctl[2]++;
    } else {
      ctl[3] = NIF_ERR_WORKSPACE;
      return false;
    }
    A[e] = make_int2(i, j);
    eflow[e] = x;
    return true;
  };
  auto arc_drop = [&](int e) {
    A[e] = make_int2(-1, ctl[1]);
    ctl[1] = e;
  };

  // greedy start: every row ships what it can along its tight arc (i, jbest_i), in row order
  if (tid == 0) {
    ctl[1] = -1;
    ctl[2] = 0;
    ctl[3] = 0;
    for (int i = 0; i < n0; ++i) {
      double s = a[i];
      const int j = jbest[i];
      if (j < 0 || j >= n1) {  // a cost row without a finite entry has no tight arc
        sup[i] = s > tol ? s : 0.0;
        continue;
      }
      const double d = dem[j];
      const double x = fmin(s, d);
      if (x > tol && arc_new(i, j, x)) {
        const double left = d - x;
        dem[j] = left > tol ? left : 0.0;
        s -= x;
      }
      sup[i] = s > tol ? s : 0.0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    if (j < n1 && dem[j] > tol) atomicOr(&dset[j >> 5], 1u << (j & 31));
  }
  __syncthreads();

  int step = 0, n_aug = 0, n_expand = 0;
  int lvl = 0;  // sweep level counter, never reused
#ifdef NIF_EMD_TIMING  // cycles per phase into status[4..9] (status must then hold 16 words): tools/emd_timing.py
  long long tk[6] = {0, 0, 0, 0, 0, 0}, tk0 = wall_clock64();
#define EMD_MARK(p) do { const long long t_ = wall_clock64(); tk[p] += t_ - tk0; tk0 = t_; } while (0)  // 100 MHz
#else
#define EMD_MARK(p) do {} while (0)
#endif
  long budget = 64L * (n0 + n1) + 1024;  // augmentations; far above the n0 + n1 a non-degenerate problem needs
  for (int i0 = 0; i0 < n0; ++i0) {
    double r = sup[i0];
    while (r > tol) {
      if (--budget < 0 || ctl[3] != 0) break;
      settled = 0;
#pragma unroll
      for (int c = 0; c < CPT; ++c) dist[c] = INFINITY;
      ++lvl;
      if (tid == 0) {
        pred_arc_row[i0] = -1;
        rstamp[i0] = lvl;
        rowd[i0] = 0.0;
        ctl[4] = INT_MAX;
        ctl[5] = 0;
        ctl[6] = 0;
      }
      __syncthreads();

      int sink = -1;
      bool sink_by_sweep = false;
      double D = 0.0;
      bool from_rows = true;  // the first sweep starts from the root row, later ones from a settled column
      for (int it = 0; it <= n1 + 1; ++it) {
        // ---- sweep the support component at distance D -----------------------------------------------
        EMD_MARK(4);
        const int first_lvl = from_rows ? lvl : lvl + 1;  // levels whose rows are new in this sweep
        const int hwm = ctl[2];
        for (;;) {
          if (!from_rows) {
            // columns settled so far -> rows that ship to them
            ++lvl;
            for (int e = tid; e < hwm; e += T) {
              const int2 rc = A[e];
              if (rc.x >= 0 && ((cset[rc.y >> 5] >> (rc.y & 31)) & 1u) && pred_arc_row[rc.x] == -2) {
                pred_arc_row[rc.x] = e;
                rstamp[rc.x] = lvl;
                rowd[rc.x] = D;
              }
            }
            __syncthreads();
          }
          from_rows = false;
          // rows reached at this level -> the other columns they ship to
          const int flag = 5 + (lvl & 1);
          for (int e = tid; e < hwm; e += T) {
            const int2 rc = A[e];
            if (rc.x >= 0 && rstamp[rc.x] == lvl && pred_arc_row[rc.x] != -2 &&
                !((cset[rc.y >> 5] >> (rc.y & 31)) & 1u)) {
              atomicOr(&cset[rc.y >> 5], 1u << (rc.y & 31));
              pred_arc_col[rc.y] = e;
              pred_row[rc.y] = rc.x;
              ctl[flag] = 1;
              if ((dset[rc.y >> 5] >> (rc.y & 31)) & 1u) atomicMin(&ctl[4], rc.y);
            }
          }
          __syncthreads();
          const bool grew = ctl[flag] != 0;
          const int found = ctl[4];
          if (tid == 0) ctl[5 + ((lvl + 1) & 1)] = 0;  // the other parity's flag, for the next level
          if (found != INT_MAX) {
            sink = found;
            sink_by_sweep = true;
            break;
          }
          if (!grew) break;
        }
        // columns the sweep settled: distance D, out of the race
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const int j = tid + c * T;
          if (j < n1 && !((settled >> c) & 1u) && ((cset[j >> 5] >> (j & 31)) & 1u)) {
            settled |= 1u << c;
            dist[c] = D;
          }
        }
        EMD_MARK(0);
        if (sink >= 0) break;

        // ---- rows reached by this sweep, in row order (wave 0 compacts them) --------------------------
        if (wave == 0) {
          int cnt = 0;
          for (int base = 0; base < n0; base += 64) {
            const int k = base + lane;
            const bool isnew = k < n0 && pred_arc_row[k] != -2 && rstamp[k] >= first_lvl;
            const unsigned long long m = __ballot(isnew);
            if (isnew) newrows[cnt + __popcll(m & ((1ull << lane) - 1ull))] = k;
            cnt += __popcll(m);
          }
          if (lane == 0) ctl[7] = cnt;
        }
        __syncthreads();
        const int nnew = ctl[7];
        n_expand += nnew;
        EMD_MARK(1);
        int k = 0;
        constexpr int RB = CPT <= 2 ? 8 : 4;  // cost rows in flight
        for (; k + RB - 1 < nnew; k += RB) {
          int ri[RB];
          double ru[RB];
          float cv[RB][CPT];
#pragma unroll
          for (int q = 0; q < RB; ++q) ri[q] = newrows[k + q];
#pragma unroll
          for (int q = 0; q < RB; ++q) {
            ru[q] = u[ri[q]];
            const float* __restrict__ crow = C + (int64_t)ri[q] * ldc;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
              // clamped address, no branch: a guarded load per row makes the compiler wait for every load in flight
              // (s_waitcnt vmcnt(0)) before the next row's, which serialised the four rows of a batch
              const int j = tid + c * T;
              cv[q][c] = crow[j < n1 ? j : n1 - 1];
            }
          }
#pragma unroll
          for (int q = 0; q < RB; ++q) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
              const int j = tid + c * T;
              if (j < n1 && !((settled >> c) & 1u)) {
                const double rr = D + (double)cv[q][c] - ru[q] - v[c];
                if (rr < dist[c]) {
                  dist[c] = rr;
                  pred_row[j] = ri[q];
                }
              }
            }
          }
        }
        for (; k < nnew; ++k) {
          const int i = newrows[k];
          const double ui = u[i];
          const float* __restrict__ crow = C + (int64_t)i * ldc;
#pragma unroll
          for (int c = 0; c < CPT; ++c) {
            const int j = tid + c * T;
            if (j < n1 && !((settled >> c) & 1u)) {
              const double rr = D + (double)crow[j] - ui - v[c];
              if (rr < dist[c]) {
                dist[c] = rr;
                pred_row[j] = i;
              }
            }
          }
        }

        EMD_MARK(2);
        // ---- nearest unsettled column ---------------------------------------------------------------
        double bv = INFINITY;
        int bj = INT_MAX, bf = 0;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const int j = tid + c * T;
          if (j < n1 && !((settled >> c) & 1u)) {
            const int f = (dset[j >> 5] >> (j & 31)) & 1u;
            if (emd_better(dist[c], f, j, bv, bf, bj)) {
              bv = dist[c];
              bf = f;
              bj = j;
            }
          }
        }
        // two scalar DPP reductions per wave (see nif_lap.hip): minimal distance, then the smallest key
        // (no demand left) << 31 | column among the lanes that hold it
        const bool have = bj != INT_MAX;
        const unsigned mykey = have ? ((bf ? 0u : 0x80000000u) | (unsigned)bj) : 0xffffffffu;
        const double wmin = wave_min_f64(have ? bv : INFINITY);
        const unsigned wkey = wave_min_u32((have && bv == wmin) ? mykey : 0xffffffffu);
        const int pb = (step & 1) * 16;
        ++step;
        if (lane == 0) {
          redv[pb + wave] = wmin;
          redj[pb + wave] = (int)wkey;
        }
        __syncthreads();
        bv = redv[pb];
        unsigned gkey = (unsigned)redj[pb];
        for (int w = 1; w < nw; ++w) {
          const double ov = redv[pb + w];
          const unsigned ok = (unsigned)redj[pb + w];
          if (ov < bv || (ov == bv && ok < gkey)) {
            bv = ov;
            gkey = ok;
          }
        }
        bj = gkey == 0xffffffffu ? INT_MAX : (int)(gkey & 0x7fffffffu);
        bf = (gkey & 0x80000000u) ? 0 : 1;
        EMD_MARK(3);
        if (bj == INT_MAX || !(bv < INFINITY)) break;  // nothing reachable is left
        D = bv;
        if (bf) {
          sink = bj;
          break;
        }
        // saturated: settle it and sweep what hangs off it
        if (tid == 0) {
          cset[bj >> 5] |= 1u << (bj & 31);
          pred_arc_col[bj] = -1;
        }
        __syncthreads();
      }

      if (sink >= 0 && tid == 0) {
        // path: sink <- row <- column <- row ... <- root; columns the sweep reached carry their support arc
        double delta = fmin(r, dem[sink]);
        const int first_arc = sink_by_sweep ? pred_arc_col[sink] : -1;
        for (int j = sink, ea = first_arc;;) {
          const int i = ea >= 0 ? A[ea].x : pred_row[j];
          const int eb = pred_arc_row[i];  // arc (i, column the row was reached from)
          if (eb < 0) break;
          delta = fmin(delta, eflow[eb]);
          j = A[eb].y;
          ea = pred_arc_col[j];
        }
        for (int j = sink, ea = first_arc;;) {
          const int i = ea >= 0 ? A[ea].x : pred_row[j];
          if (ea >= 0) {
            eflow[ea] += delta;
          } else if (!arc_new(i, j, delta)) {
            break;
          }
          const int eb = pred_arc_row[i];
          if (eb < 0) break;
          j = A[eb].y;
          ea = pred_arc_col[j];
          eflow[eb] -= delta;
          if (eflow[eb] <= tol) arc_drop(eb);  // the arc leaves the support
        }
        double left = dem[sink] - delta;
        if (left <= tol) {
          left = 0.0;
          dset[sink >> 5] &= ~(1u << (sink & 31));
        }
        dem[sink] = left;
        dctl[0] = delta;
      }
      __syncthreads();
      if (sink >= 0) {
        ++n_aug;
        r -= dctl[0];
        // duals: settled columns move down, reached rows move up, so the new support stays tight
#pragma unroll
        for (int c = 0; c < CPT; ++c)
          if ((settled >> c) & 1u) v[c] += dist[c] - D;
        for (int k = tid; k < n0; k += T)
          if (pred_arc_row[k] != -2) u[k] += D - rowd[k];
      } else {
        r = 0.0;  // rounding leftovers with every demand met: this source is done
      }
      for (int k = tid; k < n0; k += T) pred_arc_row[k] = -2;
      for (int k = tid; k < bw; k += T) cset[k] = 0u;
      __syncthreads();
    }
    if (budget < 0 || ctl[3] != 0) break;
  }

  // ---- emit the plan column by column (CSC) and the duals --------------------------------------------
  __syncthreads();
  int err = ctl[3];
  if (budget < 0 && err == 0) err = NIF_ERR_UNSUPPORTED;
  const int hwm = ctl[2];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    if (j < n1) {
      vg[j] = v[c];
      pred_row[j] = 0;
    }
  }
  __syncthreads();
  for (int e = tid; e < hwm; e += T)
    if (A[e].x >= 0) atomicAdd(&pred_row[A[e].y], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int j = 0; j < n1; ++j) {
      const int cnt = pred_row[j];
      colptr[j] = run;
      pred_arc_col[j] = run;  // write cursor of the column
      run += cnt;
    }
    colptr[n1] = run;
    if (err == 0 && run > out_cap) err = NIF_ERR_WORKSPACE;
    status[0] = err != 0 ? err : run;
    status[1] = n_aug;  // work counters: augmentations, reductions, rows relaxed
    status[2] = step;
    status[3] = n_expand;
#ifdef NIF_EMD_TIMING
    for (int q = 0; q < 6; ++q) status[4 + q] = (int)tk[q];
    status[10] = lvl;
#endif
    ctl[3] = err;
  }
  __syncthreads();
  if (ctl[3] != 0) return;
  // one ordered pass over the pool, so the arcs of a column come out in pool order on every run
  if (tid == 0) {
    for (int e = 0; e < hwm; ++e) {
      const int2 rc = A[e];
      if (rc.x < 0) continue;
      const int at = pred_arc_col[rc.y]++;
      rowidx[at] = rc.x;
      val[at] = eflow[e];
    }
  }
}

template <int CPT, bool BIG_LDS>
static void launch_emd(int threads, const EmdLds& L, hipStream_t st, const float* C, int64_t ldc, int n0, int n1,
                       const double* a, const double* b, double tol, double* u, double* v, unsigned char* base,
                       const EmdWs& w, int out_cap, int* colptr, int* rowidx, double* val, int* status) {
  auto kfn = emd_kernel<CPT, BIG_LDS>;
  if (L.bytes > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes);
  hipLaunchKernelGGL(kfn, dim3(1), dim3(threads), L.bytes, st, C, ldc, n0, n1, a, b, tol, u, v,
                     reinterpret_cast<const int*>(base + w.off_jbest), reinterpret_cast<int2*>(base + w.off_arc),
                     reinterpret_cast<double*>(base + w.off_eflow), reinterpret_cast<double*>(base + w.off_dem),
                     reinterpret_cast<double*>(base + w.off_sup), reinterpret_cast<double*>(base + w.off_rowd),
                     reinterpret_cast<int*>(base + w.off_newrows), BIG_LDS ? L.cap_lds : w.cap, L, out_cap, colptr,
                     rowidx, val, status);
}

// =====================================================================================================================
// Second solver (the default wherever it fits: n1 <= 2048 and the LDS plan below): the same successive shortest
// paths, with the support kept as a ROOTED FOREST instead of an arc pool.
//
// A vertex of the transport polytope is a forest over rows + columns, and every support arc is tight and carries
// flow, so it can be walked both ways at no cost: for the search, a tree of the forest is ONE node.  The kernel
// above finds "everything that hangs off this column" by a level-synchronous sweep over the arc pool in every
// Dijkstra step (48 % of its time at 384 x 384, 6.9 levels of two barriers per step) and relaxes the rows it reached
// eight at a time with all lanes on one row (21 %).  Here
//   * every node carries the label of its tree (comp = the tree's root), so "settle the tree of column j" is a label
//     comparison per column and per row -- no sweep, no list, no barrier;
//   * every WAVE owns all the columns (lane + 64 c, distances and predecessors in registers) and a share of the rows
//     (row i belongs to wave i mod NW): a step's rows are relaxed by all waves at once, each streaming its own cost
//     rows, and the true distance of a column is the minimum over the waves' partial ones -- which the argmin
//     reduction takes anyway, so the partial distances are never combined;
//   * the path inside a tree comes from parent pointers (entry node -> lowest common ancestor -> exit node), the flow of
//     an arc is stored at its child, an arc that runs empty cuts its child loose, a new arc re-roots the entered tree at
//     its entry column and hangs it under the row it was relaxed from; labels of the trees on the path are then rebuilt
//     by pointer jumping.  All of it is LDS work; one barrier per Dijkstra step.
// Duals, tolerances, the start and the plan that comes out follow the kernel above (same optimal cost; where the
// optimum is not unique the vertex may differ).
// =====================================================================================================================
struct __attribute__((aligned(16))) EmdNode {  // one ds_read_b128 per hop of a walk
  double flow;  // flow of the arc (node, parent)
  int parent;   // rows 0..n0-1, column j = n0 + j; -1: root
  int stamp;    // on a root: the search that settled its tree; on any node: tags of the path walks
};

struct EmdTreeLds {  // byte offsets into dynamic LDS
  int off_nodes, off_comp, off_entry, off_pred, off_dset, off_plist, off_cxy, off_red, off_v, off_cdist, off_dem;
  int big;  // 1: cdist and dem live in the workspace
  size_t bytes;
};

constexpr int EMD_PL = 4096;  // path records kept in LDS (64 per lane of wave 0); a longer path spills into the workspace

static EmdTreeLds emd_tree_lds(int n0, int n1, int cpl, bool big) {
  EmdTreeLds L;
  const int N = n0 + n1;
  const int n1p = 64 * cpl;  // one slot per lane and column group: v, the column labels and the demand bits are padded
                             // to it, so that slot (lane, c) is read at a constant offset from a per-lane address
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 15) & ~(size_t)15;
    return (int)at;
  };
  L.off_nodes = take((size_t)N * sizeof(EmdNode));
  L.off_v = take((size_t)n1p * 8);
  L.off_red = take(2 * 16 * 8 + 2 * 16 * 4 + 16 * 4 + 4 * 8);  // redv, redk, ctl, dctl
  L.big = big ? 1 : 0;
  L.off_cdist = big ? 0 : take((size_t)N * 8);
  L.off_dem = big ? 0 : take((size_t)n1 * 8);
  L.off_comp = take((size_t)(n0 + n1p) * 4);
  L.off_entry = take((size_t)N * 4);
  L.off_pred = take((size_t)n1 * 4);
  L.off_dset = take((size_t)(n1p / 32) * 4);
  L.off_plist = take((size_t)EMD_PL * 4);
  L.off_cxy = take(64 * 4);
  L.bytes = o;
  return L;
}

#ifdef NIF_EMD_TIMING  // cycles per phase into status[4..9], hop counts into [11..13] (status must hold 16 words)
#define EMD_COUNT(p) ++hops[p]
#else
#define EMD_COUNT(p) do {} while (0)
#endif

// Strided batch of independent problems of one size (nif_emd_solve_batch_f32): problem blockIdx.z reads / writes its own
// slice of every array.  All zero (and gridDim.z = 1) for a single problem.
struct EmdBatch {
  int64_t c = 0;        // cost matrices (floats)
  int64_t jbest = 0;    // the starting arcs (ints)
  int64_t ws = 0;       // per-problem workspace regions (bytes, a multiple of 256)
  const double* tol = nullptr;  // per-problem tolerance (device); null: the scalar argument
};

template <int CPL, int NW, bool BIG>
__global__ __launch_bounds__(NW * 64) void emd_tree_kernel(const float* __restrict__ C, int64_t ldc, int n0, int n1,
                                                           const double* __restrict__ a, const double* __restrict__ b,
                                                           double tol, double* __restrict__ u, double* __restrict__ vg,
                                                           const int* __restrict__ jbest, double* __restrict__ dem_g,
                                                           double* __restrict__ sup, double* __restrict__ cdist_g,
                                                           int* __restrict__ plist_g, int* __restrict__ order_g,
                                                           EmdTreeLds L, int one_lane,
                                                           int out_cap, int* __restrict__ colptr,
                                                           int* __restrict__ rowidx, double* __restrict__ val,
                                                           int* __restrict__ status, EmdBatch bt) {
  if (bt.tol) {  // a batched launch (uniform test; scalar address arithmetic only)
    const int64_t pz = blockIdx.z;
    C += pz * bt.c;
    a += pz * n0;
    b += pz * n1;
    u += pz * n0;
    vg += pz * n1;
    jbest += pz * bt.jbest;
    dem_g += pz * (bt.ws / 8);
    sup += pz * (bt.ws / 8);
    cdist_g += pz * (bt.ws / 8);
    plist_g += pz * (bt.ws / 4);
    order_g += pz * (bt.ws / 4);
    colptr += pz * (n1 + 1);
    rowidx += pz * out_cap;
    val += pz * out_cap;
    status += pz * 16;
    if (bt.tol) tol = bt.tol[pz];
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char emd_tsm[];
  const int N = n0 + n1;
  EmdNode* nodes = reinterpret_cast<EmdNode*>(emd_tsm + L.off_nodes);  // [N]
  double* vl = reinterpret_cast<double*>(emd_tsm + L.off_v);           // [n1] column duals
  double* redv = reinterpret_cast<double*>(emd_tsm + L.off_red);       // [2][16]
  unsigned* redk = reinterpret_cast<unsigned*>(redv + 32);             // [2][16]
  // 0: error, 1: tick after the augmentation, 2: rows relaxed, 3: trees on the path / path records, 4..6: relabel
  // round flags, 7: the path needs the one-lane code, 8: an arc was cut
  int* ctl = reinterpret_cast<int*>(redk + 32);
  double* dctl = reinterpret_cast<double*>(ctl + 16);  // 0: amount pushed by the last augmentation
  double* cdist = BIG ? cdist_g : reinterpret_cast<double*>(emd_tsm + L.off_cdist);  // [N] by root: distance of a settled tree
  double* dem = BIG ? dem_g : reinterpret_cast<double*>(emd_tsm + L.off_dem);        // [n1] demand left
  int* comp = reinterpret_cast<int*>(emd_tsm + L.off_comp);      // [N] root of the node's tree
  int* entry = reinterpret_cast<int*>(emd_tsm + L.off_entry);    // [N] by root: node through which the search entered the tree
  int* pred_row = reinterpret_cast<int*>(emd_tsm + L.off_pred);  // [n1] row a column was relaxed from (when it was chosen)
  unsigned* dset = reinterpret_cast<unsigned*>(emd_tsm + L.off_dset);  // columns with demand left
  int* plist = reinterpret_cast<int*>(emd_tsm + L.off_plist);  // path records: LCAP per lane of wave 0
  int* cxy = reinterpret_cast<int*>(emd_tsm + L.off_cxy);      // (entry node, exit node) of the trees on the path

  constexpr int T = NW * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- start: v from the caller's duals, every row ships along its tight arc (i, jbest_i) in row order ------------
  for (int k = tid; k < N; k += T) {
    EmdNode nd;
    nd.flow = 0.0;
    nd.parent = -1;
    nd.stamp = 0;
    nodes[k] = nd;
    entry[k] = (k < n0) ? jbest[k] : -1;  // (borrowed for the start)
  }
  for (int j = tid; j < 64 * CPL; j += T) {
    vl[j] = j < n1 ? vg[j] : 0.0;
    if (j >= n1) comp[n0 + j] = -2;  // (no tree)
  }
  for (int k = tid; k < 2 * CPL; k += T) dset[k] = 0u;
  if (tid < 16) ctl[tid] = 0;
  __syncthreads();
  for (int j = tid; j < n1; j += T) {
    double d = b[j];
    for (int i = 0; i < n0; ++i) {
      if (entry[i] != j) continue;
      double sl = a[i];
      const double x = fmin(sl, d);
      if (x > tol) {
        nodes[i].parent = n0 + j;
        nodes[i].flow = x;
        const double left = d - x;
        d = left > tol ? left : 0.0;
        sl -= x;
      }
      sup[i] = sl > tol ? sl : 0.0;
    }
    dem[j] = d;
    if (d > tol) atomicOr(&dset[j >> 5], 1u << (j & 31));
  }
  for (int i = tid; i < n0; i += T) {
    const int j = entry[i];
    if (j < 0 || j >= n1) {  // a cost row without a finite entry has no tight arc
      const double sl = a[i];
      sup[i] = sl > tol ? sl : 0.0;
    }
  }
  __syncthreads();
  for (int k = tid; k < N; k += T) {
    const int p = nodes[k].parent;
    comp[k] = p >= 0 ? p : k;  // stars: depth one
    entry[k] = -1;
  }
  __syncthreads();

  // the cost matrix as a buffer resource (range-checked loads)
  const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(C), 0, (int)((((unsigned)(n0 - 1)) * (unsigned)ldc + (unsigned)n1) * 4u), 0x00020000);
  unsigned invalid = 0;  // column slots of this lane beyond n1
#pragma unroll
  for (int c = 0; c < CPL; ++c)
    if (lane + 64 * c >= n1) invalid |= 1u << c;

  int tick = 0, step = 0, n_aug = 0, n_rel = 0;
#ifdef NIF_EMD_TIMING
  long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk0 = wall_clock64();
  int hops[3] = {0, 0, 0};
#endif
  long budget = 64L * N + 1024;  // augmentations; far above the n0 + n1 a non-degenerate problem needs
  bool stop = false;
  // ---- the source rows in ascending order of their surplus (ties: by row).  Small surpluses first means the early
  // searches end in the first or second tree and the large ones run when most demand is already placed: the same
  // augmentations within a few per cent, but half the Dijkstra steps at 1024 x 1024 (tools/emd_order_probe.py: 9281 ->
  // 4898, ascending row numbers -> ascending surplus).  Rank by counting, the surpluses staged through LDS a chunk at a
  // time; the list itself lives in the workspace and is read 64 entries at a time.
  int nroots = 0;
  {
    double* stage = reinterpret_cast<double*>(plist);  // EMD_PL / 2 doubles
    constexpr int CH = EMD_PL / 2;
    int rank[8];   // ranks of the rows tid, tid + T, ... (n0 <= NIF_EMD_MAX_N = 8192 >= 8 T for T >= 256... see the loop)
    for (int base = 0; base < n0; base += 8 * T) {
      double mine[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = base + tid + q * T;
        mine[q] = i < n0 ? sup[i] : 0.0;
        rank[q] = 0;
      }
      int total = 0;
      for (int c0 = 0; c0 < n0; c0 += CH) {
        __syncthreads();
        for (int k = tid; k < CH && c0 + k < n0; k += T) stage[k] = sup[c0 + k];
        __syncthreads();
        const int cn = min(CH, n0 - c0);
        for (int k = 0; k < cn; ++k) {
          const double sk = stage[k];  // (one address for the whole wave: a broadcast read)
          if (sk > tol) {
            ++total;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int i = base + tid + q * T;
              rank[q] += (sk < mine[q] || (sk == mine[q] && c0 + k < i)) ? 1 : 0;
            }
          }
        }
      }
      nroots = total;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = base + tid + q * T;
        if (i < n0 && mine[q] > tol) order_g[rank[q]] = i;
      }
    }
    __syncthreads();
  }
  for (int g0 = 0; g0 < nroots && !stop; g0 += 64) {
    const int gi = g0 + lane;
    const int myroot = gi < nroots ? order_g[gi] : 0;
    const double sv = gi < nroots ? sup[myroot] : 0.0;
    unsigned long long roots = __ballot(sv > tol);
    while (roots && !stop) {
      const int bit = __ffsll((long long)roots) - 1;
      roots &= roots - 1;
      const int i0 = __shfl(myroot, bit, 64);
      double r = __shfl(sv, bit, 64);
      while (r > tol) {
        if (--budget < 0 || ctl[0] != 0) {
          stop = true;
          break;
        }
        // ---- one search: Dijkstra over the trees of the forest ----------------------------------------------------
        unsigned smask = invalid, defm = 0;
        double dist[CPL];
        int pred[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          dist[c] = ((invalid >> c) & 1u) ? 0.0 : INFINITY;
          pred[c] = -1;
          if ((dset[2 * c + (lane >> 5)] >> (lane & 31)) & 1u) defm |= 1u << c;  // column lane + 64 c
        }
        if (tick > (1 << 30)) {  // (uniform) the tags would wrap: start them over
          __syncthreads();
          for (int k = tid; k < N; k += T) nodes[k].stamp = 0;
          tick = 0;
          __syncthreads();
        }
        const int s = ++tick;
        int K = comp[i0], e = i0, sink = -1;
        double D = 0.0;
        for (int it = 0; it <= N; ++it) {
          if (tid == 0) {
            nodes[K].stamp = s;
            cdist[K] = D;
            entry[K] = e;
          }
          // columns of tree K: settled at distance D (labels read without a guard: a guarded LDS read per column
          // makes the reads wait for one another)
          int lab[CPL];
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            lab[c] = comp[n0 + lane + 64 * c];
          }
          unsigned newly = 0;
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            if (lab[c] == K) {
              newly |= 1u << c;
            }
          }
          newly &= ~smask;
#pragma unroll
          for (int c = 0; c < CPL; ++c)
            if ((newly >> c) & 1u) dist[c] = D;
          smask |= newly;
          const unsigned hit = newly & defm;
          const unsigned cand = wave_min_u32(hit ? (unsigned)(lane + 64 * (__ffs((int)hit) - 1)) : 0xffffffffu);
          if (cand != 0xffffffffu) {  // the tree holds a column with demand left: the search ends inside it
            sink = (int)cand;
            break;
          }
          EMD_MARK(0);
          // rows of tree K that belong to this wave (row i: wave i mod NW): their cost rows are relaxed RB at a time, all
          // loads of a batch in flight together
          constexpr int RB = CPL <= 8 ? 4 : CPL <= 16 ? 2 : 1;  // (what the register file of NW waves leaves room for)
          for (int gb = 0; gb < n0; gb += 64 * NW) {
            const int i = gb + wave + NW * lane;
            unsigned long long m = __ballot(comp[i < n0 ? i : n0 - 1] == K && i < n0);
            n_rel += __popcll(m);
            while (m) {
              int ri[RB];
              int nb = 0;  // rows in this batch (wave-uniform, like everything about m)
#pragma unroll
              for (int q = 0; q < RB; ++q) {
                ri[q] = 0;
                if (m) {
                  ri[q] = gb + wave + NW * (__ffsll((long long)m) - 1);
                  m &= m - 1;
                  ++nb;
                }
              }
              float cv[RB][CPL];
              double ub[RB];
#pragma unroll
              for (int q = 0; q < RB; ++q) {
                if (q < nb) {
                  ub[q] = u[ri[q]];
                  // bounds-checked buffer loads: slot (lane, c) of the row at a constant offset from one per-lane
                  // address; slots beyond the matrix read 0 (beyond n1 but inside it: the next row), and are never used
                  const int rowoff = (int)((unsigned)ri[q] * (unsigned)ldc * 4u);
#pragma unroll
                  for (int c = 0; c < CPL; ++c)
                    cv[q][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crsrc, lane * 4 + 256 * c, rowoff, 0));
                }
              }
              double vj[CPL];
#pragma unroll
              for (int c = 0; c < CPL; ++c) vj[c] = vl[lane + 64 * c];
#pragma unroll
              for (int q = 0; q < RB; ++q) {
                if (q < nb) {
                  const double base = D - ub[q];
                  // in stages over the columns, so that independent operations follow one another (one column at a time
                  // is a chain of seven dependent instructions)
                  double rr[CPL];
#pragma unroll
                  for (int c = 0; c < CPL; ++c) rr[c] = (double)cv[q][c] - vj[c];
#pragma unroll
                  for (int c = 0; c < CPL; ++c) rr[c] += base;
#pragma unroll
                  for (int c = 0; c < CPL; ++c) {
                    // settled and out-of-range slots hold a distance <= D <= rr: they never take the update
                    const bool better = rr[c] < dist[c];
                    dist[c] = better ? rr[c] : dist[c];
                    pred[c] = better ? ri[q] : pred[c];
                  }
                }
              }
            }
          }
          EMD_MARK(1);
          // nearest unsettled column over all waves' partial distances
          double bvl = INFINITY;
          unsigned bk = 0xffffffffu;
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            if (!((smask >> c) & 1u)) {
              const unsigned key = (((defm >> c) & 1u) ? 0u : 0x80000000u) | (unsigned)(lane + 64 * c);
              if (dist[c] < bvl || (dist[c] == bvl && key < bk)) {
                bvl = dist[c];
                bk = key;
              }
            }
          }
          const double wmin = wave_min_f64(bvl);
          const unsigned wkey = wave_min_u32((bvl == wmin && bvl < INFINITY) ? bk : 0xffffffffu);
          const int pb = (step & 1) * 16;
          ++step;
          if (lane == 0) {
            redv[pb + wave] = wmin;
            redk[pb + wave] = wkey;
          }
          EMD_MARK(2);
          __syncthreads();
          EMD_MARK(4);
          double rv[NW];
          unsigned rk[NW];
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            rv[w] = redv[pb + w];
            rk[w] = redk[pb + w];
          }
          double bv = rv[0];
          unsigned gkey = rk[0];
          int wwin = 0;
#pragma unroll
          for (int w = 1; w < NW; ++w) {
            if (rv[w] < bv || (rv[w] == bv && rk[w] < gkey)) {
              bv = rv[w];
              gkey = rk[w];
              wwin = w;
            }
          }
          if (gkey == 0xffffffffu || !(bv < INFINITY)) break;  // nothing reachable is left
          const int bj = (int)(gkey & 0x7fffffffu);
          if (wave == wwin && lane == (bj & 63)) {
            int pv = -1;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
              if (c == (bj >> 6)) pv = pred[c];
            pred_row[bj] = pv;
          }
          D = bv;
          e = n0 + bj;
          K = comp[e];
          EMD_MARK(2);
          if (!(gkey & 0x80000000u)) {  // it has demand left: the path ends at this column, its tree stays unsettled
            if (tid == 0) entry[K] = e;
            sink = bj;
            break;
          }
        }
        __syncthreads();
        EMD_MARK(0);
        if (sink < 0) {
          r = 0.0;  // rounding leftovers with every demand met: this source is done
          break;
        }
        // ---- duals: settled columns move down, reached rows move up, so the support stays tight -------------------
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int j = lane + 64 * c;
          if ((c % NW) == wave && ((smask & ~invalid) >> c) & 1u) vl[j] += dist[c] - D;
        }
        for (int k = tid; k < n0; k += T) {
          const int kc = comp[k];
          if (nodes[kc].stamp == s) u[k] += D - cdist[kc];
        }
        __syncthreads();
        EMD_MARK(3);
        // ---- augmentation ---------------------------------------------------------------------------------------------
        // The path crosses m trees; inside tree t it runs from the entry node x_t to the exit node y_t along tree arcs
        // (up to their lowest common ancestor and down again).  Fast path, wave 0: lane 0 lists the (x_t, y_t); lanes
        // 2t / 2t+1 climb from x_t / y_t in lock step, tagging what they pass with their position, until one steps on
        // the other's tag (or both step on the same node): max-depth steps for ALL trees at once instead of the sum of
        // the walks; then every lane prices and updates its own arcs, and lane t re-roots tree t at x_t and hangs it
        // under the row it was relaxed from.  More than 32 trees or more than LCAP arcs on one side: one lane does it.
        constexpr int LCAP = EMD_PL / 64;
        if (tid == 0) {
          const int K0 = comp[i0];
          int m = 0, slow = 0;
          for (int y = n0 + sink;;) {
            const int Kc = comp[y];
            const int x = entry[Kc];
            nodes[Kc].stamp = tick + 1;  // a tag of this augmentation on the root: the tree is relabelled afterwards
            if (m < 32) {
              cxy[2 * m] = x;
              cxy[2 * m + 1] = y;
            }
            ++m;
            if (Kc == K0 || x < n0) break;  // the tree of the source row
            y = pred_row[x - n0];
            if (y < 0 || m > N) {
              ctl[0] = NIF_ERR_UNSUPPORTED;  // cannot happen: every entered tree was relaxed from a row
              break;
            }
          }
          if (m > 32 || one_lane) slow = 1;
          ctl[3] = m;
          ctl[7] = slow;
          ctl[8] = 0;
        }
        __syncthreads();
        EMD_MARK(6);
        if (wave == 0 && ctl[7] == 0 && ctl[0] == 0) {
          const int m = ctl[3];
          const int t = lane >> 1, side = lane & 1;
          const bool act = t < m;
          const int base = tick + 2 + 256 * t;
          int q = act ? cxy[lane] : -1;
          int cnt = 0, valid = -1, partner_valid = -1;
          bool walking = act, over = false;
          for (int itn = 0; itn <= LCAP + 1; ++itn) {
            const EmdNode nd = nodes[walking ? q : 0];  // (read before it is known to be needed: one LDS round trip
            const int oq = __shfl_xor(q, 1, 64);        //  per step together with the two exchanges)
            const bool owalking = __shfl_xor((int)walking, 1, 64) != 0;
            if (walking && owalking && q == oq) {  // both arrive here now: the lowest common ancestor
              walking = false;
              valid = cnt;
            }
            if (!__any(walking)) break;
            if (walking) {
              EMD_COUNT(0);
              const int st = nd.stamp - base;
              if (st >= 0 && st < 256 && (st & 1) != side) {  // the partner has been here, after st / 2 arcs
                partner_valid = st >> 1;
                valid = cnt;
                walking = false;
              } else if (cnt >= LCAP) {
                over = true;
                walking = false;
              } else {
                nodes[q].stamp = base + 2 * cnt + side;
                plist[lane * LCAP + cnt] = q;
                ++cnt;
                if (nd.parent < 0) walking = false;  // at the root: the partner will find the tags
                else q = nd.parent;
              }
            }
          }
          const int pv = __shfl_xor(partner_valid, 1, 64);
          if (valid < 0 && pv >= 0) valid = pv;
          if (act && valid < 0) over = true;
          if (__any(over)) {
            if (lane == 0) ctl[7] = 1;
          } else {
            if (valid < 0) valid = 0;
            // the arcs of all lists, spread over the 64 lanes: lane j takes list j mod P, entries j / P, j / P + 64 / P, ..
            // Side 0 walked x -> ancestor along the path (arc child -> parent): a column's arc is walked backwards;
            // side 1 walked y -> ancestor against it: a row's arc is.
            int lp = 1;
            while ((1 << lp) < 2 * m) ++lp;
            const int l = lane & ((1 << lp) - 1), kstep = 64 >> lp;
            const int lv = __shfl(valid, l, 64);
            const int lside = l & 1;
            double lmin = INFINITY;
            for (int k = lane >> lp; __any(k < lv); k += kstep) {
              if (k < lv) {
                const int nq = plist[l * LCAP + k];
                const bool back = lside ? (nq < n0) : (nq >= n0);
                if (back) lmin = fmin(lmin, nodes[nq].flow);
              }
            }
            if (lane == 0) lmin = fmin(lmin, fmin(r, dem[sink]));
            const double delta = wave_min_f64(lmin);
            int cut = 0;
            for (int k = lane >> lp; __any(k < lv); k += kstep) {
              if (k < lv) {
                EMD_COUNT(1);
                const int nq = plist[l * LCAP + k];
                const bool back = lside ? (nq < n0) : (nq >= n0);
                if (back) {
                  const double f = nodes[nq].flow - delta;
                  nodes[nq].flow = f;
                  if (f <= tol) {  // the arc leaves the support: nq's subtree is a tree of its own
                    nodes[nq].parent = -1;
                    cut = 1;
                  }
                } else {
                  nodes[nq].flow += delta;
                }
              }
            }
            if (__any(cut) && lane == 0) ctl[8] = 1;
            if (lane < m) {  // new arcs: the entered tree is re-rooted at its entry column and hung under the row
              const int x = cxy[2 * lane];
              if (x >= n0) {
                int prev = -1, guard = N + 1;
                double pf = 0.0;
                for (int nq = x; nq >= 0 && --guard > 0;) {
                  EMD_COUNT(2);
                  const EmdNode nd = nodes[nq];
                  nodes[nq].parent = prev;
                  nodes[nq].flow = pf;
                  prev = nq;
                  pf = nd.flow;
                  nq = nd.parent;
                }
                nodes[x].parent = pred_row[x - n0];
                nodes[x].flow = delta;
              }
            }
            if (lane == 0) {
              double left = dem[sink] - delta;
              if (left <= tol) {
                left = 0.0;
                dset[sink >> 5] &= ~(1u << (sink & 31));
              }
              dem[sink] = left;
              dctl[0] = delta;
              ctl[1] = tick + 2 + 256 * m;
            }
          }
        }
        EMD_MARK(7);
        __syncthreads();
        if (ctl[7] != 0 && ctl[0] == 0) {  // the general path, one lane
          auto put = [&](int idx, int rec) {
            if (idx < EMD_PL) plist[idx] = rec;
            else plist_g[idx - EMD_PL] = rec;
          };
          auto get = [&](int idx) { return idx < EMD_PL ? plist[idx] : plist_g[idx - EMD_PL]; };
          if (tid == 0) {
            const int K0 = comp[i0];
            int tk = tick + 2 + 256 * 33, np = 0;
            double delta = fmin(r, dem[sink]);
            // every walk is over a forest; `guard` only keeps a corrupted structure from spinning (it cannot run out
            // otherwise: a path visits a node once, and the trees on it are disjoint)
            int guard = 4 * N + 16;
            for (int y = n0 + sink; guard > 0;) {
              const int Kc = comp[y];
              const int x = entry[Kc];
              const int tag = ++tk;
              for (int q = x; q >= 0 && --guard > 0;) {  // x -> root
                const int up = nodes[q].parent;
                nodes[q].stamp = tag;
                q = up;
              }
              int q = y;
              while (q >= 0 && --guard > 0) {  // y -> lowest common ancestor: arcs walked against the path
                const EmdNode nd = nodes[q];
                if (nd.stamp == tag) break;
                const bool back = q < n0;  // column -> row takes flow off the arc
                if (back) delta = fmin(delta, nd.flow);
                put(np++, (q << 1) | (back ? 1 : 0));
                q = nd.parent;
              }
              if (q < 0) guard = 0;  // y and its entry node in different trees: cannot happen
              const int lca = q;
              for (q = x; q != lca && --guard > 0;) {  // x -> lowest common ancestor: arcs child -> parent
                const EmdNode nd = nodes[q];
                const bool back = q >= n0;
                if (back) delta = fmin(delta, nd.flow);
                put(np++, (q << 1) | (back ? 1 : 0));
                q = nd.parent;
              }
              if (Kc == K0 || x < n0) break;
              y = pred_row[x - n0];
              if (y < 0) guard = 0;
            }
            if (guard <= 0) {
              ctl[0] = NIF_ERR_UNSUPPORTED;
              np = 0;
              delta = 0.0;
            }
            ctl[1] = tk;
            ctl[3] = np;
            ctl[8] = 1;  // (treated as if an arc had been cut: full relabelling)
            dctl[0] = delta;
          }
          __syncthreads();
          {
            const int np = ctl[3];
            const double delta = dctl[0];
            for (int k = tid; k < np; k += T) {
              const int rec = get(k);
              const int q = rec >> 1;
              if (rec & 1) {
                const double f = nodes[q].flow - delta;
                nodes[q].flow = f;
                if (f <= tol) nodes[q].parent = -1;
              } else {
                nodes[q].flow += delta;
              }
            }
          }
          __syncthreads();
          if (tid == 0 && ctl[0] == 0) {
            const double delta = dctl[0];
            int guard = 4 * N + 16;
            for (int y = n0 + sink; guard > 0;) {
              const int x = entry[comp[y]];
              if (x < n0) break;
              const int i = pred_row[x - n0];
              int prev = -1;
              double pf = 0.0;
              for (int q = x; q >= 0 && --guard > 0;) {
                const EmdNode nd = nodes[q];
                nodes[q].parent = prev;
                nodes[q].flow = pf;
                prev = q;
                pf = nd.flow;
                q = nd.parent;
              }
              nodes[x].parent = i;
              nodes[x].flow = delta;
              y = i;
            }
            double left = dem[sink] - delta;
            if (left <= tol) {
              left = 0.0;
              dset[sink >> 5] &= ~(1u << (sink & 31));
            }
            dem[sink] = left;
          }
          __syncthreads();
        }
        EMD_MARK(4);
        if (ctl[0] != 0) {
          stop = true;
          break;
        }
        ++n_aug;
        r -= dctl[0];
        // ---- labels of the trees on the path (their old roots carry a tag of this augmentation).  No arc cut: they
        // have all become part of the source row's tree.  Otherwise every node of them starts from its parent and
        // jumps three ancestors up per round until all of them name a root -------------------------------------------
        {
          const int t0 = tick;
          tick = ctl[1];
          const bool cut = ctl[8] != 0;
          const int K0 = comp[i0];
          unsigned aff = 0;
          int q = 0;
          for (int k = tid; k < N; k += T, ++q) {
            if (nodes[comp[k]].stamp > t0) {
              aff |= 1u << q;
              if (cut) {
                const int p = nodes[k].parent;
                comp[k] = p >= 0 ? p : k;  // (nobody else reads comp[k] in this pass)
              }
            }
          }
          if (tid == 0) ctl[4] = 0;
          __syncthreads();
          if (!cut) {
            q = 0;
            for (int k = tid; k < N; k += T, ++q)
              if ((aff >> q) & 1u) comp[k] = K0;
            __syncthreads();
          } else {
            for (int round = 0; round < 40; ++round) {  // the depth shrinks fourfold per round
              int changed = 0;
              q = 0;
              for (int k = tid; k < N; k += T, ++q) {
                if ((aff >> q) & 1u) {
                  const int c1 = comp[k];
                  const int c2 = comp[c1];
                  const int c3 = comp[c2];
                  const int c4 = comp[c3];
                  if (c4 != c1) {
                    comp[k] = c4;
                    changed = 1;
                  }
                }
              }
              if (changed) ctl[4 + round % 3] = 1;
              if (tid == 0) ctl[4 + (round + 1) % 3] = 0;
              __syncthreads();
              if (!ctl[4 + round % 3]) break;
            }
          }
        }
        EMD_MARK(5);
      }
    }
  }

  // ---- emit the plan column by column (CSC, rows ascending inside a column) and the duals ------------------------
  __syncthreads();
  int err = ctl[0];
  if (budget < 0 && err == 0) err = NIF_ERR_UNSUPPORTED;
  if (lane == 0) atomicAdd(&ctl[2], n_rel);
  for (int j = tid; j < n1; j += T) {
    vg[j] = vl[j];
    pred_row[j] = 0;
  }
  __syncthreads();
  for (int k = tid; k < N; k += T) {
    const int p = nodes[k].parent;
    if (p >= 0) atomicAdd(&pred_row[(k < n0 ? p : k) - n0], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int j = 0; j < n1; ++j) {
      const int cnt = pred_row[j];
      colptr[j] = run;
      entry[j] = run;  // write cursor of the column
      run += cnt;
    }
    colptr[n1] = run;
    if (err == 0 && run > out_cap) err = NIF_ERR_WORKSPACE;
    status[0] = err != 0 ? err : run;
    status[1] = n_aug;  // work counters: augmentations, Dijkstra steps, rows relaxed
    status[2] = step;
    status[3] = ctl[2];
#ifdef NIF_EMD_TIMING
    for (int q = 0; q < 6; ++q) status[4 + q] = (int)tk[q];
    status[10] = (int)tk[6];
    status[11] = (int)tk[7];
    for (int q = 0; q < 3; ++q) status[12 + q] = hops[q];
#endif
    ctl[0] = err;
  }
  __syncthreads();
  if (ctl[0] != 0) return;
  for (int k = tid; k < N; k += T) {
    const EmdNode nd = nodes[k];
    if (nd.parent < 0) continue;
    const int row = k < n0 ? k : nd.parent, col = (k < n0 ? nd.parent : k) - n0;
    const int at = atomicAdd(&entry[col], 1);
    rowidx[at] = row;
    val[at] = nd.flow;
  }
  __syncthreads();
  for (int j = tid; j < n1; j += T) {  // insertion sort of the column's few arcs by row: the same plan on every run
    const int lo = colptr[j], hi = lo + pred_row[j];
    for (int p = lo + 1; p < hi; ++p) {
      const int rr = rowidx[p];
      const double vv = val[p];
      int q = p - 1;
      for (; q >= lo && rowidx[q] > rr; --q) {
        rowidx[q + 1] = rowidx[q];
        val[q + 1] = val[q];
      }
      rowidx[q + 1] = rr;
      val[q + 1] = vv;
    }
  }
}

template <int CPL, int NW, bool BIG>
static void launch_emd_tree_v(const EmdTreeLds& L, hipStream_t st, const float* C, int64_t ldc, int n0, int n1,
                              const double* a, const double* b, double tol, double* u, double* v, unsigned char* base,
                              const EmdWs& w, int one_lane, int out_cap, int* colptr, int* rowidx, double* val,
                              int* status, int batch, const EmdBatch& bt, const int* jbest) {
  auto kfn = emd_tree_kernel<CPL, NW, BIG>;
  static PerDeviceOnce attr_once;
  attr_once.run([&] {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  });
  hipLaunchKernelGGL(kfn, dim3(1, 1, batch), dim3(NW * 64), L.bytes, st, C, ldc, n0, n1, a, b, tol, u, v,
                     jbest ? jbest : reinterpret_cast<const int*>(base + w.off_jbest), reinterpret_cast<double*>(base + w.off_dem),
                     reinterpret_cast<double*>(base + w.off_sup), reinterpret_cast<double*>(base + w.off_eflow),
                     reinterpret_cast<int*>(base + w.off_arc), reinterpret_cast<int*>(base + w.off_newrows), L, one_lane,
                     out_cap, colptr, rowidx, val, status, bt);
}

template <int CPL, int NW>
static void launch_emd_tree(const EmdTreeLds& L, hipStream_t st, const float* C, int64_t ldc, int n0, int n1,
                            const double* a, const double* b, double tol, double* u, double* v, unsigned char* base,
                            const EmdWs& w, int one_lane, int out_cap, int* colptr, int* rowidx, double* val,
                            int* status, int batch = 1, const EmdBatch& bt = EmdBatch(), const int* jbest = nullptr) {
  if (L.big)
    launch_emd_tree_v<CPL, NW, true>(L, st, C, ldc, n0, n1, a, b, tol, u, v, base, w, one_lane, out_cap, colptr, rowidx,
                                     val, status, batch, bt, jbest);
  else
    launch_emd_tree_v<CPL, NW, false>(L, st, C, ldc, n0, n1, a, b, tol, u, v, base, w, one_lane, out_cap, colptr,
                                      rowidx, val, status, batch, bt, jbest);
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
                                 double tol, int flags, int32_t* colptr, int32_t* rowidx, double* val,
                                 int32_t out_cap, int32_t* status, double* u, double* v, void* ws, size_t ws_bytes,
                                 void* stream) {
  nif::Range nif_range_("nif_emd_solve_f32");
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
  const bool warm = w.lap_bytes != 0 && n0 >= 2 && !(flags & NIF_EMD_COLD_START) && !env_int("NIF_EMD_COLD_START");
  if (warm) {
    // Square problem: start from the ASSIGNMENT solution of the same costs (uniform marginals) -- its duals are
    // feasible for any marginals and tight on a perfect matching, so the greedy shipment of the kernel sends
    // min(a_i, b_j) along every matched pair and only the differences |a_i - b_j| are left to the augmenting paths
    // (at most one row or column saturates per augmentation: ~n / 2 of them instead of the ~3 n the column-minimum /
    // row-minimum start needs, and the searches are short because the duals are already near the optimum's).
    const int rc = nif_lap_solve_f32(C, ldc, n0, 0, reinterpret_cast<int32_t*>(base + w.off_jbest),
                                     reinterpret_cast<int32_t*>(base + w.off_r4c), uu, vv, base + w.off_lap, w.lap_bytes,
                                     stream);
    if (rc != NIF_OK) return rc;
  } else {
    hipLaunchKernelGGL(emd_colmin_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, C, ldc, n0, n1, vv);
    hipLaunchKernelGGL(emd_rowmin_kernel, dim3((n0 + 3) / 4), dim3(256), 0, st, C, ldc, n0, n1, vv, uu,
                       reinterpret_cast<int*>(base + w.off_jbest));
  }
  // the forest solver wherever its register and LDS plan fits (NIF_EMD_LEGACY / NIF_EMD_ARC_POOL: the arc-pool kernel)
  static const int legacy = env_int("NIF_EMD_LEGACY");
  if (!legacy && !(flags & (NIF_EMD_ARC_POOL | NIF_EMD_EDGES_GLOBAL)) && n1 <= 2048 &&
      (size_t)w.cap * sizeof(double) >= (size_t)(n0 + n1) * sizeof(double)) {
    const int cpl = (n1 + 63) / 64;
    const int cplp = cpl <= 2 ? 2 : cpl <= 4 ? 4 : cpl <= 8 ? 8 : cpl <= 16 ? 16 : 32;
    EmdTreeLds TL = emd_tree_lds(n0, n1, cplp, false);
    if (TL.bytes > 158 * 1024) TL = emd_tree_lds(n0, n1, cplp, true);
    if (TL.bytes <= 158 * 1024 && ((uint64_t)(n0 - 1) * (uint64_t)ldc + (uint64_t)n1) * 4 < 0xffffffffull) {
#define NIF_EMD_TREE(CPL, NW)                                                                                          \
  launch_emd_tree<CPL, NW>(TL, st, C, ldc, n0, n1, a, b, tol, uu, vv, base, w, (flags & NIF_EMD_ONE_LANE_PATHS) ? 1 : 0,   \
                           out_cap, colptr, rowidx, val, status)
      if (cpl <= 2) NIF_EMD_TREE(2, 8);
      else if (cpl <= 4) NIF_EMD_TREE(4, 8);
      else if (cpl <= 8) NIF_EMD_TREE(8, 8);
      else if (cpl <= 16) NIF_EMD_TREE(16, 8);
      else NIF_EMD_TREE(32, 4);
#undef NIF_EMD_TREE
      return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
    }
  }
  int cpt = 1;
  while (cpt < 8 && (n1 + cpt - 1) / cpt > 1024) cpt *= 2;
  const int threads = ((n1 + cpt - 1) / cpt + 63) / 64 * 64;
  if (threads > 1024) return NIF_ERR_UNSUPPORTED;
  // arc records in LDS when they fit (a forest has < n0 + n1 arcs); the caller retries with
  // NIF_EMD_EDGES_GLOBAL if a degenerate problem outgrows that
  bool in_lds = !(flags & NIF_EMD_EDGES_GLOBAL);
  EmdLds L = emd_lds(n0, n1, in_lds);
  if (in_lds && L.bytes > 150 * 1024) {
    in_lds = false;
    L = emd_lds(n0, n1, false);
  }
  if (L.bytes > 160 * 1024) return NIF_ERR_UNSUPPORTED;
#define NIF_EMD_GO(CPT)                                                                                              \
  do {                                                                                                               \
    if (in_lds) launch_emd<CPT, true>(threads, L, st, C, ldc, n0, n1, a, b, tol, uu, vv, base, w, out_cap, colptr,   \
                                      rowidx, val, status);                                                          \
    else launch_emd<CPT, false>(threads, L, st, C, ldc, n0, n1, a, b, tol, uu, vv, base, w, out_cap, colptr, rowidx, \
                                val, status);                                                                        \
  } while (0)
  switch (cpt) {
    case 1: NIF_EMD_GO(1); break;
    case 2: NIF_EMD_GO(2); break;
    case 4: NIF_EMD_GO(4); break;
    default: NIF_EMD_GO(8); break;
  }
#undef NIF_EMD_GO
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

namespace nif {
struct EmdBatchWs {
  size_t off_jbest, off_r4c, off_lap, total;
};
static EmdBatchWs emd_batch_ws(int n, int batch) {
  const EmdWs w = emd_ws(n, n);
  EmdBatchWs bw;
  size_t o = w.total * (size_t)batch;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 255) & ~(size_t)255;
    return at;
  };
  bw.off_jbest = take((size_t)batch * n * sizeof(int));
  bw.off_r4c = take((size_t)batch * n * sizeof(int));
  bw.off_lap = take(nif_lap_batch_workspace_bytes(n, batch));
  bw.total = o;
  return bw;
}
}  // namespace nif

extern "C" size_t nif_emd_batch_workspace_bytes(int n, int batch) {
  if (n <= 0 || n > NIF_EMD_MAX_N || batch <= 0 || batch > 65535) return 0;
  return nif::emd_batch_ws(n, batch).total;
}

extern "C" int nif_emd_solve_batch_f32(const float* C, int64_t ldc, int64_t stride_c, int n, int batch, const double* a,
                                       const double* b, const double* tol, int flags, int32_t* colptr, int32_t* rowidx,
                                       double* val, int32_t out_cap, int32_t* status, double* u, double* v, void* ws,
                                       size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_emd_solve_batch_f32");
  using namespace nif;
  if (!C || !a || !b || !tol || !colptr || !rowidx || !val || !status || !u || !v || n <= 0 || ldc < n || out_cap <= 0 ||
      batch <= 0 || batch > 65535)
    return NIF_ERR_INVALID_ARG;
  if (batch > 1 && stride_c < (int64_t)(n - 1) * ldc + n) return NIF_ERR_INVALID_ARG;
  if (n > NIF_EMD_MAX_N) return NIF_ERR_UNSUPPORTED;
  // the forest kernel from the assignment start only (what nif_emd_solve_f32 runs by default on square problems that
  // fit it); anything else -- wider problems, the tuning flags -- goes through nif_emd_solve_f32 problem by problem
  if ((flags & ~NIF_EMD_ONE_LANE_PATHS) != 0 || n < 2 || n > 2048 || env_int("NIF_EMD_LEGACY") || env_int("NIF_EMD_COLD_START"))
    return NIF_ERR_UNSUPPORTED;
  const EmdWs w = emd_ws(n, n);
  const EmdBatchWs bw = emd_batch_ws(n, batch);
  if (!ws || ws_bytes < bw.total) return NIF_ERR_WORKSPACE;
  const int cpl = (n + 63) / 64;
  const int cplp = cpl <= 2 ? 2 : cpl <= 4 ? 4 : cpl <= 8 ? 8 : cpl <= 16 ? 16 : 32;
  EmdTreeLds TL = emd_tree_lds(n, n, cplp, false);
  if (TL.bytes > 158 * 1024) TL = emd_tree_lds(n, n, cplp, true);
  if (TL.bytes > 158 * 1024 || ((uint64_t)(n - 1) * (uint64_t)ldc + (uint64_t)n) * 4 >= 0xffffffffull) return NIF_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned char* base = reinterpret_cast<unsigned char*>(ws);
  int32_t* jb = reinterpret_cast<int32_t*>(base + bw.off_jbest);
  clear_err();
  // the assignment start of every problem in one batched call: col4row = the tight arc each row ships along
  const int rc = nif_lap_solve_batch_f32(C, ldc, stride_c, n, batch, 0, jb, reinterpret_cast<int32_t*>(base + bw.off_r4c), u, v,
                                         base + bw.off_lap, nif_lap_batch_workspace_bytes(n, batch), stream);
  if (rc != NIF_OK) return rc;
  EmdBatch bt;
  bt.c = stride_c;
  bt.jbest = n;
  bt.ws = (int64_t)w.total;
  bt.tol = tol;
#define NIF_EMD_TREE(CPL, NW)                                                                                          \
  launch_emd_tree<CPL, NW>(TL, st, C, ldc, n, n, a, b, 0.0, u, v, base, w, (flags & NIF_EMD_ONE_LANE_PATHS) ? 1 : 0,     \
                           out_cap, colptr, rowidx, val, status, batch, bt, jb)
  if (cpl <= 2) NIF_EMD_TREE(2, 8);
  else if (cpl <= 4) NIF_EMD_TREE(4, 8);
  else if (cpl <= 8) NIF_EMD_TREE(8, 8);
  else if (cpl <= 16) NIF_EMD_TREE(16, 8);
  else NIF_EMD_TREE(32, 4);
#undef NIF_EMD_TREE
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_refit_sparse(const float* W0, const float* W1, int64_t ldw1, const int32_t* rptr, const int32_t* ri,
                                const float* rv, const int32_t* cptr, const int32_t* ci, const float* cv, float alpha,
                                float beta, int n_out, int n_in, int sp, float* out, void* stream) {
  nif::Range nif_range_("nif_refit_sparse");
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
