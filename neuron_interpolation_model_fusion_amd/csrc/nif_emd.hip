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
