// nif_lap.hip -- exact linear assignment (min-cost perfect matching) on gfx950.
//
// Two phases, both on the device, no host round trip:
//  1. parallel dual initialisation over the whole chip (HBM-bound, two passes over C):
//       v[j] = min_i C[i][j];  u[i] = min_j (C[i][j] - v[j]);  row i claims its argmin
//       column, the lowest row wins a contested column.  This leaves a dual-feasible
//       (u[i] + v[j] <= C[i][j]) partial matching on tight edges.
//  2. shortest augmenting paths for the rows left free (Jonker-Volgenant / the scheme
//       scipy's rectangular_lsap follows), one workgroup: every thread owns up to 8
//       columns (distance, dual v, scanned bit in registers), each Dijkstra step is one
//       coalesced read of a cost row plus one block-wide argmin over wavefront shuffles.
// Duals are f64, costs are widened exactly, so the optimum found is the true optimum of
// the f32 cost matrix; it equals scipy's wherever that optimum is unique.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

constexpr int LAP_CHUNKS_MAX = 64;

__global__ void lap_colmin_partial_kernel(const float* __restrict__ C, int64_t ldc, int n, int rows_per_chunk,
                                          float* __restrict__ part) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  float mn = INFINITY;
  for (int r = r0; r < r1; ++r) mn = fminf(mn, C[(int64_t)r * ldc + j]);
  part[(int64_t)blockIdx.y * n + j] = mn;
}

__global__ void lap_colmin_final_kernel(const float* __restrict__ part, int chunks, int n, double* __restrict__ v,
                                        int* __restrict__ claim) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float mn = INFINITY;
  for (int c = 0; c < chunks; ++c) mn = fminf(mn, part[(int64_t)c * n + j]);
  v[j] = (double)mn;
  claim[j] = INT_MAX;
}

// one wavefront per row
__global__ void lap_rowmin_kernel(const float* __restrict__ C, int64_t ldc, int n, const double* __restrict__ v,
                                  double* __restrict__ u, int* __restrict__ jbest, int* __restrict__ claim) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* row = C + (int64_t)i * ldc;
  double bv = INFINITY;
  int bj = INT_MAX;
  for (int j = lane; j < n; j += 64) {
    const double r = (double)row[j] - v[j];
    if (r < bv) {  // strict: keeps the lowest j of this lane on ties
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
    atomicMin(&claim[bj], i);
  }
}

__global__ void lap_greedy_finish_kernel(int n, const int* __restrict__ jbest, const int* __restrict__ claim,
                                         int* __restrict__ col4row, int* __restrict__ row4col) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int jb = jbest[t];
  col4row[t] = (claim[jb] == t) ? jb : -1;
  const int c = claim[t];
  row4col[t] = (c == INT_MAX) ? -1 : c;
}

// candidate ordering of one Dijkstra step: smaller distance, then a free column, then the lower index
__device__ __forceinline__ bool lap_better(double v, int f, int j, double bv, int bf, int bj) {
  if (v < bv) return true;
  if (v > bv) return false;
  if (f != bf) return f > bf;
  return j < bj;
}

// scipy-order variant (rectangular_lsap.cpp, augmenting_path): the sequential scan over its
// `remaining` vector keeps the FIRST column of minimal distance, except that every later FREE
// column of equal distance replaces the incumbent -- i.e. among the minima: the free column of
// highest position if any is free, else the column of lowest position.  p = position of the
// column in `remaining`.
__device__ __forceinline__ bool lap_better_scipy(double v, int f, int p, double bv, int bf, int bp) {
  if (v < bv) return true;
  if (v > bv) return false;
  if (f != bf) return f > bf;
  return f ? (p > bp) : (p < bp);
}

__global__ void lap_zero_init_kernel(int n, double* __restrict__ u, double* __restrict__ v, int* __restrict__ col4row,
                                     int* __restrict__ row4col) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u[t] = 0.0;
  v[t] = 0.0;
  col4row[t] = -1;
  row4col[t] = -1;
}

template <int CPT, bool SCIPY>
__global__ __launch_bounds__(1024) void lap_augment_kernel(const float* __restrict__ C, int64_t ldc, int n,
                                                           double* __restrict__ u, double* __restrict__ vg,
                                                           int* __restrict__ col4row_g, int* __restrict__ row4col_g,
                                                           int* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) int lap_sm[];
  int* path = lap_sm;
  int* row4col = lap_sm + n;
  int* col4row = lap_sm + 2 * n;
  int* remaining = lap_sm + 3 * n;  // position -> column (scipy-order mode only)
  double* redv = reinterpret_cast<double*>(lap_sm + (((SCIPY ? 4 : 3) * n + 3) & ~3));  // [2][16]
  int* redj = reinterpret_cast<int*>(redv + 32);                                         // [2][16]
  int* redf = redj + 32;                                                                 // [2][16]

  const int T = blockDim.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nw = T >> 6;

  double v[CPT], spc[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    v[c] = 0.0;
    if (j < n) {
      v[c] = vg[j];
      row4col[j] = row4col_g[j];
      col4row[j] = col4row_g[j];
    }
  }
  __syncthreads();

  int step = 0;
  for (int cur = 0; cur < n; ++cur) {
    if (col4row[cur] != -1) continue;  // same LDS word for every thread: uniform

    unsigned scanned = 0;
    int pos[CPT];
    int num_remaining = n;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      spc[c] = INFINITY;
      const int j = tid + c * T;
      pos[c] = n - 1 - j;  // scipy fills `remaining` in reverse order
      if (SCIPY && j < n) remaining[n - 1 - j] = j;
    }
    int i = cur;
    double minval = 0.0;
    int sink = -1;
    // every iteration scans one more column, so the loop ends after at most n steps
    for (int it = 0; it < n; ++it) {
      const double ui = u[i];
      const float* __restrict__ crow = C + (int64_t)i * ldc;
      double bv = INFINITY;
      int bj = INT_MAX, bf = 0, bp = 0;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int j = tid + c * T;
        if (j < n && !((scanned >> c) & 1u)) {
          const double r = minval + (double)crow[j] - ui - v[c];
          if (r < spc[c]) {
            spc[c] = r;
            path[j] = i;
          }
          const int f = row4col[j] == -1 ? 1 : 0;
          if (SCIPY ? (bj == INT_MAX || lap_better_scipy(spc[c], f, pos[c], bv, bf, bp))
                    : lap_better(spc[c], f, j, bv, bf, bj)) {
            bv = spc[c];
            bf = f;
            bj = j;
            bp = pos[c];
          }
        }
      }
      // block-wide argmin in two scalar reductions: the minimal distance, then -- among the lanes that hold
      // it -- the smallest tie-break key.  key = (column is matched) << 31 | rank: fast mode ranks by column
      // index; scipy order ranks free columns by descending position and matched ones by ascending position
      // (lap_better / lap_better_scipy as a total order).
      const bool have = bj != INT_MAX;
      const unsigned mykey = !have ? 0xffffffffu
                                   : ((bf ? 0u : 0x80000000u) |
                                      (unsigned)(SCIPY ? (bf ? n - 1 - bp : bp) : bj));
      const double wmin = wave_min_f64(have ? bv : INFINITY);
      const unsigned wkey = wave_min_u32((have && bv == wmin) ? mykey : 0xffffffffu);
      const int pb = (step & 1) * 16;
      ++step;
      if (lane == 0) {
        redv[pb + wave] = wmin;
        redj[pb + wave] = (int)wkey;
      }
      if (have && bv == wmin && mykey == wkey) redf[pb + wave] = bj;  // keys are unique: exactly one lane
      __syncthreads();
      double gmin = redv[pb];
      unsigned gkey = (unsigned)redj[pb];
      int gcol = redf[pb];
      for (int w = 1; w < nw; ++w) {
        const double ov = redv[pb + w];
        const unsigned ok = (unsigned)redj[pb + w];
        const int oc = redf[pb + w];
        if (ov < gmin || (ov == gmin && ok < gkey)) {
          gmin = ov;
          gkey = ok;
          gcol = oc;
        }
      }
      if (gkey == 0xffffffffu || !(gmin < INFINITY)) break;  // no finite candidate (non-finite costs): give up on this row
      bv = gmin;
      bf = (gkey & 0x80000000u) ? 0 : 1;
      bj = gcol;
      if (SCIPY) {
        const int r = (int)(gkey & 0x7fffffffu);
        bp = bf ? n - 1 - r : r;
      }
      minval = bv;
      const int inext = row4col[bj];
      if (SCIPY) {
        // remaining[index] = remaining[--num_remaining]
        --num_remaining;
        const int last = remaining[num_remaining];
        if (last != bj) {
          if (tid == 0) remaining[bp] = last;
#pragma unroll
          for (int c = 0; c < CPT; ++c)
            if (tid + c * T == last) pos[c] = bp;
        }
      }
      if (inext == -1) {
        sink = bj;
        break;
      }
      if (tid == bj % T) scanned |= 1u << (bj / T);
      i = inext;
    }
    if (sink < 0) continue;  // uniform; leaves the row unmatched (reported by the caller's check)

    // dual update (scipy rectangular_lsap: u[i] += minVal - spc[col4row[i]], v[j] -= minVal - spc[j])
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      if ((scanned >> c) & 1u) {
        const int j = tid + c * T;
        const double delta = minval - spc[c];
        v[c] -= delta;
        u[row4col[j]] += delta;
      }
    }
    if (tid == 0) u[cur] += minval;
    __syncthreads();
    if (tid == 0) {
      int j = sink;
      for (int guard = 0; guard <= n; ++guard) {
        const int ii = path[j];
        row4col[j] = ii;
        const int t = col4row[ii];
        col4row[ii] = j;
        j = t;
        if (ii == cur) break;
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    if (j < n) {
      vg[j] = v[c];
      row4col_g[j] = row4col[j];
      col4row_g[j] = col4row[j];
    }
  }
  if (tid == 0) stats[0] = step;  // Dijkstra steps taken (work counter, read by tools/lap_bench.py)
}

struct LapWs {
  size_t off_part, off_u, off_v, off_jbest, off_claim, off_r4c, total;
  int chunks, rows_per_chunk;
};

static inline size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

static LapWs lap_ws(int n) {
  LapWs w;
  w.chunks = (n + 63) / 64;
  if (w.chunks > LAP_CHUNKS_MAX) w.chunks = LAP_CHUNKS_MAX;
  w.rows_per_chunk = (n + w.chunks - 1) / w.chunks;
  w.chunks = (n + w.rows_per_chunk - 1) / w.rows_per_chunk;
  size_t o = 0;
  w.off_part = o; o += a256((size_t)w.chunks * n * sizeof(float));
  w.off_u = o; o += a256((size_t)n * sizeof(double));
  w.off_v = o; o += a256((size_t)n * sizeof(double));
  w.off_jbest = o; o += a256((size_t)n * sizeof(int));
  w.off_claim = o; o += a256((size_t)n * sizeof(int));
  w.off_r4c = o; o += a256((size_t)n * sizeof(int));
  w.total = o;
  return w;
}

template <int CPT, bool SCIPY>
static void launch_augment(int threads, size_t lds, hipStream_t st, const float* C, int64_t ldc, int n, double* u,
                           double* v, int* c4r, int* r4c, int* stats) {
  auto kfn = lap_augment_kernel<CPT, SCIPY>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(1), dim3(threads), lds, st, C, ldc, n, u, v, c4r, r4c, stats);
}

}  // namespace nif

extern "C" size_t nif_lap_workspace_bytes(int n) {
  if (n <= 0 || n > NIF_LAP_MAX_N) return 0;
  return nif::lap_ws(n).total;
}

extern "C" int nif_lap_solve_f32(const float* C, int64_t ldc, int n, int flags, int32_t* col4row, int32_t* row4col,
                                 double* u, double* v, void* ws, size_t ws_bytes, void* stream) {
  using namespace nif;
  if (!C || !col4row || n <= 0 || ldc < n) return NIF_ERR_INVALID_ARG;
  if (n > NIF_LAP_MAX_N) return NIF_ERR_UNSUPPORTED;
  const LapWs w = lap_ws(n);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  float* part = reinterpret_cast<float*>(b + w.off_part);
  double* uu = u ? u : reinterpret_cast<double*>(b + w.off_u);
  double* vv = v ? v : reinterpret_cast<double*>(b + w.off_v);
  int* jbest = reinterpret_cast<int*>(b + w.off_jbest);
  int* claim = reinterpret_cast<int*>(b + w.off_claim);
  int* r4c = row4col ? row4col : reinterpret_cast<int*>(b + w.off_r4c);

  const int nb = (n + 255) / 256;
  const bool scipy_order = (flags & NIF_LAP_SCIPY_ORDER) != 0;
  if (scipy_order) {
    hipLaunchKernelGGL(lap_zero_init_kernel, dim3(nb), dim3(256), 0, st, n, uu, vv, col4row, r4c);
  } else {
  hipLaunchKernelGGL(lap_colmin_partial_kernel, dim3(nb, w.chunks), dim3(256), 0, st, C, ldc, n, w.rows_per_chunk,
                     part);
  hipLaunchKernelGGL(lap_colmin_final_kernel, dim3(nb), dim3(256), 0, st, part, w.chunks, n, vv, claim);
  hipLaunchKernelGGL(lap_rowmin_kernel, dim3((n + 3) / 4), dim3(256), 0, st, C, ldc, n, vv, uu, jbest, claim);
  hipLaunchKernelGGL(lap_greedy_finish_kernel, dim3(nb), dim3(256), 0, st, n, jbest, claim, col4row, r4c);
  }
  if (!launch_ok()) return NIF_ERR_LAUNCH;

  // columns per lane: a Dijkstra step is a chain of dependent latencies (cost row -> wave reduce -> barrier),
  // and the barrier / cross-wave part grows with the wave count: 4-8 waves measured best (tools/lap_bench.py)
  int cpt = 1;
  while (cpt < 4 && (n + cpt - 1) / cpt > 256) cpt *= 2;
  while (cpt < 8 && (n + cpt - 1) / cpt > 1024) cpt *= 2;
  if (const char* e = getenv("NIF_LAP_CPT")) {  // tuning aid
    const int c = atoi(e);
    if ((c == 1 || c == 2 || c == 4 || c == 8) && (n + c - 1) / c <= 1024) cpt = c;
  }
  int threads = ((n + cpt - 1) / cpt + 63) / 64 * 64;
  if (threads > 1024) return NIF_ERR_UNSUPPORTED;
  const size_t lds =
      (size_t)(((scipy_order ? 4 : 3) * n + 3) & ~3) * sizeof(int) + 32 * sizeof(double) + 96 * sizeof(int);
  prof_begin(NIF_PROF_LAP, st);
#define NIF_LAP_GO(CPT)                                                                       \
  do {                                                                                        \
    if (scipy_order) launch_augment<CPT, true>(threads, lds, st, C, ldc, n, uu, vv, col4row, r4c, jbest); \
    else launch_augment<CPT, false>(threads, lds, st, C, ldc, n, uu, vv, col4row, r4c, jbest);       \
  } while (0)
  switch (cpt) {
    case 1: NIF_LAP_GO(1); break;
    case 2: NIF_LAP_GO(2); break;
    case 4: NIF_LAP_GO(4); break;
    default: NIF_LAP_GO(8); break;
  }
#undef NIF_LAP_GO
  prof_end(NIF_PROF_LAP, st);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
