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
#include <mutex>
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
  int r = r0;
  for (; r + 8 <= r1; r += 8) {  // eight independent loads in flight: the loop is latency-bound otherwise
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = C[(int64_t)(r + e) * ldc + j];
#pragma unroll
    for (int e = 0; e < 8; ++e) mn = fminf(mn, x[e]);
  }
  for (; r < r1; ++r) mn = fminf(mn, C[(int64_t)r * ldc + j]);
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
    // a row without a single finite entry (bj still INT_MAX) claims nothing: it stays free and the augmenting
    // phase reports it unmatched
    u[i] = bj == INT_MAX ? 0.0 : bv;
    jbest[i] = bj;
    if (bj != INT_MAX) atomicMin(&claim[bj], i);
  }
}

__global__ void lap_greedy_finish_kernel(int n, const int* __restrict__ jbest, const int* __restrict__ claim,
                                         int* __restrict__ col4row, int* __restrict__ row4col) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int jb = jbest[t];
  col4row[t] = (jb != INT_MAX && claim[jb] == t) ? jb : -1;
  const int c = claim[t];
  row4col[t] = (c == INT_MAX) ? -1 : c;
}

// Candidate ordering of one Dijkstra step (the tie-break key of lap_augment_kernel).  Fast mode: smaller
// distance, then a free column, then the lower index.  Scipy order (rectangular_lsap.cpp, augmenting_path): the
// sequential scan over its `remaining` vector keeps the FIRST column of minimal distance, except that every
// later FREE column of equal distance replaces the incumbent -- i.e. among the minima: the free column of
// highest position if any is free, else the column of lowest position.

__global__ void lap_zero_init_kernel(int n, double* __restrict__ u, double* __restrict__ v, int* __restrict__ col4row,
                                     int* __restrict__ row4col) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u[t] = 0.0;
  v[t] = 0.0;
  col4row[t] = -1;
  row4col[t] = -1;
}

// One mailbox per wave and step parity: {distance hi, distance lo (ordered-integer form), tie-break key,
// column, row matched to that column}; 8 words apart so a lane fetches the first four as one 16-byte read.
constexpr int LAP_MBOX_WORDS = 8;
constexpr int LAP_MBOX_TOTAL = 2 * 16 * LAP_MBOX_WORDS;

// ONE: the block is a single wave (n <= 128): its winner is the block's winner, so the mailbox, the barrier and the
// second reduction tree drop out.  A template parameter rather than a test of blockDim: with the test compiled into
// the multi-wave kernel that kernel ran 7 % slower.
template <int CPT, bool SCIPY, bool ONE = false>
__global__ __launch_bounds__(1024) void lap_augment_kernel(const float* __restrict__ C, int64_t ldc, int n,
                                                           double* __restrict__ u, double* __restrict__ vg,
                                                           int* __restrict__ col4row_g, int* __restrict__ row4col_g,
                                                           int* __restrict__ stats, int max_steps) {
  extern __shared__ __attribute__((aligned(16))) int lap_sm[];
  int* path = lap_sm;
  int* row4col = lap_sm + n;
  int* col4row = lap_sm + 2 * n;
  unsigned* mbox = reinterpret_cast<unsigned*>(lap_sm + ((3 * n + 3) & ~3));

  const int T = blockDim.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  constexpr unsigned NONE = 0xffffffffu;
  constexpr unsigned long long ORD_INF = 0xfff0000000000000ull;  // f64_to_ordered(+inf)

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
  for (int q = tid; q < LAP_MBOX_TOTAL; q += T) mbox[q] = NONE;  // absent waves never win
  __syncthreads();

  unsigned invalid = 0;
  int jcl[CPT];  // column index clamped into the matrix: loads need no guard
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int j = tid + c * T;
    if (j >= n) invalid |= 1u << c;
    jcl[c] = j < n ? j : 0;
  }

  int step = 0;
  // free rows in ascending order, 64 at a time: one LDS read and a ballot per block of rows instead of a dependent
  // LDS read per row (that scan alone took 17 us at n = 384).  A search matches only its own row, so the mask of a
  // block stays valid while its rows are processed.
  for (int rbase = 0; rbase < n; rbase += 64) {
    unsigned long long freemask = __ballot(rbase + lane < n && col4row[rbase + lane] == -1);  // same in every wave
    while (freemask) {
    // step budget (RESUME callers that only want cheap instances finished): the rows not reached stay free
    if (max_steps > 0 && step >= max_steps) {  // uniform
      rbase = n;
      break;
    }
    const int cur = rbase + __ffsll(freemask) - 1;
    freemask &= freemask - 1;

    // per search: the columns of a lane keep their distance, scanned bit, matched row and tie-break key in
    // registers -- nothing of a Dijkstra step but the `path` store touches LDS before the reduction.
    // key = (column is matched) << 31 | rank: fast mode ranks by column index; scipy order ranks free columns
    // by descending position in scipy's `remaining` vector and matched ones by ascending position.
    unsigned scanned = invalid;  // columns past n count as scanned
    unsigned key[CPT];
    int r4c[CPT];
    int num_remaining = n;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      spc[c] = INFINITY;
      r4c[c] = row4col[jcl[c]];
      const int j = tid + c * T;
      // scipy fills `remaining` in reverse order: column j starts at position n - 1 - j
      key[c] = r4c[c] == -1 ? (unsigned)j : (0x80000000u | (unsigned)(SCIPY ? n - 1 - j : j));
    }
    int i = cur;
    double minval = 0.0;
    int sink = -1;
    // every iteration scans one more column, so the loop ends after at most n steps
    for (int it = 0; it < n; ++it) {
      const double ui = u[i];
      const float* __restrict__ crow = C + (int64_t)i * ldc;
      float cj[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) cj[c] = crow[jcl[c]];  // all loads in flight together
      // lane-level candidate: smallest (distance, key)
      double bv = INFINITY;
      unsigned bk = NONE;
      int bj = -1, bi = -1;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const bool active = !((scanned >> c) & 1u);
        const double r = minval + (double)cj[c] - ui - v[c];
        if (active && r < spc[c]) {
          spc[c] = r;
          path[tid + c * T] = i;
        }
        const double cand = active ? spc[c] : INFINITY;
        const unsigned k = active ? key[c] : NONE;
        if (cand < bv || (cand == bv && k < bk)) {
          bv = cand;
          bk = k;
          bj = tid + c * T;
          bi = r4c[c];
        }
      }
      // wave-level argmin as integer DPP minima over the ordered form of the distance: high word, and only if
      // several lanes share it the low word, and only if several still tie the key.  Keys are unique, so one
      // lane is left; it posts the wave's candidate.
      const unsigned long long bo = bk == NONE ? ~0ull : f64_to_ordered(bv + 0.0);  // + 0.0: no negative zero
      const unsigned hi = (unsigned)(bo >> 32), lo = (unsigned)bo;
      const unsigned whi = wave_min_u32(hi);
      bool holder = hi == whi;
      unsigned long long m = __ballot(holder);
      if (whi != NONE && (m & (m - 1))) {
        const unsigned wlo = wave_min_u32(holder ? lo : NONE);
        holder = holder && lo == wlo;
        m = __ballot(holder);
        if (m & (m - 1)) {
          const unsigned wkey = wave_min_u32(holder ? bk : NONE);
          holder = holder && bk == wkey;
          m = __ballot(holder);
        }
      }
      const int poster = whi == NONE ? 0 : __ffsll(m) - 1;
      unsigned ghi, glo, gkey;
      int gcol, inext;
      if constexpr (ONE) {
        // one wave owns every column: its winner is the block's winner -- no mailbox, no barrier, no second tree
        ++step;
        if (whi == NONE) break;  // no candidate left
        ghi = whi;
        glo = (unsigned)__builtin_amdgcn_readlane((int)lo, poster);
        gkey = (unsigned)__builtin_amdgcn_readlane((int)bk, poster);
        gcol = __builtin_amdgcn_readlane(bj, poster);
        inext = __builtin_amdgcn_readlane(bi, poster);
      } else {
      unsigned* mine = mbox + ((step & 1) * 16 + wave) * LAP_MBOX_WORDS;
      const unsigned* all = mbox + ((step & 1) * 16 + (lane & 15)) * LAP_MBOX_WORDS;
      ++step;
      if (lane == poster) {
        *reinterpret_cast<uint4*>(mine) = make_uint4(hi, lo, bk, (unsigned)bj);
        mine[4] = (unsigned)bi;
      }
      __syncthreads();
      // block-level: lane l of every row of 16 takes wave l's mailbox, same minima over the row
      const uint4 e = *reinterpret_cast<const uint4*>(all);
      const unsigned erow = all[4];
      ghi = row16_min_u32(e.x);
      if (ghi == NONE) break;  // no candidate left
      bool gholder = e.x == ghi;
      unsigned gm = (unsigned)__ballot(gholder) & 0xffffu;
      if (gm & (gm - 1)) {
        const unsigned glo2 = row16_min_u32(gholder ? e.y : NONE);
        gholder = gholder && e.y == glo2;
        gm = (unsigned)__ballot(gholder) & 0xffffu;
        if (gm & (gm - 1)) {
          const unsigned gk = row16_min_u32(gholder ? e.z : NONE);
          gholder = gholder && e.z == gk;
          gm = (unsigned)__ballot(gholder) & 0xffffu;
        }
      }
      const int src = __ffs(gm) - 1;
      glo = (unsigned)__builtin_amdgcn_readlane((int)e.y, src);
      gkey = (unsigned)__builtin_amdgcn_readlane((int)e.z, src);
      gcol = __builtin_amdgcn_readlane((int)e.w, src);
      inext = __builtin_amdgcn_readlane((int)erow, src);
      }
      const unsigned long long gord = ((unsigned long long)ghi << 32) | glo;
      if (!(gord < ORD_INF)) break;  // no finite candidate (non-finite costs): give up on this row
      minval = ordered_to_f64(gord);
      if (SCIPY) {
        // remaining[index] = remaining[--num_remaining]: the column that sat last takes the freed position
        const int r = (int)(gkey & 0x7fffffffu);
        const int bp = (gkey & 0x80000000u) ? r : n - 1 - r;
        --num_remaining;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const bool is_free = r4c[c] == -1;
          const unsigned was_last = is_free ? (unsigned)(n - 1 - num_remaining) : (0x80000000u | (unsigned)num_remaining);
          const unsigned moved = is_free ? (unsigned)(n - 1 - bp) : (0x80000000u | (unsigned)bp);
          if (key[c] == was_last) key[c] = moved;
        }
      }
      if (inext == -1) {
        sink = gcol;
        break;
      }
#pragma unroll
      for (int c = 0; c < CPT; ++c)
        if (tid + c * T == gcol) scanned |= 1u << c;
      i = inext;
    }
    if (sink < 0) continue;  // uniform; leaves the row unmatched (reported by the caller's check)

    // dual update (scipy rectangular_lsap: u[i] += minVal - spc[col4row[i]], v[j] -= minVal - spc[j])
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      if (((scanned & ~invalid) >> c) & 1u) {
        const double delta = minval - spc[c];
        v[c] -= delta;
        u[r4c[c]] += delta;
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

// Uniqueness certificate for an optimal (col4row, u, v): counts the edges OUTSIDE the matching whose reduced cost
// c_ij - u_i - v_j is not clearly positive (<= tol_rel * (|c_ij| + |u_i| + |v_j|); a negative one, i.e. infeasible
// duals, counts too).  Zero such edges means no other assignment reaches the optimal cost, so every exact solver --
// scipy's included -- returns this matching.  One wavefront per row, HBM-bound single pass over C.
__global__ void lap_count_tight_kernel(const float* __restrict__ C, int64_t ldc, int n, const int* __restrict__ col4row,
                                       const double* __restrict__ u, const double* __restrict__ v, double tol_rel,
                                       int* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* row = C + (int64_t)i * ldc;
  const double ui = u[i];
  const int mate = col4row[i];
  int bad = mate < 0 || mate >= n ? (lane == 0) : 0;  // an unmatched row is no certificate either
  for (int j = lane; j < n; j += 64) {
    const double c = (double)row[j], vj = v[j];
    const double rc = c - ui - vj;
    const double tol = tol_rel * (fabs(c) + fabs(ui) + fabs(vj));
    if (j == mate) {
      // the matching must sit on tight edges, and tighter than the margin by more than a factor n: any other
      // assignment then costs more by at least one margin minus the sum of these residues
      if (fabs(rc) > 1e-5 * tol) ++bad;
    } else if (!(rc > tol)) {
      ++bad;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if (lane == 0 && bad) atomicAdd(count, bad);
}

struct LapWs {
  size_t off_part, off_u, off_v, off_jbest, off_claim, off_r4c, total;
  int chunks, rows_per_chunk;
};

static inline size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

static LapWs lap_ws(int n) {
  LapWs w;
  w.chunks = (n + 15) / 16;  // many short row chunks: the pass is latency-bound, not bandwidth-bound, at these sizes
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

template <int CPT, bool SCIPY, bool ONE = false>
static void launch_augment(int threads, size_t lds, hipStream_t st, const float* C, int64_t ldc, int n, double* u,
                           double* v, int* c4r, int* r4c, int* stats, int max_steps) {
  auto kfn = lap_augment_kernel<CPT, SCIPY, ONE>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(1), dim3(threads), lds, st, C, ldc, n, u, v, c4r, r4c, stats, max_steps);
}

}  // namespace nif

extern "C" size_t nif_lap_workspace_bytes(int n) {
  if (n <= 0 || n > NIF_LAP_MAX_N) return 0;
  return nif::lap_ws(n).total;
}

namespace nif {
// Both entry points below run this: `ws` is the caller's workspace (nif_lap_solve_f32) or the per-device address of
// the library's static scratch array (nif_lap_solve_f32_static).  `timed` = record the NIF_PROF_LAP span (creates
// HIP events, which allocates: the static entry point never asks for it).
static int lap_solve_on(const float* C, int64_t ldc, int n, int flags, int32_t* col4row, int32_t* row4col, double* u,
                        double* v, void* ws, hipStream_t st, bool timed);
}  // namespace nif

extern "C" int nif_lap_solve_f32(const float* C, int64_t ldc, int n, int flags, int32_t* col4row, int32_t* row4col,
                                 double* u, double* v, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_lap_solve_f32");
  using namespace nif;
  if (!C || !col4row || n <= 0 || ldc < n) return NIF_ERR_INVALID_ARG;
  if (n > NIF_LAP_MAX_N) return NIF_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < lap_ws(n).total) return NIF_ERR_WORKSPACE;
  return lap_solve_on(C, ldc, n, flags, col4row, row4col, u, v, ws, reinterpret_cast<hipStream_t>(stream), true);
}

// ---- the entry point that allocates nothing -------------------------------------------------------------------
// Scratch of the assignment kernels (column-minimum partials, claim / best-column words, stand-ins for optional
// outputs) for n <= NIF_LAP_STATIC_MAX_N, in the code object's own memory: every GPU the library runs on holds one
// copy until the process ends.  In host code the array's NAME is only a handle; its device address on the current
// GPU comes from hipGetSymbolAddress, once per device, and is what the kernels receive.
namespace nif {
constexpr size_t LAP_STATIC_BYTES = NIF_LAP_STATIC_BYTES;  // >= lap_ws(NIF_LAP_STATIC_MAX_N).total, checked at run time
static __device__ __align__(256) unsigned char scratch[LAP_STATIC_BYTES];

constexpr int LAP_MAX_DEVICES = 64;
static std::mutex scratch_mu;
static void* scratch_ptr[LAP_MAX_DEVICES];  // per device: each GPU has its own copy at its own address

static int lap_static_scratch(void** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LAP_MAX_DEVICES) return NIF_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lock(scratch_mu);
  if (!scratch_ptr[dev]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // with `dev` current
    if (e != hipSuccess || !p) {
      fprintf(stderr, "[libnif] hipGetSymbolAddress(scratch) on device %d: %s\n", dev, hipGetErrorString(e));
      return NIF_ERR_NO_DEVICE;
    }
    scratch_ptr[dev] = p;
  }
  *out = scratch_ptr[dev];
  return NIF_OK;
}
}  // namespace nif

extern "C" int nif_lap_solve_f32_static(const float* C, int64_t ldc, int n, int flags, int32_t* col4row,
                                        int32_t* row4col, double* u, double* v, void* stream) {
  nif::Range nif_range_("nif_lap_solve_f32_static");
  using namespace nif;
  if (!C || !col4row || n <= 0 || ldc < n) return NIF_ERR_INVALID_ARG;
  if (n > NIF_LAP_STATIC_MAX_N) {
    fprintf(stderr, "[libnif] nif_lap_solve_f32_static: n = %d exceeds NIF_LAP_STATIC_MAX_N = %d (the static scratch "
                    "is sized for that; use nif_lap_solve_f32 with a caller workspace)\n", n, NIF_LAP_STATIC_MAX_N);
    return NIF_ERR_UNSUPPORTED;
  }
  if (lap_ws(n).total > LAP_STATIC_BYTES) return NIF_ERR_WORKSPACE;  // cannot happen for n <= the maximum
  void* ws = nullptr;
  const int rc = lap_static_scratch(&ws);
  if (rc != NIF_OK) return rc;
  // every word the kernels read from the scratch is written by an earlier kernel of the same call, so nothing
  // left by the previous call (possibly for a larger n) has to be cleared
  return lap_solve_on(C, ldc, n, flags, col4row, row4col, u, v, ws, reinterpret_cast<hipStream_t>(stream), false);
}

namespace nif {
static int lap_solve_on(const float* C, int64_t ldc, int n, int flags, int32_t* col4row, int32_t* row4col, double* u,
                        double* v, void* ws, hipStream_t st, bool timed) {
  const LapWs w = lap_ws(n);
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
  const bool init_only = (flags & NIF_LAP_INIT_ONLY) != 0, resume = (flags & NIF_LAP_RESUME) != 0;
  const int max_steps = (int)((unsigned)flags >> NIF_LAP_BUDGET_SHIFT);  // fast mode only; 0 = no budget
  if ((init_only || resume) && (scipy_order || (init_only && resume) || !u || !v || !row4col)) return NIF_ERR_INVALID_ARG;
  if (resume) {
    // u, v, col4row, row4col hold the state an INIT_ONLY call left: go straight to the augmenting paths
  } else if (scipy_order) {
    hipLaunchKernelGGL(lap_zero_init_kernel, dim3(nb), dim3(256), 0, st, n, uu, vv, col4row, r4c);
  } else {
  hipLaunchKernelGGL(lap_colmin_partial_kernel, dim3(nb, w.chunks), dim3(256), 0, st, C, ldc, n, w.rows_per_chunk,
                     part);
  hipLaunchKernelGGL(lap_colmin_final_kernel, dim3(nb), dim3(256), 0, st, part, w.chunks, n, vv, claim);
  hipLaunchKernelGGL(lap_rowmin_kernel, dim3((n + 3) / 4), dim3(256), 0, st, C, ldc, n, vv, uu, jbest, claim);
  hipLaunchKernelGGL(lap_greedy_finish_kernel, dim3(nb), dim3(256), 0, st, n, jbest, claim, col4row, r4c);
  }
  if (!launch_ok()) return NIF_ERR_LAUNCH;
  if (init_only) return NIF_OK;

  // columns per lane: a Dijkstra step is a chain of dependent latencies (cost row -> relax -> wave minima ->
  // mailbox + barrier -> row-of-16 minima); the relax part grows with the columns a lane owns while the
  // cross-wave part no longer depends on the wave count, so as many waves as fit win (tools/lap_bench.py)
  int cpt = 1;
  while (cpt < 8 && (n + cpt - 1) / cpt > 1024) cpt *= 2;
  static const int cpt_override = env_int("NIF_LAP_CPT");  // tuning aid
  if ((cpt_override == 1 || cpt_override == 2 || cpt_override == 4 || cpt_override == 8) &&
      (n + cpt_override - 1) / cpt_override <= 1024)
    cpt = cpt_override;
  // up to 128 columns a single wave (two columns per lane) is quicker still: its winner is the block's winner, so the
  // mailbox, the barrier and the second reduction tree drop out (0.99 -> 0.89 us per step at n = 128); from four
  // columns per lane on the relax work outweighs that (n = 512 on one wave: 1.3 us per step against 0.95)
  if (n > 64 && n <= 128 && cpt_override == 0) cpt = 2;
  int threads = ((n + cpt - 1) / cpt + 63) / 64 * 64;
  if (threads > 1024) return NIF_ERR_UNSUPPORTED;
  const size_t lds = (size_t)((3 * n + 3) & ~3) * sizeof(int) + LAP_MBOX_TOTAL * sizeof(int);
  if (timed) prof_begin(NIF_PROF_LAP, st);
#define NIF_LAP_GO(CPT, ONE)                                                                       \
  do {                                                                                        \
    if (scipy_order) launch_augment<CPT, true, ONE>(threads, lds, st, C, ldc, n, uu, vv, col4row, r4c, jbest, 0); \
    else launch_augment<CPT, false, ONE>(threads, lds, st, C, ldc, n, uu, vv, col4row, r4c, jbest, max_steps);       \
  } while (0)
  if (threads == 64 && cpt == 1) NIF_LAP_GO(1, true);
  else if (threads == 64 && cpt == 2) NIF_LAP_GO(2, true);
  else
  switch (cpt) {
    case 1: NIF_LAP_GO(1, false); break;
    case 2: NIF_LAP_GO(2, false); break;
    case 4: NIF_LAP_GO(4, false); break;
    default: NIF_LAP_GO(8, false); break;
  }
#undef NIF_LAP_GO
  if (timed) prof_end(NIF_PROF_LAP, st);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
}  // namespace nif

extern "C" int nif_lap_count_tight_f32(const float* C, int64_t ldc, int n, const int32_t* col4row, const double* u,
                                       const double* v, double tol_rel, int32_t* count, void* stream) {
  nif::Range nif_range_("nif_lap_count_tight_f32");
  using namespace nif;
  if (!C || !col4row || !u || !v || !count || n <= 0 || ldc < n || !(tol_rel >= 0.0)) return NIF_ERR_INVALID_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  if (hipMemsetAsync(count, 0, sizeof(int32_t), st) != hipSuccess) return NIF_ERR_LAUNCH;
  hipLaunchKernelGGL(lap_count_tight_kernel, dim3((n + 3) / 4), dim3(256), 0, st, C, ldc, n, col4row, u, v, tol_rel, count);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
