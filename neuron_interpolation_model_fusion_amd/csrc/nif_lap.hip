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

// Strided batch of independent problems of one size (nif_lap_solve_batch_f32): problem blockIdx.z of a launch reads /
// writes its own slice of every array.  Strides in elements of the array's type; all zero (and gridDim.z = 1) for a
// single problem.  The per-problem workspace arrays (partials, best columns, claims) are `ws` elements of 4 bytes apart.
struct LapBatch {
  int64_t c = 0;    // cost matrices
  int64_t vec = 0;  // u, v, col4row, row4col (n entries per problem)
  int64_t ws = 0;   // partials / jbest / claim inside the workspace
};

__global__ void lap_colmin_partial_kernel(const float* __restrict__ C, int64_t ldc, int n, int rows_per_chunk,
                                          float* __restrict__ part, LapBatch bt) {
  C += blockIdx.z * bt.c;
  part += blockIdx.z * bt.ws;
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
                                        int* __restrict__ claim, LapBatch bt) {
  part += blockIdx.z * bt.ws;
  v += blockIdx.z * bt.vec;
  claim += blockIdx.z * bt.ws;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float mn = INFINITY;
  for (int c = 0; c < chunks; ++c) mn = fminf(mn, part[(int64_t)c * n + j]);
  v[j] = (double)mn;
  claim[j] = INT_MAX;
}

// one wavefront per row
__global__ void lap_rowmin_kernel(const float* __restrict__ C, int64_t ldc, int n, const double* __restrict__ v,
                                  double* __restrict__ u, int* __restrict__ jbest, int* __restrict__ claim, LapBatch bt) {
  C += blockIdx.z * bt.c;
  v += blockIdx.z * bt.vec;
  u += blockIdx.z * bt.vec;
  jbest += blockIdx.z * bt.ws;
  claim += blockIdx.z * bt.ws;
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
                                         int* __restrict__ col4row, int* __restrict__ row4col, LapBatch bt) {
  jbest += blockIdx.z * bt.ws;
  claim += blockIdx.z * bt.ws;
  col4row += blockIdx.z * bt.vec;
  row4col += blockIdx.z * bt.vec;
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
                                     int* __restrict__ row4col, LapBatch bt) {
  u += blockIdx.z * bt.vec;
  v += blockIdx.z * bt.vec;
  col4row += blockIdx.z * bt.vec;
  row4col += blockIdx.z * bt.vec;
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
                                                           int* __restrict__ stats, int max_steps, LapBatch bt) {
  C += blockIdx.z * bt.c;
  u += blockIdx.z * bt.vec;
  vg += blockIdx.z * bt.vec;
  col4row_g += blockIdx.z * bt.vec;
  row4col_g += blockIdx.z * bt.vec;
  stats += blockIdx.z * bt.ws;
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

// ---- augmenting row reduction: the auction with zero slack of Jonker-Volgenant's initialisation, one workgroup ----
// The greedy start leaves about a third of the rows of an unrelated instance free, and each of them costs a
// shortest-path search of dozens to hundreds of Dijkstra steps (one dependent chain of ~0.9 us per step).  This phase
// empties most of that list first.  A free row i bids for its best column j1 (smallest reduced cost r_j = C[i][j] -
// v[j]): the price v[j1] drops by gamma = second-best - best, so that the row's best and second-best columns tie --
// its new edge stays tight, v stays dual feasible (prices only ever fall), exact complementary slackness is kept --
// and the previous owner of j1, if any, is free now and bids next: a chain that ends at a free column.  Sixteen
// chains run side by side, one per wave, in lock-step passes: bids of one pass are formed from the same prices; per
// column the largest gamma wins (lowest wave on ties) and the losers bid again with the same row.  gamma = 0 (costs
// are f32, so reduced costs live on a lattice and exact ties do occur) is accepted for an occupied column too, up to
// max_wait times in a row per chain: the owner it frees usually answers with a positive bid that breaks the tie; a
// chain of zero bids that goes on (structural ties, e.g. duplicated columns) leaves its row to the augmenting phase.
// The phase ends when the list of free rows is handed out and at most min_chains chains are left (the tail is no
// faster per bid than a Dijkstra step), or after max_passes.  Whatever is still free goes to lap_augment_kernel as
// before, from much better duals: unrelated activations, n = 384: 8958 Dijkstra steps / 7.2 ms without this phase,
// 413 passes + 1479 steps / 2.0 ms with it; n = 1024: 36.7 -> 13.7 ms; n = 2048: 141 -> 51 ms
// (profiles/r3_b_lap_arr_sweep.txt).  A pass costs ~1.5 us: the sixteen waves share one CU's four SIMDs, and a bid
// is ~250 dependent vector instructions (f64 scan of the row, two wave-wide minima, the exchange).
// Nothing in a pass waits for global memory: a wave keeps the cost row of its bidder in registers (CPT columns per
// lane; CPT = 0 re-reads the row every pass, n > 1024), and as soon as it knows its best column it requests the row
// of that column's owner -- the row that bids next if the bid wins -- into a second register set.  A wave that takes
// a new row from the list of free rows sits one pass out while the row arrives.  Rows are handed out by wave order
// from flags every wave reads, so the run is deterministic.  v / row4col / the list live in LDS; col4row is rebuilt at
// the end.
constexpr int ARR_W = 16;

template <int CPT>
__global__ __launch_bounds__(1024) void lap_arr_kernel(const float* __restrict__ C, int64_t ldc, int n,
                                                       double* __restrict__ u, double* __restrict__ vg,
                                                       int* __restrict__ col4row_g, int* __restrict__ row4col_g,
                                                       int* __restrict__ stats, int max_passes, int min_chains,
                                                       int max_wait, LapBatch bt) {
  C += blockIdx.z * bt.c;
  u += blockIdx.z * bt.vec;
  vg += blockIdx.z * bt.vec;
  col4row_g += blockIdx.z * bt.vec;
  row4col_g += blockIdx.z * bt.vec;
  stats += blockIdx.z * bt.ws;
  extern __shared__ __attribute__((aligned(16))) int lap_sm[];
  // v is padded to one slot per (lane, column group) when rows are register-resident: the lane's prices are then read
  // at constant offsets from ONE per-lane address (a clamped index per column costs an address register per column)
  const int nv = CPT > 0 ? 64 * CPT : n;
  double* v = reinterpret_cast<double*>(lap_sm);
  int* row4col = reinterpret_cast<int*>(v + nv);
  int* queue = row4col + n;
  double* bid_gamma = reinterpret_cast<double*>(queue + n + (n & 1));
  int* bid_col = reinterpret_cast<int*>(bid_gamma + ARR_W);
  int* flags = bid_col + ARR_W;
  int* ctl = flags + ARR_W;  // per-wave counts of a compaction

  const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, nw = T >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the per-wave state below stays in SGPRs
  constexpr unsigned NONE = 0xffffffffu;
  constexpr int R = CPT > 0 ? CPT : 1;

  // free rows, ascending, into `queue` (also used for the final pass over the rows still free); returns their number
  auto compact_free = [&]() -> int {
    int total = 0;
    for (int base = 0; base < n; base += T) {
      const int r = base + tid;
      const bool fr = r < n && col4row_g[r] == -1;
      const unsigned long long m = __ballot(fr);
      if (lane == 0) ctl[wave] = __popcll(m);
      __syncthreads();
      int before = 0, all = 0;
      for (int w = 0; w < nw; ++w) {
        const int c = ctl[w];
        if (w < wave) before += c;
        all += c;
      }
      if (fr) queue[total + before + __popcll(m & ((1ull << lane) - 1ull))] = r;
      total += all;
      __syncthreads();
    }
    return total;
  };

  for (int j = tid; j < n; j += T) {
    v[j] = vg[j];
    row4col[j] = row4col_g[j];
  }
  for (int j = n + tid; j < nv; j += T) v[j] = 0.0;
  const int total = compact_free();  // ends with a barrier: v / row4col / queue are visible

  float bufA[R], bufB[R];
#pragma unroll
  for (int q = 0; q < R; ++q) bufA[q] = bufB[q] = 0.f;
  // The cost matrix as a buffer resource: a row's columns are loaded at constant offsets from one per-lane offset plus
  // the row's scalar offset -- every load in flight, no guards, no per-column address registers.  Slots past column n
  // read into the next row (discarded by the bid scan) or, past the end of the matrix, the range check's zero.
  const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(C), 0, (int)((((unsigned)(n - 1)) * (unsigned)ldc + (unsigned)n) * 4u), 0x00020000);
  auto request_row = [&](int row, float* buf) __attribute__((always_inline)) {
    const unsigned rowoff = (unsigned)row * (unsigned)ldc * 4u;  // row is wave-uniform: a scalar
#pragma unroll
    for (int q = 0; q < R; ++q)
      buf[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(crsrc, lane * 4 + 256 * q, rowoff, 0));
  };

  int cur = -1;    // the row this wave bids for
  int state = 0;   // 0: no row, 1: row requested this pass (bids from the next pass on), 2: bidding
  int waited = 0;  // passes in a row in which this wave's bid was a zero bid for an occupied column
  int pending = 0; // bufB holds the bidder's row (requested during the previous pass): move it over before use
  int next = 0;    // entries of the list handed out so far (the same number in every wave)
  unsigned need = (1u << ARR_W) - 1u;  // waves that want a row (the same mask in every wave)
  int passes = 0;
#ifdef NIF_LAP_ARR_TIMING
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tm = clock64();
#define NIF_ARR_MARK(k) do { const long long t_ = clock64(); tacc[k] += t_ - tm; tm = t_; } while (0)
#else
#define NIF_ARR_MARK(k) do { } while (0)
#endif

  while (passes < max_passes && !(need == (1u << ARR_W) - 1u && next >= total)) {  // uniform over the block
    if ((need >> wave) & 1u) {
      const int at = next + __popc(need & ((1u << wave) - 1u));
      if (at < total) {
        cur = __builtin_amdgcn_readfirstlane(queue[at]);
        state = CPT > 0 ? 1 : 2;
        pending = 0;
        if (CPT > 0) request_row(cur, bufA);  // not waited for here: the wave sits this pass out
      }
    }
    next += __popc(need);
    if (next > total) next = total;
    NIF_ARR_MARK(0);

    const bool bidding = state == 2;
    int j1 = -1, prev = -1;
    double g = 0.0;
    if (bidding) {
      double b1 = INFINITY, b2 = INFINITY;
      int jl = INT_MAX;
      // branch-free: a guarded LDS read or a branch per column would serialise the columns of a lane (one LDS
      // round trip each); r = +inf for the columns past n
      auto take = [&](float c, double vj, int j) __attribute__((always_inline)) {
        const double r = j < n ? (double)c - vj : INFINITY;
        const bool lt1 = r < b1, lt2 = r < b2;  // strict: the lowest column of a lane keeps a tie, the other copy
        b2 = lt1 ? b1 : (lt2 ? r : b2);         // becomes the second best
        b1 = lt1 ? r : b1;
        jl = lt1 ? j : jl;
      };
      if (CPT > 0) {
        if (pending) {  // the wait for last pass's request lands here, a whole pass after it was issued
#pragma unroll
          for (int q = 0; q < R; ++q) bufA[q] = bufB[q];
          pending = 0;
        }
#ifdef NIF_LAP_ARR_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NIF_ARR_MARK(1);
#endif
        // the prices of the lane's columns eight at a time: all R at once would put 2 R more registers beside the two row
        // sets (R = 16 spilled that way, and wider rows had to be re-read from L2 every pass)
        constexpr int VC = R < 8 ? R : 8;
#pragma unroll
        for (int q0 = 0; q0 < R; q0 += VC) {
          double vj[VC];
#pragma unroll
          for (int q = 0; q < VC; ++q) vj[q] = v[lane + 64 * (q0 + q)];  // (padded: no clamp)
#pragma unroll
          for (int q = 0; q < VC; ++q) take(bufA[q0 + q], vj[q], lane + 64 * (q0 + q));
        }
      } else {
        const float* __restrict__ crow = C + (int64_t)cur * ldc;
        for (int t0 = 0; t0 < n; t0 += 512) {
          float c[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int j = t0 + q * 64 + lane;
            c[q] = crow[j < n ? j : n - 1];
          }
          double vj[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int j = t0 + q * 64 + lane;
            vj[q] = v[j < n ? j : n - 1];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(c[q]));
#pragma unroll
          for (int q = 0; q < 8; ++q) take(c[q], vj[q], t0 + q * 64 + lane);
        }
      }
      // wave-wide best (lowest column on ties) through the ordered-integer minima, then the second best: the minimum
      // over the best lane's second and everybody else's best
      const unsigned long long o1 = jl == INT_MAX ? ~0ull : f64_to_ordered(b1 + 0.0);
      const unsigned hi = (unsigned)(o1 >> 32), lo = (unsigned)o1;
      const unsigned whi = wave_min_u32(hi);
      bool holder = hi == whi;
      unsigned long long m = __ballot(holder);
      if (whi != NONE && (m & (m - 1))) {
        const unsigned wlo = wave_min_u32(holder ? lo : NONE);
        holder = holder && lo == wlo;
        m = __ballot(holder);
        if (m & (m - 1)) {
          const unsigned wj = wave_min_u32(holder ? (unsigned)jl : NONE);
          holder = holder && (unsigned)jl == wj;
          m = __ballot(holder);
        }
      }
      if (whi != NONE) {
        const int poster = __ffsll(m) - 1;
        const double w2 = wave_min_f64(lane == poster ? b2 : b1);
        j1 = __builtin_amdgcn_readlane(jl, poster);
        const double w1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(b1), poster),
                                           __builtin_amdgcn_readlane(__double2loint(b1), poster));
        g = w2 - w1;
        prev = __builtin_amdgcn_readfirstlane(row4col[j1]);
        if (CPT > 0 && prev >= 0) request_row(prev, bufB);  // the row that bids next if this bid wins
      }
    }
    NIF_ARR_MARK(2);
    if (lane == 0) {
      bid_col[wave] = j1;
      bid_gamma[wave] = g;
    }
    __syncthreads();
    NIF_ARR_MARK(3);

    int flag = 0;  // 1: wants a row, 2: a bid was accepted
    if (bidding && j1 >= 0) {
      const int oc = lane < ARR_W ? bid_col[lane] : -1;
      const double og = lane < ARR_W ? bid_gamma[lane] : 0.0;
      const bool lost = __ballot(lane < ARR_W && lane != wave && oc == j1 && (og > g || (og == g && lane < wave))) != 0ull;
      if (!lost) {
        const int finite = __builtin_amdgcn_readfirstlane((int)(g < INFINITY));  // 0 for +inf (one finite entry) / NaN
        const int positive = __builtin_amdgcn_readfirstlane((int)(g > 0.0));
        // a zero bid for an occupied column is accepted too, a few times in a row: the owner it frees usually answers
        // with a positive bid for the same column, which breaks the tie the row sat on
        if ((finite && positive) || prev == -1 || (finite && waited < max_wait)) {
          if (!positive && prev >= 0) ++waited;
          else waited = 0;
          if (lane == 0) {
            if (finite) v[j1] -= g;
            row4col[j1] = cur;  // col4row is rebuilt from row4col at the end: no global store inside the loop
          }
          flag = 2;
          if (prev >= 0) {
            cur = prev;  // its row was requested into the second register set
            pending = 1;
          } else {
            cur = -1;
            state = 0;
            flag |= 1;
          }
        } else {
          // zero bids went back and forth (exact structural ties, e.g. duplicated columns): the row is left to the
          // augmenting phase, so that such rows cannot take the sixteen chains out of work one by one
          waited = 0;
          cur = -1;
          state = 0;
          flag = 1;
        }
      }
    } else if (bidding) {  // a row without a finite entry: nothing to bid for
      cur = -1;
      state = 0;
      flag = 1;
    } else if (state == 1) {
      state = 2;  // its row arrives while this pass runs
    } else {
      flag = 1;
    }
    NIF_ARR_MARK(4);
    if (lane == 0) flags[wave] = flag;
    __syncthreads();
    NIF_ARR_MARK(5);
    const int f = lane < ARR_W ? flags[lane] : 0;
    need = (unsigned)__ballot((f & 1) != 0);
    ++passes;
    // the tail -- the list handed out and only a few chains left -- is no faster per bid than the Dijkstra steps of
    // the augmenting kernel, which also ends: hand the last rows over (their col4row is -1)
    if (next >= total && ARR_W - __popc(need) <= min_chains) break;
  }

  for (int j = tid; j < n; j += T) {
    vg[j] = v[j];
    row4col_g[j] = row4col[j];
    col4row_g[j] = -1;
  }
  __syncthreads();
  // col4row from row4col; u from v: the matched edge is tight, a free row gets its smallest reduced cost
  for (int j = tid; j < n; j += T) {
    const int i = row4col[j];
    if (i >= 0) {
      col4row_g[i] = j;
      u[i] = (double)C[(int64_t)i * ldc + j] - v[j];
    }
  }
  __syncthreads();
  const int nfree = compact_free();
  for (int e = wave; e < nfree; e += nw) {
    const int i = queue[e];
    const float* __restrict__ crow = C + (int64_t)i * ldc;
    double b1 = INFINITY;
    for (int j = lane; j < n; j += 64) b1 = fmin(b1, (double)crow[j] - v[j]);
    b1 = wave_min_f64(b1 + 0.0);
    if (lane == 0) u[i] = b1 < INFINITY ? b1 : 0.0;
  }
  if (tid == 0) {
    stats[1] = passes;
    stats[2] = nfree;
#ifdef NIF_LAP_ARR_TIMING
    for (int k = 0; k < 6; ++k) stats[4 + k] = (int)(tacc[k] / (passes > 0 ? passes : 1));
#endif
  }
}
#undef NIF_ARR_MARK

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

// ---- uniqueness certificate that tolerates ties: no alternating cycle through near-tight edges --------------------
// lap_count_tight_kernel's test (NO near-tight edge outside the matching) is sufficient but far from necessary: the
// augmenting row reduction and the shortest-path dual updates leave tight edges behind by construction (a bidder's
// best and second-best columns tie), and the optimum is unique all the same unless such edges close a cycle.  Another
// optimal assignment differs from this one by alternating cycles, i.e. by a directed cycle in the graph on the rows
// with an arc i -> row4col[j] for every near-tight edge (i, j) outside the matching; if that graph is acyclic, every
// alternating cycle contains an edge whose reduced cost exceeds the margin, and every other assignment costs more.
// Pass 1 (all CUs): collect the arcs (reduced cost <= tol_rel * (|c| + |u| + |v|)) and count what voids a certificate
// outright (a matched edge that is not tight, a clearly negative reduced cost, an unmatched row).  Pass 2 (one
// workgroup): peel rows without outgoing arcs until nothing changes; rows left over sit on or lead into a cycle.
__global__ void lap_cert_arcs_kernel(const float* __restrict__ C, int64_t ldc, int n, const int* __restrict__ col4row,
                                     const int* __restrict__ row4col, const double* __restrict__ u,
                                     const double* __restrict__ v, double tol_rel, int* __restrict__ ctl,
                                     int2* __restrict__ arcs, int cap) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* row = C + (int64_t)i * ldc;
  const double ui = u[i];
  const int mate = col4row[i];
  int bad = mate < 0 || mate >= n ? (lane == 0) : 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    bool arc = false;
    if (j < n) {
      const double c = (double)row[j], vj = v[j];
      const double rc = c - ui - vj;
      const double tol = tol_rel * (fabs(c) + fabs(ui) + fabs(vj));
      if (j == mate) {
        if (fabs(rc) > 1e-5 * tol) ++bad;
      } else if (!(rc > -tol)) {
        ++bad;  // infeasible duals (or NaN)
      } else if (!(rc > tol)) {
        arc = true;
      }
    }
    const unsigned long long m = __ballot(arc);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&ctl[1], __popcll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      if (arc) {
        const int at = base + __popcll(m & ((1ull << lane) - 1ull));
        if (at < cap) arcs[at] = make_int2(i, row4col[j]);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if (lane == 0 && bad) atomicAdd(&ctl[0], bad);
}

// ctl[0] = disqualifying edges, ctl[1] = arcs found; result -> ctl[2]: 0 = unique optimum certified, else not
__global__ __launch_bounds__(1024) void lap_cert_acyclic_kernel(int n, int* __restrict__ ctl, const int2* __restrict__ arcs,
                                                                int cap) {
  extern __shared__ int cert_sm[];
  int* outdeg = cert_sm;       // arcs still leaving row i; -1 once the row is peeled
  int* changed = cert_sm + n;  // [2]: flag per parity
  const int tid = threadIdx.x, T = blockDim.x;
  const int narcs = ctl[1];
  if (ctl[0] != 0 || narcs > cap) {  // uniform
    if (tid == 0) ctl[2] = 1;
    return;
  }
  for (int i = tid; i < n; i += T) outdeg[i] = 0;
  if (tid < 2) changed[tid] = 0;
  __syncthreads();
  for (int e = tid; e < narcs; e += T) atomicAdd(&outdeg[arcs[e].x], 1);
  __syncthreads();
  // a row with no outgoing arc cannot be on a cycle: peel it, which removes the arcs that point to it
  for (int round = 0; round <= n; ++round) {
    int* flag = changed + (round & 1);
    for (int i = tid; i < n; i += T)
      if (outdeg[i] == 0) {
        outdeg[i] = -2;  // peeled in this round
        *flag = 1;
      }
    __syncthreads();
    if (*flag == 0) break;  // uniform
    for (int e = tid; e < narcs; e += T) {
      const int2 a = arcs[e];
      if (outdeg[a.y] == -2 && outdeg[a.x] > 0) atomicSub(&outdeg[a.x], 1);
    }
    __syncthreads();
    for (int i = tid; i < n; i += T)
      if (outdeg[i] == -2) outdeg[i] = -1;
    if (tid == 0) changed[(round + 1) & 1] = 0;
    __syncthreads();
  }
  int left = 0;
  for (int i = tid; i < n; i += T) left |= outdeg[i] > 0;
  if (left) ctl[2] = 1;  // any number of threads may store the same value
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
                           double* v, int* c4r, int* r4c, int* stats, int max_steps, int batch, LapBatch bt) {
  auto kfn = lap_augment_kernel<CPT, SCIPY, ONE>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(1, 1, batch), dim3(threads), lds, st, C, ldc, n, u, v, c4r, r4c, stats, max_steps, bt);
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
// batch > 1: `batch` problems of size n, cost matrices stride_c floats apart, outputs n entries apart, workspaces
// lap_ws(n).total bytes apart; every kernel runs once with the problems along gridDim.z (u, v, row4col required).
static int lap_solve_on(const float* C, int64_t ldc, int n, int flags, int32_t* col4row, int32_t* row4col, double* u,
                        double* v, void* ws, hipStream_t st, bool timed, int batch = 1, int64_t stride_c = 0);
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

extern "C" size_t nif_lap_batch_workspace_bytes(int n, int batch) {
  if (n <= 0 || n > NIF_LAP_MAX_N || batch <= 0) return 0;
  return nif::lap_ws(n).total * (size_t)batch;
}

extern "C" int nif_lap_solve_batch_f32(const float* C, int64_t ldc, int64_t stride_c, int n, int batch, int flags,
                                       int32_t* col4row, int32_t* row4col, double* u, double* v, void* ws,
                                       size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_lap_solve_batch_f32");
  using namespace nif;
  if (!C || !col4row || !row4col || !u || !v || n <= 0 || ldc < n || batch <= 0 || batch > 65535) return NIF_ERR_INVALID_ARG;
  if (batch > 1 && stride_c < (int64_t)(n - 1) * ldc + n) return NIF_ERR_INVALID_ARG;  // problems must not overlap
  if (n > NIF_LAP_MAX_N) return NIF_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < lap_ws(n).total * (size_t)batch) return NIF_ERR_WORKSPACE;
  return lap_solve_on(C, ldc, n, flags, col4row, row4col, u, v, ws, reinterpret_cast<hipStream_t>(stream), true, batch,
                      stride_c);
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
                        double* v, void* ws, hipStream_t st, bool timed, int batch, int64_t stride_c) {
  const LapWs w = lap_ws(n);
  LapBatch bt;
  if (batch > 1) {
    if (!u || !v || !row4col) return NIF_ERR_INVALID_ARG;
    bt.c = stride_c;
    bt.vec = n;
    bt.ws = (int64_t)(w.total / 4);
  }
  const unsigned bz = (unsigned)batch;
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
    hipLaunchKernelGGL(lap_zero_init_kernel, dim3(nb, 1, bz), dim3(256), 0, st, n, uu, vv, col4row, r4c, bt);
  } else {
  hipLaunchKernelGGL(lap_colmin_partial_kernel, dim3(nb, w.chunks, bz), dim3(256), 0, st, C, ldc, n, w.rows_per_chunk,
                     part, bt);
  hipLaunchKernelGGL(lap_colmin_final_kernel, dim3(nb, 1, bz), dim3(256), 0, st, part, w.chunks, n, vv, claim, bt);
  hipLaunchKernelGGL(lap_rowmin_kernel, dim3((n + 3) / 4, 1, bz), dim3(256), 0, st, C, ldc, n, vv, uu, jbest, claim, bt);
  hipLaunchKernelGGL(lap_greedy_finish_kernel, dim3(nb, 1, bz), dim3(256), 0, st, n, jbest, claim, col4row, r4c, bt);
  }
  if (!launch_ok()) return NIF_ERR_LAUNCH;
  if (init_only) return NIF_OK;
  if (timed) prof_begin(NIF_PROF_LAP, st);  // the span covers the row reduction and the augmenting kernel
  if (!scipy_order && n >= 2) {
    // augmenting row reduction before the shortest-path searches: at most 4 n passes of up to 16 bids each
    // (NIF_LAP_ARR_PASSES: tuning aid, -1 = skip the phase, k > 0 = that many passes)
    const int arr_env = env_int("NIF_LAP_ARR_PASSES");
    const int arr_passes = arr_env < 0 ? 0 : (arr_env > 0 ? arr_env : 4 * n);
    // measured on unrelated-activation costs (tools/lap_arr_sweep.py, profiles/r3_b_lap_arr_sweep.txt): stop once the
    // list is handed out and at most 6 chains are left, accept at most 4 zero bids in a row per chain
    const int arr_min_chains = getenv("NIF_LAP_ARR_MIN_CHAINS") ? env_int("NIF_LAP_ARR_MIN_CHAINS") : 6;
    const int arr_max_wait = getenv("NIF_LAP_ARR_MAX_WAIT") ? env_int("NIF_LAP_ARR_MAX_WAIT") : 4;
    if (arr_passes > 0) {
      // LDS: v (padded to 64 CPT slots when rows are register-resident) + row4col + queue + bids / flags
#define NIF_ARR_GO(CPT)                                                                                               \
  do {                                                                                                                \
    const size_t nv = CPT > 0 ? (size_t)64 * CPT : (size_t)n;                                                         \
    const size_t arr_lds = nv * 8 + (size_t)n * 8 + (n & 1) * 4 + ARR_W * 16 + 32 * sizeof(int);                      \
    auto kfn = lap_arr_kernel<CPT>;                                                                                   \
    if (arr_lds > 48 * 1024)                                                                                          \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)arr_lds);          \
    hipLaunchKernelGGL(kfn, dim3(1, 1, bz), dim3(1024), arr_lds, st, C, ldc, n, uu, vv, col4row, r4c, jbest, arr_passes, \
                       arr_min_chains, arr_max_wait, bt);                                                           \
  } while (0)
      int cols = (n + 63) / 64;  // columns a lane holds of its bidder's row
      // (the register-resident kernels address the matrix through a 32-bit buffer offset)
      if (((uint64_t)(n - 1) * (uint64_t)ldc + (uint64_t)n) * 4 >= 0xffffffffull) cols = 1 << 20;
      if (cols <= 2) NIF_ARR_GO(2);
      else if (cols <= 4) NIF_ARR_GO(4);
      else if (cols <= 6) NIF_ARR_GO(6);
      else if (cols <= 8) NIF_ARR_GO(8);
      else if (cols <= 12) NIF_ARR_GO(12);
      else if (cols <= 16) NIF_ARR_GO(16);
      else NIF_ARR_GO(0);  // two register sets of more than 16 columns spill (24: 340 B, 32: 4.6 KB of scratch a lane with
                           // sixteen waves on the CU): these sizes re-read the row from L2
#undef NIF_ARR_GO
    }
  }

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
  if (threads > 1024) {
    if (timed) prof_end(NIF_PROF_LAP, st);
    return NIF_ERR_UNSUPPORTED;
  }
  const size_t lds = (size_t)((3 * n + 3) & ~3) * sizeof(int) + LAP_MBOX_TOTAL * sizeof(int);
#define NIF_LAP_GO(CPT, ONE)                                                                       \
  do {                                                                                        \
    if (scipy_order) launch_augment<CPT, true, ONE>(threads, lds, st, C, ldc, n, uu, vv, col4row, r4c, jbest, 0, batch, bt); \
    else launch_augment<CPT, false, ONE>(threads, lds, st, C, ldc, n, uu, vv, col4row, r4c, jbest, max_steps, batch, bt); \
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

extern "C" size_t nif_lap_certify_workspace_bytes(int n) {
  if (n <= 0 || n > NIF_LAP_MAX_N) return 0;
  return 256 + (size_t)8 * n * sizeof(int2);
}

extern "C" int nif_lap_certify_unique_f32(const float* C, int64_t ldc, int n, const int32_t* col4row, const int32_t* row4col,
                                          const double* u, const double* v, double tol_rel, int32_t* result, void* ws,
                                          size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_lap_certify_unique_f32");
  using namespace nif;
  if (!C || !col4row || !row4col || !u || !v || !result || n <= 0 || ldc < n || !(tol_rel >= 0.0)) return NIF_ERR_INVALID_ARG;
  if (n > NIF_LAP_MAX_N) return NIF_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < nif_lap_certify_workspace_bytes(n)) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  int* ctl = reinterpret_cast<int*>(ws);
  int2* arcs = reinterpret_cast<int2*>(reinterpret_cast<unsigned char*>(ws) + 256);
  const int cap = 8 * n;
  if (hipMemsetAsync(ctl, 0, 256, st) != hipSuccess) return NIF_ERR_LAUNCH;
  hipLaunchKernelGGL(lap_cert_arcs_kernel, dim3((n + 3) / 4), dim3(256), 0, st, C, ldc, n, col4row, row4col, u, v, tol_rel,
                     ctl, arcs, cap);
  const size_t lds = (size_t)(n + 2) * sizeof(int);
  hipLaunchKernelGGL(lap_cert_acyclic_kernel, dim3(1), dim3(1024), lds, st, n, ctl, arcs, cap);
  if (hipMemcpyAsync(result, ctl + 2, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess) return NIF_ERR_LAUNCH;
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
