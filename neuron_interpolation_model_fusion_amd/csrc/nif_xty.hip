// nif_xty.hip -- column cross-Gram G = A^T B over the sample rows, for gfx950.
//
// The contraction axis is the ROW axis of the row-major activation matrices, so one output
// tile is a sum over m rows: the row range is split over workgroups (split-K), each workgroup
// owns one output tile x one row block and writes its partial tile to a slab; a second kernel
// sums the slabs in a fixed order in f64 (bitwise reproducible, no atomics), mirrors the lower
// triangle in symmetric mode and reduces the column-norm partials in the same launch.
//
// MFMA mapping (wave64):
//   f32 accumulate: v_mfma_f32_32x32x2_f32.  Lane l feeds A[i = l&31][k = l>>5]: with row-major
//     activations, lanes 0..31 read 32 consecutive columns of sample row k and lanes 32..63 the
//     same columns of row k+1 -- the layout the tiles already have in HBM and in LDS, so there
//     is no transpose anywhere.
//   f64 accumulate: v_mfma_f64_16x16x4_f64, lane l feeds A[i = l&15][k = l>>4].
// A workgroup is 4 waves (2 x 2); a wave owns (32 FI) x (32 FJ) outputs, the workgroup tile is
// (64 FI) x (64 FJ): FI = FJ = 2 (128 x 128, 64 accumulator registers, 4 waves/SIMD) or, f32
// only, FI = FJ = 3 (192 x 192, 144 accumulator registers, 2 waves/SIMD: 9 MFMAs per 6 operand
// reads and per barrier instead of 4 per 4).  Tiles are staged through LDS as [BK = 16][TM + 16]
// images (the +16 keeps the 16-lane groups of the f64 form on distinct banks), double-buffered;
// the global loads of tile t+1 are issued before the MFMAs of tile t and its LDS writes and
// norm updates are spread between the MFMA groups of tile t (one barrier per tile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

constexpr int BK = 16;
constexpr int NT = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct XtyArgs {
  const void* A;
  const void* B;
  int64_t lda, ldb, m, rows_per_split;
  int n0, n1, n0p, n1p;
  int tiles_i, tiles_j, ntiles, splits;
  int flags;
  int vec_a, vec_b;  // 16-byte vector loads are legal for this operand
  void* slab;        // [splits][n0p][n1p] accumulator type
  void* nslab_a;     // [splits][n0p] partial squared column norms, accumulator type
  void* nslab_b;     // [splits][n1p]
};

template <typename T>
struct V4 {
  T v[4];
};

template <typename T>
__device__ __forceinline__ void load4(V4<T>& x, const T* __restrict__ p) {
  if constexpr (sizeof(T) == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    x.v[0] = q.x; x.v[1] = q.y; x.v[2] = q.z; x.v[3] = q.w;
  } else {
    const double2 q0 = *reinterpret_cast<const double2*>(p);
    const double2 q1 = *reinterpret_cast<const double2*>(p + 2);
    x.v[0] = q0.x; x.v[1] = q0.y; x.v[2] = q1.x; x.v[3] = q1.y;
  }
}

// ---- global -> register staging of one [BK][W] operand tile ------------------------------------
// W = tile width in columns.  Row-major operand: element (k, c) = P[(k0 + k) * ld + c0 + c]; the
// tile is BK * W / 4 four-element pieces, piece idx = tid + 256 h -> row idx / (W/4), columns
// 4 (idx % (W/4)) .. +3.  Interior tiles (all rows and columns in range, 16-byte loads legal)
// take a branch-free path; that test is block-uniform (a scalar branch).
template <typename T, int W, bool RELU>
struct StageRows {
  static constexpr int C4 = W / 4;
  static constexpr int PASSES = BK * C4 / NT;
  static_assert(BK * C4 % NT == 0, "tile must split evenly over the workgroup");
  static constexpr int PARTS = BK;  // norm partials per column: one per tile row

  static __device__ __forceinline__ void load(V4<T> (&r)[PASSES], const T* __restrict__ P, int64_t ld, int64_t k0,
                                              int64_t kend, int c0, int n, int vec_ok, int tid) {
    const bool interior = vec_ok && (c0 + W <= n) && (k0 + BK <= kend);
    if (interior) {
#pragma unroll
      for (int h = 0; h < PASSES; ++h) {
        const int idx = tid + NT * h;
        load4(r[h], P + (k0 + idx / C4) * ld + c0 + 4 * (idx % C4));
      }
    } else {
#pragma unroll
      for (int h = 0; h < PASSES; ++h) {
        const int idx = tid + NT * h;
        const int64_t row = k0 + idx / C4;
        const int col = c0 + 4 * (idx % C4);
        V4<T> x;
        x.v[0] = x.v[1] = x.v[2] = x.v[3] = T(0);
        if (row < kend && col < n) {
          const T* p = P + row * ld + col;
          if (vec_ok && col + 3 < n) {
            load4(x, p);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < n) x.v[e] = p[e];
          }
        }
        r[h] = x;
      }
    }
  }

  // ReLU on load (Git Re-Basin's post-activation Gram): applied when the staged registers are COMMITTED, not where
  // they are loaded -- clamping right behind the loads made the kernel wait for them at once, and the tile's global
  // latency, which the MFMA groups between load and commit are there to cover, was exposed (C3: 103 -> 121 TF)
  static __device__ __forceinline__ void activate(V4<T> (&r)[PASSES]) {
    if (RELU) {
#pragma unroll
      for (int h = 0; h < PASSES; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) r[h].v[e] = r[h].v[e] > T(0) ? r[h].v[e] : T(0);
    }
  }

  template <int LD>
  static __device__ __forceinline__ void put(T* __restrict__ L, const V4<T> (&r)[PASSES], int tid) {
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
      const int idx = tid + NT * h;
      T* d = L + (idx / C4) * LD + 4 * (idx % C4);  // 16-byte aligned: LD and 4 are multiples of 4 elements
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(d) = make_float4(r[h].v[0], r[h].v[1], r[h].v[2], r[h].v[3]);
      } else {
        *reinterpret_cast<double2*>(d) = make_double2(r[h].v[0], r[h].v[1]);
        *reinterpret_cast<double2*>(d + 2) = make_double2(r[h].v[2], r[h].v[3]);
      }
    }
  }

  // squared-norm partials: sq[h][e] belongs to column 4 (idx % C4) + e, part idx / C4
  template <typename acc_t>
  static __device__ __forceinline__ void norms(acc_t (&sq)[PASSES][4], const V4<T> (&r)[PASSES]) {
#pragma unroll
    for (int h = 0; h < PASSES; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) sq[h][e] += (acc_t)r[h].v[e] * (acc_t)r[h].v[e];
  }
  template <typename acc_t>
  static __device__ __forceinline__ void norms_to_lds(acc_t* __restrict__ red, const acc_t (&sq)[PASSES][4], int tid) {
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
      const int idx = tid + NT * h;
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(idx / C4) * W + 4 * (idx % C4) + e] = sq[h][e];
    }
  }
};

// Transposed operand (stored n x m): element (k, c) = P[(c0 + c) * ld + k0 + k]; pieces run along
// k: piece idx -> column idx / (BK/4), k = 4 (idx % (BK/4)) .. +3.
template <typename T, int W, bool RELU>
struct StageCols {
  static constexpr int K4 = BK / 4;
  static constexpr int PASSES = W * K4 / NT;
  static_assert(W * K4 % NT == 0, "tile must split evenly over the workgroup");
  static constexpr int PARTS = K4;

  static __device__ __forceinline__ void load(V4<T> (&r)[PASSES], const T* __restrict__ P, int64_t ld, int64_t k0,
                                              int64_t kend, int c0, int n, int vec_ok, int tid) {
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
      const int idx = tid + NT * h;
      const int c = c0 + idx / K4;
      const int64_t k = k0 + 4 * (idx % K4);
      V4<T> x;
      x.v[0] = x.v[1] = x.v[2] = x.v[3] = T(0);
      if (c < n && k < kend) {
        const T* p = P + (int64_t)c * ld + k;
        if (vec_ok && k + 3 < kend) {
          load4(x, p);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < kend) x.v[e] = p[e];
        }
      }
      r[h] = x;
    }
  }

  static __device__ __forceinline__ void activate(V4<T> (&)[PASSES]) {}  // ReLU on load: row-major operands only

  template <int LD>
  static __device__ __forceinline__ void put(T* __restrict__ L, const V4<T> (&r)[PASSES], int tid) {
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
      const int idx = tid + NT * h;
#pragma unroll
      for (int e = 0; e < 4; ++e) L[(4 * (idx % K4) + e) * LD + idx / K4] = r[h].v[e];
    }
  }

  // sq[h][0] sums the piece's four k; column idx / K4, part idx % K4
  template <typename acc_t>
  static __device__ __forceinline__ void norms(acc_t (&sq)[PASSES][4], const V4<T> (&r)[PASSES]) {
#pragma unroll
    for (int h = 0; h < PASSES; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) sq[h][0] += (acc_t)r[h].v[e] * (acc_t)r[h].v[e];
  }
  template <typename acc_t>
  static __device__ __forceinline__ void norms_to_lds(acc_t* __restrict__ red, const acc_t (&sq)[PASSES][4], int tid) {
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
      const int idx = tid + NT * h;
      red[(idx % K4) * W + idx / K4] = sq[h][0];
    }
  }
};

// ---- matrix-core micro kernels: a wave's (32 FI) x (32 FJ) tile ----------------------------------

template <int FI, int FJ>
struct MmaF32 {
  typedef float acc_t;
  typedef f32x16 frag_t;
  static constexpr int RI = FI, RJ = FJ;  // 32-wide fragments
  static constexpr int KSTEPS = BK / 2;
  typedef float op_t;
  // operand fragments of k-step kk: lane l reads column (l & 31) + 32 i of tile row 2 kk + (l >> 5)
  template <typename T, int LDA, int LDB>
  static __device__ __forceinline__ void load(const T* __restrict__ la, const T* __restrict__ lb, int kk,
                                              op_t (&a)[RI], op_t (&b)[RJ], int lane) {
    const int krow = 2 * kk + (lane >> 5);
#pragma unroll
    for (int i = 0; i < RI; ++i) a[i] = (float)la[krow * LDA + (lane & 31) + 32 * i];
#pragma unroll
    for (int j = 0; j < RJ; ++j) b[j] = (float)lb[krow * LDB + (lane & 31) + 32 * j];
  }
  static __device__ __forceinline__ void mma(const op_t (&a)[RI], const op_t (&b)[RJ], frag_t (&acc)[RI][RJ]) {
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  static __device__ __forceinline__ void zero(frag_t (&acc)[RI][RJ]) {
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < RJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
  // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  static __device__ __forceinline__ void store(acc_t* __restrict__ S, int ld, const frag_t (&acc)[RI][RJ], int lane) {
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < RJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          S[(int64_t)row * ld + 32 * j + (lane & 31)] = acc[i][j][e];
        }
  }
};

template <int FI, int FJ>
struct MmaF64 {
  typedef double acc_t;
  typedef f64x4 frag_t;
  static constexpr int RI = 2 * FI, RJ = 2 * FJ;  // 16-wide fragments
  static constexpr int KSTEPS = BK / 4;
  typedef double op_t;
  template <typename T, int LDA, int LDB>
  static __device__ __forceinline__ void load(const T* __restrict__ la, const T* __restrict__ lb, int kk,
                                              op_t (&a)[RI], op_t (&b)[RJ], int lane) {
    const int krow = 4 * kk + (lane >> 4);
#pragma unroll
    for (int i = 0; i < RI; ++i) a[i] = (double)la[krow * LDA + (lane & 15) + 16 * i];
#pragma unroll
    for (int j = 0; j < RJ; ++j) b[j] = (double)lb[krow * LDB + (lane & 15) + 16 * j];
  }
  static __device__ __forceinline__ void mma(const op_t (&a)[RI], const op_t (&b)[RJ], frag_t (&acc)[RI][RJ]) {
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  static __device__ __forceinline__ void zero(frag_t (&acc)[RI][RJ]) {
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < RJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
  }
  // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ void store(acc_t* __restrict__ S, int ld, const frag_t (&acc)[RI][RJ], int lane) {
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < RJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * i + (lane >> 4) + 4 * e;
          S[(int64_t)row * ld + 16 * j + (lane & 15)] = acc[i][j][e];
        }
  }
};

// ---- main kernel -----------------------------------------------------------------------------------

// waves per SIMD the register budget is sized for: 3 for the 128-wide f32 tile (64 accumulator registers; at 4 waves
// = 128 registers it spilled 60-76 bytes per lane and was no faster: 105 vs 113 TF at n = 6144, 102 vs 106 at 1024),
// 2 for the 192-wide f32 tile and for the f64 forms
template <template <int, int> class MmaT, int FI>
constexpr int xty_waves_per_simd() {
  return (sizeof(typename MmaT<FI, FI>::acc_t) == 4 && FI == 2) ? 3 : 2;
}

template <typename Tin, template <int, int> class MmaT, int FI, int FJ, bool RELU_A, bool RELU_B, bool TA, bool TB>
__global__ __launch_bounds__(NT, (xty_waves_per_simd<MmaT, FI>())) void xty_kernel(const XtyArgs p) {
  typedef MmaT<FI, FJ> Mma;
  typedef typename Mma::acc_t acc_t;
  constexpr int TM = 64 * FI, TN = 64 * FJ;
  constexpr int LDA = TM + 16, LDB = TN + 16;
  constexpr int TILE_A = BK * LDA, TILE_B = BK * LDB;
  typedef typename std::conditional<TA, StageCols<Tin, TM, RELU_A>, StageRows<Tin, TM, RELU_A>>::type SA;
  typedef typename std::conditional<TB, StageCols<Tin, TN, RELU_B>, StageRows<Tin, TN, RELU_B>>::type SB;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Tin* smem = reinterpret_cast<Tin*>(smem_raw);  // [buf]{A[BK][LDA], B[BK][LDB]}

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;

  // XCD-aware id remap (bijective for any grid size): blocks b and b + 8 share an XCD, so give
  // each XCD a contiguous range of ids; consecutive ids are the tiles of one row block and reuse
  // its A / B panels out of that XCD's L2.
  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int xcd = orig & 7;
  const int q = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);

  const int split = wg / p.ntiles;
  int tidx = wg - split * p.ntiles;
  int ti, tj;
  if (p.flags & NIF_SYMMETRIC) {
    ti = 0;
    int cnt = p.tiles_j;
    while (tidx >= cnt) {
      tidx -= cnt;
      ++ti;
      --cnt;
    }
    tj = ti + tidx;
  } else {
    ti = tidx / p.tiles_j;
    tj = tidx - ti * p.tiles_j;
  }
  const int ci = ti * TM, cj = tj * TN;

  const int64_t k_begin = (int64_t)split * p.rows_per_split;
  int64_t k_end = k_begin + p.rows_per_split;
  if (k_end > p.m) k_end = p.m;
  const int nt = k_begin < k_end ? (int)((k_end - k_begin + BK - 1) / BK) : 0;

  const Tin* A = reinterpret_cast<const Tin*>(p.A);
  const Tin* B = reinterpret_cast<const Tin*>(p.B);

  typename Mma::frag_t acc[Mma::RI][Mma::RJ];
  Mma::zero(acc);

  // column-norm partials are only kept by the blocks that publish them: A's from tile column 0
  // (the diagonal in symmetric mode), B's from tile row 0
  const bool want_a = (p.flags & NIF_SYMMETRIC) ? (ti == tj) : (tj == 0);
  const bool want_b = !(p.flags & NIF_SYMMETRIC) && (ti == 0);
  acc_t sqa[SA::PASSES][4], sqb[SB::PASSES][4];
#pragma unroll
  for (int h = 0; h < SA::PASSES; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) sqa[h][e] = 0;
#pragma unroll
  for (int h = 0; h < SB::PASSES; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) sqb[h][e] = 0;

  // A's and B's staging registers have disjoint lifetimes in the main loop (A: loaded at the top of
  // an iteration, written to LDS at k-step KA; B: loaded right after that, written at KB2), so they
  // share physical registers.
  V4<Tin> ra[SA::PASSES], rb[SB::PASSES];
  auto stage_a = [&](int64_t k0) { SA::load(ra, A, p.lda, k0, k_end, ci, p.n0, p.vec_a, tid); };
  auto stage_b = [&](int64_t k0) { SB::load(rb, B, p.ldb, k0, k_end, cj, p.n1, p.vec_b, tid); };
  auto commit_a = [&](int buf) {
    SA::activate(ra);
    SA::template put<LDA>(smem + buf * (TILE_A + TILE_B), ra, tid);
    if (want_a) SA::norms(sqa, ra);
  };
  auto commit_b = [&](int buf) {
    SB::activate(rb);
    SB::template put<LDB>(smem + buf * (TILE_A + TILE_B) + TILE_A, rb, tid);
    if (want_b) SB::norms(sqb, rb);
  };

  if (nt > 0) {
    stage_a(k_begin);
    commit_a(0);
    stage_b(k_begin);
    commit_b(0);
  }
  __syncthreads();
  // Software pipeline inside every wave.  (1) All waves of a CU run this program in near lock-step,
  // so a separate staging phase would idle the matrix pipe in every wave at once: the LDS writes and
  // norm updates of tile t+1 sit between the MFMA groups of tile t.  (2) A wave issues in order, so
  // operand reads placed after an MFMA group only start once its last MFMA has ISSUED and their
  // latency is exposed: the fragments of k-step kk+1 are requested before the MFMAs of step kk.
  constexpr int KS = Mma::KSTEPS;
  constexpr int KA = KS / 4, KB2 = KS - 1;
  typename Mma::op_t fa[2][Mma::RI], fb[2][Mma::RJ];
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const bool more = t + 1 < nt;
    const int64_t knext = k_begin + (int64_t)(t + 1) * BK;
    if (more) stage_a(knext);
    const Tin* la = smem + buf * (TILE_A + TILE_B) + wi * (TM / 2);
    const Tin* lb = smem + buf * (TILE_A + TILE_B) + TILE_A + wj * (TN / 2);
    Mma::template load<Tin, LDA, LDB>(la, lb, 0, fa[0], fb[0], lane);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if (kk + 1 < KS) Mma::template load<Tin, LDA, LDB>(la, lb, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1], lane);
      if (kk == KA && more) {
        commit_a(buf ^ 1);
        stage_b(knext);
      }
      if (kk == KB2 && more) commit_b(buf ^ 1);
      Mma::mma(fa[kk & 1], fb[kk & 1], acc);
      __builtin_amdgcn_sched_barrier(0);  // keep the one-step prefetch distance: no hoisting of later reads
    }
    __syncthreads();
  }

  // partial tile -> slab (padded to whole tiles, so no guards)
  acc_t* S = reinterpret_cast<acc_t*>(p.slab) + ((int64_t)split * p.n0p + ci + wi * (TM / 2)) * p.n1p + cj + wj * (TN / 2);
  Mma::store(S, p.n1p, acc, lane);

  // column-norm partials, fixed summation order; the tile images are dead after the last barrier
  acc_t* red = reinterpret_cast<acc_t*>(smem_raw);
  if (want_a) {
    SA::norms_to_lds(red, sqa, tid);
    __syncthreads();
    if (tid < TM) {
      acc_t s = 0;
      for (int r = 0; r < SA::PARTS; ++r) s += red[r * TM + tid];
      reinterpret_cast<acc_t*>(p.nslab_a)[(int64_t)split * p.n0p + ci + tid] = s;
    }
    __syncthreads();
  }
  if (want_b) {
    SB::norms_to_lds(red, sqb, tid);
    __syncthreads();
    if (tid < TN) {
      acc_t s = 0;
      for (int r = 0; r < SB::PARTS; ++r) s += red[r * TN + tid];
      reinterpret_cast<acc_t*>(p.nslab_b)[(int64_t)split * p.n1p + cj + tid] = s;
    }
  }
}

// ---- split-bf16 variant (flag NIF_XTY_SPLIT_BF16) -----------------------------------------------------
// The f32 matrix instruction peaks at 157 TFLOP/s, the bf16 one at 2.5 PFLOP/s.  An f32 value is the exact sum
// of three bf16 values (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 significand bits), and a
// product needs the six cross terms down to 2^-16 (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi; the dropped
// ones are below one f32 rounding), each exact in the f32 accumulator of v_mfma_f32_32x32x16_bf16: six bf16
// matrix instructions of 16 samples each replace eight f32 ones of 2 samples -- 2.65x the f32 peak for a result
// that is as close to the exact Gram entry as the f32 fma chain is (tools/mfma_probe, tests).
//
// Data path: 16-byte global loads of the row-major f32 tile (a dword-per-lane load moves 16 B/clk/CU through the
// texture addresser whatever its coalescing; only dwordx4 reaches 64 B/clk) -> split in registers -> three
// row-major bf16 planes [16 samples][128 columns] in LDS -> ds_read_b64_tr_b16, the transposing read of gfx950,
// hands every lane the 8 consecutive samples of its column that the 32x32x16 operand layout wants.  Plane rows
// are padded by 64 B so that the four rows one transposed read gathers fall on different banks.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ unsigned pack_bf16_rn(float a, float b) {
  f32x2 v = {a, b};
  union {
    bf16x2 h;
    unsigned u;
  } r;
  r.h = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32, round to nearest even
  return r.u;
}
// single-instruction f32 helpers: left to itself the compiler packs neighbouring f32 operations into v_pk_*_f32,
// which stall the matrix pipe they are meant to run beside (MI355X_MICROARCH.md, "price of one filler")
__device__ __forceinline__ float sub_f32(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void fmac_f32(float& acc, float a, float b) {
  asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}

template <int FI, bool RELU_A, bool RELU_B>
__global__ __launch_bounds__(NT, (FI == 3 ? 1 : 2)) void xty_bf16_kernel(const XtyArgs p) {
  constexpr int TM = 64 * FI;
  constexpr int PITCH = TM * 2 + 64;   // bytes per plane row: the pad keeps the four rows a transposed read
                                       // gathers on different banks (pitch = 64 or 192 mod 256), no swizzle
  constexpr int PLANE = BK * PITCH;    // one split plane of one operand tile: [16 samples][TM columns] bf16
  constexpr int OPB = 3 * PLANE;       // one operand: hi, mid, lo
  constexpr int BUFB = 2 * OPB;        // one buffer: A, B
  constexpr int NP = TM / 64;          // 16-byte pieces per thread and operand tile
  constexpr int RC4 = TM / 4;          // pieces per tile row
  typedef MmaF32<FI, FI> Acc;          // same 32x32 accumulator layout
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;

  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int xcd = orig & 7;
  const int q8 = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (orig >> 3);

  const int split = wg / p.ntiles;
  int tidx = wg - split * p.ntiles;
  int ti, tj;
  if (p.flags & NIF_SYMMETRIC) {
    ti = 0;
    int cnt = p.tiles_j;
    while (tidx >= cnt) {
      tidx -= cnt;
      ++ti;
      --cnt;
    }
    tj = ti + tidx;
  } else {
    ti = tidx / p.tiles_j;
    tj = tidx - ti * p.tiles_j;
  }
  const int ci = ti * TM, cj = tj * TM;

  const int64_t k_begin = (int64_t)split * p.rows_per_split;
  int64_t k_end = k_begin + p.rows_per_split;
  if (k_end > p.m) k_end = p.m;
  const int nt = k_begin < k_end ? (int)((k_end - k_begin + BK - 1) / BK) : 0;
  const int ntf = k_begin < k_end ? (int)((k_end - k_begin) / BK) : 0;  // full tiles; nt - ntf is 0 or 1
  const int last = ntf - 1;

  const float* A = reinterpret_cast<const float*>(p.A);
  const float* B = reinterpret_cast<const float*>(p.B);

  typename Acc::frag_t acc[FI][FI];
  Acc::zero(acc);

  const bool want_a = (p.flags & NIF_SYMMETRIC) ? (ti == tj) : (tj == 0);
  const bool want_b = !(p.flags & NIF_SYMMETRIC) && (ti == 0);

  // Staging map: piece q = tid + 256 h of a [16][TM] tile = row q / (TM/4), columns 4 (q % (TM/4)) .. + 3, the same
  // for both operands.  Columns past the matrix edge re-read the last valid piece: the padded rows / columns of
  // the slab then hold finite garbage that nothing reads, and the loop needs no guards.
  unsigned goffa[NP], goffb[NP], woff[NP];
  int prow[NP], pcol[NP];
#pragma unroll
  for (int h = 0; h < NP; ++h) {
    const int q = tid + NT * h;
    prow[h] = q / RC4;
    pcol[h] = 4 * (q - prow[h] * RC4);
    goffa[h] = (unsigned)((prow[h] * p.lda + min(ci + pcol[h], p.n0 - 4)) * 4);
    goffb[h] = (unsigned)((prow[h] * p.ldb + min(cj + pcol[h], p.n1 - 4)) * 4);
    woff[h] = (unsigned)(prow[h] * PITCH + pcol[h] * 2);
  }
  const char* rowa = reinterpret_cast<const char*>(A + k_begin * p.lda);  // scalar
  const char* rowb = reinterpret_cast<const char*>(B + k_begin * p.ldb);
  const int64_t tile_a = (int64_t)BK * p.lda * 4, tile_b = (int64_t)BK * p.ldb * 4;
  float sqa[NP][4], sqb[NP][4];
#pragma unroll
  for (int h = 0; h < NP; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) sqa[h][e] = sqb[h][e] = 0.f;

  // full tile tt (samples k_begin + 16 tt .. + 15) -> registers.  The empty asm keeps the 32-bit offsets opaque
  // inside the loop: hoisted, their zero-extensions would turn the scalar-base addressing into 64-bit vector adds.
  auto stage = [&](float4 (&x)[2][NP], int tt) {
    const char* ba = rowa + tt * tile_a;
    const char* bb = rowb + tt * tile_b;
#pragma unroll
    for (int h = 0; h < NP; ++h) {
      unsigned oa = goffa[h], ob = goffb[h];
      asm volatile("" : "+v"(oa), "+v"(ob));
      x[0][h] = *reinterpret_cast<const float4*>(ba + oa);
      x[1][h] = *reinterpret_cast<const float4*>(bb + ob);
    }
  };
  // the last, partial tile of the last split: rows guarded
  auto stage_tail = [&](float4 (&x)[2][NP], int tt) {
    const int64_t k0 = k_begin + (int64_t)tt * BK;
    const char* ba = rowa + tt * tile_a;
    const char* bb = rowb + tt * tile_b;
#pragma unroll
    for (int h = 0; h < NP; ++h) {
      const bool rok = k0 + prow[h] < k_end;
      x[0][h] = rok ? *reinterpret_cast<const float4*>(ba + goffa[h]) : make_float4(0.f, 0.f, 0.f, 0.f);
      x[1][h] = rok ? *reinterpret_cast<const float4*>(bb + goffb[h]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // one 16-byte piece -> 8 bytes in each of the three planes
  auto split_put = [&](const float4& xi, bool relu, float (&sq)[4], unsigned char* dst) {
    float v[4] = {xi.x, xi.y, xi.z, xi.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (relu) v[e] = fmaxf(v[e], 0.f);
      fmac_f32(sq[e], v[e], v[e]);
    }
#pragma unroll
    for (int sidx = 0; sidx < 3; ++sidx) {
      const unsigned p01 = pack_bf16_rn(v[0], v[1]);
      const unsigned p23 = pack_bf16_rn(v[2], v[3]);
      *reinterpret_cast<uint2*>(dst + sidx * PLANE) = make_uint2(p01, p23);
      if (sidx < 2) {  // exact residuals
        v[0] = sub_f32(v[0], __uint_as_float(p01 << 16));
        v[1] = sub_f32(v[1], __uint_as_float(p01 & 0xffff0000u));
        v[2] = sub_f32(v[2], __uint_as_float(p23 << 16));
        v[3] = sub_f32(v[3], __uint_as_float(p23 & 0xffff0000u));
      }
    }
  };
  auto commit = [&](const float4 (&x)[2][NP], int buf) {
    unsigned char* L = smem_raw + buf * BUFB;
#pragma unroll
    for (int h = 0; h < NP; ++h) {
      split_put(x[0][h], RELU_A, sqa[h], L + woff[h]);
      split_put(x[1][h], RELU_B, sqb[h], L + OPB + woff[h]);
    }
  };

  // transposed-read base addresses: lane (column l & 31 of a 32-wide fragment, k half l >> 5) takes rows
  // 8 (l >> 5) + 4 jj + 0..3; inside its 16-lane group, lane 4 q + pp supplies row q, columns 4 pp .. + 3.
  // Fragment i, row block jj, plane and buffer are immediates on top of one base per operand.
  unsigned rbase_a, rbase_b;
  {
    const int kh = lane >> 5, g1 = (lane >> 4) & 1, l16 = lane & 15, q = l16 >> 2, pp = l16 & 3;
    const int rowoff = (8 * kh + q) * PITCH + (16 * g1 + 4 * pp) * 2;
    rbase_a = (unsigned)(rowoff + wi * (TM / 2) * 2);
    rbase_b = (unsigned)(OPB + rowoff + wj * (TM / 2) * 2);
  }
  auto frag8 = [&](int buf, unsigned rbase, int i, int sidx) -> bf16x8 {
    const unsigned char* base = smem_raw + rbase + buf * BUFB + sidx * PLANE + i * 64;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * PITCH));
    union {
      s16x8 s;
      bf16x8 v;
    } u;
    u.s = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return u.v;
  };
  // the 6 FI^2 matrix instructions of one tile, term by term (smallest first) so that consecutive ones hit
  // different accumulators; `mid` (the split of a later tile) is spread between them by the issue-order directives
  auto mma_tile = [&](int buf, auto&& mid) {
    bf16x8 a[3][FI], b[3][FI];
#pragma unroll
    for (int sidx = 0; sidx < 3; ++sidx)
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        a[sidx][i] = frag8(buf, rbase_a, i, sidx);
        b[sidx][i] = frag8(buf, rbase_b, i, sidx);
      }
    constexpr int TA_[6] = {2, 0, 1, 1, 0, 0};
    constexpr int TB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < 6; ++term) {
      if (term == 1) mid();
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA_[term]][i], b[TB_[term]][j], acc[i][j], 0, 0, 0);
    }
    // Issue order inside this block: the split arithmetic of the next tile is spread through the matrix
    // instructions (a wave issues in order; a 32x32x16 instruction occupies the matrix pipe for 32 cycles but
    // the issue port for 8, so a few vector instructions fit in its shadow) instead of sitting in one lump
    // during which the matrix pipe drains.
    __builtin_amdgcn_sched_group_barrier(0x100, 12 * FI, 0);  // the fragment reads first
#pragma unroll
    for (int q = 0; q < 6 * FI * FI; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, FI == 3 ? 4 : 5, 0);  // a few VALU
      if (q % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // one LDS write
    }
  };

  // Two register sets: while tile t computes, tile t + 1 is committed from one set and tile t + 2 is in flight
  // into the other (a tile of bf16 matrix work is a quarter of the f32 kernel's: a distance of one is too short).
  float4 xa[2][NP], xb[2][NP];
  if (ntf > 0) {
    stage(xa, 0);
    stage(xb, min(1, last));
    commit(xa, 0);
  }
  __syncthreads();
  // tile t reads buffer t & 1 and commits tile t + 1 (unconditionally: one basic block); loads past the end
  // re-read the last tile and are never committed.  The last full tile runs after the loop without a commit.
  for (int t = 0; t < last; t += 2) {
    stage(xa, min(t + 2, last));
    mma_tile(0, [&] { commit(xb, 1); });
    __syncthreads();
    if (t + 1 >= last) break;
    stage(xb, min(t + 3, last));
    mma_tile(1, [&] { commit(xa, 0); });
    __syncthreads();
  }
  if (ntf > 0) {
    mma_tile(last & 1, [] {});
    __syncthreads();
  }
  if (nt > ntf) {  // block-uniform: the partial tile, not pipelined (at most once per launch and split)
    stage_tail(xa, ntf);
    commit(xa, 0);
    __syncthreads();
    mma_tile(0, [] {});
    __syncthreads();
  }

  float* S = reinterpret_cast<float*>(p.slab) + ((int64_t)split * p.n0p + ci + wi * (TM / 2)) * p.n1p + cj + wj * (TM / 2);
  Acc::store(S, p.n1p, acc, lane);

  // column-norm partials: every (tile row, column) slot of the [16][TM] image belongs to one thread and piece
  float* red = reinterpret_cast<float*>(smem_raw);  // [operand][16][TM]; the tile images are dead
#pragma unroll
  for (int h = 0; h < NP; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[prow[h] * TM + pcol[h] + e] = sqa[h][e];
      red[(BK + prow[h]) * TM + pcol[h] + e] = sqb[h][e];
    }
  __syncthreads();
  if (tid < TM) {
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int r = 0; r < BK; ++r) {
      sa += red[r * TM + tid];
      sb += red[(BK + r) * TM + tid];
    }
    if (want_a) reinterpret_cast<float*>(p.nslab_a)[(int64_t)split * p.n0p + ci + tid] = sa;
    if (want_b) reinterpret_cast<float*>(p.nslab_b)[(int64_t)split * p.n1p + cj + tid] = sb;
  }
}

// ---- slab + norm reduction, one launch --------------------------------------------------------------
// blockIdx.y < n0: row i of G (threads along j); blockIdx.y == n0: na; == n0 + 1: nb.
// OT: storage type of G (double; float for the k-means storage, whose exact f64 diagonal goes to gdiag).
template <typename acc_t, typename OT>
__global__ void xty_reduce_kernel(const acc_t* __restrict__ slab, const acc_t* __restrict__ nsa,
                                  const acc_t* __restrict__ nsb, int splits, int n0, int n1, int n0p, int n1p,
                                  int tm, int symmetric, OT* __restrict__ G, int64_t ldg,
                                  double* __restrict__ na, double* __restrict__ nb, int ybase,
                                  double* __restrict__ gdiag) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y + ybase;  // ybase = n0: the two norm rows only
  if (i < n0) {
    if (j >= n1) return;
    // symmetric: the lower triangle is the mirror of the upper one, inside diagonal tiles too (the split-bf16
    // kernel's term order is not symmetric in i and j).  The thread of (i, j), i <= j, writes both entries: the mirror
    // costs one strided store instead of `splits` strided loads (the lower triangle's threads used to read the slab
    // down a column, one cache line per lane and split).
    if (symmetric && i > j) return;
    const acc_t* s = slab + (int64_t)i * n1p + j;
    const int64_t stride = (int64_t)n0p * n1p;
    // eight chains: fixed order, eight loads in flight per thread (the pass is a latency-bound stream over
    // splits x n0p x n1p partials -- 113 MB at the C2 level -- and four in flight left it at half the bandwidth)
    double c[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int k = 0;
    for (; k + 7 < splits; k += 8) {
      acc_t t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = s[(k + q) * stride];
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] += (double)t[q];
    }
    for (; k < splits; ++k) c[0] += (double)s[k * stride];
    const double g = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    G[(int64_t)i * ldg + j] = (OT)g;
    if (symmetric && i != j) G[(int64_t)j * ldg + i] = (OT)g;
    if (gdiag && i == j) gdiag[i] = g;
  } else {
    const bool is_a = i == n0;
    double* out = is_a ? na : nb;
    const int n = is_a ? n0 : n1;
    if (out == nullptr || j >= n) return;
    const acc_t* ns = (is_a || symmetric) ? nsa : nsb;
    const int npad = (is_a || symmetric) ? n0p : n1p;
    double sum = 0.0;  // one chain, in split order; the loads go out eight at a time
    int k = 0;
    for (; k + 7 < splits; k += 8) {
      acc_t t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = ns[(int64_t)(k + q) * npad + j];
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += (double)t[q];
    }
    for (; k < splits; ++k) sum += (double)ns[(int64_t)k * npad + j];
    out[j] = sum;
  }
}

// Symmetric mode for large n: one workgroup per 64 x 64 tile of the upper triangle sums the tile into LDS and writes
// it twice, as it stands and transposed, both in full rows -- the mirror is then a coalesced store (the general kernel's
// mirror is one 8-byte store per cache line: 4.2 ms of a 5 ms reduction at n = 24 576).  Same chains per entry.
template <typename acc_t, typename OT>
__global__ __launch_bounds__(256) void xty_reduce_sym_kernel(const acc_t* __restrict__ slab, int splits, int n, int np,
                                                             OT* __restrict__ G, int64_t ldg, double* __restrict__ gdiag) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) return;
  __shared__ double tile[64][65];
  const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  const int64_t stride = (int64_t)np * np;
  const int j = bj * 64 + c;
#pragma unroll 4
  for (int q = 0; q < 16; ++q) {
    const int r = r0 + 4 * q, i = bi * 64 + r;
    double g = 0.0;
    if (i < n && j < n) {
      const acc_t* s = slab + (int64_t)i * np + j;
      double ch[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      int k = 0;
      for (; k + 7 < splits; k += 8) {
        acc_t t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = s[(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) ch[u] += (double)t[u];
      }
      for (; k < splits; ++k) ch[0] += (double)s[k * stride];
      g = ((ch[0] + ch[1]) + (ch[2] + ch[3])) + ((ch[4] + ch[5]) + (ch[6] + ch[7]));
    }
    tile[r][c] = g;
  }
  __syncthreads();
#pragma unroll 4
  for (int q = 0; q < 16; ++q) {
    const int r = r0 + 4 * q;
    const int i = bi * 64 + r;
    // the tile as it stands; inside a diagonal tile the lower entries mirror the upper ones
    if (i < n && j < n) {
      G[(int64_t)i * ldg + j] = (OT)((bi == bj && r > c) ? tile[c][r] : tile[r][c]);
      if (gdiag && bi == bj && r == c) gdiag[i] = tile[r][c];
    }
    // and transposed: row bj * 64 + r of G, columns of block bi
    if (bi != bj) {
      const int mi = bj * 64 + r, mj = bi * 64 + c;
      if (mi < n && mj < n) G[(int64_t)mi * ldg + mj] = (OT)tile[c][r];
    }
  }
}

// ---- packed block upper triangle: what a sample-sharded run sends through its all-reduce ----------------------------
// Rows are grouped in block rows of `block` rows; block row b (first row i0 = b * block) keeps its columns from i0 on,
// row-major with row length n - i0, block rows back to back: the diagonal blocks in full, nothing left of them --
// about half of the n^2 entries for n >> block.  The partial Grams of the ranks are summed in this layout and mirrored
// afterwards (xty_unpack_upper_kernel): entry for entry the result of reducing the full matrix.
__host__ __device__ __forceinline__ int64_t packed_upper_offset(int i, int j, int n, int block) {
  const int64_t b = i / block, i0 = b * block;
  const int64_t base = (int64_t)block * (b * (int64_t)n - (int64_t)block * (b * (b - 1) / 2));
  return base + (int64_t)(i - i0) * (n - i0) + (j - i0);
}

// split-K reduction of a symmetric Gram straight into the packed layout (same chains per entry as xty_reduce_kernel)
template <typename acc_t>
__global__ void xty_reduce_packed_kernel(const acc_t* __restrict__ slab, int splits, int n, int np, int block,
                                         double* __restrict__ packed) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n || i > j) return;
  const acc_t* s = slab + (int64_t)i * np + j;
  const int64_t stride = (int64_t)np * np;
  double c[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int k = 0;
  for (; k + 7 < splits; k += 8) {
    acc_t t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = s[(k + q) * stride];
#pragma unroll
    for (int q = 0; q < 8; ++q) c[q] += (double)t[q];
  }
  for (; k < splits; ++k) c[0] += (double)s[k * stride];
  const double g = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
  packed[packed_upper_offset(i, j, n, block)] = g;
  if (i != j && j / block == i / block) packed[packed_upper_offset(j, i, n, block)] = g;  // inside a diagonal block
}

// G (n x n, f64 or f32 storage) from the packed layout; gdiag (optional): the diagonal in f64
template <typename OT>
__global__ void xty_unpack_upper_kernel(const double* __restrict__ packed, int n, int block, OT* __restrict__ G,
                                        int64_t ldg, double* __restrict__ gdiag) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n) return;
  const int i0 = i / block * block;
  // left of the block row's first column the entry lives in the block row of j, transposed
  const double g = j >= i0 ? packed[packed_upper_offset(i, j, n, block)] : packed[packed_upper_offset(j, i, n, block)];
  G[(int64_t)i * ldg + j] = (OT)g;
  if (gdiag && i == j) gdiag[i] = g;
}

// ---- host side -------------------------------------------------------------------------------------

struct XtyPlan {
  int f;  // fragments per wave per axis: 2 (128-wide tiles) or 3 (192-wide)
  int tm, tiles_i, tiles_j, ntiles, splits, n0p, n1p, slots;
  int64_t rows_per_split;
  size_t slab_bytes, nsa_bytes, nsb_bytes, lds_bytes;
};

static XtyPlan plan_for(int f, int64_t m, int n0, int n1, int in_f64, int acc, int flags, int slot_div = 1,
                        int force_splits = 0) {
  XtyPlan pl;
  pl.f = f;
  pl.tm = 64 * f;
  pl.tiles_i = (n0 + pl.tm - 1) / pl.tm;
  pl.tiles_j = (n1 + pl.tm - 1) / pl.tm;
  pl.n0p = pl.tiles_i * pl.tm;
  pl.n1p = pl.tiles_j * pl.tm;
  pl.ntiles = (flags & NIF_SYMMETRIC) ? pl.tiles_i * (pl.tiles_i + 1) / 2 : pl.tiles_i * pl.tiles_j;
  // fill the chip (256 CUs x resident workgroups per CU) without leaving a second, nearly empty
  // round of workgroups; keep >= 4 K-tiles per row block
  const bool split_bf16 = (flags & NIF_XTY_SPLIT_BF16) != 0;
  pl.slots = 256 * (split_bf16 ? (f == 3 ? 1 : 2) : (f == 3 ? 2 : (in_f64 || acc == NIF_F64 ? 2 : 3)));
  pl.slots /= slot_div;
  static const int slots_override = env_int("NIF_XTY_SLOTS");  // tuning aid: workgroups the split-K aims for
  if (slots_override >= 1) pl.slots = slots_override;
  int64_t target = pl.slots;
  if (pl.ntiles > target) target = pl.ntiles;
  int64_t splits = force_splits > 0 ? force_splits : target / pl.ntiles;
  const int64_t max_splits = (m + 4 * BK - 1) / (4 * BK);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int64_t rps = (m + splits - 1) / splits;
  rps = (rps + BK - 1) / BK * BK;
  if (rps < BK) rps = BK;
  splits = (m + rps - 1) / rps;
  if (splits < 1) splits = 1;
  pl.splits = (int)splits;
  pl.rows_per_split = rps;
  const size_t esz = acc == NIF_F64 ? 8 : 4;
  pl.slab_bytes = (size_t)pl.splits * pl.n0p * pl.n1p * esz;
  pl.nsa_bytes = (size_t)pl.splits * pl.n0p * esz;
  pl.nsb_bytes = (size_t)pl.splits * pl.n1p * esz;
  const size_t tiles = (size_t)2 * 2 * BK * (pl.tm + 16) * (in_f64 ? 8 : 4);
  const size_t red = (size_t)BK * pl.tm * esz;  // norm reduction scratch aliases the tile images
  pl.lds_bytes = tiles > red ? tiles : red;
  if (split_bf16) pl.lds_bytes = (size_t)2 * 2 * 3 * BK * (pl.tm * 2 + 64);  // [buf][operand][3 planes][16][tm + pad] bf16
  return pl;
}

// Estimated duration in seconds of the Gram kernel plus the slab reduction that follows it.  Rates are the
// measured ones on MI355X (tools/xty_bench.py): 192-wide f32 tiles 114 TFLOP/s, 128-wide 102, f64 46, with all
// resident slots filled; leaving half / a quarter of the slots of a CU empty costs 4.5 % / 12 % of that rate,
// but halves / quarters the split-K slabs the reduction has to stream (1.7 TB/s measured).
static double plan_cost(const XtyPlan& pl, int slot_div, bool f64_path, size_t esz, bool split_bf16) {
  const int64_t nblocks = (int64_t)pl.ntiles * pl.splits;
  const int64_t rounds = (nblocks + pl.slots - 1) / pl.slots;
  const double rate = f64_path ? 46e12 : (split_bf16 ? (pl.f == 3 ? 260e12 : 220e12) : (pl.f == 3 ? 114e12 : 102e12));
  const double slow = slot_div == 1 ? 1.0 : (slot_div == 2 ? 1.045 : 1.12);
  const double gemm = 2.0 * (double)rounds * pl.slots * (double)pl.rows_per_split * pl.tm * pl.tm / rate * slow;
  const double reduce = (double)pl.splits * pl.n0p * pl.n1p * (double)esz / 1.7e12;
  return gemm + reduce;
}

static XtyPlan make_plan(int64_t m, int n0, int n1, int in_f64, int acc, int flags) {
  const bool f64_path = in_f64 || acc != NIF_F32;
  const size_t esz = acc == NIF_F64 ? 8 : 4;
  static const int forced = env_int("NIF_XTY_FRAGS");  // tuning aid
  XtyPlan best{};
  double best_cost = 0.0;
  bool have = false;
  for (int f = 2; f <= 3; ++f) {
    if (f == 3 && acc != NIF_F32) continue;  // the 192-wide tile exists for the f32-accumulate paths only
    // split-bf16: the 192-wide instantiation compiles, but with one wave per SIMD the compiler shuttles the 144
    // accumulator registers between the two register files every tile (0.76 ms against 0.55 at C2): 128 only
    if (f == 3 && (flags & NIF_XTY_SPLIT_BF16) && forced != 3) continue;
    if ((forced == 2 || forced == 3) && f != forced) continue;
    for (int div = 1; div <= 4; div *= 2) {
      if (f == 3 && div == 4) continue;
      const XtyPlan pl = plan_for(f, m, n0, n1, in_f64, acc, flags, div);
      const double c = plan_cost(pl, div, f64_path, esz, (flags & NIF_XTY_SPLIT_BF16) != 0);
      if (!have || c < best_cost) {
        best = pl;
        best_cost = c;
        have = true;
      }
      // When the output tiles alone fill the chip the rows are not split, and a last, partly filled round of
      // workgroups can cost up to half the kernel (symmetric N = 6144: 1176 tiles on 768 slots = 2 rounds for 1.53
      // rounds of work).  A few row splits put the block count next to a multiple of the slots; the price is their
      // slabs in the reduction (and in the workspace: capped at 1 GiB of slabs).
      if (pl.splits == 1 && pl.ntiles >= pl.slots / 2) {
        for (int sp = 2; sp <= 12; ++sp) {
          const XtyPlan ps = plan_for(f, m, n0, n1, in_f64, acc, flags, div, sp);
          if (ps.splits != sp || ps.slab_bytes > ((size_t)1 << 30)) continue;
          const double cs = plan_cost(ps, div, f64_path, esz, (flags & NIF_XTY_SPLIT_BF16) != 0);
          if (cs < best_cost) {
            best = ps;
            best_cost = cs;
          }
        }
      }
    }
  }
  return best;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename Tin, template <int, int> class MmaT, int F>
static void launch_variant(const XtyArgs& a, int nblocks, size_t lds, hipStream_t st) {
  const bool ra = a.flags & NIF_RELU_A, rb = a.flags & NIF_RELU_B;
  const bool ta = a.flags & NIF_TRANS_A, tb = a.flags & NIF_TRANS_B;
#define NIF_XTY_GO(RA, RB, TA_, TB_)                                                                     \
  do {                                                                                                   \
    auto kfn = xty_kernel<Tin, MmaT, F, F, RA, RB, TA_, TB_>;                                            \
    if (lds > 48 * 1024)                                                                                 \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(NT), lds, st, a);                                        \
  } while (0)
  if (!ta && !tb) {
    if (ra && rb) NIF_XTY_GO(true, true, false, false);
    else if (ra) NIF_XTY_GO(true, false, false, false);
    else if (rb) NIF_XTY_GO(false, true, false, false);
    else NIF_XTY_GO(false, false, false, false);
  } else {
    // ReLU-on-load is only used on activation (row-major) operands
    if (ta && tb) NIF_XTY_GO(false, false, true, true);
    else if (ta) NIF_XTY_GO(false, false, true, false);
    else NIF_XTY_GO(false, false, false, true);
  }
#undef NIF_XTY_GO
}

template <int F>
static void launch_split_bf16(const XtyArgs& a, int nblocks, size_t lds, hipStream_t st) {
  const bool ra = a.flags & NIF_RELU_A, rb = a.flags & NIF_RELU_B;
#define NIF_XTY_GO(RA, RB)                                                                               \
  do {                                                                                                   \
    auto kfn = xty_bf16_kernel<F, RA, RB>;                                                               \
    if (lds > 48 * 1024)                                                                                 \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(NT), lds, st, a);                                        \
  } while (0)
  if (ra && rb) NIF_XTY_GO(true, true);
  else if (ra) NIF_XTY_GO(true, false);
  else if (rb) NIF_XTY_GO(false, true);
  else NIF_XTY_GO(false, false);
#undef NIF_XTY_GO
}

}  // namespace nif

extern "C" size_t nif_xty_workspace_bytes(int64_t m, int n0, int n1, int acc, int flags) {
  if (m < 0 || n0 <= 0 || n1 <= 0) return 0;
  // the plan (tile size, splits) does not depend on the input type except through the resident
  // workgroups per CU: size for the larger of the two
  size_t need = 0;
  for (int variant = 0; variant < 4; ++variant) {
    const int in_f64 = variant & 1;
    // a split-bf16 request may fall back to the f32 matrix path (operand alignment): size for both plans
    const int fl = (variant & 2) ? (flags & ~NIF_XTY_SPLIT_BF16) : flags;
    const nif::XtyPlan pl = nif::make_plan(m, n0, n1, in_f64, acc, fl);
    const size_t b = nif::align256(pl.slab_bytes) + nif::align256(pl.nsa_bytes) + nif::align256(pl.nsb_bytes);
    if (b > need) need = b;
  }
  return need;
}

namespace nif {
template <typename OT>
static int xty_impl(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t m, int n0, int n1, int in_dtype,
                    int acc, int flags, OT* G, int64_t ldg, double* gdiag, double* na, double* nb, void* ws,
                    size_t ws_bytes, void* stream, int pack_block = 0) {
  // m == 0 (an empty shard): the operand pointers are never dereferenced and may be null
  // pack_block > 0 (symmetric, OT = double): G is the packed block-upper-triangle buffer, ldg is not used
  if ((m > 0 && (!A || !B)) || !G || m < 0 || n0 <= 0 || n1 <= 0 || (pack_block <= 0 && ldg < n1)) return NIF_ERR_INVALID_ARG;
  if (pack_block > 0 && (!(flags & NIF_SYMMETRIC) || sizeof(OT) != 8 || gdiag)) return NIF_ERR_INVALID_ARG;
  if (in_dtype != NIF_F32 && in_dtype != NIF_F64) return NIF_ERR_INVALID_ARG;
  if (acc != NIF_F32 && acc != NIF_F64) return NIF_ERR_INVALID_ARG;
  if (in_dtype == NIF_F64 && acc != NIF_F64) return NIF_ERR_INVALID_ARG;
  const bool ta = flags & NIF_TRANS_A, tb = flags & NIF_TRANS_B;
  if (lda < (ta ? m : (int64_t)n0) || ldb < (tb ? m : (int64_t)n1)) return NIF_ERR_INVALID_ARG;
  if ((flags & NIF_SYMMETRIC) && (A != B || n0 != n1 || lda != ldb || ta != tb)) return NIF_ERR_INVALID_ARG;
  if ((ta || tb) && (flags & (NIF_RELU_A | NIF_RELU_B))) return NIF_ERR_UNSUPPORTED;
  if ((flags & NIF_SYMMETRIC) && (((flags & NIF_RELU_A) != 0) != ((flags & NIF_RELU_B) != 0)))
    return NIF_ERR_INVALID_ARG;
  if ((flags & NIF_XTY_SPLIT_BF16) && (in_dtype != NIF_F32 || acc != NIF_F32 || ta || tb)) return NIF_ERR_UNSUPPORTED;
  if (flags & NIF_XTY_SPLIT_BF16) {
    // the split kernel moves 16-byte pieces of rows: operands it cannot address that way take the f32 matrix path
    const bool ok = reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0 && lda % 4 == 0 &&
                    ldb % 4 == 0 && n0 % 4 == 0 && n1 % 4 == 0 && lda < (1 << 24) && ldb < (1 << 24);
    if (!ok) flags &= ~NIF_XTY_SPLIT_BF16;
  }
  const XtyPlan pl = make_plan(m, n0, n1, in_dtype == NIF_F64, acc, flags);
  const size_t need = align256(pl.slab_bytes) + align256(pl.nsa_bytes) + align256(pl.nsb_bytes);
  if (!ws || ws_bytes < need) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();

  const size_t esz_in = in_dtype == NIF_F64 ? 8 : 4;
  XtyArgs a;
  a.A = A; a.B = B; a.lda = lda; a.ldb = ldb; a.m = m; a.rows_per_split = pl.rows_per_split;
  a.n0 = n0; a.n1 = n1; a.n0p = pl.n0p; a.n1p = pl.n1p;
  a.tiles_i = pl.tiles_i; a.tiles_j = pl.tiles_j; a.ntiles = pl.ntiles; a.splits = pl.splits;
  a.flags = flags;
  a.vec_a = ((reinterpret_cast<uintptr_t>(A) % 16) == 0 && ((size_t)lda * esz_in) % 16 == 0) ? 1 : 0;
  a.vec_b = ((reinterpret_cast<uintptr_t>(B) % 16) == 0 && ((size_t)ldb * esz_in) % 16 == 0) ? 1 : 0;
  unsigned char* w = reinterpret_cast<unsigned char*>(ws);
  a.slab = w;
  a.nslab_a = w + align256(pl.slab_bytes);
  a.nslab_b = w + align256(pl.slab_bytes) + align256(pl.nsa_bytes);

  const int nblocks = pl.ntiles * pl.splits;
  size_t lds = pl.lds_bytes;
  static const int lds_pad = env_int("NIF_XTY_LDS_PAD");  // tuning aid: caps workgroups per CU
  if (lds_pad > 0) lds += (size_t)lds_pad;
  prof_begin(NIF_PROF_XTY, st);
  if (flags & NIF_XTY_SPLIT_BF16) {
    if (pl.f == 3) launch_split_bf16<3>(a, nblocks, lds, st);
    else launch_split_bf16<2>(a, nblocks, lds, st);
  } else if (in_dtype == NIF_F32 && acc == NIF_F32) {
    if (pl.f == 3) launch_variant<float, MmaF32, 3>(a, nblocks, lds, st);
    else launch_variant<float, MmaF32, 2>(a, nblocks, lds, st);
  } else if (in_dtype == NIF_F32) {
    launch_variant<float, MmaF64, 2>(a, nblocks, lds, st);
  } else {
    launch_variant<double, MmaF64, 2>(a, nblocks, lds, st);
  }
  prof_end(NIF_PROF_XTY, st);
  if (!launch_ok()) return NIF_ERR_LAUNCH;

  const int sym = (flags & NIF_SYMMETRIC) ? 1 : 0;
  const int nmax = n0 > n1 ? n0 : n1;
  dim3 rg((nmax + 255) / 256, n0 + 2);
  const int nt64 = (n0 + 63) / 64;
  const bool sym_tiles = sym && nt64 >= 45 && pack_block <= 0;  // >= 1000 upper tiles: enough workgroups to fill the chip
  int ybase = 0;
  if (pack_block > 0) {
    double* packed = reinterpret_cast<double*>(G);
    if (acc == NIF_F32)
      hipLaunchKernelGGL((xty_reduce_packed_kernel<float>), dim3((n0 + 255) / 256, n0), dim3(256), 0, st,
                         (const float*)a.slab, pl.splits, n0, pl.n0p, pack_block, packed);
    else
      hipLaunchKernelGGL((xty_reduce_packed_kernel<double>), dim3((n0 + 255) / 256, n0), dim3(256), 0, st,
                         (const double*)a.slab, pl.splits, n0, pl.n0p, pack_block, packed);
    ybase = n0;
    rg = dim3((nmax + 255) / 256, 2);
  } else if (sym_tiles) {
    if (acc == NIF_F32)
      hipLaunchKernelGGL((xty_reduce_sym_kernel<float, OT>), dim3(nt64, nt64), dim3(256), 0, st, (const float*)a.slab,
                         pl.splits, n0, pl.n0p, G, ldg, gdiag);
    else
      hipLaunchKernelGGL((xty_reduce_sym_kernel<double, OT>), dim3(nt64, nt64), dim3(256), 0, st, (const double*)a.slab,
                         pl.splits, n0, pl.n0p, G, ldg, gdiag);
    ybase = n0;  // the norm rows, if asked for, through the general kernel
    rg = dim3((nmax + 255) / 256, 2);
  }
  if ((!sym_tiles && pack_block <= 0) || na || nb) {
    if (acc == NIF_F32) {
      hipLaunchKernelGGL((xty_reduce_kernel<float, OT>), rg, dim3(256), 0, st, (const float*)a.slab,
                         (const float*)a.nslab_a, (const float*)a.nslab_b, pl.splits, n0, n1, pl.n0p, pl.n1p, pl.tm, sym,
                         G, ldg, na, nb, ybase, gdiag);
    } else {
      hipLaunchKernelGGL((xty_reduce_kernel<double, OT>), rg, dim3(256), 0, st, (const double*)a.slab,
                         (const double*)a.nslab_a, (const double*)a.nslab_b, pl.splits, n0, n1, pl.n0p, pl.n1p, pl.tm,
                         sym, G, ldg, na, nb, ybase, gdiag);
    }
  }
  if (!launch_ok()) return NIF_ERR_LAUNCH;
  return NIF_OK;
}
}  // namespace nif

extern "C" int nif_xty(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t m, int n0, int n1, int in_dtype,
                       int acc, int flags, double* G, int64_t ldg, double* na, double* nb, void* ws, size_t ws_bytes,
                       void* stream) {
  nif::Range nif_range_("nif_xty");
  return nif::xty_impl<double>(A, lda, B, ldb, m, n0, n1, in_dtype, acc, flags, G, ldg, nullptr, na, nb, ws, ws_bytes,
                               stream);
}

extern "C" int nif_xty_sym_gf32(const void* A, int64_t lda, int64_t m, int n, int in_dtype, int acc, int flags, float* G,
                                int64_t ldg, double* gdiag, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_xty_sym_gf32");
  if (!gdiag) return NIF_ERR_INVALID_ARG;
  return nif::xty_impl<float>(A, lda, A, lda, m, n, n, in_dtype, acc, flags | NIF_SYMMETRIC, G, ldg, gdiag, nullptr, nullptr,
                              ws, ws_bytes, stream);
}

extern "C" size_t nif_packed_upper_len(int n, int block) {
  if (n <= 0 || block <= 0) return 0;
  const int last = (n - 1) / block * block;  // first row of the last block row
  return (size_t)nif::packed_upper_offset(last, last, n, block) + (size_t)(n - last) * (size_t)(n - last);
}

extern "C" int nif_xty_sym_packed(const void* A, int64_t lda, int64_t m, int n, int in_dtype, int acc, int flags, int block,
                                  double* packed, double* na, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_xty_sym_packed");
  if (block <= 0 || !packed) return NIF_ERR_INVALID_ARG;
  return nif::xty_impl<double>(A, lda, A, lda, m, n, n, in_dtype, acc, flags | NIF_SYMMETRIC, packed, n, nullptr, na, nullptr,
                               ws, ws_bytes, stream, block);
}

extern "C" int nif_unpack_upper(const double* packed, int n, int block, void* G, int64_t ldg, int out_dtype, double* gdiag,
                                void* stream) {
  nif::Range nif_range_("nif_unpack_upper");
  using namespace nif;
  if (!packed || !G || n <= 0 || block <= 0 || ldg < n || (out_dtype != NIF_F32 && out_dtype != NIF_F64))
    return NIF_ERR_INVALID_ARG;
  if (n > 65535) return NIF_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  const dim3 grid((n + 255) / 256, n);
  if (out_dtype == NIF_F64)
    hipLaunchKernelGGL(xty_unpack_upper_kernel<double>, grid, dim3(256), 0, st, packed, n, block, (double*)G, ldg, gdiag);
  else
    hipLaunchKernelGGL(xty_unpack_upper_kernel<float>, grid, dim3(256), 0, st, packed, n, block, (float*)G, ldg, gdiag);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
