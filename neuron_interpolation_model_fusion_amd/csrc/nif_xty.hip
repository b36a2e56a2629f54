// nif_xty.hip -- column cross-Gram G = A^T B over the sample rows, for gfx950.
//
// The contraction axis is the ROW axis of the row-major activation matrices, so
// one output tile is a sum over m rows: the row range is split over workgroups
// (split-K), each workgroup owns one 128x128 output tile x one row block and
// writes its partial tile to a slab; a second kernel sums the slabs in a fixed
// order in f64 (bitwise reproducible, no atomics) and mirrors the lower
// triangle in symmetric mode.
//
// MFMA mapping (wave64):
//   f32 accumulate: v_mfma_f32_32x32x2_f32. Lane l feeds A[i = l&31][k = l>>5]:
//     with row-major activations, lanes 0..31 read 32 consecutive columns of
//     sample row k and lanes 32..63 the same columns of row k+1 -- the layout
//     the tiles already have in HBM and LDS, so no transpose anywhere.
//   f64 accumulate: v_mfma_f64_16x16x4_f64, lane l feeds A[i = l&15][k = l>>4].
// A workgroup is 4 waves (2x2), each wave a 64x64 sub-tile; tiles are staged
// through LDS in [BK=16][144]-element images (row stride 144 keeps the 16-lane
// groups of the f64 path on distinct banks), double-buffered, global loads for
// tile t+1 issued before the MFMAs of tile t and written to LDS after them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

constexpr int TM = 128;
constexpr int TN = 128;
constexpr int BK = 16;
constexpr int LDS_LD = 144;
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
  float* nslab_a;    // [splits][n0p] partial squared column norms (f32 in / f32 acc) ...
  float* nslab_b;    // [splits][n1p]
  double* dslab_a;   // ... or f64 when accumulating in f64
  double* dslab_b;
};

template <typename T>
struct V4 {
  T v[4];
};

// ---- global -> register staging ------------------------------------------------

// Non-transposed operand: tile element (k, c) = P[(k0 + k) * ld + c0 + c].
// Thread t stages columns 4*(t%32)..+3 of rows t/32 and t/32 + 8.
template <typename T, bool RELU>
__device__ __forceinline__ void stage_rows(V4<T> (&r)[2], const T* __restrict__ P, int64_t ld, int64_t k0,
                                           int64_t kend, int c0, int n, int vec_ok, int tid) {
  const int col = c0 + 4 * (tid & 31);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t row = k0 + (tid >> 5) + 8 * h;
    V4<T> x;
    x.v[0] = x.v[1] = x.v[2] = x.v[3] = T(0);
    if (row < kend && col < n) {
      const T* p = P + row * ld + col;
      if (vec_ok && col + 3 < n) {
        if constexpr (sizeof(T) == 4) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          x.v[0] = q.x; x.v[1] = q.y; x.v[2] = q.z; x.v[3] = q.w;
        } else {
          const double2 q0 = *reinterpret_cast<const double2*>(p);
          const double2 q1 = *reinterpret_cast<const double2*>(p + 2);
          x.v[0] = q0.x; x.v[1] = q0.y; x.v[2] = q1.x; x.v[3] = q1.y;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < n) x.v[e] = p[e];
      }
      if (RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x.v[e] = x.v[e] > T(0) ? x.v[e] : T(0);
      }
    }
    r[h] = x;
  }
}

// Transposed operand (stored n x m): tile element (k, c) = P[(c0 + c) * ld + k0 + k].
// Thread t stages k = 4*(t%4)..+3 of columns t/4 and t/4 + 64.
template <typename T, bool RELU>
__device__ __forceinline__ void stage_cols(V4<T> (&r)[2], const T* __restrict__ P, int64_t ld, int64_t k0,
                                           int64_t kend, int c0, int n, int vec_ok, int tid) {
  const int64_t k = k0 + 4 * (tid & 3);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = c0 + (tid >> 2) + 64 * h;
    V4<T> x;
    x.v[0] = x.v[1] = x.v[2] = x.v[3] = T(0);
    if (c < n && k < kend) {
      const T* p = P + (int64_t)c * ld + k;
      if (vec_ok && k + 3 < kend) {
        if constexpr (sizeof(T) == 4) {
          const float4 q = *reinterpret_cast<const float4*>(p);
          x.v[0] = q.x; x.v[1] = q.y; x.v[2] = q.z; x.v[3] = q.w;
        } else {
          const double2 q0 = *reinterpret_cast<const double2*>(p);
          const double2 q1 = *reinterpret_cast<const double2*>(p + 2);
          x.v[0] = q0.x; x.v[1] = q0.y; x.v[2] = q1.x; x.v[3] = q1.y;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < kend) x.v[e] = p[e];
      }
      if (RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x.v[e] = x.v[e] > T(0) ? x.v[e] : T(0);
      }
    }
    r[h] = x;
  }
}

template <typename T>
__device__ __forceinline__ void lds_put_rows(T* __restrict__ L, const V4<T> (&r)[2], int tid) {
  // 16-byte aligned: LDS_LD * sizeof(T) and 4 * sizeof(T) are multiples of 16
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    T* d = L + ((tid >> 5) + 8 * h) * LDS_LD + 4 * (tid & 31);
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(d) = make_float4(r[h].v[0], r[h].v[1], r[h].v[2], r[h].v[3]);
    } else {
      *reinterpret_cast<double2*>(d) = make_double2(r[h].v[0], r[h].v[1]);
      *reinterpret_cast<double2*>(d + 2) = make_double2(r[h].v[2], r[h].v[3]);
    }
  }
}

template <typename T>
__device__ __forceinline__ void lds_put_cols(T* __restrict__ L, const V4<T> (&r)[2], int tid) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = (tid >> 2) + 64 * h;
#pragma unroll
    for (int e = 0; e < 4; ++e) L[(4 * (tid & 3) + e) * LDS_LD + c] = r[h].v[e];
  }
}

// ---- matrix-core micro kernels ---------------------------------------------------

struct MmaF32 {
  typedef float acc_t;
  typedef f32x16 frag_t;
  static constexpr int R = 2;  // 32-wide fragments per 64-wide wave tile
  template <typename T>
  static __device__ __forceinline__ void tile(const T* __restrict__ la, const T* __restrict__ lb, frag_t (&acc)[R][R],
                                              int lane) {
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int off = (2 * kk + (lane >> 5)) * LDS_LD + (lane & 31);
      float a[R], b[R];
#pragma unroll
      for (int i = 0; i < R; ++i) a[i] = (float)la[off + 32 * i];
#pragma unroll
      for (int j = 0; j < R; ++j) b[j] = (float)lb[off + 32 * j];
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  static __device__ __forceinline__ void zero(frag_t (&acc)[R][R]) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
  // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  static __device__ __forceinline__ void store(acc_t* __restrict__ S, int ld, const frag_t (&acc)[R][R], int lane) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          S[(int64_t)row * ld + 32 * j + (lane & 31)] = acc[i][j][e];
        }
  }
};

struct MmaF64 {
  typedef double acc_t;
  typedef f64x4 frag_t;
  static constexpr int R = 4;  // 16-wide fragments per 64-wide wave tile
  template <typename T>
  static __device__ __forceinline__ void tile(const T* __restrict__ la, const T* __restrict__ lb, frag_t (&acc)[R][R],
                                              int lane) {
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int off = (4 * kk + (lane >> 4)) * LDS_LD + (lane & 15);
      double a[R], b[R];
#pragma unroll
      for (int i = 0; i < R; ++i) a[i] = (double)la[off + 16 * i];
#pragma unroll
      for (int j = 0; j < R; ++j) b[j] = (double)lb[off + 16 * j];
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  static __device__ __forceinline__ void zero(frag_t (&acc)[R][R]) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
  }
  // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ void store(acc_t* __restrict__ S, int ld, const frag_t (&acc)[R][R], int lane) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * i + (lane >> 4) + 4 * e;
          S[(int64_t)row * ld + 16 * j + (lane & 15)] = acc[i][j][e];
        }
  }
};

// ---- main kernel -----------------------------------------------------------------

template <typename Tin, typename Mma, bool RELU_A, bool RELU_B, bool TA, bool TB>
__global__ __launch_bounds__(NT) void xty_kernel(const XtyArgs p) {
  typedef typename Mma::acc_t acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Tin* smem = reinterpret_cast<Tin*>(smem_raw);
  // [buf][operand][BK][LDS_LD]
  constexpr int TILE = BK * LDS_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;

  // XCD-aware id remap (bijective for any grid size): blocks b and b + 8 share an
  // XCD, so give each XCD a contiguous range of ids; consecutive ids are the tiles
  // of one row block and reuse its A / B panels out of that XCD's L2.
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

  typename Mma::frag_t acc[Mma::R][Mma::R];
  Mma::zero(acc);

  // squared-norm partials of the staged elements
  acc_t sqa[4] = {0, 0, 0, 0}, sqb[4] = {0, 0, 0, 0};

  V4<Tin> ra[2], rb[2];
  auto stage = [&](int64_t k0) {
    if (TA) stage_cols<Tin, RELU_A>(ra, A, p.lda, k0, k_end, ci, p.n0, p.vec_a, tid);
    else stage_rows<Tin, RELU_A>(ra, A, p.lda, k0, k_end, ci, p.n0, p.vec_a, tid);
    if (TB) stage_cols<Tin, RELU_B>(rb, B, p.ldb, k0, k_end, cj, p.n1, p.vec_b, tid);
    else stage_rows<Tin, RELU_B>(rb, B, p.ldb, k0, k_end, cj, p.n1, p.vec_b, tid);
  };
  auto commit = [&](int buf) {
    Tin* la = smem + (buf * 2 + 0) * TILE;
    Tin* lb = smem + (buf * 2 + 1) * TILE;
    if (TA) lds_put_cols<Tin>(la, ra, tid); else lds_put_rows<Tin>(la, ra, tid);
    if (TB) lds_put_cols<Tin>(lb, rb, tid); else lds_put_rows<Tin>(lb, rb, tid);
    if (TA) {
      // two columns per thread: sq[h] sums that column's 4 staged k
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) sqa[h] += (acc_t)ra[h].v[e] * (acc_t)ra[h].v[e];
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) sqa[e] += (acc_t)ra[h].v[e] * (acc_t)ra[h].v[e];
    }
    if (TB) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) sqb[h] += (acc_t)rb[h].v[e] * (acc_t)rb[h].v[e];
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) sqb[e] += (acc_t)rb[h].v[e] * (acc_t)rb[h].v[e];
    }
  };

  if (nt > 0) {
    stage(k_begin);
    commit(0);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) stage(k_begin + (int64_t)(t + 1) * BK);
    const Tin* la = smem + (buf * 2 + 0) * TILE + wi * 64;
    const Tin* lb = smem + (buf * 2 + 1) * TILE + wj * 64;
    Mma::template tile<Tin>(la, lb, acc, lane);
    if (t + 1 < nt) commit(buf ^ 1);
    __syncthreads();
  }

  // partial tile -> slab (padded to whole tiles, so no guards)
  acc_t* S = reinterpret_cast<acc_t*>(p.slab) + ((int64_t)split * p.n0p + ci + wi * 64) * p.n1p + cj + wj * 64;
  Mma::store(S, p.n1p, acc, lane);

  // column-norm partials: A's from the blocks of tile column 0 (or the diagonal
  // in symmetric mode), B's from tile row 0.  Fixed summation order.
  const bool want_a = (p.flags & NIF_SYMMETRIC) ? (ti == tj) : (tj == 0);
  const bool want_b = !(p.flags & NIF_SYMMETRIC) && (ti == 0);
  acc_t* red = reinterpret_cast<acc_t*>(smem_raw);  // [8][128], the tile images are dead now
  auto reduce_norm = [&](const acc_t (&sq)[4], bool trans, int c0, int npad, float* fslab, double* dslab) {
    int parts;
    if (trans) {
      parts = 4;
#pragma unroll
      for (int h = 0; h < 2; ++h) red[(tid & 3) * 128 + (tid >> 2) + 64 * h] = sq[h];
    } else {
      parts = 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(tid >> 5) * 128 + 4 * (tid & 31) + e] = sq[e];
    }
    __syncthreads();
    if (tid < 128) {
      acc_t s = 0;
      for (int r = 0; r < parts; ++r) s += red[r * 128 + tid];
      if constexpr (sizeof(acc_t) == 4) fslab[(int64_t)split * npad + c0 + tid] = s;
      else dslab[(int64_t)split * npad + c0 + tid] = s;
    }
    __syncthreads();
  };
  if (want_a) reduce_norm(sqa, TA, ci, p.n0p, p.nslab_a, p.dslab_a);
  if (want_b) reduce_norm(sqb, TB, cj, p.n1p, p.nslab_b, p.dslab_b);
}

// ---- slab reduction --------------------------------------------------------------

template <typename acc_t>
__global__ void xty_reduce_kernel(const acc_t* __restrict__ slab, int splits, int n0, int n1, int n0p, int n1p,
                                  int symmetric, double* __restrict__ G, int64_t ldg) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n1 || i >= n0) return;
  int si = i, sj = j;
  if (symmetric && (i / TM) > (j / TN)) {
    si = j;
    sj = i;
  }
  const acc_t* s = slab + (int64_t)si * n1p + sj;
  const int64_t stride = (int64_t)n0p * n1p;
  double sum = 0.0;
  for (int k = 0; k < splits; ++k) sum += (double)s[k * stride];
  G[(int64_t)i * ldg + j] = sum;
}

template <typename acc_t>
__global__ void xty_norm_reduce_kernel(const acc_t* __restrict__ ns, int splits, int n, int npad,
                                       double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double sum = 0.0;
  for (int k = 0; k < splits; ++k) sum += (double)ns[(int64_t)k * npad + i];
  out[i] = sum;
}

// ---- host side -------------------------------------------------------------------

struct XtyPlan {
  int tiles_i, tiles_j, ntiles, splits, n0p, n1p;
  int64_t rows_per_split;
  size_t slab_bytes, nsa_bytes, nsb_bytes;
};

static XtyPlan make_plan(int64_t m, int n0, int n1, int acc, int flags) {
  XtyPlan pl;
  pl.tiles_i = (n0 + TM - 1) / TM;
  pl.tiles_j = (n1 + TN - 1) / TN;
  pl.n0p = pl.tiles_i * TM;
  pl.n1p = pl.tiles_j * TN;
  pl.ntiles = (flags & NIF_SYMMETRIC) ? pl.tiles_i * (pl.tiles_i + 1) / 2 : pl.tiles_i * pl.tiles_j;
  // fill the chip (256 CUs, up to 4 resident workgroups each) without leaving a
  // second, nearly empty round of workgroups; keep >= 4 K-tiles per row block.
  const int64_t target = 1024;
  int64_t splits = target / pl.ntiles;
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
  return pl;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename Tin, typename Mma>
static void launch_variant(const XtyArgs& a, int nblocks, size_t lds, hipStream_t st) {
  const bool ra = a.flags & NIF_RELU_A, rb = a.flags & NIF_RELU_B;
  const bool ta = a.flags & NIF_TRANS_A, tb = a.flags & NIF_TRANS_B;
#define NIF_XTY_GO(RA, RB, TA_, TB_)                                                                          \
  do {                                                                                                        \
    auto kfn = xty_kernel<Tin, Mma, RA, RB, TA_, TB_>;                                                        \
    if (lds > 48 * 1024)                                                                                      \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
    hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(NT), lds, st, a);                                             \
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

}  // namespace nif

extern "C" size_t nif_xty_workspace_bytes(int64_t m, int n0, int n1, int acc, int flags) {
  if (m < 0 || n0 <= 0 || n1 <= 0) return 0;
  const nif::XtyPlan pl = nif::make_plan(m, n0, n1, acc, flags);
  return nif::align256(pl.slab_bytes) + nif::align256(pl.nsa_bytes) + nif::align256(pl.nsb_bytes);
}

extern "C" int nif_xty(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t m, int n0, int n1, int in_dtype,
                       int acc, int flags, double* G, int64_t ldg, double* na, double* nb, void* ws, size_t ws_bytes,
                       void* stream) {
  using namespace nif;
  if (!A || !B || !G || m < 0 || n0 <= 0 || n1 <= 0 || ldg < n1) return NIF_ERR_INVALID_ARG;
  if (in_dtype != NIF_F32 && in_dtype != NIF_F64) return NIF_ERR_INVALID_ARG;
  if (acc != NIF_F32 && acc != NIF_F64) return NIF_ERR_INVALID_ARG;
  if (in_dtype == NIF_F64 && acc != NIF_F64) return NIF_ERR_INVALID_ARG;
  const bool ta = flags & NIF_TRANS_A, tb = flags & NIF_TRANS_B;
  if (lda < (ta ? m : (int64_t)n0) || ldb < (tb ? m : (int64_t)n1)) return NIF_ERR_INVALID_ARG;
  if ((flags & NIF_SYMMETRIC) && (A != B || n0 != n1 || lda != ldb || ta != tb)) return NIF_ERR_INVALID_ARG;
  if ((ta || tb) && (flags & (NIF_RELU_A | NIF_RELU_B))) return NIF_ERR_UNSUPPORTED;
  if ((flags & NIF_SYMMETRIC) && (((flags & NIF_RELU_A) != 0) != ((flags & NIF_RELU_B) != 0)))
    return NIF_ERR_INVALID_ARG;
  const XtyPlan pl = make_plan(m, n0, n1, acc, flags);
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
  void* nsa = w + align256(pl.slab_bytes);
  void* nsb = w + align256(pl.slab_bytes) + align256(pl.nsa_bytes);
  a.nslab_a = reinterpret_cast<float*>(nsa); a.nslab_b = reinterpret_cast<float*>(nsb);
  a.dslab_a = reinterpret_cast<double*>(nsa); a.dslab_b = reinterpret_cast<double*>(nsb);

  const int nblocks = pl.ntiles * pl.splits;
  const size_t lds = (size_t)2 * 2 * BK * LDS_LD * esz_in;
  prof_begin(NIF_PROF_XTY, st);
  if (in_dtype == NIF_F32 && acc == NIF_F32) launch_variant<float, MmaF32>(a, nblocks, lds, st);
  else if (in_dtype == NIF_F32) launch_variant<float, MmaF64>(a, nblocks, lds, st);
  else launch_variant<double, MmaF64>(a, nblocks, lds, st);
  prof_end(NIF_PROF_XTY, st);
  if (!launch_ok()) return NIF_ERR_LAUNCH;

  const int sym = (flags & NIF_SYMMETRIC) ? 1 : 0;
  dim3 rg((n1 + 255) / 256, n0);
  if (acc == NIF_F32) {
    hipLaunchKernelGGL((xty_reduce_kernel<float>), rg, dim3(256), 0, st, (const float*)a.slab, pl.splits, n0, n1,
                       pl.n0p, pl.n1p, sym, G, ldg);
    if (na) hipLaunchKernelGGL((xty_norm_reduce_kernel<float>), dim3((n0 + 255) / 256), dim3(256), 0, st,
                               (const float*)nsa, pl.splits, n0, pl.n0p, na);
    if (nb) hipLaunchKernelGGL((xty_norm_reduce_kernel<float>), dim3((n1 + 255) / 256), dim3(256), 0, st,
                               sym ? (const float*)nsa : (const float*)nsb, pl.splits, n1, sym ? pl.n0p : pl.n1p, nb);
  } else {
    hipLaunchKernelGGL((xty_reduce_kernel<double>), rg, dim3(256), 0, st, (const double*)a.slab, pl.splits, n0, n1,
                       pl.n0p, pl.n1p, sym, G, ldg);
    if (na) hipLaunchKernelGGL((xty_norm_reduce_kernel<double>), dim3((n0 + 255) / 256), dim3(256), 0, st,
                               (const double*)nsa, pl.splits, n0, pl.n0p, na);
    if (nb) hipLaunchKernelGGL((xty_norm_reduce_kernel<double>), dim3((n1 + 255) / 256), dim3(256), 0, st,
                               sym ? (const double*)nsa : (const double*)nsb, pl.splits, n1, sym ? pl.n0p : pl.n1p, nb);
  }
  if (!launch_ok()) return NIF_ERR_LAUNCH;
  return NIF_OK;
}
