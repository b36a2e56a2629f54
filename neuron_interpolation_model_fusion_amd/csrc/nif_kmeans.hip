// nif_kmeans.hip -- neuron k-means in Gram form (KF: hf_kf.py:405-490, ALF: alf.py:216-283).
//
// The reference clusters the N neurons of all models by their activation vectors (length m):
// every Lloyd step recomputes k centroids with a Python loop and a k x N cdist over m.  Every
// distance it needs is a function of inner products of neuron vectors, so after ONE N x N Gram
// matrix (nif_xty, MFMA) a Lloyd step only streams that matrix:
//     centroid c = sum_{i in c} s_i x_i / S_c            (x_i: neuron i's -- projected -- vector)
//     H[c][j]    = <centroid c, y_j> = sum_{i in c} s_i G[i][j] / S_c
//     |c|^2      = sum_{i in c} s_i H[c][i] / S_c
//     d^2(c, j)  = |c|^2 - 2 H[c][j] + nn[j]             (nn[j] = |y_j|^2, the neuron compared)
// For KF x_i = X w_i is the LS projection of y_i = a_i and <X w_i, a_j> = <X w_i, X w_j>, so
// G = (XW)^T (XW) and nn = |a_j|^2; for ALF x = y and nn = diag(G).
// HBM-bound kernels: one pass over G per step, 8 N^2 bytes (f64 storage) or 4 N^2 (f32 storage, _gf32 entries).
// Every reduction runs in a fixed order (members sorted by index), so results are reproducible.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

// Entry (i, j) of the Gram matrix: with f32 storage the diagonal comes from its own f64 array (gd), so that the
// distance of a neuron to a centroid made of that neuron alone cancels exactly, as it does with f64 storage.
template <typename GT>
__device__ __forceinline__ double gram_at(const GT* __restrict__ row, int i, int j, const double* __restrict__ gd) {
  if (sizeof(GT) == 4) return i == j ? gd[j] : (double)row[j];
  return (double)row[j];
}

// ---- block-wide arg-extremum over (value, index): larger value wins, lower index on ties -----
__device__ __forceinline__ void block_argmax(double& v, int& j, double* sv, int* sj) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // two scalar DPP reductions per wave: the maximum, then the lowest index among the lanes that hold it
  const double wmax = -wave_min_f64(-v);
  const unsigned widx = wave_min_u32(v == wmax ? (unsigned)j : 0xffffffffu);
  __syncthreads();  // protects sv/sj against readers of the previous call
  if (lane == 0) {
    sv[wave] = wmax;
    sj[wave] = (int)widx;
  }
  __syncthreads();
  v = sv[0];
  j = sj[0];
  for (int w = 1; w < nw; ++w) {
    const double ov = sv[w];
    const int oj = sj[w];
    if (ov > v || (ov == v && (unsigned)oj < (unsigned)j)) {
      v = ov;
      j = oj;
    }
  }
}

// Up to 16 initialisations (KF's tries: same G, different first neuron) ride in one launch: blockIdx.y is the try,
// its outputs are rows of [tries][k] picks and [tries][N] assign / mind.
constexpr int FI_MAX_TRIES = 16;
struct FiFirsts {
  int first[FI_MAX_TRIES];
};

// The Lloyd-step kernels take a batch of k-means problems over the SAME G, nn, s (KF's tries): blockIdx.y walks the
// batch, idx[] names the try each entry works on (the tries still alive), every per-try array is a row of a
// [tries][...] tensor and every workspace pointer belongs to try 0's workspace, ws_stride bytes before the next one's.
struct KmBatch {
  int idx[FI_MAX_TRIES];
  long long ws_stride;
  int blk0;  // first 64-neuron block of the launch (the two passes over G of a neuron-sharded Lloyd step); else 0
};
template <typename T>
__device__ __forceinline__ T* kb_ws(T* p, const KmBatch& kb) {
  typedef typename std::remove_const<T>::type U;
  return reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(const_cast<U*>(p)) +
                              (size_t)kb.idx[blockIdx.y] * (size_t)kb.ws_stride);
}
template <typename T>
__device__ __forceinline__ T* kb_row(T* p, const KmBatch& kb, size_t n) {
  return p + (size_t)kb.idx[blockIdx.y] * n;
}

// ---- farthest-point initialisation (hf_kf.py:424-441), one workgroup ----------------------------
// picks[0] = first; picks[c] = argmax_j min_{c' < c} dist(centroid c', neuron j), where centroid c'
// is the PROJECTED neuron picks[c'] and dist^2 = G[p][p] - 2 G[p][j] + nn[j].  Also leaves the
// first assignment (argmin over all k centroids, lowest centroid on ties, :440-441).
template <typename GT>
__global__ __launch_bounds__(1024) void kmeans_farthest_init_kernel(const GT* __restrict__ G, int64_t ldg, const double* __restrict__ gd,
                                                                    const double* __restrict__ nn, int N, int k,
                                                                    FiFirsts fb, int* __restrict__ picks,
                                                                    int* __restrict__ assign,
                                                                    double* __restrict__ mind,
                                                                    unsigned* __restrict__ redo_count) {
  __shared__ double sv[16];
  __shared__ int sj[16];
  picks += (size_t)blockIdx.y * k;
  assign += (size_t)blockIdx.y * N;
  mind += (size_t)blockIdx.y * N;
  if (redo_count) {
    // second launch behind the multi-workgroup kernel: only a try whose exchange timed out (last pick left at -1) is
    // done again, from scratch, by this one workgroup -- same arithmetic, same picks, no exchange to wait for
    const bool failed = picks[k - 1] < 0;
    __syncthreads();  // every thread has read the flag before thread 0 overwrites the picks
    if (!failed) return;
    if (threadIdx.x == 0) atomicAdd(redo_count, 1u);
  }
  int idx = fb.first[blockIdx.y];
  for (int c = 0; c < k; ++c) {
    if (threadIdx.x == 0) picks[c] = idx;
    const GT* __restrict__ row = G + (int64_t)idx * ldg;
    const double gii = gram_at(row, idx, idx, gd);
    double bv = -INFINITY;
    int bj = INT_MAX;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      double d2 = gii - 2.0 * gram_at(row, idx, j, gd) + nn[j];
      if (!(d2 > 0.0)) d2 = 0.0;
      const double d = sqrt(d2);
      double m = d;
      if (c == 0) {
        assign[j] = 0;
      } else {
        m = mind[j];
        if (d < m) {
          m = d;
          assign[j] = c;
        }
      }
      mind[j] = m;
      if (m > bv) {  // j ascends within a thread: strict > keeps the lowest index
        bv = m;
        bj = j;
      }
    }
    block_argmax(bv, bj, sv, sj);
    idx = bj;
  }
}

// Same, with the running minimum distance, |y_j|^2 and the label of every neuron held in registers (EPT
// neurons per lane, N <= 1024 * EPT): a pick then only streams one Gram row.
template <int EPT, typename GT>
__global__ __launch_bounds__(1024) void kmeans_farthest_init_reg_kernel(const GT* __restrict__ G, int64_t ldg, const double* __restrict__ gd,
                                                                        const double* __restrict__ nn, int N, int k,
                                                                        FiFirsts fb, int* __restrict__ picks,
                                                                        int* __restrict__ assign,
                                                                        double* __restrict__ mind) {
  __shared__ double sv[16];
  __shared__ int sj[16];
  picks += (size_t)blockIdx.y * k;
  assign += (size_t)blockIdx.y * N;
  mind += (size_t)blockIdx.y * N;
  const int first = fb.first[blockIdx.y];
  double md[EPT], nj[EPT];
  int lab[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int j = threadIdx.x + q * 1024;
    nj[q] = j < N ? nn[j] : 0.0;
    md[q] = 0.0;
    lab[q] = 0;
  }
  int idx = first;
  for (int c = 0; c < k; ++c) {
    if (threadIdx.x == 0) picks[c] = idx;
    const GT* __restrict__ row = G + (int64_t)idx * ldg;
    const double gii = gram_at(row, idx, idx, gd);
    // clamped addresses, eight loads in flight together (see the multi-workgroup kernel)
    constexpr int LB = EPT < 8 ? EPT : 8;
    double bv = -INFINITY;
    int bj = INT_MAX;
#pragma unroll
    for (int q0 = 0; q0 < EPT; q0 += LB) {
      double gv[LB];
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const int j = threadIdx.x + (q0 + u) * 1024;
        gv[u] = (double)row[j < N ? j : idx];
      }
      // pinned: the compiler otherwise sinks each load into the `j < N` block that uses it and waits for it there
#pragma unroll
      for (int u = 0; u < LB; ++u) asm volatile("" : "+v"(gv[u]));
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const int q = q0 + u;
        const int j = threadIdx.x + q * 1024;
        if (j < N) {
          double d2 = gii - 2.0 * ((sizeof(GT) == 4 && j == idx) ? gii : gv[u]) + nj[q];
          if (!(d2 > 0.0)) d2 = 0.0;
          const double d = sqrt(d2);
          if (c == 0 || d < md[q]) {
            md[q] = d;
            lab[q] = c;
          }
          if (md[q] > bv) {  // j ascends within a thread: strict > keeps the lowest index
            bv = md[q];
            bj = j;
          }
        }
      }
    }
    block_argmax(bv, bj, sv, sj);
    idx = bj;
  }
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int j = threadIdx.x + q * 1024;
    if (j < N) {
      mind[j] = md[q];
      assign[j] = lab[q];
    }
  }
}

// ---- the same initialisation spread over several workgroups (N > 4096) ------------------------------------------
// Workgroup b owns the neurons [b * span, (b + 1) * span) -- running minimum distance, |y_j|^2 and label in registers
// -- and streams only its slice of the picked Gram row.  One exchange per pick: every workgroup posts its local
// (maximum, index) as three self-tagged 8-byte granules {tag = pick + 1, 32 payload bits} (value high word, low word,
// index; one agent-scope store each, so a granule is never torn), wave 0 of every workgroup sweeps all 3 * B granules
// with agent-scope loads until every tag shows the current pick, and each workgroup reduces them the same way
// (larger value, lower index on ties), so all continue with the same pick.  Slots are double-buffered by pick parity:
// a workgroup can be at most one pick ahead of the slowest reader.  The mailbox is zeroed by the launch function
// before every call; spins are bounded.  The exchange needs the workgroups of a try co-resident, which nothing
// guarantees (another stream or process may hold the CUs): a timeout marks the remaining picks -1, every workgroup of
// the try leaves, and the launch function ALWAYS queues the one-workgroup kernel behind this one in its
// "only where picks[k - 1] < 0" mode, which redoes such a try from scratch -- the caller never sees a partial
// initialisation, and nif_kmeans_farthest_init_redo_count says how often that happened.  `inject` (tests only)
// makes the last workgroup declare a timeout at pick 2.
constexpr int FI_MAX_WG = 16;
constexpr int FI_GRANULES = 2 * FI_MAX_WG * 3;          // per call region: [parity][workgroup][hi, lo, idx]
constexpr int FI_REGION_WORDS = FI_GRANULES + 32;       // + timeout word, padded: 128 x 8 bytes = 1 KiB
constexpr int FI_SLOTS = 8;                             // calls in flight at once per device (one slot each) ...
constexpr int FI_REGIONS = FI_SLOTS * 16;               // ... of FI_MAX_TRIES regions, one per try
static __device__ __align__(256) unsigned long long fi_mailbox[FI_REGIONS * FI_REGION_WORDS];
static __device__ unsigned fi_redo_count;               // tries redone by the one-workgroup kernel on this device

template <int EPT, typename GT>
__global__ __launch_bounds__(1024) void kmeans_farthest_init_multi_kernel(const GT* __restrict__ G, int64_t ldg, const double* __restrict__ gd,
                                                                          const double* __restrict__ nn, int N, int k,
                                                                          FiFirsts fb, int span, int* __restrict__ picks,
                                                                          int* __restrict__ assign,
                                                                          double* __restrict__ mind,
                                                                          unsigned long long* mbox, int inject) {
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  __shared__ double sv[16];
  __shared__ int sj[16];
  __shared__ int s_next;
  picks += (size_t)blockIdx.y * k;
  assign += (size_t)blockIdx.y * N;
  mind += (size_t)blockIdx.y * N;
  mbox += (size_t)blockIdx.y * FI_REGION_WORDS;  // every try has a mailbox region of its own
  const int first = fb.first[blockIdx.y];
  const int B = gridDim.x, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int j0 = b * span;
  double md[EPT], nj[EPT];
  int lab[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int j = j0 + tid + q * 1024;
    nj[q] = (tid + q * 1024 < span && j < N) ? nn[j] : 0.0;
    md[q] = 0.0;
    lab[q] = 0;
  }
  gu64* box = (gu64*)mbox;
  gu64* tmo = box + FI_GRANULES;
  int idx = first;
  int c = 0;
  for (; c < k; ++c) {
    if (b == 0 && tid == 0) picks[c] = idx;
    const GT* __restrict__ row = G + (int64_t)idx * ldg;
    const double gii = gram_at(row, idx, idx, gd);
    // the row's entries through clamped addresses, all in flight together (a guarded load per element is waited for
    // before the next is issued: EPT dependent round trips per pick instead of one)
    double gv[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int j = j0 + tid + q * 1024;
      gv[q] = (double)row[(tid + q * 1024 < span && j < N) ? j : idx];
    }
    // pinned: the compiler otherwise sinks the loads into the guarded blocks that use them, one round trip each
#pragma unroll
    for (int q = 0; q < EPT; ++q) asm volatile("" : "+v"(gv[q]));
    double bv = -INFINITY;
    int bj = INT_MAX;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int j = j0 + tid + q * 1024;
      if (tid + q * 1024 < span && j < N) {
        double d2 = gii - 2.0 * ((sizeof(GT) == 4 && j == idx) ? gii : gv[q]) + nj[q];
        if (!(d2 > 0.0)) d2 = 0.0;
        const double d = sqrt(d2);
        if (c == 0 || d < md[q]) {
          md[q] = d;
          lab[q] = c;
        }
        if (md[q] > bv) {  // j ascends within a thread: strict > keeps the lowest index
          bv = md[q];
          bj = j;
        }
      }
    }
    if (c + 1 == k) break;  // nothing left to pick
    block_argmax(bv, bj, sv, sj);
    const unsigned epoch = (unsigned)c + 1u;
    gu64* slots = box + (c & 1) * (FI_MAX_WG * 3);
    if (tid == 0) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(bv);
      const unsigned long long tag = (unsigned long long)epoch << 32;
      __hip_atomic_store(slots + b * 3 + 0, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(slots + b * 3 + 1, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(slots + b * 3 + 2, tag | (unsigned)bj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 64) {  // wave 0 sweeps every workgroup's granules
      const int ng = 3 * B;
      unsigned val = 0;
      bool ok = false, failed = false;
      for (unsigned spins = 0;; ++spins) {
        if (lane < ng) {
          const unsigned long long x = __hip_atomic_load(slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          val = (unsigned)x;
          ok = (unsigned)(x >> 32) == epoch;
        } else {
          ok = true;
        }
        if (__all(ok) && !(inject && c == 2 && b == B - 1)) break;
        if (spins > (1u << 22) || (inject && c == 2 && b == B - 1) ||
            (spins % 64 == 63 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          failed = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      int nxt = -1;
      if (!failed) {
        const int src = lane < B ? 3 * lane : 0;
        const unsigned hi = (unsigned)__shfl((int)val, src, 64), lo = (unsigned)__shfl((int)val, src + 1, 64);
        const int oj = __shfl((int)val, src + 2, 64);
        double v = lane < B ? __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) : -INFINITY;
        const double wmax = -wave_min_f64(-v);
        nxt = (int)wave_min_u32((lane < B && v == wmax) ? (unsigned)oj : 0xffffffffu);
      } else if (lane == 0) {
        __hip_atomic_store(tmo, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) s_next = nxt;
    }
    __syncthreads();
    idx = s_next;
    if (idx < 0) break;  // timed out: uniform over the workgroup
  }
  if (idx < 0 && b == 0)
    for (int cc = c + 1 + tid; cc < k; cc += 1024) picks[cc] = -1;
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int j = j0 + tid + q * 1024;
    if (tid + q * 1024 < span && j < N) {
      mind[j] = md[q];
      assign[j] = lab[q];
    }
  }
}

// ---- first assignment from neurons used as centroids (alf.py:221-229) ------------------------------
// centroid c = neuron c0 + c; assign[j] = argmin_c G[cc][cc] - 2 G[cc][j] + nn[j] (lowest c on ties).
template <typename GT>
__global__ void kmeans_assign_to_neurons_kernel(const GT* __restrict__ G, int64_t ldg, const double* __restrict__ gd,
                                                const double* __restrict__ nn, int N, int k, int c0,
                                                int* __restrict__ assign) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const double nj = nn[j];
  double best = INFINITY;
  int bc = 0;
  // eight centroids' entries in flight, the diagonal (a uniform load) selected in rather than fetched behind a branch:
  // with a guarded load per centroid the k loads arrived one by one (126 us at N = 768, k = 384)
  auto dist2 = [&](int c, double gcj) {
    const int cc = c0 + c;
    const double gcc = sizeof(GT) == 4 ? gd[cc] : (double)G[(int64_t)cc * ldg + cc];
    double d2 = gcc - 2.0 * ((sizeof(GT) == 4 && cc == j) ? gcc : gcj) + nj;
    if (!(d2 > 0.0)) d2 = 0.0;
    if (d2 < best) {  // ascending c: the lowest centroid wins ties
      best = d2;
      bc = c;
    }
  };
  int c = 0;
  for (; c + 8 <= k; c += 8) {
    double g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = (double)G[(int64_t)(c0 + c + q) * ldg + j];
#pragma unroll
    for (int q = 0; q < 8; ++q) dist2(c + q, g[q]);
  }
  for (; c < k; ++c) dist2(c, (double)G[(int64_t)(c0 + c) * ldg + j]);
  assign[j] = bc;
}

// ---- Lloyd step -----------------------------------------------------------------------------------
// clusters above KM_BIG members are summed in a chunked order (see "clusters that have swallowed ..." below)
constexpr int KM_BIG = 1024;      // members
constexpr int KM_BIG_MAX = 64;    // big clusters at most (the rest, if a layer had more, keep the sequential order)
static inline int km_big_slots(int N) { return N / (KM_BIG + 1) < KM_BIG_MAX ? N / (KM_BIG + 1) : KM_BIG_MAX; }

// counts[label]++ (counts zeroed by the caller)
__global__ void kmeans_count_kernel(const int* __restrict__ assign, int N, int* __restrict__ counts, KmBatch kb) {
  assign = kb_row(assign, kb, N);
  counts = kb_ws(counts, kb);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lab = i < N ? assign[i] : -1;
  // the lanes that share the first lane's label add once together: with one label on most of a layer the plain
  // per-lane atomics all hit one address (0.27 ms for 24 576 neurons)
  const int lead = __builtin_amdgcn_readfirstlane(lab);
  const unsigned long long m = __ballot(lab == lead);
  if (lab < 0) return;
  if (lab != lead) atomicAdd(&counts[lab], 1);
  else if ((threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&counts[lab], __popcll(m));
}

// seg = exclusive scan of counts; compact[c] = index of cluster c among the non-empty ones (or -1);
// *kprime = number of non-empty clusters.  One workgroup: lane t owns a run of consecutive clusters.
__global__ __launch_bounds__(1024) void kmeans_scan_kernel(const int* __restrict__ counts, int k,
                                                           int* __restrict__ seg, int* __restrict__ compact,
                                                           int* __restrict__ kprime, int big_slots,
                                                           int* __restrict__ bigflag, int* __restrict__ biglist,
                                                           KmBatch kb) {
  bigflag = kb_ws(bigflag, kb);
  biglist = kb_ws(biglist, kb);
  counts = kb_ws(counts, kb);
  seg = kb_ws(seg, kb);
  compact = kb_ws(compact, kb);
  kprime = kb_row(kprime, kb, 1);
  __shared__ int wsum[16], wlive[16], wbig[16];
  const int per = (k + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(lo + per, k);
  int sum = 0, live = 0, big = 0;
  for (int c = lo; c < hi; ++c) {
    const int n = counts[c];
    sum += n;
    live += n > 0 ? 1 : 0;
    big += n > KM_BIG ? 1 : 0;
  }
  // inclusive scan inside the wave, then over the 16 wave totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int isum = sum, ilive = live, ibig = big;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(isum, o, 64), b = __shfl_up(ilive, o, 64), g = __shfl_up(ibig, o, 64);
    if (lane >= o) {
      isum += a;
      ilive += b;
      ibig += g;
    }
  }
  if (lane == 63) {
    wsum[wave] = isum;
    wlive[wave] = ilive;
    wbig[wave] = ibig;
  }
  __syncthreads();
  int bsum = 0, blive = 0, bbig = 0;
  for (int w = 0; w < wave; ++w) {
    bsum += wsum[w];
    blive += wlive[w];
    bbig += wbig[w];
  }
  int run = bsum + isum - sum, lv = blive + ilive - live, bg = bbig + ibig - big;
  for (int c = lo; c < hi; ++c) {
    const int n = counts[c];
    seg[c] = run;
    run += n;
    compact[c] = n > 0 ? lv++ : -1;
    int slot = -1;
    if (n > KM_BIG) {
      if (bg < big_slots) {
        slot = bg;
        biglist[1 + slot] = c;
      }
      ++bg;
    }
    bigflag[c] = slot;
  }
  if (threadIdx.x == 1023) {
    seg[k] = run;
    *kprime = lv;
    biglist[0] = min(bg, big_slots);
  }
}

// order[seg[c] ...] = members of cluster c in ascending index order; one wavefront per cluster sweeps the
// labels (L2-resident) and compacts its matches with a ballot.
__global__ __launch_bounds__(256) void kmeans_members_kernel(const int* __restrict__ assign, const int* __restrict__ seg,
                                                             int N, int k, int* __restrict__ order, KmBatch kb) {
  assign = kb_row(assign, kb, N);
  seg = kb_ws(seg, kb);
  order = kb_ws(order, kb);
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= k) return;
  const int lane = threadIdx.x & 63;
  int at = seg[c];
  const int end = seg[c + 1];
  for (int i0 = 0; i0 < N && at < end; i0 += 512) {  // eight label loads in flight per lane
    int lab[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 64 + lane;
      lab[u] = i < N ? assign[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool mine = lab[u] == c;
      const unsigned long long m = __ballot(mine);
      if (mine) order[at + __popcll(m & ((1ull << lane) - 1ull))] = i0 + u * 64 + lane;
      at += __popcll(m);
    }
  }
}

// S[cc] = sum of the member weights of cluster c, added one by one in member order.  One WAVE per cluster: the lanes
// fetch 64 members at a time (one dependent load chain per 64 members instead of per member -- a cluster that has
// swallowed half a wide layer took 6 ms with one lane per cluster), then the values are added in lane order.
__global__ __launch_bounds__(256) void kmeans_S_kernel(const float* __restrict__ s, const int* __restrict__ order,
                                                       const int* __restrict__ seg, const int* __restrict__ compact,
                                                       int k, double* __restrict__ S, KmBatch kb) {
  order = kb_ws(order, kb);
  seg = kb_ws(seg, kb);
  compact = kb_ws(compact, kb);
  S = kb_ws(S, kb);
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= k) return;
  const int cc = compact[c];
  if (cc < 0) return;
  const int lane = threadIdx.x & 63;
  const int b = seg[c], e = seg[c + 1];
  double tot = 0.0;
  if (e - b > KM_BIG) {  // big cluster: lane-strided partial sums, then the fixed tree
    for (int t = b + lane; t < e; t += 64) tot += (double)s[order[t]];
    tot = wave_sum_f64(tot);
    if (lane == 0) S[cc] = tot;
    return;
  }
  for (int t0 = b; t0 < e; t0 += 64) {
    const int t = t0 + lane;
    const double val = t < e ? (double)s[order[t]] : 0.0;  // padding adds an exact +0 (tot never holds -0)
    const int lo = __double2loint(val), hi = __double2hiint(val);
#pragma unroll
    for (int l = 0; l < 64; ++l)
      tot += __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
  }
  if (lane == 0) S[cc] = tot;
}

// cost = sum of the per-block partials: lane l adds partial[l], partial[l + 64], ... in order, then a fixed tree
__global__ __launch_bounds__(64) void kmeans_cost_kernel(const double* __restrict__ partial, int nblocks,
                                                         double* __restrict__ cost, KmBatch kb) {
  partial = kb_ws(partial, kb);
  cost = kb_row(cost, kb, 1);
  double t = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) t += partial[b];
  t = wave_sum_f64(t);
  if (threadIdx.x == 0) *cost = t;
}

// Neuron-sharded step: the partials of this rank's blocks leave the workspace as a slice of a [tries][nblocks] tensor
// (the other entries are the other ranks'), and the cost is formed from the exchanged tensor in kmeans_cost_kernel's
// order, so a sharded step returns the bits of an unsharded one.
__global__ void kmeans_partial_export_kernel(const double* __restrict__ partial, int nblocks, int lo, int hi,
                                             double* __restrict__ out, KmBatch kb) {
  partial = kb_ws(partial, kb);
  out = kb_row(out, kb, nblocks);
  const int b = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (b < hi) out[b] = partial[b];
}
__global__ __launch_bounds__(64) void kmeans_cost_rows_kernel(const double* __restrict__ partial, int nblocks,
                                                              double* __restrict__ cost, KmBatch kb) {
  partial = kb_row(partial, kb, nblocks);
  cost = kb_row(cost, kb, 1);
  double t = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) t += partial[b];
  t = wave_sum_f64(t);
  if (threadIdx.x == 0) *cost = t;
}

// ---- Lloyd step: H is never stored ---------------------------------------------------------------------------
// H[c][j] = sum_{i in c} s_i G[i][j] / S_c (k x N) is needed for two things only: the k centroid norms, which use
// H[c][i] for i INSIDE cluster c alone (sum of n_c^2 Gram entries, a sliver of G), and one running minimum per
// neuron.  So: hself[i] = H[c(i)][i] first, the norms from it, then ONE pass over G that forms every H[c][j] in
// registers, turns it into a distance and keeps the minimum: 8 N^2 (f64 G) or 4 N^2 (f32 G) bytes per step.  (The
// first version of this file wrote H and read it back, 16 k N bytes more; this pass gives the same bits.)
// ---- clusters that have swallowed a large part of a layer --------------------------------------------------
// Every sum over a cluster's members above runs in member order on one lane or wave: reproducible bit for bit, and
// fine while clusters are small.  A cluster of n_c members costs n_c^2 sequential work in the in-cluster pass and one
// wave's n_c-row walk in the main pass (a cluster holding half of a 24 576-neuron layer: 11 ms instead of 1 ms per
// step).  Clusters above KM_BIG members therefore take a chunked order, still fixed and deterministic: members in
// chunks of 64, chunk sums combined in chunk order.  Below the threshold nothing changes.

// kmeans_scan_kernel numbers them: bigflag[c] = slot of cluster c (or -1), biglist[0] = count, biglist[1 + slot] = c.
//
// bigtot[slot][j] = sum over the members i of the big cluster of s_i G[i][j], for every neuron j.  Block = 64 neurons
// x 16 waves as in the main pass; wave y walks the member chunks y, y + 16, ... (64 rows each) and the 16 partial
// sums are added in wave order.  The main pass reads its distances to the big clusters from here, and the clusters'
// norms come from the same numbers (|centroid|^2 S^2 = sum over members j of s_j bigtot[j]), so G's rows of a big
// cluster are read once per step.
template <typename GT>
__global__ __launch_bounds__(1024) void kmeans_bigtot_kernel(const GT* __restrict__ G, int64_t ldg,
                                                             const double* __restrict__ gd,
                                                             const float* __restrict__ ssorted,
                                                             const int* __restrict__ order, const int* __restrict__ seg,
                                                             const int* __restrict__ biglist, int N,
                                                             double* __restrict__ bigtot, KmBatch kb) {
  ssorted = kb_ws(ssorted, kb);
  order = kb_ws(order, kb);
  seg = kb_ws(seg, kb);
  biglist = kb_ws(biglist, kb);
  bigtot = kb_ws(bigtot, kb);
  const int nbig = biglist[0];
  if (nbig == 0) return;
  __shared__ double part[16][64];
  const int tx = threadIdx.x & 63;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = (kb.blk0 + (int)blockIdx.x) * 64 + tx;
  const int jc = j < N ? j : N - 1;
  const double gjj = sizeof(GT) == 4 ? gd[jc] : 0.0;
  const GT* __restrict__ col = G + jc;
  for (int bb = 0; bb < nbig; ++bb) {  // uniform over the block
    const int c = biglist[1 + bb];
    const int b = seg[c], e = seg[c + 1];
    double acc = 0.0;
    for (int t0 = b + 64 * ty; t0 < e; t0 += 64 * 16) {
      const int t1 = min(e, t0 + 64);
      int t = t0;
      for (; t + 8 <= t1; t += 8) {
        double g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = order[t + u];
          const double gv = (double)col[(int64_t)i * ldg];
          g[u] = (sizeof(GT) == 4 && i == jc) ? gjj : gv;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)ssorted[t + u] * g[u];
      }
      for (; t < t1; ++t) {
        const int i = order[t];
        const double gv = (double)col[(int64_t)i * ldg];
        acc += (double)ssorted[t] * ((sizeof(GT) == 4 && i == jc) ? gjj : gv);
      }
    }
    __syncthreads();  // part is reused by every big cluster
    part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && j < N) {
      double tot = 0.0;
#pragma unroll
      for (int y = 0; y < 16; ++y) tot += part[y][tx];
      bigtot[(size_t)bb * N + j] = tot;
    }
  }
}

// cn of the big clusters: block = one slot; thread-strided partial sums over the members, then a fixed tree
__global__ __launch_bounds__(256) void kmeans_cn_big_kernel(const double* __restrict__ bigtot, int N,
                                                            const float* __restrict__ ssorted,
                                                            const int* __restrict__ order, const int* __restrict__ seg,
                                                            const int* __restrict__ compact,
                                                            const int* __restrict__ biglist,
                                                            const double* __restrict__ S, double* __restrict__ cn,
                                                            KmBatch kb) {
  bigtot = kb_ws(bigtot, kb);
  ssorted = kb_ws(ssorted, kb);
  order = kb_ws(order, kb);
  seg = kb_ws(seg, kb);
  compact = kb_ws(compact, kb);
  biglist = kb_ws(biglist, kb);
  S = kb_ws(S, kb);
  cn = kb_ws(cn, kb);
  const int slot = blockIdx.x;
  if (slot >= biglist[0]) return;
  __shared__ double wsum[4];
  const int c = biglist[1 + slot];
  const double* __restrict__ tot = bigtot + (size_t)slot * N;
  double acc = 0.0;
  for (int t = seg[c] + threadIdx.x; t < seg[c + 1]; t += 256) acc += (double)ssorted[t] * tot[order[t]];
  acc = wave_sum_f64(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int cc = compact[c];
    cn[cc] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) / S[cc] / S[cc];
  }
}

template <typename GT>
__global__ void kmeans_hself_kernel(const GT* __restrict__ G, int64_t ldg, const double* __restrict__ gd,
                                    const float* __restrict__ s,
                                    const int* __restrict__ order, const int* __restrict__ assign,
                                    const int* __restrict__ seg, const int* __restrict__ compact,
                                    const double* __restrict__ S, int N, double* __restrict__ hself,
                                    const int* __restrict__ bigflag, KmBatch kb) {
  bigflag = kb_ws(bigflag, kb);
  order = kb_ws(order, kb);
  seg = kb_ws(seg, kb);
  compact = kb_ws(compact, kb);
  S = kb_ws(S, kb);
  hself = kb_ws(hself, kb);
  assign = kb_row(assign, kb, N);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // sorted position: a wave's lanes share clusters
  if (t >= N) return;
  const int i = order[t];
  const int c = assign[i];
  if (bigflag[c] >= 0) return;  // big clusters get their norm from kmeans_bigtot_kernel's sums
  double acc = 0.0;
  int u = seg[c];
  const int ue = seg[c + 1];
  // f32 storage: the diagonal entry comes from its f64 array -- loaded once and SELECTED in, not fetched behind a
  // branch (a guarded load per member makes the compiler wait for each load before the next is issued)
  const double gii = sizeof(GT) == 4 ? gd[i] : 0.0;
  for (; u + 8 <= ue; u += 8) {  // eight independent loads in flight, added in member order
    double g[8], w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ip = order[u + q];
      w[q] = (double)s[ip];
      const double gv = (double)G[(int64_t)ip * ldg + i];
      g[q] = (sizeof(GT) == 4 && ip == i) ? gii : gv;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += w[q] * g[q];
  }
  for (; u < ue; ++u) {
    const int ip = order[u];
    const double gv = (double)G[(int64_t)ip * ldg + i];
    acc += (double)s[ip] * ((sizeof(GT) == 4 && ip == i) ? gii : gv);
  }
  hself[i] = acc / S[compact[c]];
}

// cn[cc] = sum_{i in c} s_i hself[i] / S_c in kmeans_cn_kernel's order; one wavefront per cluster
__global__ void kmeans_cn_self_kernel(const double* __restrict__ hself, const float* __restrict__ s,
                                      const int* __restrict__ order, const int* __restrict__ seg,
                                      const int* __restrict__ compact, const double* __restrict__ S, int k,
                                      double* __restrict__ cn, const int* __restrict__ bigflag, KmBatch kb) {
  bigflag = kb_ws(bigflag, kb);
  hself = kb_ws(hself, kb);
  order = kb_ws(order, kb);
  seg = kb_ws(seg, kb);
  compact = kb_ws(compact, kb);
  S = kb_ws(S, kb);
  cn = kb_ws(cn, kb);
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= k) return;
  const int cc = compact[c];
  if (cc < 0 || bigflag[c] >= 0) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (int t = seg[c] + lane; t < seg[c + 1]; t += 64) {
    const int i = order[t];
    acc += (double)s[i] * hself[i];
  }
  acc = wave_sum_f64(acc);
  if (lane == 0) cn[cc] = acc / S[cc];
}

// ssorted[t] = s[order[t]]: the member weights in the order the pass below walks them
__global__ void kmeans_ssorted_kernel(const float* __restrict__ s, const int* __restrict__ order, int N,
                                      float* __restrict__ ssorted, KmBatch kb) {
  order = kb_ws(order, kb);
  ssorted = kb_ws(ssorted, kb);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < N) ssorted[t] = s[order[t]];
}

// Block = 64 neurons x 16 wavefronts; wavefront y takes the y-th sixteenth of the labels (a contiguous run of
// `order`, read through the scalar unit), forms H[c][j] for its 64 neurons member row by member row (256 or 512
// contiguous bytes of G per row) and keeps the smallest distance; the 16 candidates are merged in LDS, lowest compact
// label on ties as in kmeans_nearest_kernel.
template <typename GT>
__global__ __launch_bounds__(1024) void kmeans_nearest_fused_kernel(
    const GT* __restrict__ G, int64_t ldg, const double* __restrict__ gd, const float* __restrict__ ssorted,
    const int* __restrict__ order, const int* __restrict__ seg, const int* __restrict__ compact,
    const double* __restrict__ S,
    const double* __restrict__ cn, const double* __restrict__ nn, const float* __restrict__ s, int N, int k,
    int* __restrict__ assign_out, double* __restrict__ nearest, double* __restrict__ partial,
    const int* __restrict__ bigflag, const int* __restrict__ biglist, const double* __restrict__ bigtot, KmBatch kb) {
  bigflag = kb_ws(bigflag, kb);
  biglist = kb_ws(biglist, kb);
  bigtot = kb_ws(bigtot, kb);
  ssorted = kb_ws(ssorted, kb);
  order = kb_ws(order, kb);
  seg = kb_ws(seg, kb);
  compact = kb_ws(compact, kb);
  S = kb_ws(S, kb);
  cn = kb_ws(cn, kb);
  partial = kb_ws(partial, kb);
  assign_out = kb_row(assign_out, kb, N);
  nearest = kb_row(nearest, kb, N);
  __shared__ double cand_v[16][64];
  __shared__ int cand_c[16][64];
  const int tx = threadIdx.x & 63;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blk = kb.blk0 + (int)blockIdx.x;
  const int j = blk * 64 + tx;
  const int jc = j < N ? j : N - 1;  // lanes past the end compute a valid column and drop it
  const int per = (k + 15) / 16;
  const int c_lo = ty * per, c_hi = min(k, c_lo + per);
  const double nj = nn[jc];
  const double gjj = sizeof(GT) == 4 ? gd[jc] : 0.0;  // f32 storage: the diagonal entry comes from its f64 array
  const GT* __restrict__ col = G + jc;
  double best = INFINITY;
  int bc = INT_MAX;
  for (int c = c_lo; c < c_hi; ++c) {
    const int cc = compact[c];
    if (cc < 0 || bigflag[c] >= 0) continue;  // big clusters: from bigtot, below
    const int b = seg[c], e = seg[c + 1];
    double acc = 0.0;
    int t = b;
    for (; t + 16 <= e; t += 8) {  // chunks of eight while sixteen or more rows are left ...
      double g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = order[t + u];
        const double gv = (double)col[(int64_t)i * ldg];
        g[u] = (sizeof(GT) == 4 && i == jc) ? gjj : gv;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)ssorted[t + u] * g[u];
    }
    // the remainder: exactly the rows that exist, in flight together, summed in the same sequence
    auto tail = [&](auto width) {
      constexpr int R = decltype(width)::value;
      int idx[R];
      float wv[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        idx[u] = order[t + u];
        wv[u] = ssorted[t + u];
      }
      GT gr[R];
#pragma unroll
      for (int u = 0; u < R; ++u) gr[u] = col[(int64_t)idx[u] * ldg];
      // the values are pinned here: left alone, the compiler sinks the last load into the branch of the lanes whose own
      // row it is not and waits for it there, after the others -- a second round trip
#pragma unroll
      for (int u = 0; u < R; ++u) asm volatile("" : "+v"(gr[u]));
#pragma unroll
      for (int u = 0; u < R; ++u)
        acc += (double)wv[u] * ((sizeof(GT) == 4 && idx[u] == jc) ? gjj : (double)gr[u]);
    };
    switch (e - t) {  // ... then all the rest at once (at most fifteen rows: one round trip for nearly every cluster)
#define NIF_KM_TAIL(R) case R: tail(std::integral_constant<int, R>()); break;
      NIF_KM_TAIL(1) NIF_KM_TAIL(2) NIF_KM_TAIL(3) NIF_KM_TAIL(4) NIF_KM_TAIL(5)
      NIF_KM_TAIL(6) NIF_KM_TAIL(7) NIF_KM_TAIL(8) NIF_KM_TAIL(9) NIF_KM_TAIL(10)
      NIF_KM_TAIL(11) NIF_KM_TAIL(12) NIF_KM_TAIL(13) NIF_KM_TAIL(14) NIF_KM_TAIL(15)
#undef NIF_KM_TAIL
      default: break;
    }
    const double d2 = cn[cc] - 2.0 * (acc / S[cc]) + nj;
    if (d2 < best) {
      best = d2;
      bc = cc;
    }
  }
  cand_v[ty][tx] = best;
  cand_c[ty][tx] = bc;
  __syncthreads();
  if (ty != 0) return;
  for (int y = 1; y < 16; ++y) {
    const double ov = cand_v[y][tx];
    const int oc = cand_c[y][tx];
    if (ov < best || (ov == best && oc < bc)) {
      best = ov;
      bc = oc;
    }
  }
  const int nbig = biglist[0];
  for (int bb = 0; bb < nbig; ++bb) {  // big clusters: their sums were formed by kmeans_bigtot_kernel
    const int cc = compact[biglist[1 + bb]];
    const double d2 = cn[cc] - 2.0 * (bigtot[(size_t)bb * N + jc] / S[cc]) + nj;
    if (d2 < best || (d2 == best && cc < bc)) {
      best = d2;
      bc = cc;
    }
  }
  double contrib = 0.0;
  if (j < N) {
    if (bc == INT_MAX) bc = 0;
    if (!(best > 0.0)) best = 0.0;
    const double d = sqrt(best);
    assign_out[j] = bc;
    nearest[j] = d;
    contrib = (double)s[j] * d;
  }
  contrib = wave_sum_f64(contrib);
  if (tx == 0) partial[blk] = contrib;
}

// ---- KF empty-cluster repair (hf_kf.py:466-471), one workgroup --------------------------------------
// for c in 0..k-1: if no neuron carries label c, the neuron with the largest s * distance moves to c
// and its distance becomes 0.  Also reports whether the repaired assignment differs from `prev`.
__global__ __launch_bounds__(1024) void kmeans_repair_kf_kernel(int* __restrict__ assign,
                                                                double* __restrict__ nearest,
                                                                const float* __restrict__ s,
                                                                const int* __restrict__ prev, int N, int k,
                                                                int* __restrict__ counts,
                                                                int* __restrict__ changed, KmBatch kb) {
  assign = kb_row(assign, kb, N);
  nearest = kb_row(nearest, kb, N);
  prev = kb_row(prev, kb, N);
  counts = kb_ws(counts, kb);
  changed = kb_row(changed, kb, 1);
  __shared__ double sv[16];
  __shared__ int sj[16];
  __shared__ int flag;
  for (int c = threadIdx.x; c < k; c += blockDim.x) counts[c] = 0;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += blockDim.x) atomicAdd(&counts[assign[j]], 1);
  __syncthreads();
  for (int c = 0; c < k; ++c) {
    if (counts[c] != 0) continue;  // same word for every thread; writers below are fenced by barriers
    double bv = -INFINITY;
    int bj = INT_MAX;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      const double v = (double)s[j] * nearest[j];
      if (v > bv) {
        bv = v;
        bj = j;
      }
    }
    block_argmax(bv, bj, sv, sj);
    if (threadIdx.x == 0) {
      counts[assign[bj]] -= 1;
      counts[c] += 1;
      assign[bj] = c;
      nearest[bj] = 0.0;
    }
    __syncthreads();
  }
  int diff = 0;
  for (int j = threadIdx.x; j < N; j += blockDim.x) diff |= (assign[j] != prev[j]) ? 1 : 0;
  if (diff) atomicOr(&flag, 1);
  __syncthreads();
  if (threadIdx.x == 0) *changed = flag;
}

// weighted cluster sums of LS solution columns -> rows of the fused weight (hf_kf.py:494-504):
// out[cc][r] = sum_{i in c} s_i W[r][i] / S_c, cc = compact label; rows cc >= k' stay zero.
__global__ void kmeans_centroid_weights_kernel(const double* __restrict__ W, int64_t ldw, const float* __restrict__ s,
                                               const int* __restrict__ order, const int* __restrict__ seg,
                                               const int* __restrict__ compact, int d, float* __restrict__ out) {
  const int c = blockIdx.y;
  const int cc = compact[c];
  if (cc < 0) return;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= d) return;
  const int b = seg[c], e = seg[c + 1];
  double tot = 0.0, acc = 0.0;
  for (int t = b; t < e; ++t) {
    const int i = order[t];
    const double si = (double)s[i];
    tot += si;
    acc += si * W[(int64_t)r * ldw + i];
  }
  out[(int64_t)cc * d + r] = (float)(acc / tot);
}

// ALF targets (alf.py:288-289): T[r][cc] = sum_{i in c} w_i A[r][i] / S_c for every sample row r --
// the importance-weighted centroid of the member neurons' (un-normalised) activations.
// One thread per (row, cluster); lanes walk consecutive clusters of one row, whose members all
// lie inside that row's N floats (served from L1/L2 after the first touch), and write T coalesced.
__global__ void kmeans_targets_kernel(const float* __restrict__ A, int64_t lda, int64_t m, const float* __restrict__ w,
                                      const int* __restrict__ order, const int* __restrict__ seg,
                                      const int* __restrict__ compact, int k, float* __restrict__ T, int64_t ldt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  const int cc = compact[c];
  const int b = seg[c], e = seg[c + 1];
  double tot = 0.0;
  for (int t = b; t < e; ++t) tot += (double)w[order[t]];
  for (int64_t r = blockIdx.y; r < m; r += gridDim.y) {
    if (cc < 0) continue;
    const float* __restrict__ row = A + r * lda;
    double acc = 0.0;
    for (int t = b; t < e; ++t) {
      const int i = order[t];
      acc += (double)w[i] * (double)row[i];
    }
    T[r * ldt + cc] = (float)(acc / tot);
  }
}

// Same result through LDS: a block stages R whole rows of A (coalesced, each element read from HBM once), then
// every lane owns (row, cluster) pairs and walks the cluster's members inside the staged row.
__global__ void kmeans_targets_lds_kernel(const float* __restrict__ A, int64_t lda, int64_t m,
                                          const float* __restrict__ w, const int* __restrict__ order,
                                          const int* __restrict__ seg, const int* __restrict__ compact,
                                          const double* __restrict__ S, int N, int k, int R,
                                          float* __restrict__ T, int64_t ldt) {
  extern __shared__ __attribute__((aligned(16))) float km_tile[];  // [R][N]
  const int nthreads = blockDim.x;
  const bool vec = (N % 4 == 0) && (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  for (int64_t r0 = (int64_t)blockIdx.x * R; r0 < m; r0 += (int64_t)gridDim.x * R) {
    const int rows = (int)((m - r0) < R ? (m - r0) : R);
    if (vec) {
      const int n4 = N >> 2;
      for (int idx = threadIdx.x; idx < rows * n4; idx += nthreads) {
        const int rr = idx / n4, q = idx - rr * n4;
        reinterpret_cast<float4*>(km_tile)[rr * n4 + q] =
            reinterpret_cast<const float4*>(A + (r0 + rr) * lda)[q];
      }
    } else {
      for (int idx = threadIdx.x; idx < rows * N; idx += nthreads) {
        const int rr = idx / N, q = idx - rr * N;
        km_tile[rr * N + q] = A[(r0 + rr) * lda + q];
      }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < rows * k; p += nthreads) {
      const int rr = p / k, c = p - rr * k;
      const int cc = compact[c];
      if (cc < 0) continue;
      const float* __restrict__ row = km_tile + rr * N;
      double acc = 0.0;
      for (int t = seg[c]; t < seg[c + 1]; ++t) {
        const int i = order[t];
        acc += (double)w[i] * (double)row[i];
      }
      T[(r0 + rr) * ldt + cc] = (float)(acc / S[cc]);
    }
    __syncthreads();
  }
}

// The wide-load version.  What bounds the kernel above is not HBM but the CU's vector-memory pipe: every (row,
// cluster) pair re-reads seg / compact / S and chases order[t] -> w[i] through scattered 4-byte global loads, and the
// loads of the next tile only start after the walk.  Here
//   * a row is staged in LDS in CLUSTER-SORTED order (element i goes to position pos[i], its place in `order`; the
//     positions of the elements a lane stages are the same for every tile and live in registers), bank-swizzled so
//     lanes that walk neighbouring clusters do not collide;
//   * a lane owns CLUSTERS, not pairs: its clusters' segment, output column and weight total are read once, and it
//     walks its clusters in every row of the tile it is assigned;
//   * the weights sit in LDS next to the tile, in the same sorted, swizzled order, as far as the CU's 160 KB allow
//     (WL sorted positions).  Clusters that reach past WL (only wide levels have any) keep their first four weights
//     in registers and take the rest from the member-major table wT[t * k + c] (coalesced across a wave);
//   * the NEXT tile is fetched into registers (PF float4 per lane) as soon as no other global load can follow: after
//     the clusters that read wT, before the ones that live in LDS (the memory counter retires in order: a later wT
//     load would make the walk wait for the whole prefetch).  Every load and LDS read is unconditional (clamped
//     address + select), so nothing serialises behind a branch.
// Same members in the same order with the same f64 arithmetic: same bits as above.  Columns >= *kprime are zeroed here.
// Requires N % 4 == 0, lda % 4 == 0, A 16-byte aligned, R * N / 4 <= PF * blockDim.x, k <= CPT * blockDim.x,
// R * N < 65536, R * lda < 2^31; LDS holds R * N floats rounded up to 64, then WL floats (WL % 64 == 0).
__global__ void kmeans_sorted_kernel(const int* __restrict__ order, const int* __restrict__ assign,
                                     const int* __restrict__ seg, const float* __restrict__ w, int N, int k,
                                     int* __restrict__ pos, float* __restrict__ wsorted, float* __restrict__ wT) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const int i = order[t];
  const int c = assign[i];
  const float wi = w[i];
  pos[i] = t;
  wsorted[t] = wi;
  wT[(int64_t)(t - seg[c]) * k + c] = wi;
}

__device__ __forceinline__ int km_sw(int a) { return a ^ ((a >> 6) & 63); }

template <int PF, int CPT>
__global__ __launch_bounds__(1024) void kmeans_targets_sw_kernel(
    const float* __restrict__ A, int64_t lda, int64_t m, const float* __restrict__ wsorted,
    const float* __restrict__ wT, const int* __restrict__ pos, const int* __restrict__ seg,
    const int* __restrict__ compact, const double* __restrict__ S, const int* __restrict__ kprime, int N, int k, int R,
    int WL, float* __restrict__ T, int64_t ldt) {
  extern __shared__ __attribute__((aligned(16))) float km_tile[];  // [R][N] sorted + swizzled rows, then WL weights
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const int n4 = N >> 2;
  const int kp = kprime[0];
  float* __restrict__ wl = km_tile + ((R * N + 63) & ~63);
  for (int p = tid; p < WL; p += nthreads) wl[km_sw(p)] = p < N ? wsorted[p] : 0.0f;
  // per staging slot: the float4 in flight, its four LDS positions (16 bits each) and its element offset inside the
  // tile's block of rows
  float4 pf[PF];
  unsigned pxy[PF], pzw[PF];
  int foff[PF];
#pragma unroll
  for (int e = 0; e < PF; ++e) {
    const int idx = tid + e * nthreads;
    pxy[e] = pzw[e] = 0u;
    foff[e] = -1;
    if (idx < R * n4) {
      const int rr = idx / n4, q = idx - rr * n4;
      const int4 pq = reinterpret_cast<const int4*>(pos)[q];
      pxy[e] = (unsigned)km_sw(rr * N + pq.x) | ((unsigned)km_sw(rr * N + pq.y) << 16);
      pzw[e] = (unsigned)km_sw(rr * N + pq.z) | ((unsigned)km_sw(rr * N + pq.w) << 16);
      foff[e] = rr * (int)lda + 4 * q;
    }
  }
  const int full_lim = (R - 1) * (int)lda + N;  // element offsets below rows * lda belong to rows of a ragged last tile
  auto fetch = [&](int64_t r0) {
    const int rows = (int)((m - r0) < R ? (m - r0) : R);
    const int lim = rows == R ? full_lim : (rows - 1) * (int)lda + N;
    const float* __restrict__ base = A + r0 * lda;
#pragma unroll
    for (int e = 0; e < PF; ++e) {  // branch-free: a slot without an element re-reads the tile's first one
      const int off = (foff[e] >= 0 && foff[e] < lim) ? foff[e] : 0;
      pf[e] = *reinterpret_cast<const float4*>(base + off);
    }
  };
  // cluster ownership: with k <= blockDim the workgroup splits into G groups of k lanes, group g takes rows g, g + G, ...
  int G = 1, g = 0, cbase = tid;
  if (CPT == 1) {
    G = nthreads / k;
    g = tid / k;
    cbase = tid - g * k;
  }
  int ccc[CPT];
  unsigned cseg[CPT];  // first member | member count << 16
  double cS[CPT];
  float wfar[8];       // the first eight weights of this lane's first far cluster
  unsigned far = 0u;  // bit j: cluster j reaches past the weights held in LDS ("far")
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = cbase + j * nthreads;
    ccc[j] = -1; cseg[j] = 0u; cS[j] = 1.0;
    const bool mine = CPT == 1 ? g < G : c < k;
    if (mine) {
      const int cc = compact[c];
      if (cc >= 0) {
        const int t0 = seg[c], n = seg[c + 1] - t0;
        ccc[j] = cc;
        cseg[j] = (unsigned)t0 | ((unsigned)n << 16);
        cS[j] = S[cc];
        if (t0 + n > WL) far |= 1u << j;
      }
    }
  }
  const int jf = far ? __builtin_ctz(far) : 0;
  {
    unsigned sf = 0u;
#pragma unroll
    for (int j = 0; j < CPT; ++j) sf = j == jf ? cseg[j] : sf;
    const int c = cbase + jf * nthreads, n = far ? (int)(sf >> 16) : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) wfar[u] = u < n ? wT[(int64_t)u * k + c] : 0.0f;
  }
  const int64_t stride = (int64_t)gridDim.x * R;
  int64_t r0 = (int64_t)blockIdx.x * R;
  if (r0 < m) fetch(r0);
  for (; r0 < m; r0 += stride) {
    const int rows = (int)((m - r0) < R ? (m - r0) : R);
    {
      const int lim = rows == R ? full_lim : (rows - 1) * (int)lda + N;
#pragma unroll
      for (int e = 0; e < PF; ++e) {
        if (foff[e] >= 0 && foff[e] < lim) {
          km_tile[pxy[e] & 0xffffu] = pf[e].x;
          km_tile[pxy[e] >> 16] = pf[e].y;
          km_tile[pzw[e] & 0xffffu] = pf[e].z;
          km_tile[pzw[e] >> 16] = pf[e].w;
        }
      }
    }
    __syncthreads();
    // absent members add an exact +0 in both walks (acc never holds -0: it starts at +0)
    if (__syncthreads_or(far != 0u)) {
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        if (ccc[j] < 0 || !((far >> j) & 1u)) continue;
        const int t0 = (int)(cseg[j] & 0xffffu), n = (int)(cseg[j] >> 16);
        const int c = cbase + j * nthreads;
        for (int rr = g; rr < rows; rr += G) {
          const int base = rr * N + t0;
          double acc = 0.0;
          int t = 0;
          if (j == jf) {
            float aq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float v = km_tile[km_sw(base + (u < n ? u : 0))];
              aq[u] = u < n ? v : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)wfar[u] * (double)aq[u];
            t = 8;
          }
          for (; t < n; t += 4) {
            float wq[4], aq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const bool in = t + u < n;
              const int tu = in ? t + u : t;
              const float wv = wT[(int64_t)tu * k + c];
              const float v = km_tile[km_sw(base + tu)];
              wq[u] = in ? wv : 0.0f;
              aq[u] = in ? v : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += (double)wq[u] * (double)aq[u];
          }
          T[(r0 + rr) * ldt + ccc[j]] = (float)(acc / cS[j]);
        }
      }
    }
    if (r0 + stride < m) fetch(r0 + stride);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      if (ccc[j] < 0 || ((far >> j) & 1u)) continue;
      const int t0 = (int)(cseg[j] & 0xffffu), n = (int)(cseg[j] >> 16);
      float w4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (CPT == 1) {  // narrow levels: the lane's one cluster serves several rows of the tile with the same weights
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float wv = wl[km_sw(t0 + (u < n ? u : 0))];
          w4[u] = u < n ? wv : 0.0f;
        }
      }
      for (int rr = g; rr < rows; rr += G) {
        const int base = rr * N + t0;
        double acc = 0.0;
        if (CPT == 1) {
          float aq[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float v = km_tile[km_sw(base + (u < n ? u : 0))];
            aq[u] = u < n ? v : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc += (double)w4[u] * (double)aq[u];
        }
        for (int t = CPT == 1 ? 4 : 0; t < n; t += 4) {
          float wq[4], aq[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool in = t + u < n;
            const int tu = in ? t + u : t;
            const float wv = wl[km_sw(t0 + tu)];
            const float v = km_tile[km_sw(base + tu)];
            wq[u] = in ? wv : 0.0f;
            aq[u] = in ? v : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc += (double)wq[u] * (double)aq[u];
        }
        T[(r0 + rr) * ldt + ccc[j]] = (float)(acc / cS[j]);
      }
    }
    const int nz = k - kp;  // columns of empty clusters stay zero (hf_kf.py:494-504 / alf.py:288-289 compaction)
    for (int p = tid; p < rows * nz; p += nthreads) {
      const int rr = p / nz, cz = p - rr * nz;
      T[(r0 + rr) * ldt + kp + cz] = 0.0f;
    }
    __syncthreads();
  }
}

constexpr size_t KM_LDS_BUDGET = 159 * 1024;  // of the CU's 160 KB

template <int PF, int CPT>
static bool km_launch_sw(int64_t blocks, hipStream_t st, const float* A, int64_t lda, int64_t m, const float* wsorted,
                         const float* wT, const int* pos, const int* seg, const int* compact, const double* S,
                         const int* kprime, int N, int k, int R, float* T, int64_t ldt) {
  auto kern = kmeans_targets_sw_kernel<PF, CPT>;
  const size_t tile = (((size_t)R * N + 63) / 64) * 64 * sizeof(float);
  size_t wl = (((size_t)N + 63) / 64) * 64;
  if (tile + wl * sizeof(float) > KM_LDS_BUDGET) wl = (KM_LDS_BUDGET - tile) / sizeof(float) / 64 * 64;
  const size_t lds = tile + wl * sizeof(float);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int per_cu = 0;  // resident workgroups per CU walk the row tiles, each with its next tile in flight
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 1024, lds) != hipSuccess || per_cu < 1) return false;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    cus = 256;
  int64_t pb = (int64_t)cus * per_cu;
  if (pb > blocks) pb = blocks;
  hipLaunchKernelGGL(kern, dim3((unsigned)pb), dim3(1024), lds, st, A, lda, m, wsorted, wT, pos, seg, compact, S, kprime,
                     N, k, R, (int)wl, T, ldt);
  return true;
}

struct KmWs {
  size_t off_rank, off_counts, off_seg, off_compact, off_order, off_wT, off_S, off_cn, off_partial, off_pos, off_wsorted, off_hself, off_bigflag, off_biglist, off_bigtot, total;
  int nblk;
};
static inline size_t ka(size_t x) { return (x + 255) & ~(size_t)255; }
static KmWs km_ws(int N, int k) {
  KmWs w;
  w.nblk = (N + 255) / 256;
  size_t o = 0;
  w.off_rank = o; o += ka((size_t)N * 4);
  w.off_counts = o; o += ka((size_t)k * 4);
  w.off_seg = o; o += ka((size_t)(k + 1) * 4);
  w.off_compact = o; o += ka((size_t)k * 4);
  w.off_order = o; o += ka((size_t)N * 4);
  w.off_S = o; o += ka((size_t)k * 8);
  w.off_cn = o; o += ka((size_t)k * 8);
  w.off_partial = o; o += ka((size_t)((N + 63) / 64) * 8);
  w.off_pos = o; o += ka((size_t)N * 4);
  w.off_wsorted = o; o += ka((size_t)N * 4);
  w.off_hself = o; o += ka((size_t)N * 8);
  w.off_bigflag = o; o += ka((size_t)k * 4);
  w.off_biglist = o; o += ka((size_t)(1 + KM_BIG_MAX) * 4);
  w.off_bigtot = o; o += ka((size_t)km_big_slots(N) * N * 8);
  w.off_wT = o; o += ka((size_t)k * N * 4);  // member-major weights of nif_kmeans_targets: (largest cluster) x k
  w.total = o;
  return w;
}

// builds order / seg / compact / kprime for `assign` inside the workspace
static KmBatch km_single() {
  KmBatch kb;
  for (int t = 0; t < FI_MAX_TRIES; ++t) kb.idx[t] = 0;
  kb.ws_stride = 0;
  kb.blk0 = 0;
  return kb;
}

// builds order / seg / compact / kprime for `assign` inside the workspace (of every try of the batch)
static void km_group(const int* assign, int N, int k, const KmWs& w, unsigned char* b, int* kprime, hipStream_t st,
                     const KmBatch& kb = km_single(), int L = 1) {
  int* counts = reinterpret_cast<int*>(b + w.off_counts);
  int* seg = reinterpret_cast<int*>(b + w.off_seg);
  int* compact = reinterpret_cast<int*>(b + w.off_compact);
  int* order = reinterpret_cast<int*>(b + w.off_order);
  for (int e = 0; e < L; ++e)
    (void)hipMemsetAsync(reinterpret_cast<unsigned char*>(counts) + (size_t)kb.idx[e] * (size_t)kb.ws_stride, 0,
                         (size_t)k * 4, st);
  hipLaunchKernelGGL(kmeans_count_kernel, dim3(w.nblk, L), dim3(256), 0, st, assign, N, counts, kb);
  hipLaunchKernelGGL(kmeans_scan_kernel, dim3(1, L), dim3(1024), 0, st, counts, k, seg, compact, kprime, km_big_slots(N),
                     reinterpret_cast<int*>(b + w.off_bigflag), reinterpret_cast<int*>(b + w.off_biglist), kb);
  hipLaunchKernelGGL(kmeans_members_kernel, dim3((k + 3) / 4, L), dim3(256), 0, st, assign, seg, N, k, order, kb);
}

}  // namespace nif

extern "C" size_t nif_kmeans_workspace_bytes(int N, int k) {
  if (N <= 0 || k <= 0) return 0;
  return nif::km_ws(N, k).total;
}

namespace nif {

// mailbox slots of the multi-workgroup farthest-point initialisation, per device (shared by the f32 and f64 storage
// instantiations of the launch function, hence not inside the template)
struct FiDevice {
  unsigned long long* base = nullptr;
  unsigned* redo = nullptr;
  hipEvent_t done[FI_SLOTS] = {};
  bool used[FI_SLOTS] = {};
  unsigned next = 0;
};
static std::mutex fi_mu;
static FiDevice fi_devs[64];

static int fi_device(int* dev_out) {  // fi_mu held
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return NIF_ERR_NO_DEVICE;
  FiDevice& fd = fi_devs[dev];
  if (!fd.base) {
    void *p = nullptr, *q = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(fi_mailbox)) != hipSuccess || !p) return NIF_ERR_NO_DEVICE;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(fi_redo_count)) != hipSuccess || !q) return NIF_ERR_NO_DEVICE;
    fd.base = reinterpret_cast<unsigned long long*>(p);
    fd.redo = reinterpret_cast<unsigned*>(q);
  }
  *dev_out = dev;
  return NIF_OK;
}

static int fi_acquire(int* dev_out, int* slot_out) {  // fi_mu held
  const int rc = fi_device(dev_out);
  if (rc != NIF_OK) return rc;
  FiDevice& fd = fi_devs[*dev_out];
  int slot = -1;
  for (int i = 0; i < FI_SLOTS && slot < 0; ++i) {
    const int c = (int)((fd.next + (unsigned)i) % FI_SLOTS);
    if (!fd.used[c] || hipEventQuery(fd.done[c]) == hipSuccess) slot = c;
  }
  if (slot < 0) {
    slot = (int)(fd.next % FI_SLOTS);
    if (hipEventSynchronize(fd.done[slot]) != hipSuccess) return NIF_ERR_LAUNCH;
  }
  (void)hipGetLastError();  // hipEventQuery leaves hipErrorNotReady behind
  fd.next = (unsigned)slot + 1u;
  if (!fd.done[slot] && hipEventCreateWithFlags(&fd.done[slot], hipEventDisableTiming) != hipSuccess)
    return NIF_ERR_LAUNCH;
  *slot_out = slot;
  return NIF_OK;
}

template <typename GT>
static int km_farthest_init(const GT* G, int64_t ldg, const double* gd, const double* nn, int N, int k,
                            const int32_t* firsts, int tries, int32_t* picks, int32_t* assign, double* mind,
                            void* stream) {
  if (!G || !nn || !firsts || !picks || !assign || !mind || N <= 0 || k <= 0 || ldg < N || tries < 1 ||
      tries > FI_MAX_TRIES || (sizeof(GT) == 4 && !gd))
    return NIF_ERR_INVALID_ARG;
  FiFirsts fb;
  for (int t = 0; t < FI_MAX_TRIES; ++t) {
    fb.first[t] = t < tries ? firsts[t] : 0;
    if (fb.first[t] < 0 || fb.first[t] >= N) return NIF_ERR_INVALID_ARG;
  }
  clear_err();
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static const int fi_single = env_int("NIF_KMEANS_INIT_SINGLE_WG");  // tuning aid: keep the one-workgroup kernel
  if (N > 4096 && N <= FI_MAX_WG * 4096 && !fi_single) {
    // several workgroups per try, one mailbox exchange per pick (see kmeans_farthest_init_multi_kernel).  Each call
    // takes one of FI_SLOTS mailbox slots of its device; a slot is handed out again only once the event recorded
    // behind its last user has completed (waited for if all slots are busy), so calls in flight on different streams
    // never share or zero a live region.
    std::lock_guard<std::mutex> lk(fi_mu);  // held over the launches: the slot's event must be recorded before a re-use
    int dev = -1, slot = -1;
    const int rc = fi_acquire(&dev, &slot);
    if (rc != NIF_OK) return rc;
    FiDevice& fd = fi_devs[dev];
    unsigned long long* region = fd.base + (size_t)slot * FI_MAX_TRIES * FI_REGION_WORDS;
    int B = (N + 2047) / 2048;            // two neurons per lane where that fills <= 16 workgroups
    if (B > FI_MAX_WG) B = FI_MAX_WG;
    const int span = (N + B - 1) / B;
    const int ept = (span + 1023) / 1024;
    const int inject = env_int("NIF_KMEANS_INIT_INJECT_TIMEOUT");  // tests: force the timeout path
    if (hipMemsetAsync(region, 0, (size_t)tries * FI_REGION_WORDS * sizeof(unsigned long long), st) != hipSuccess)
      return NIF_ERR_LAUNCH;
#define NIF_FIM_GO(E)                                                                                                  \
  hipLaunchKernelGGL((kmeans_farthest_init_multi_kernel<E, GT>), dim3(B, tries), dim3(1024), 0, st, G, ldg, gd, nn, N, \
                     k, fb, span, picks, assign, mind, region, inject)
    if (ept <= 1) NIF_FIM_GO(1);
    else if (ept <= 2) NIF_FIM_GO(2);
    else NIF_FIM_GO(4);
#undef NIF_FIM_GO
    // a try whose exchange timed out is redone by one workgroup; the others return at once
    hipLaunchKernelGGL(kmeans_farthest_init_kernel<GT>, dim3(1, tries), dim3(1024), 0, st, G, ldg, gd, nn, N, k, fb, picks,
                       assign, mind, fd.redo);
    const bool ok = launch_ok();
    fd.used[slot] = hipEventRecord(fd.done[slot], st) == hipSuccess;
    if (!fd.used[slot]) (void)hipStreamSynchronize(st);  // no event to guard the slot: drain before it can be re-used
    return ok ? NIF_OK : NIF_ERR_LAUNCH;
  }
  const int ept = (N + 1023) / 1024;
#define NIF_FI_GO(E)                                                                                                   \
  hipLaunchKernelGGL((kmeans_farthest_init_reg_kernel<E, GT>), dim3(1, tries), dim3(1024), 0, st, G, ldg, gd, nn, N, k, \
                     fb, picks, assign, mind)
  if (ept <= 1) NIF_FI_GO(1);
  else if (ept <= 2) NIF_FI_GO(2);
  else if (ept <= 4) NIF_FI_GO(4);
  else if (ept <= 8) NIF_FI_GO(8);
  else if (ept <= 16) NIF_FI_GO(16);
  else if (ept <= 32) NIF_FI_GO(32);
  else
    hipLaunchKernelGGL(kmeans_farthest_init_kernel<GT>, dim3(1, tries), dim3(1024), 0, st, G, ldg, gd, nn, N, k, fb, picks,
                       assign, mind, (unsigned*)nullptr);
#undef NIF_FI_GO
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

template <typename GT>
static int km_assign_to_neurons(const GT* G, int64_t ldg, const double* gd, const double* nn, int N, int k, int c0,
                                int32_t* assign, void* stream) {
  if (!G || !nn || !assign || N <= 0 || k <= 0 || ldg < N || c0 < 0 || c0 + k > N || (sizeof(GT) == 4 && !gd))
    return NIF_ERR_INVALID_ARG;
  clear_err();
  hipLaunchKernelGGL(kmeans_assign_to_neurons_kernel<GT>, dim3((N + 255) / 256), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), G, ldg, gd, nn, N, k, c0, assign);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

// tries_idx (host, L entries, each in [0, FI_MAX_TRIES)): the tries this call advances; nullptr = the single problem
static int km_batch(const int32_t* tries_idx, int L, size_t stride, size_t ws_bytes, KmBatch* kb) {
  *kb = km_single();
  if (!tries_idx) return L == 1 ? NIF_OK : NIF_ERR_INVALID_ARG;
  if (L < 1 || L > FI_MAX_TRIES) return NIF_ERR_INVALID_ARG;
  int top = 0;
  for (int e = 0; e < L; ++e) {
    if (tries_idx[e] < 0 || tries_idx[e] >= FI_MAX_TRIES) return NIF_ERR_INVALID_ARG;
    kb->idx[e] = tries_idx[e];
    if (tries_idx[e] > top) top = tries_idx[e];
  }
  kb->ws_stride = (long long)stride;
  return ws_bytes >= (size_t)(top + 1) * stride ? NIF_OK : NIF_ERR_WORKSPACE;
}

template <typename GT>
static int km_lloyd_step(const GT* G, int64_t ldg, const double* gd, const double* nn, const float* s,
                         const int32_t* assign, int N, int k, const int32_t* tries_idx, int L, int32_t* assign_out,
                         double* nearest, double* cost, int32_t* kprime, void* ws, size_t ws_bytes, void* stream,
                         int blk_lo = 0, int blk_hi = -1, double* partial_out = nullptr) {
  // partial_out set: the neuron-sharded form.  Only the 64-neuron blocks [blk_lo, blk_hi) go through the two passes
  // over G; their slices of assign_out / nearest / partial_out are written and `cost` is not (see
  // nif_kmeans_lloyd_step_range in nif.h).  The grouping and the per-cluster kernels run in full on every rank: they
  // read O(N) labels and the member-by-member sliver of G, not G.
  const bool ranged = partial_out != nullptr;
  if (!G || !nn || !s || !assign || !assign_out || !nearest || (!cost && !ranged) || !kprime || N <= 0 || k <= 0 ||
      ldg < N || (sizeof(GT) == 4 && !gd))
    return NIF_ERR_INVALID_ARG;
  const int nb64 = (N + 63) / 64;
  if (!ranged) {
    blk_lo = 0;
    blk_hi = nb64;
  } else if (blk_lo < 0 || blk_hi < blk_lo || blk_hi > nb64) {
    return NIF_ERR_INVALID_ARG;
  }
  const KmWs w = km_ws(N, k);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  KmBatch kb;
  const int brc = km_batch(tries_idx, L, w.total, ws_bytes, &kb);
  if (brc != NIF_OK) return brc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  int* seg = reinterpret_cast<int*>(b + w.off_seg);
  int* compact = reinterpret_cast<int*>(b + w.off_compact);
  int* order = reinterpret_cast<int*>(b + w.off_order);
  double* S = reinterpret_cast<double*>(b + w.off_S);
  double* cn = reinterpret_cast<double*>(b + w.off_cn);
  double* partial = reinterpret_cast<double*>(b + w.off_partial);
  double* hself = reinterpret_cast<double*>(b + w.off_hself);
  float* ssorted = reinterpret_cast<float*>(b + w.off_wsorted);
  km_group(assign, N, k, w, b, kprime, st, kb, L);
  hipLaunchKernelGGL(kmeans_S_kernel, dim3((k + 3) / 4, L), dim3(256), 0, st, s, order, seg, compact, k, S, kb);
  hipLaunchKernelGGL(kmeans_ssorted_kernel, dim3(w.nblk, L), dim3(256), 0, st, s, order, N, ssorted, kb);
  int* bigflag = reinterpret_cast<int*>(b + w.off_bigflag);
  int* biglist = reinterpret_cast<int*>(b + w.off_biglist);
  double* bigtot = reinterpret_cast<double*>(b + w.off_bigtot);
  hipLaunchKernelGGL(kmeans_hself_kernel<GT>, dim3(w.nblk, L), dim3(256), 0, st, G, ldg, gd, s, order, assign, seg, compact,
                     S, N, hself, bigflag, kb);
  hipLaunchKernelGGL(kmeans_cn_self_kernel, dim3((k + 3) / 4, L), dim3(256), 0, st, hself, s, order, seg, compact, S, k, cn,
                     bigflag, kb);
  if (km_big_slots(N) > 0) {  // both leave at once when the step has no big cluster
    // (a big cluster's norm needs its sums at its own members, which lie anywhere: this pass stays whole when sharded)
    hipLaunchKernelGGL(kmeans_bigtot_kernel<GT>, dim3(nb64, L), dim3(1024), 0, st, G, ldg, gd, ssorted, order, seg, biglist,
                       N, bigtot, kb);
    hipLaunchKernelGGL(kmeans_cn_big_kernel, dim3(km_big_slots(N), L), dim3(256), 0, st, bigtot, N, ssorted, order, seg,
                       compact, biglist, S, cn, kb);
  }
  if (blk_hi > blk_lo) {
    KmBatch kr = kb;
    kr.blk0 = blk_lo;
    prof_begin(NIF_PROF_KMEANS, st);
    hipLaunchKernelGGL(kmeans_nearest_fused_kernel<GT>, dim3(blk_hi - blk_lo, L), dim3(1024), 0, st, G, ldg, gd, ssorted,
                       order, seg, compact, S, cn, nn, s, N, k, assign_out, nearest, partial, bigflag, biglist, bigtot, kr);
    prof_end(NIF_PROF_KMEANS, st);
  }
  if (!ranged)
    hipLaunchKernelGGL(kmeans_cost_kernel, dim3(1, L), dim3(64), 0, st, partial, nb64, cost, kb);
  else if (blk_hi > blk_lo)
    hipLaunchKernelGGL(kmeans_partial_export_kernel, dim3((blk_hi - blk_lo + 255) / 256, L), dim3(256), 0, st, partial,
                       nb64, blk_lo, blk_hi, partial_out, kb);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

static int km_repair_kf(int32_t* assign, double* nearest, const float* s, const int32_t* prev, int N, int k,
                        const int32_t* tries_idx, int L, int32_t* changed, void* ws, size_t ws_bytes, void* stream) {
  if (!assign || !nearest || !s || !prev || !changed || N <= 0 || k <= 0) return NIF_ERR_INVALID_ARG;
  const KmWs w = km_ws(N, k);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  KmBatch kb;
  const int brc = km_batch(tries_idx, L, w.total, ws_bytes, &kb);
  if (brc != NIF_OK) return brc;
  clear_err();
  int* counts = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(ws) + w.off_counts);
  int threads = (N + 63) / 64 * 64;
  if (threads > 1024) threads = 1024;
  hipLaunchKernelGGL(kmeans_repair_kf_kernel, dim3(1, L), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), assign,
                     nearest, s, prev, N, k, counts, changed, kb);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

}  // namespace nif

extern "C" int nif_kmeans_farthest_init_redo_count(uint32_t* count_host, void* stream) {
  if (!count_host) return NIF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(nif::fi_mu);
  int dev = -1;
  const int rc = nif::fi_device(&dev);
  if (rc != NIF_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(count_host, nif::fi_devs[dev].redo, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
    return NIF_ERR_LAUNCH;
  return hipStreamSynchronize(st) == hipSuccess ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_kmeans_farthest_init(const double* G, int64_t ldg, const double* nn, int N, int k, int first,
                                        int32_t* picks, int32_t* assign, double* mind, void* stream) {
  nif::Range nif_range_("nif_kmeans_farthest_init");
  return nif::km_farthest_init<double>(G, ldg, nullptr, nn, N, k, &first, 1, picks, assign, mind, stream);
}
extern "C" int nif_kmeans_farthest_init_batch(const double* G, int64_t ldg, const double* nn, int N, int k,
                                              const int32_t* firsts, int tries, int32_t* picks, int32_t* assign,
                                              double* mind, void* stream) {
  nif::Range nif_range_("nif_kmeans_farthest_init_batch");
  return nif::km_farthest_init<double>(G, ldg, nullptr, nn, N, k, firsts, tries, picks, assign, mind, stream);
}
extern "C" int nif_kmeans_farthest_init_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn, int N,
                                             int k, int first, int32_t* picks, int32_t* assign, double* mind,
                                             void* stream) {
  nif::Range nif_range_("nif_kmeans_farthest_init_gf32");
  return nif::km_farthest_init<float>(G, ldg, gdiag, nn, N, k, &first, 1, picks, assign, mind, stream);
}
extern "C" int nif_kmeans_farthest_init_batch_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn,
                                                   int N, int k, const int32_t* firsts, int tries, int32_t* picks,
                                                   int32_t* assign, double* mind, void* stream) {
  nif::Range nif_range_("nif_kmeans_farthest_init_batch_gf32");
  return nif::km_farthest_init<float>(G, ldg, gdiag, nn, N, k, firsts, tries, picks, assign, mind, stream);
}

extern "C" int nif_kmeans_assign_to_neurons(const double* G, int64_t ldg, const double* nn, int N, int k, int c0,
                                            int32_t* assign, void* stream) {
  nif::Range nif_range_("nif_kmeans_assign_to_neurons");
  return nif::km_assign_to_neurons<double>(G, ldg, nullptr, nn, N, k, c0, assign, stream);
}
extern "C" int nif_kmeans_assign_to_neurons_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn,
                                                 int N, int k, int c0, int32_t* assign, void* stream) {
  nif::Range nif_range_("nif_kmeans_assign_to_neurons_gf32");
  return nif::km_assign_to_neurons<float>(G, ldg, gdiag, nn, N, k, c0, assign, stream);
}

extern "C" int nif_kmeans_lloyd_step(const double* G, int64_t ldg, const double* nn, const float* s,
                                     const int32_t* assign, int N, int k, int32_t* assign_out, double* nearest,
                                     double* cost, int32_t* kprime, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_kmeans_lloyd_step");
  return nif::km_lloyd_step<double>(G, ldg, nullptr, nn, s, assign, N, k, nullptr, 1, assign_out, nearest, cost, kprime, ws,
                                    ws_bytes, stream);
}
extern "C" int nif_kmeans_lloyd_step_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn,
                                          const float* s,
                                          const int32_t* assign, int N, int k, int32_t* assign_out, double* nearest,
                                          double* cost, int32_t* kprime, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_kmeans_lloyd_step_gf32");
  return nif::km_lloyd_step<float>(G, ldg, gdiag, nn, s, assign, N, k, nullptr, 1, assign_out, nearest, cost, kprime, ws,
                                   ws_bytes, stream);
}

extern "C" int nif_kmeans_repair_kf(int32_t* assign, double* nearest, const float* s, const int32_t* prev, int N,
                                    int k, int32_t* changed, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_kmeans_repair_kf");
  return nif::km_repair_kf(assign, nearest, s, prev, N, k, nullptr, 1, changed, ws, ws_bytes, stream);
}

// KF's tries advance together (hf_kf.py:420-475 runs them one after the other; they share G, nn and s): one launch
// per kernel for all the tries named in tries_idx.
extern "C" int nif_kmeans_lloyd_step_batch(const double* G, int64_t ldg, const double* nn, const float* s,
                                           const int32_t* assign, int N, int k, const int32_t* tries_idx, int n_tries,
                                           int32_t* assign_out, double* nearest, double* cost, int32_t* kprime,
                                           void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_kmeans_lloyd_step_batch");
  if (!tries_idx) return NIF_ERR_INVALID_ARG;
  return nif::km_lloyd_step<double>(G, ldg, nullptr, nn, s, assign, N, k, tries_idx, n_tries, assign_out, nearest, cost,
                                    kprime, ws, ws_bytes, stream);
}
extern "C" int nif_kmeans_lloyd_step_batch_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn,
                                                const float* s, const int32_t* assign, int N, int k,
                                                const int32_t* tries_idx, int n_tries, int32_t* assign_out,
                                                double* nearest, double* cost, int32_t* kprime, void* ws,
                                                size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_kmeans_lloyd_step_batch_gf32");
  if (!tries_idx) return NIF_ERR_INVALID_ARG;
  return nif::km_lloyd_step<float>(G, ldg, gdiag, nn, s, assign, N, k, tries_idx, n_tries, assign_out, nearest, cost,
                                   kprime, ws, ws_bytes, stream);
}
extern "C" int nif_kmeans_repair_kf_batch(int32_t* assign, double* nearest, const float* s, const int32_t* prev, int N,
                                          int k, const int32_t* tries_idx, int n_tries, int32_t* changed, void* ws,
                                          size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_kmeans_repair_kf_batch");
  if (!tries_idx) return NIF_ERR_INVALID_ARG;
  return nif::km_repair_kf(assign, nearest, s, prev, N, k, tries_idx, n_tries, changed, ws, ws_bytes, stream);
}

extern "C" int nif_kmeans_lloyd_step_range(const double* G, int64_t ldg, const double* nn, const float* s,
                                           const int32_t* assign, int N, int k, const int32_t* tries_idx, int n_tries,
                                           int blk_lo, int blk_hi, int32_t* assign_out, double* nearest,
                                           double* partial, int32_t* kprime, void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_kmeans_lloyd_step_range");
  if (!partial) return NIF_ERR_INVALID_ARG;
  return nif::km_lloyd_step<double>(G, ldg, nullptr, nn, s, assign, N, k, tries_idx, tries_idx ? n_tries : 1, assign_out,
                                    nearest, nullptr, kprime, ws, ws_bytes, stream, blk_lo, blk_hi, partial);
}
extern "C" int nif_kmeans_lloyd_step_range_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn,
                                                const float* s, const int32_t* assign, int N, int k,
                                                const int32_t* tries_idx, int n_tries, int blk_lo, int blk_hi,
                                                int32_t* assign_out, double* nearest, double* partial, int32_t* kprime,
                                                void* ws, size_t ws_bytes, void* stream) {
  nif::Range nif_range_("nif_kmeans_lloyd_step_range_gf32");
  if (!partial) return NIF_ERR_INVALID_ARG;
  return nif::km_lloyd_step<float>(G, ldg, gdiag, nn, s, assign, N, k, tries_idx, tries_idx ? n_tries : 1, assign_out,
                                   nearest, nullptr, kprime, ws, ws_bytes, stream, blk_lo, blk_hi, partial);
}
extern "C" int nif_kmeans_cost_from_partials(const double* partial, int nblocks, const int32_t* tries_idx, int n_tries,
                                             double* cost, void* stream) {
  nif::Range nif_range_("nif_kmeans_cost_from_partials");
  using namespace nif;
  if (!partial || !cost || nblocks <= 0) return NIF_ERR_INVALID_ARG;
  const int L = tries_idx ? n_tries : 1;
  KmBatch kb;
  const int brc = km_batch(tries_idx, L, 0, 0, &kb);
  if (brc != NIF_OK) return brc;
  clear_err();
  hipLaunchKernelGGL(kmeans_cost_rows_kernel, dim3(1, L), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), partial,
                     nblocks, cost, kb);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_kmeans_centroid_weights(const double* W, int64_t ldw, const float* s, const int32_t* assign, int N,
                                           int k, int d, float* out, int32_t* kprime, void* ws, size_t ws_bytes,
                                           void* stream) {
  nif::Range nif_range_("nif_kmeans_centroid_weights");
  using namespace nif;
  if (!W || !s || !assign || !out || !kprime || N <= 0 || k <= 0 || d <= 0 || ldw < N) return NIF_ERR_INVALID_ARG;
  const KmWs w = km_ws(N, k);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  km_group(assign, N, k, w, b, kprime, st);
  (void)hipMemsetAsync(out, 0, (size_t)k * d * sizeof(float), st);
  hipLaunchKernelGGL(kmeans_centroid_weights_kernel, dim3((d + 255) / 256, k), dim3(256), 0, st, W, ldw, s,
                     reinterpret_cast<int*>(b + w.off_order), reinterpret_cast<int*>(b + w.off_seg),
                     reinterpret_cast<int*>(b + w.off_compact), d, out);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_kmeans_targets(const float* A, int64_t lda, int64_t m, const float* w, const int32_t* assign, int N,
                                  int k, float* T, int64_t ldt, int32_t* kprime, void* ws, size_t ws_bytes,
                                  void* stream) {
  nif::Range nif_range_("nif_kmeans_targets");
  using namespace nif;
  if (!A || !w || !assign || !T || !kprime || m <= 0 || N <= 0 || k <= 0 || lda < N || ldt < k)
    return NIF_ERR_INVALID_ARG;
  const KmWs ww = km_ws(N, k);
  if (!ws || ws_bytes < ww.total) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  km_group(assign, N, k, ww, b, kprime, st);
  int* order = reinterpret_cast<int*>(b + ww.off_order);
  int* seg = reinterpret_cast<int*>(b + ww.off_seg);
  int* compact = reinterpret_cast<int*>(b + ww.off_compact);
  double* S = reinterpret_cast<double*>(b + ww.off_S);
  const size_t row_bytes = (size_t)N * sizeof(float);
  if (row_bytes <= 144 * 1024) {
    // stage whole rows in LDS: up to 16 rows / 64 KB per block for narrow levels, one row for wide ones
    int R = (int)((64 * 1024) / row_bytes);
    if (R < 1) R = 1;
    if (R > 16) R = 16;
    const size_t lds = (size_t)R * row_bytes;
    const int threads_l = ((size_t)R * k >= 1024 || lds > 64 * 1024) ? 1024 : 256;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kmeans_targets_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
    hipLaunchKernelGGL(kmeans_S_kernel, dim3((k + 3) / 4), dim3(256), 0, st, w, order, seg, compact, k, S, km_single());
    int64_t blocks = (m + R - 1) / R;
    if (blocks > 4096) blocks = 4096;
    const bool vec = (N % 4 == 0) && (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    static const int no_pipe = env_int("NIF_KMEANS_TARGETS_PLAIN");  // tuning aid: the simple kernel
    const size_t slots = (size_t)R * (N / 4);
    const int pf = slots <= 3 * 1024 ? 3 : slots <= 4 * 1024 ? 4 : slots <= 6 * 1024 ? 6 : slots <= 9 * 1024 ? 9 : 0;
    const int cpt = k <= 1024 ? 1 : k <= 2048 ? 2 : k <= 3072 ? 3 : k <= 4096 ? 4 : 0;
    if (vec && !no_pipe && pf && cpt && !(pf == 9 && cpt == 4) && (int64_t)R * lda < (int64_t)1 << 31 && (size_t)R * N < 65536) {
      int* pos = reinterpret_cast<int*>(b + ww.off_pos);
      float* wT = reinterpret_cast<float*>(b + ww.off_wT);
      float* wsorted = reinterpret_cast<float*>(b + ww.off_wsorted);
      hipLaunchKernelGGL(kmeans_sorted_kernel, dim3((N + 255) / 256), dim3(256), 0, st, order, assign, seg, w, N, k, pos,
                         wsorted, wT);
      bool ok = false;
#define NIF_KM_SW(P, C)                                                                                            \
  if (pf == P && cpt == C)                                                                                         \
    ok = km_launch_sw<P, C>(blocks, st, A, lda, m, wsorted, wT, pos, seg, compact, S, kprime, N, k, R, T, ldt);
      NIF_KM_SW(3, 1) NIF_KM_SW(3, 2) NIF_KM_SW(3, 3) NIF_KM_SW(3, 4) NIF_KM_SW(4, 1) NIF_KM_SW(4, 2) NIF_KM_SW(4, 3)
      NIF_KM_SW(4, 4) NIF_KM_SW(6, 1) NIF_KM_SW(6, 2) NIF_KM_SW(6, 3) NIF_KM_SW(6, 4) NIF_KM_SW(9, 1) NIF_KM_SW(9, 2)
      NIF_KM_SW(9, 3)
#undef NIF_KM_SW
      if (ok) return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
    }
    (void)hipMemset2DAsync(T, (size_t)ldt * sizeof(float), 0, (size_t)k * sizeof(float), (size_t)m, st);
    hipLaunchKernelGGL(kmeans_targets_lds_kernel, dim3((unsigned)blocks), dim3(threads_l), lds, st, A, lda, m, w, order,
                       seg, compact, S, N, k, R, T, ldt);
    return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
  }
  (void)hipMemset2DAsync(T, (size_t)ldt * sizeof(float), 0, (size_t)k * sizeof(float), (size_t)m, st);
  const int threads = k >= 256 ? 256 : (k + 63) / 64 * 64;
  int64_t gy = m < 16384 ? m : 16384;
  hipLaunchKernelGGL(kmeans_targets_kernel, dim3((k + threads - 1) / threads, (unsigned)gy), dim3(threads), 0, st, A,
                     lda, m, w, reinterpret_cast<int*>(b + ww.off_order), reinterpret_cast<int*>(b + ww.off_seg),
                     reinterpret_cast<int*>(b + ww.off_compact), k, T, ldt);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
