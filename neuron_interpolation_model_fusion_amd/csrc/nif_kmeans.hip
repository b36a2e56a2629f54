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
// HBM-bound kernels: 8 N^2 bytes (G, f64) + 16 k N bytes (H written, read back) per step.
// Every reduction runs in a fixed order (members sorted by index), so results are reproducible.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

// ---- block-wide arg-extremum over (value, index): larger value wins, lower index on ties -----
__device__ __forceinline__ void block_argmax(double& v, int& j, double* sv, int* sj) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = shfl_xor_f64(v, o);
    const int oj = __shfl_xor(j, o, 64);
    if (ov > v || (ov == v && oj < j)) {
      v = ov;
      j = oj;
    }
  }
  __syncthreads();  // protects sv/sj against readers of the previous call
  if (lane == 0) {
    sv[wave] = v;
    sj[wave] = j;
  }
  __syncthreads();
  v = sv[0];
  j = sj[0];
  for (int w = 1; w < nw; ++w) {
    const double ov = sv[w];
    const int oj = sj[w];
    if (ov > v || (ov == v && oj < j)) {
      v = ov;
      j = oj;
    }
  }
}

// ---- farthest-point initialisation (hf_kf.py:424-441), one workgroup ----------------------------
// picks[0] = first; picks[c] = argmax_j min_{c' < c} dist(centroid c', neuron j), where centroid c'
// is the PROJECTED neuron picks[c'] and dist^2 = G[p][p] - 2 G[p][j] + nn[j].  Also leaves the
// first assignment (argmin over all k centroids, lowest centroid on ties, :440-441).
__global__ __launch_bounds__(1024) void kmeans_farthest_init_kernel(const double* __restrict__ G, int64_t ldg,
                                                                    const double* __restrict__ nn, int N, int k,
                                                                    int first, int* __restrict__ picks,
                                                                    int* __restrict__ assign,
                                                                    double* __restrict__ mind) {
  __shared__ double sv[16];
  __shared__ int sj[16];
  int idx = first;
  for (int c = 0; c < k; ++c) {
    if (threadIdx.x == 0) picks[c] = idx;
    const double* __restrict__ row = G + (int64_t)idx * ldg;
    const double gii = row[idx];
    double bv = -INFINITY;
    int bj = INT_MAX;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
      double d2 = gii - 2.0 * row[j] + nn[j];
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

// ---- first assignment from neurons used as centroids (alf.py:221-229) ------------------------------
// centroid c = neuron c0 + c; assign[j] = argmin_c G[cc][cc] - 2 G[cc][j] + nn[j] (lowest c on ties).
__global__ void kmeans_assign_to_neurons_kernel(const double* __restrict__ G, int64_t ldg,
                                                const double* __restrict__ nn, int N, int k, int c0,
                                                int* __restrict__ assign) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const double nj = nn[j];
  double best = INFINITY;
  int bc = 0;
  for (int c = 0; c < k; ++c) {
    const int cc = c0 + c;
    double d2 = G[(int64_t)cc * ldg + cc] - 2.0 * G[(int64_t)cc * ldg + j] + nj;
    if (!(d2 > 0.0)) d2 = 0.0;
    if (d2 < best) {
      best = d2;
      bc = c;
    }
  }
  assign[j] = bc;
}

// ---- Lloyd step -----------------------------------------------------------------------------------

// rank[i] = number of i' < i with the same label (stable order inside a cluster); counts[label]++
__global__ void kmeans_rank_kernel(const int* __restrict__ assign, int N, int* __restrict__ rank,
                                   int* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int lab = assign[i];
  int r = 0;
  for (int t = 0; t < i; ++t) r += (assign[t] == lab) ? 1 : 0;
  rank[i] = r;
  atomicAdd(&counts[lab], 1);
}

// seg = exclusive scan of counts; compact[c] = index of cluster c among the non-empty ones (or -1);
// *kprime = number of non-empty clusters.  k is small: one thread.
__global__ void kmeans_scan_kernel(const int* __restrict__ counts, int k, int* __restrict__ seg,
                                   int* __restrict__ compact, int* __restrict__ kprime) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int run = 0, live = 0;
  for (int c = 0; c < k; ++c) {
    seg[c] = run;
    run += counts[c];
    compact[c] = counts[c] > 0 ? live++ : -1;
  }
  seg[k] = run;
  *kprime = live;
}

__global__ void kmeans_scatter_kernel(const int* __restrict__ assign, const int* __restrict__ rank,
                                      const int* __restrict__ seg, int N, int* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  order[seg[assign[i]] + rank[i]] = i;
}

// H[cc][j] = sum_{i in c} s_i G[i][j] / S_c for the non-empty cluster c = blockIdx.y (cc = compact[c])
__global__ void kmeans_H_kernel(const double* __restrict__ G, int64_t ldg, const float* __restrict__ s,
                                const int* __restrict__ order, const int* __restrict__ seg,
                                const int* __restrict__ compact, int N, double* __restrict__ H,
                                double* __restrict__ S) {
  const int c = blockIdx.y;
  const int cc = compact[c];
  if (cc < 0) return;
  const int b = seg[c], e = seg[c + 1];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  double tot = 0.0;
  for (int t = b; t < e; ++t) tot += (double)s[order[t]];
  if (j == 0) S[cc] = tot;
  if (j >= N) return;
  double acc = 0.0;
  for (int t = b; t < e; ++t) {
    const int i = order[t];
    acc += (double)s[i] * G[(int64_t)i * ldg + j];
  }
  H[(int64_t)cc * N + j] = acc / tot;
}

// cn[cc] = sum_{i in c} s_i H[cc][i] / S_c; one wavefront per cluster
__global__ void kmeans_cn_kernel(const double* __restrict__ H, const float* __restrict__ s,
                                 const int* __restrict__ order, const int* __restrict__ seg,
                                 const int* __restrict__ compact, const double* __restrict__ S, int N, int k,
                                 double* __restrict__ cn) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= k) return;
  const int cc = compact[c];
  if (cc < 0) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (int t = seg[c] + lane; t < seg[c + 1]; t += 64) {
    const int i = order[t];
    acc += (double)s[i] * H[(int64_t)cc * N + i];
  }
  acc = wave_sum_f64(acc);
  if (lane == 0) cn[cc] = acc / S[cc];
}

// nearest centroid of every neuron (lowest compact label on ties, torch.min's first index),
// distance, and per-block partial sums of s_j * distance in a fixed order
__global__ void kmeans_nearest_kernel(const double* __restrict__ H, const double* __restrict__ cn,
                                      const double* __restrict__ nn, const float* __restrict__ s,
                                      const int* __restrict__ kprime, int N, int* __restrict__ assign_out,
                                      double* __restrict__ nearest, double* __restrict__ partial) {
  __shared__ double red[256];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int kp = *kprime;
  double contrib = 0.0;
  if (j < N) {
    const double nj = nn[j];
    double best = INFINITY;
    int bc = 0;
    for (int cc = 0; cc < kp; ++cc) {
      const double d2 = cn[cc] - 2.0 * H[(int64_t)cc * N + j] + nj;
      if (d2 < best) {
        best = d2;
        bc = cc;
      }
    }
    if (!(best > 0.0)) best = 0.0;
    const double d = sqrt(best);
    assign_out[j] = bc;
    nearest[j] = d;
    contrib = (double)s[j] * d;
  }
  red[threadIdx.x] = contrib;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void kmeans_cost_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ cost) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double t = 0.0;
  for (int b = 0; b < nblocks; ++b) t += partial[b];
  *cost = t;
}

// ---- KF empty-cluster repair (hf_kf.py:466-471), one workgroup --------------------------------------
// for c in 0..k-1: if no neuron carries label c, the neuron with the largest s * distance moves to c
// and its distance becomes 0.  Also reports whether the repaired assignment differs from `prev`.
__global__ __launch_bounds__(1024) void kmeans_repair_kf_kernel(int* __restrict__ assign,
                                                                double* __restrict__ nearest,
                                                                const float* __restrict__ s,
                                                                const int* __restrict__ prev, int N, int k,
                                                                int* __restrict__ counts,
                                                                int* __restrict__ changed) {
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

struct KmWs {
  size_t off_rank, off_counts, off_seg, off_compact, off_order, off_H, off_S, off_cn, off_partial, total;
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
  w.off_partial = o; o += ka((size_t)w.nblk * 8);
  w.off_H = o; o += ka((size_t)k * N * 8);
  w.total = o;
  return w;
}

// builds order / seg / compact / kprime for `assign` inside the workspace
static void km_group(const int* assign, int N, int k, const KmWs& w, unsigned char* b, int* kprime, hipStream_t st) {
  int* rank = reinterpret_cast<int*>(b + w.off_rank);
  int* counts = reinterpret_cast<int*>(b + w.off_counts);
  int* seg = reinterpret_cast<int*>(b + w.off_seg);
  int* compact = reinterpret_cast<int*>(b + w.off_compact);
  int* order = reinterpret_cast<int*>(b + w.off_order);
  (void)hipMemsetAsync(counts, 0, (size_t)k * 4, st);
  hipLaunchKernelGGL(kmeans_rank_kernel, dim3(w.nblk), dim3(256), 0, st, assign, N, rank, counts);
  hipLaunchKernelGGL(kmeans_scan_kernel, dim3(1), dim3(64), 0, st, counts, k, seg, compact, kprime);
  hipLaunchKernelGGL(kmeans_scatter_kernel, dim3(w.nblk), dim3(256), 0, st, assign, rank, seg, N, order);
}

}  // namespace nif

extern "C" size_t nif_kmeans_workspace_bytes(int N, int k) {
  if (N <= 0 || k <= 0) return 0;
  return nif::km_ws(N, k).total;
}

extern "C" int nif_kmeans_farthest_init(const double* G, int64_t ldg, const double* nn, int N, int k, int first,
                                        int32_t* picks, int32_t* assign, double* mind, void* stream) {
  if (!G || !nn || !picks || !assign || !mind || N <= 0 || k <= 0 || ldg < N || first < 0 || first >= N)
    return NIF_ERR_INVALID_ARG;
  nif::clear_err();
  int threads = (N + 63) / 64 * 64;
  if (threads > 1024) threads = 1024;
  hipLaunchKernelGGL(nif::kmeans_farthest_init_kernel, dim3(1), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream),
                     G, ldg, nn, N, k, first, picks, assign, mind);
  return nif::launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_kmeans_assign_to_neurons(const double* G, int64_t ldg, const double* nn, int N, int k, int c0,
                                            int32_t* assign, void* stream) {
  if (!G || !nn || !assign || N <= 0 || k <= 0 || ldg < N || c0 < 0 || c0 + k > N) return NIF_ERR_INVALID_ARG;
  nif::clear_err();
  hipLaunchKernelGGL(nif::kmeans_assign_to_neurons_kernel, dim3((N + 255) / 256), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), G, ldg, nn, N, k, c0, assign);
  return nif::launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_kmeans_lloyd_step(const double* G, int64_t ldg, const double* nn, const float* s,
                                     const int32_t* assign, int N, int k, int32_t* assign_out, double* nearest,
                                     double* cost, int32_t* kprime, void* ws, size_t ws_bytes, void* stream) {
  using namespace nif;
  if (!G || !nn || !s || !assign || !assign_out || !nearest || !cost || !kprime || N <= 0 || k <= 0 || ldg < N)
    return NIF_ERR_INVALID_ARG;
  const KmWs w = km_ws(N, k);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  int* seg = reinterpret_cast<int*>(b + w.off_seg);
  int* compact = reinterpret_cast<int*>(b + w.off_compact);
  int* order = reinterpret_cast<int*>(b + w.off_order);
  double* H = reinterpret_cast<double*>(b + w.off_H);
  double* S = reinterpret_cast<double*>(b + w.off_S);
  double* cn = reinterpret_cast<double*>(b + w.off_cn);
  double* partial = reinterpret_cast<double*>(b + w.off_partial);
  km_group(assign, N, k, w, b, kprime, st);
  prof_begin(NIF_PROF_KMEANS, st);
  hipLaunchKernelGGL(kmeans_H_kernel, dim3(w.nblk, k), dim3(256), 0, st, G, ldg, s, order, seg, compact, N, H, S);
  prof_end(NIF_PROF_KMEANS, st);
  hipLaunchKernelGGL(kmeans_cn_kernel, dim3((k + 3) / 4), dim3(256), 0, st, H, s, order, seg, compact, S, N, k, cn);
  hipLaunchKernelGGL(kmeans_nearest_kernel, dim3(w.nblk), dim3(256), 0, st, H, cn, nn, s, kprime, N, assign_out,
                     nearest, partial);
  hipLaunchKernelGGL(kmeans_cost_kernel, dim3(1), dim3(64), 0, st, partial, w.nblk, cost);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_kmeans_repair_kf(int32_t* assign, double* nearest, const float* s, const int32_t* prev, int N,
                                    int k, int32_t* changed, void* ws, size_t ws_bytes, void* stream) {
  using namespace nif;
  if (!assign || !nearest || !s || !prev || !changed || N <= 0 || k <= 0) return NIF_ERR_INVALID_ARG;
  const KmWs w = km_ws(N, k);
  if (!ws || ws_bytes < w.total) return NIF_ERR_WORKSPACE;
  clear_err();
  int* counts = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(ws) + w.off_counts);
  int threads = (N + 63) / 64 * 64;
  if (threads > 1024) threads = 1024;
  hipLaunchKernelGGL(kmeans_repair_kf_kernel, dim3(1), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), assign,
                     nearest, s, prev, N, k, counts, changed);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}

extern "C" int nif_kmeans_centroid_weights(const double* W, int64_t ldw, const float* s, const int32_t* assign, int N,
                                           int k, int d, float* out, int32_t* kprime, void* ws, size_t ws_bytes,
                                           void* stream) {
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
  using namespace nif;
  if (!A || !w || !assign || !T || !kprime || m <= 0 || N <= 0 || k <= 0 || lda < N || ldt < k)
    return NIF_ERR_INVALID_ARG;
  const KmWs ww = km_ws(N, k);
  if (!ws || ws_bytes < ww.total) return NIF_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  clear_err();
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  km_group(assign, N, k, ww, b, kprime, st);
  (void)hipMemset2DAsync(T, (size_t)ldt * sizeof(float), 0, (size_t)k * sizeof(float), (size_t)m, st);
  const int threads = k >= 256 ? 256 : (k + 63) / 64 * 64;
  int64_t gy = m < 16384 ? m : 16384;
  hipLaunchKernelGGL(kmeans_targets_kernel, dim3((k + threads - 1) / threads, (unsigned)gy), dim3(threads), 0, st, A,
                     lda, m, w, reinterpret_cast<int*>(b + ww.off_order), reinterpret_cast<int*>(b + ww.off_seg),
                     reinterpret_cast<int*>(b + ww.off_compact), k, T, ldt);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
