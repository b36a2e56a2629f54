/*
 * nif.h -- C ABI of libnif.so: the MI355X (gfx950) kernels under the per-level
 * neuron matching-and-refit path of model fusion.
 *
 * The reference project (AndrewSpano/neuron-interpolation-model-fusion) is pure
 * Python and has no FFI layer; each entry point below names the reference call
 * site (path:line under the reference tree) whose arithmetic it replaces.  The
 * reference-side binding a maintainer would add is the ctypes stub shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its comment says "host";
 *   - the caller owns every buffer, including the workspace (query its size with
 *     the matching *_workspace_bytes function); the library never allocates;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     no entry point synchronises the host unless its comment says so;
 *   - return value: NIF_OK (0) or a negative NIF_ERR_* code; nothing throws;
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS.
 */
#ifndef NIF_H_
#define NIF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIF_OK 0
#define NIF_ERR_INVALID_ARG (-1)   /* null pointer, negative size, bad enum */
#define NIF_ERR_WORKSPACE (-2)     /* workspace too small */
#define NIF_ERR_UNSUPPORTED (-3)   /* size outside what the kernels cover */
#define NIF_ERR_LAUNCH (-4)        /* hipGetLastError() != hipSuccess after a launch */
#define NIF_ERR_NO_DEVICE (-5)

/* element types of nif_xty inputs */
#define NIF_F32 0
#define NIF_F64 1

/* nif_xty flags */
#define NIF_RELU_A 1      /* clamp A's elements at 0 while loading (git_rebasin.py:97) */
#define NIF_RELU_B 2      /* same for B (git_rebasin.py:98) */
#define NIF_SYMMETRIC 4   /* B is A (same pointer, n1 == n0): only the upper tile triangle is computed */
#define NIF_TRANS_A 8     /* A is stored as its transpose: n0 x m row-major (contraction over the contiguous axis) */
#define NIF_TRANS_B 16    /* same for B */
#define NIF_XTY_SPLIT_BF16 32 /* f32 in, f32 accumulate, no transposes: every f32 operand is split exactly into
                                 three bf16 values and the product runs as six bf16 matrix instructions (f32-level
                                 accuracy at 2.65x the f32 matrix peak; results differ from the default path in the
                                 last bits) */

int nif_version(void);
const char* nif_error_string(int code);
/* number of HIP devices visible, or a negative error code (host call, no allocation) */
int nif_device_count(void);

/* ---------------------------------------------------------------------------
 * Optional kernel timing for bench.py's roofline line: when enabled, the entry points
 * record HIP events on the launch stream around their dominant kernel.  Slots:
 * NIF_PROF_XTY = the MFMA cross-Gram kernel of nif_xty, NIF_PROF_LAP = the augmenting
 * kernel of nif_lap_solve_f32, NIF_PROF_KMEANS = the Lloyd reduction kernel.
 * nif_profile_collect synchronises on the recorded events (host blocks), returns the
 * summed milliseconds and launch count of a slot and clears it.  No reference counterpart
 * (the reference only prints time.time() deltas, hf_kf.py:211-214,245-247).
 * ------------------------------------------------------------------------- */
#define NIF_PROFILE_SLOTS 5
#define NIF_PROF_XTY 0
#define NIF_PROF_LAP 1
#define NIF_PROF_KMEANS 2
#define NIF_PROF_EIGH_TRD 3 /* all tridiagonalisation passes of one nif_eigh_kept_f64 call (one span per call) */
#define NIF_PROF_SINKHORN 4 /* all iterations of one nif_sinkhorn_f32 call (one span per call) */
void nif_profile_enable(int on); /* bit s set: record spans of slot s; negative: every slot; 0: off */
int nif_profile_collect(int slot, double* total_ms, int64_t* launches);
/* The same spans one by one, in record order: ms_out[0 .. min(cap, *launches)) (HOST floats); consumes them like
 * nif_profile_collect.  For per-launch minimum / median / maximum of a kernel (bench.py). */
int nif_profile_collect_spans(int slot, float* ms_out, int64_t cap, int64_t* launches);
/* Algorithmic work the entry points booked for the recorded spans of a slot since the last call (then cleared).
 * NIF_PROF_EIGH_TRD: bytes of the fused update + matrix-vector passes, 16 (n - j - 1)^2 per column j (each element of
 * the trailing matrix read once and written once).  The other slots book nothing: their callers know the shapes. */
int nif_profile_work(int slot, double* units);

/* ---------------------------------------------------------------------------
 * nif_xty: G = A^T B summed over the m sample rows, plus the squared column
 * norms na[i] = sum_r A[r][i]^2, nb[j] = sum_r B[r][j]^2.
 *
 * This is the one GEMM under every cost matrix of the path:
 *   - Git Re-Basin cost            git_rebasin.py:60-78   (acts0^T acts1, ReLU on load)
 *   - OTFusion ground metric       otfusion.py:135-150    (torch.cdist of activation columns)
 *   - HF cost matrix               hf_kf.py:93-124        (via X^T X, X^T acts: see nif_cost_hf)
 *   - KF / ALF k-means distances   hf_kf.py:429,440,455 ; alf.py:227,243,276 (Gram form)
 *   - ALF target/input matching    alf.py:126-143
 *   - LS refit normal equations    hf_kf.py:213,233-234,377,397 (X^T X and X^T acts, f64 accumulate)
 *
 * A is m x n0 (lda), B is m x n1 (ldb); with NIF_TRANS_* the operand is stored
 * n x m.  in_dtype is the element type of A and B; acc selects the matrix-core
 * path: NIF_F32 = v_mfma_f32_32x32x2_f32 (exact f32 fma chains per row block,
 * row blocks summed in f64), NIF_F64 = v_mfma_f64_16x16x4_f64 (f32 inputs are
 * widened exactly).  in_dtype NIF_F64 requires acc NIF_F64.
 * G (n0 x n1, ldg), na (n0) and nb (n1) are f64; na / nb may be NULL.
 * The row range is split over workgroups; partial tiles go through `ws`
 * and are summed in a fixed order, so results are bitwise reproducible.
 * ------------------------------------------------------------------------- */
size_t nif_xty_workspace_bytes(int64_t m, int n0, int n1, int acc, int flags);
int nif_xty(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t m, int n0, int n1,
            int in_dtype, int acc, int flags, double* G, int64_t ldg, double* na, double* nb,
            void* ws, size_t ws_bytes, void* stream);
/* The symmetric Gram A^T A in the storage the k-means entry points stream (KF hf_kf.py:429-456, ALF alf.py:227-276):
 * G (n x n, ldg) rounded once to f32, its diagonal in f64 (gdiag, n entries, required) -- the values nif_xty's f64 G
 * rounds to, without the f64 matrix ever being written.  Workspace: nif_xty_workspace_bytes(m, n, n, acc,
 * flags | NIF_SYMMETRIC).  flags: NIF_TRANS_A / NIF_RELU_A / NIF_XTY_SPLIT_BF16 as for nif_xty. */
int nif_xty_sym_gf32(const void* A, int64_t lda, int64_t m, int n, int in_dtype, int acc, int flags, float* G,
                     int64_t ldg, double* gdiag, void* ws, size_t ws_bytes, void* stream);

/* Sample-sharded runs (one process per GPU, rows split over the ranks; no counterpart in the reference, which is
 * single-process) sum the ranks' partial symmetric Grams -- k-means Gram alf.py:227-276 / hf_kf.py:429-456, normal
 * equations hf_kf.py:213,377 -- with ONE all-reduce of the block upper triangle: block rows of `block` rows, block row
 * b (first row i0 = b * block) keeps columns i0 .. n-1, row-major, block rows back to back (nif_packed_upper_len
 * doubles in all; about n^2 / 2 for n >> block).  nif_xty_sym_packed writes a rank's partial A^T A in that layout
 * straight from the split-K reduction (na, optional: the squared column norms, e.g. the tail of the same buffer);
 * nif_unpack_upper expands the reduced buffer to the full symmetric matrix, f64 or the f32 k-means storage
 * (out_dtype NIF_F32 / NIF_F64; gdiag optional: the unrounded diagonal).  Same values, entry for entry, as reducing
 * the full matrices.  Workspace as nif_xty_workspace_bytes(m, n, n, acc, flags | NIF_SYMMETRIC).  n <= 65535. */
size_t nif_packed_upper_len(int n, int block);
int nif_xty_sym_packed(const void* A, int64_t lda, int64_t m, int n, int in_dtype, int acc, int flags, int block,
                       double* packed, double* na, void* ws, size_t ws_bytes, void* stream);
int nif_unpack_upper(const double* packed, int n, int block, void* G, int64_t ldg, int out_dtype, double* gdiag,
                     void* stream);

/* ---------------------------------------------------------------------------
 * Cost-matrix epilogues: O(1) work per (i, j) on top of the Gram quantities.
 * C is n0 x n1 f32 (ldc), the layout scipy.optimize.linear_sum_assignment /
 * ot.emd receive in the reference.
 * ------------------------------------------------------------------------- */

/* C[i][j] = scale_ij * sqrt(max(na[i] + nb[j] - 2 G[i][j], 0)).
 *  s0 == s1 == NULL: scale = 1          -> OTFusion ground metric, otfusion.py:150,182
 *  otherwise scale = 2 s0[i] s1[j] / (s0[i] + s1[j]) -> Git Re-Basin cost, git_rebasin.py:70-77
 *  (closed form of s0*||avg - a0|| + s1*||avg - a1||, avg the importance-weighted mean).
 *  colscale (n1, may be NULL) multiplies column j afterwards -> alf.py:135-136 (dists * W). */
int nif_cost_dist(const double* G, int64_t ldg, const double* na, const double* nb,
                  const float* s0, const float* s1, const float* colscale, int n0, int n1,
                  float* C, int64_t ldc, void* stream);

/* HF cost, hf_kf.py:109-119 in closed form: with p = projected activations X w,
 *   D = pp0[i] + pp1[j] - 2 G01[i][j]              (||p0_i - p1_j||^2)
 *   a = s0[i] / (s0[i] + s1[j]),  b = 1 - a
 *   C = s0[i] sqrt(b^2 D + r0[i]) + s1[j] sqrt(a^2 D + r1[j])
 * G01 = P0^T P1 (n0 x n1), pp = squared norms of the projections, r = squared LS residuals. */
int nif_cost_hf(const double* G01, int64_t ldg, const double* pp0, const double* pp1,
                const double* r0, const double* r1, const float* s0, const float* s1,
                int n0, int n1, float* C, int64_t ldc, void* stream);

/* ---------------------------------------------------------------------------
 * nif_lap_solve_f32: min-cost perfect matching of an n x n f32 cost matrix;
 * replaces scipy.optimize.linear_sum_assignment at hf_kf.py:251,
 * git_rebasin.py:112, alf.py:138,383 (and ot.emd at otfusion.py:32,309 when both
 * marginals are uniform and the problem is square: the exact plan is P/n).
 * col4row[i] = column matched to row i (== scipy's col_ind, rows sorted).
 * Exact (f64 duals, shortest augmenting paths after a parallel dual
 * initialisation); equals scipy wherever the optimum is unique.
 * row4col (n), u (n), v (n) are optional outputs (NULL to skip): the dual
 * certificate u[i] + v[j] <= C[i][j] with equality on matched pairs.
 * n <= NIF_LAP_MAX_N.  Costs must be finite.
 *
 * flags = 0: fastest path (parallel dual initialisation, then augmentation of the rows
 *   left free).  Any optimal matching may come back when the optimum is not unique.
 * flags = NIF_LAP_SCIPY_ORDER: replays scipy's rectangular_lsap order exactly -- zero
 *   initial duals, rows augmented in index order, and its tie rule between columns of
 *   equal distance (first in its `remaining` vector unless a free column ties) -- so
 *   that cost matrices with exact structural ties (dead ReLU neurons give identical
 *   rows / columns) yield scipy's own matching.  Slower: n augmentations instead of a few.
 * The workspace also receives a work counter (number of Dijkstra steps) for tools/lap_bench.py.
 */
#define NIF_LAP_MAX_N 8192
#define NIF_LAP_SCIPY_ORDER 1
#define NIF_LAP_INIT_ONLY 2 /* fast mode, dual initialisation + greedy matching only: col4row[i] = -1 for rows left free
                               (u, v, row4col required) */
#define NIF_LAP_RESUME 4    /* fast mode, continue from the state an INIT_ONLY call left in u, v, col4row, row4col */
#define NIF_LAP_BUDGET_SHIFT 8 /* fast mode: flags >> 8 = budget of Dijkstra steps (0 = none); once it is spent no
                                  further row is started and the rows not reached keep col4row[i] = -1 */
size_t nif_lap_workspace_bytes(int n);
int nif_lap_solve_f32(const float* C, int64_t ldc, int n, int flags, int32_t* col4row, int32_t* row4col,
                      double* u, double* v, void* ws, size_t ws_bytes, void* stream);

/* `batch` independent problems of ONE size in one call: the layers of Git Re-Basin / OTFusion in activation mode
 * (git_rebasin.py:112 and otfusion.py:309 run once per layer, on costs that come from activations captured once, so
 * nothing orders them), or the same layer of several models against one target (otfusion.py:300-343 loops over the
 * models).  Problem p reads C + p * stride_c (stride_c >= (n - 1) * ldc + n) and writes entries [p * n, (p + 1) * n)
 * of col4row / row4col / u / v (all required here).  Every kernel of nif_lap_solve_f32 is launched once with the
 * problems along gridDim.z, so the single-workgroup phases of the problems run on `batch` compute units side by
 * side instead of one after the other; each problem's results are bit-identical to its own nif_lap_solve_f32 call.
 * `flags` as for nif_lap_solve_f32, the same for every problem.  Workspace: nif_lap_batch_workspace_bytes(n, batch)
 * (= batch * nif_lap_workspace_bytes(n)); batch <= 65535. */
size_t nif_lap_batch_workspace_bytes(int n, int batch);
int nif_lap_solve_batch_f32(const float* C, int64_t ldc, int64_t stride_c, int n, int batch, int flags,
                            int32_t* col4row, int32_t* row4col, double* u, double* v, void* ws, size_t ws_bytes,
                            void* stream);

/* The same solve for callers that cannot allocate device memory or hand over a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the reference.  The
 * scratch the kernels need (column-minimum partials, claim words, stand-ins for optional outputs) lives in a
 * static device array inside libnif.so -- NIF_LAP_STATIC_BYTES per GPU the library runs on, held until the
 * process ends; its address on the current device is looked up once per device.  The call itself allocates and
 * frees nothing and records no timing events.  n > NIF_LAP_STATIC_MAX_N returns NIF_ERR_UNSUPPORTED (with a line
 * on stderr).  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race; calls ordered on one stream are safe.  Arguments and results are those of nif_lap_solve_f32. */
#define NIF_LAP_STATIC_MAX_N 2048
#define NIF_LAP_STATIC_BYTES (640 * 1024)
int nif_lap_solve_f32_static(const float* C, int64_t ldc, int n, int flags, int32_t* col4row, int32_t* row4col,
                             double* u, double* v, void* stream);

/* Uniqueness certificate (no counterpart in the reference: it is what lets a call site that must reproduce
 * scipy.optimize.linear_sum_assignment -- hf_kf.py:251, git_rebasin.py:112, alf.py:138,383 -- take the fast solver's
 * answer): count[0] (device) = number of edges outside the matching whose reduced cost C[i][j] - u[i] - v[j] is
 * <= tol_rel * (|C[i][j]| + |u[i]| + |v[j]|), plus matched edges that are not tight to 1e-5 of that margin and
 * unmatched rows.  0 means the
 * optimal assignment is unique, so scipy returns the same one whatever its tie rule. */
int nif_lap_count_tight_f32(const float* C, int64_t ldc, int n, const int32_t* col4row, const double* u, const double* v,
                            double tol_rel, int32_t* count, void* stream);
/* The same question answered without demanding strict slackness: *result (device int) = 0 when the graph of near-tight
 * edges outside the matching (reduced cost <= tol_rel * (|c| + |u| + |v|)) contains no alternating cycle, the matched
 * edges are tight and the duals feasible -- then no other assignment reaches the optimal cost, however many ties the
 * solver's own dual updates left behind (they do, by construction: the bidder's best and second-best columns tie);
 * non-zero otherwise.  ws from nif_lap_certify_workspace_bytes(n).  No counterpart in the reference: it is what lets
 * the call sites of scipy's linear_sum_assignment (hf_kf.py:251, git_rebasin.py:112, alf.py:138) take the fast solver's
 * answer wherever the optimum is unique. */
size_t nif_lap_certify_workspace_bytes(int n);
int nif_lap_certify_unique_f32(const float* C, int64_t ldc, int n, const int32_t* col4row, const int32_t* row4col,
                               const double* u, const double* v, double tol_rel, int32_t* result, void* ws,
                               size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * nif_refit_gather: weight write-back under a matching,
 *   out[i][c*sp + e] = a_i * W0[i][c*sp + e] + b_i * W1[rowsrc[i]][colsrc[c]*sp + e]
 * with a_i = alpha_v ? alpha_v[i] : alpha (same for b).  W0 may be NULL (treated as 0).
 * rowsrc / colsrc NULL = identity.  All matrices are n_out x (n_in * sp) f32, dense.
 *   - Git Re-Basin: aligned_w1 = P w1 P_prev^T, importance-weighted average
 *     (git_rebasin.py:163-213): rowsrc = col_ind, colsrc = previous col_ind, sp = kh*kw.
 *   - OTFusion with permutation plans (uniform marginals): W_k T_prev diag(1/nu_prev) then
 *     T^T diag(1/mu), accumulated into the running sum (otfusion.py:261-343):
 *     rowsrc = row4col, colsrc = previous row4col.
 * ------------------------------------------------------------------------- */
int nif_refit_gather(const float* W0, const float* W1, const int32_t* rowsrc, const int32_t* colsrc,
                     const float* alpha_v, const float* beta_v, float alpha, float beta, int n_out,
                     int n_in, int sp, float* out, void* stream);

/* ---------------------------------------------------------------------------
 * Neuron k-means in Gram form.  G (N x N f64, ldg) holds the inner products of the neuron
 * vectors that form centroids, nn[j] the squared norm of the neuron vector that distances
 * are measured to (KF: G = (XW)^T (XW) and nn = |a_j|^2, hf_kf.py:425-441,454-456;
 * ALF: G = A^T A and nn = diag(G), alf.py:227,242-244).  s = importance weights (f32, N).
 * Labels are int32.  ws from nif_kmeans_workspace_bytes(N, k).
 * The three entry points that stream G come in two storage types: f64 (the accumulator type of nif_xty) and,
 * suffix _gf32, the same matrix rounded once to f32 -- half the bytes per Lloyd step; every product and sum is
 * still formed in f64, and the diagonal is taken from gdiag (N f64, the unrounded G[j][j]) so that the distance of
 * a neuron to a centroid made of that neuron alone still cancels exactly.  The reference itself measures these distances in f32 (torch.cdist, hf_kf.py:431,456 /
 * alf.py:227,244); the labels of every reference golden are the same with either storage.
 * ------------------------------------------------------------------------- */
size_t nif_kmeans_workspace_bytes(int N, int k);

/* KF farthest-point initialisation, hf_kf.py:424-441: picks[0] = first, then k-1 times the
 * neuron farthest from its nearest chosen centroid (centroid = projected neuron, distance to
 * the raw neuron); assign = nearest of the k centroids (lowest on ties).  mind: N f64 scratch.
 * One workgroup; replaces k full k x N x m cdist calls by k reads of one Gram row. */
int nif_kmeans_farthest_init(const double* G, int64_t ldg, const double* nn, int N, int k, int first,
                             int32_t* picks, int32_t* assign, double* mind, void* stream);
int nif_kmeans_farthest_init_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn, int N, int k,
                                  int first, int32_t* picks, int32_t* assign, double* mind, void* stream);
/* The `tries` (<= 16) initialisations of one layer (hf_kf.py:420-441: same matrices, a different first neuron per
 * try) in ONE launch: firsts is a HOST array; picks [tries][k], assign [tries][N], mind [tries][N]. */
int nif_kmeans_farthest_init_batch(const double* G, int64_t ldg, const double* nn, int N, int k,
                                   const int32_t* firsts, int tries, int32_t* picks, int32_t* assign, double* mind,
                                   void* stream);
int nif_kmeans_farthest_init_batch_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn, int N,
                                        int k, const int32_t* firsts, int tries, int32_t* picks, int32_t* assign,
                                        double* mind, void* stream);

/* Above N = 4096 the initialisation runs on up to 16 co-operating workgroups per try that exchange one (maximum,
 * index) pair per pick.  If their exchange times out (the workgroups of a try were not resident together: another
 * stream or process held the compute units) the try is done again by one workgroup inside the same call, so the
 * outputs are always complete; this diagnostic returns how many tries took that slow path on the current device
 * since the process started (*count_host: HOST uint32; blocks on `stream`).  No counterpart in the reference. */
int nif_kmeans_farthest_init_redo_count(uint32_t* count_host, void* stream);

/* ALF initial assignment, alf.py:221-229: centroids are neurons c0 .. c0+k-1. */
int nif_kmeans_assign_to_neurons(const double* G, int64_t ldg, const double* nn, int N, int k, int c0,
                                 int32_t* assign, void* stream);
int nif_kmeans_assign_to_neurons_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn, int N,
                                      int k, int c0, int32_t* assign, void* stream);

/* One Lloyd step, hf_kf.py:454-459 / alf.py:242-245: importance-weighted centroids of the
 * NON-EMPTY clusters of `assign` (compacted in label order, hf_kf.py:294-296, alf.py:155-157),
 * nearest centroid of every neuron -> assign_out (compact labels), nearest (distance, f64),
 * *cost = sum_j s_j * nearest_j, *kprime = number of non-empty clusters (device scalars). */
int nif_kmeans_lloyd_step(const double* G, int64_t ldg, const double* nn, const float* s,
                          const int32_t* assign, int N, int k, int32_t* assign_out, double* nearest,
                          double* cost, int32_t* kprime, void* ws, size_t ws_bytes, void* stream);
int nif_kmeans_lloyd_step_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn, const float* s,
                               const int32_t* assign, int N, int k, int32_t* assign_out, double* nearest,
                               double* cost, int32_t* kprime, void* ws, size_t ws_bytes, void* stream);

/* KF empty-cluster repair, hf_kf.py:466-471, in place; *changed = (assign != prev).any(). */
int nif_kmeans_repair_kf(int32_t* assign, double* nearest, const float* s, const int32_t* prev, int N,
                         int k, int32_t* changed, void* ws, size_t ws_bytes, void* stream);

/* KF's tries of one layer advance together (hf_kf.py:420-475 runs them one after the other; they share G, nn, s and
 * differ in their assignment): tries_idx (HOST array, n_tries <= 16 entries, each < 16) names the tries this call
 * advances; assign / assign_out / nearest are rows of [tries][N] tensors, cost / kprime / changed entries of [tries]
 * vectors, ws holds one nif_kmeans_workspace_bytes(N, k) block per try, back to back.  Same results as the
 * single-problem calls, one launch per kernel for the whole batch. */
int nif_kmeans_lloyd_step_batch(const double* G, int64_t ldg, const double* nn, const float* s, const int32_t* assign,
                                int N, int k, const int32_t* tries_idx, int n_tries, int32_t* assign_out,
                                double* nearest, double* cost, int32_t* kprime, void* ws, size_t ws_bytes, void* stream);
int nif_kmeans_lloyd_step_batch_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn, const float* s,
                                     const int32_t* assign, int N, int k, const int32_t* tries_idx, int n_tries,
                                     int32_t* assign_out, double* nearest, double* cost, int32_t* kprime, void* ws,
                                     size_t ws_bytes, void* stream);
int nif_kmeans_repair_kf_batch(int32_t* assign, double* nearest, const float* s, const int32_t* prev, int N, int k,
                               const int32_t* tries_idx, int n_tries, int32_t* changed, void* ws, size_t ws_bytes,
                               void* stream);

/* The Lloyd step of a run whose ranks all hold the reduced Gram (the sample-sharded levels of hf_kf.py:420-475, f1):
 * the pass over G that finds every neuron's nearest centroid is split by neuron.  This call does the whole step's
 * O(N) part (grouping, cluster weights and norms) and the G pass for the 64-neuron blocks [blk_lo, blk_hi) only: it
 * writes assign_out / nearest for the neurons 64 blk_lo .. min(N, 64 blk_hi) - 1 and partial[blk] (the blocks' shares
 * of the cost; partial is [tries][ceil(N / 64)] f64, or one row when tries_idx is NULL) for those blocks, and leaves
 * everything else of the three untouched.  The caller zeroes the three, sums them over the ranks (every entry has
 * one writer, so the sum is exact) and forms the cost with nif_kmeans_cost_from_partials, which adds the partials in
 * nif_kmeans_lloyd_step's order: a sharded step returns the bits of an unsharded one.  tries_idx / n_tries as in the
 * batch calls (NULL: the single problem). */
int nif_kmeans_lloyd_step_range(const double* G, int64_t ldg, const double* nn, const float* s, const int32_t* assign,
                                int N, int k, const int32_t* tries_idx, int n_tries, int blk_lo, int blk_hi,
                                int32_t* assign_out, double* nearest, double* partial, int32_t* kprime, void* ws,
                                size_t ws_bytes, void* stream);
int nif_kmeans_lloyd_step_range_gf32(const float* G, int64_t ldg, const double* gdiag, const double* nn, const float* s,
                                     const int32_t* assign, int N, int k, const int32_t* tries_idx, int n_tries,
                                     int blk_lo, int blk_hi, int32_t* assign_out, double* nearest, double* partial,
                                     int32_t* kprime, void* ws, size_t ws_bytes, void* stream);
int nif_kmeans_cost_from_partials(const double* partial, int nblocks, const int32_t* tries_idx, int n_tries,
                                  double* cost, void* stream);

/* Fused neurons from a clustering, hf_kf.py:283-312,494-504: out[cc][r] = sum_{i in c} s_i W[r][i] / S_c
 * for the non-empty clusters in label order (rows >= *kprime stay zero). W: d x N f64 (LS solutions
 * as columns), out: k x d f32. */
int nif_kmeans_centroid_weights(const double* W, int64_t ldw, const float* s, const int32_t* assign, int N,
                                int k, int d, float* out, int32_t* kprime, void* ws, size_t ws_bytes,
                                void* stream);

/* ALF level targets, alf.py:288-289 (with _calculate_centroids :145-169): T[r][cc] = weighted mean
 * over the member neurons of cluster c of A[r][.], for every sample row; A: m x N f32 (the
 * un-normalised level outputs of all models side by side), T: m x k f32, columns >= *kprime zero. */
int nif_kmeans_targets(const float* A, int64_t lda, int64_t m, const float* w, const int32_t* assign, int N,
                       int k, float* T, int64_t ldt, int32_t* kprime, void* ws, size_t ws_bytes, void* stream);

/* HF fused neurons, hf_kf.py:261-275: out[i][r] = a_i W[r][i] + b_i W[r][n0 + col[i]] with
 * a_i = s0_i / (s0_i + s1_col[i]), b_i = 1 - a_i.  W: d x (n0 + n1) f64, out: n0 x d f32. */
int nif_refit_pairs(const double* W, int64_t ldw, const int32_t* col, const float* s0, const float* s1,
                    int n0, int d, float* out, void* stream);

/* ---- exact optimal transport with arbitrary marginals (POT `ot.emd`, otfusion.py:18-32, call site :309) ----
 * min <T, C> s.t. T 1 = a, T^T 1 = b, T >= 0 for a dense n0 x n1 f32 cost (>= 0) and f64 marginals with equal
 * sums (POT rescales b to a's total; the caller does the same).  Successive shortest augmenting paths, f64
 * duals, one workgroup (the support kept as a rooted forest in LDS up to n1 = 2048, as an arc pool beyond; square
 * problems start from the assignment solution of the same costs).  The plan comes back sparse, by column: colptr[n1 + 1], rowidx / val[<= out_cap]
 * (a vertex of the transport polytope has at most n0 + n1 - 1 positive entries).  status (device, 4 ints):
 * [0] = number of entries, or a negative NIF_ERR_*; [1..3] = work counters (augmentations, Dijkstra steps,
 * rows expanded).  tol: flows / supplies below it count as zero (e.g. 1e-13 * sum a).
 * u / v (optional): optimal duals, u_i + v_j <= C_ij with equality on the support.  Uniform square problems
 * are assignments: use nif_lap_solve_f32 for those. */
#define NIF_EMD_MAX_N 8192
#define NIF_EMD_EDGES_GLOBAL 1 /* flags: keep the arc records in the workspace instead of LDS (room for 4 (n0 + n1)
                                  arcs; retry with it when a degenerate problem returns NIF_ERR_WORKSPACE in status) */
#define NIF_EMD_COLD_START 2   /* flags: square problems normally start from the assignment solution of the same costs
                                  (nif_lap_solve_f32: feasible duals, a tight perfect matching to ship along); this
                                  keeps the column-minimum / row-minimum start (tuning aid) */
#define NIF_EMD_ARC_POOL 4     /* flags: the first-generation kernel (support kept as an arc pool swept level by level)
                                  instead of the rooted-forest kernel that runs wherever it fits (n1 <= 2048): A/B aid */
#define NIF_EMD_ONE_LANE_PATHS 8 /* flags: the forest kernel walks every augmenting path with its general one-lane code
                                  (taken by itself only for paths over more than 32 trees or 64 arcs a side): test aid */
size_t nif_emd_workspace_bytes(int n0, int n1);
int nif_emd_solve_f32(const float* C, int64_t ldc, int n0, int n1, const double* a, const double* b, double tol,
                      int flags, int32_t* colptr, int32_t* rowidx, double* val, int32_t out_cap, int32_t* status,
                      double* u, double* v, void* ws, size_t ws_bytes, void* stream);

/* `batch` independent SQUARE problems of one size n in one call: OTFusion solves one plan per model and layer
 * (otfusion.py:300-343 loops over the models, :261 over the layers) on ground metrics that come from activations
 * captured once, so nothing orders them.  Problem p reads C + p * stride_c, a + p n, b + p n, tol[p] (DEVICE array: the
 * tolerance of nif_emd_solve_f32, one per problem) and writes colptr + p (n + 1), rowidx / val + p * out_cap, status +
 * 16 p, u / v + p n (both required).  The assignment start of every problem is one nif_lap_solve_batch_f32 call and
 * the forest kernel is launched once with the problems along gridDim.z; every problem's plan, duals and status are
 * bit-identical to its own nif_emd_solve_f32 call with flags 0.  Only what that call runs by default on such problems:
 * n <= 2048 within the forest kernel's LDS plan, flags 0 or NIF_EMD_ONE_LANE_PATHS; anything else returns
 * NIF_ERR_UNSUPPORTED (solve those problem by problem).  A problem whose status[0] comes back NIF_ERR_WORKSPACE is
 * retried by the caller through nif_emd_solve_f32 with NIF_EMD_EDGES_GLOBAL, as there. */
size_t nif_emd_batch_workspace_bytes(int n, int batch);
int nif_emd_solve_batch_f32(const float* C, int64_t ldc, int64_t stride_c, int n, int batch, const double* a,
                            const double* b, const double* tol, int flags, int32_t* colptr, int32_t* rowidx, double* val,
                            int32_t out_cap, int32_t* status, double* u, double* v, void* ws, size_t ws_bytes,
                            void* stream);

/* ---- entropic optimal transport (Sinkhorn-Knopp): an ADDITIONAL, approximate plan solver -------------------
 * No counterpart in the reference, which calls the exact ot.emd (otfusion.py:18-32, :309); north_star (2) / SURVEY D1
 * ask for it as a documented-approximate mode with its own HBM roofline.  Restates POT's `ot.sinkhorn(a, b, M, reg)`
 * (method 'sinkhorn': K = exp(-M / reg), v = b / K^T u, u = a / K v, every 10th iteration the marginal error
 * ||v * K^T u - b||_2 against tol, at most max_iter iterations; T = diag(u) K diag(v)).  C: n0 x n1 f32; a, b f64;
 * T: n0 x n1 f32 (ldt); u / v (optional, f64): the scalings; info (device, 2 ints): [0] = 0 iteration cap reached,
 * 1 converged, 2 numerical breakdown (a zero or non-finite sum: the last good u, v are returned, as POT does),
 * [1] = iterations completed; err (device f64): the last marginal error evaluated.  HBM-bound: 8 n0 n1 bytes per
 * iteration (K streamed twice).  Synchronises the stream once per 50 iterations to read the state word. */
size_t nif_sinkhorn_workspace_bytes(int n0, int n1);
int nif_sinkhorn_f32(const float* C, int64_t ldc, int n0, int n1, const double* a, const double* b, double reg,
                     int max_iter, double tol, float* T, int64_t ldt, double* u, double* v, int32_t* info, double* err,
                     void* ws, size_t ws_bytes, void* stream);

/* OTFusion alignment with general (sparse) plans, otfusion.py:261-335:
 * out[j][q*sp + s] = beta * W0[j][q*sp + s]
 *                  + alpha * sum_{e in col j of R} rv[e] * sum_{f in col q of P} cv[f] * W1[ri[e]][ci[f]*sp + s]
 * R = T^T diag(1/mu) of the current layer and P = T_prev diag(1/nu_prev), both handed over as CSC triplets of T
 * with the scaling folded into the values; a NULL triplet is the identity.  W1: n_k x ldw1, out / W0:
 * n_out x (n_in * sp) contiguous, W0 may be NULL (then beta is ignored). */
int nif_refit_sparse(const float* W0, const float* W1, int64_t ldw1, const int32_t* rptr, const int32_t* ri,
                     const float* rv, const int32_t* cptr, const int32_t* ci, const float* cv, float alpha, float beta,
                     int n_out, int n_in, int sp, float* out, void* stream);

/* ---- ALF level fit: fused cluster-weighted squared error (alf.py:414-422 `_criterion_vectorized`,
 * and nn.MSELoss of :434-435 when w is NULL) --------------------------------------------------------
 * loss = mean over all elements of w[c] * (pred - truth)^2; pred / truth: [outer][C][inner] contiguous
 * f32 (the level network's own output layout: inner = 1 for linear levels, H*W for conv, seq for
 * transformer levels), w: C f32 or NULL.  loss: one f32 on the device.  Deterministic reduction. */
size_t nif_wmse_workspace_bytes(void);
int nif_wmse_forward(const float* pred, const float* truth, const float* w, int64_t outer, int C, int64_t inner,
                     float* loss, void* ws, size_t ws_bytes, void* stream);
/* grad_pred = grad_out[0] * 2 w[c] (pred - truth) / numel (what autograd produces for the loss above). */
int nif_wmse_backward(const float* pred, const float* truth, const float* w, const float* grad_out, int64_t outer,
                      int C, int64_t inner, float* grad_pred, void* stream);

/* ---- design matrix of a conv level (hf_kf.py:196-208 `F.unfold(...)`, `:224-225` row order, `:204-208` ones
 * column for the bias) ------------------------------------------------------------------------------
 * x: [batch][C][H][W] contiguous f32.  out: (batch * OH * OW) x ldo, row (b, oy, ox), column (c, ky, kx) =
 * x[b][c][oy*sh - ph + ky][ox*sw - pw + kx] (0 outside the image), column C*kh*kw = 1.0 when add_ones, remaining
 * columns up to ldo = 0 (every one of the ldo columns is written).  OH = (H + 2 ph - kh) / sh + 1, OW likewise. */
int nif_im2row_f32(const float* x, int64_t batch, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                   int add_ones, float* out, int64_t ldo, void* stream);

/* ---- symmetric eigendecomposition with a relative cutoff: the factor step of the LS refit ----------------------
 * torch.linalg.pinv (hf_kf.py:213,377) drops the singular values <= rtol * sigma_max of the design matrix; on the
 * normal equations G = X^T X (or X X^T) that keeps the eigenpairs with lambda > rtol^2 * lambda_max.  G: n x n f64,
 * symmetric (ldg), not modified.  Returns K = number of eigenvalues > cutoff_rel * lambda_max in *kept_host (HOST
 * int), their values ascending in lam[0 .. K) and the orthonormal eigenvectors as the COLUMNS of V[:, 0 .. K) (V: n
 * rows, ldv >= n doubles per row).  Householder tridiagonalisation (one fused update + matrix-vector pass per column),
 * Sturm-count multisection, inverse iteration with in-cluster re-orthogonalisation, blocked back-transformation --
 * hand-written kernels, no vendor solver.  The call blocks the host ONCE on `stream` (to learn K, which sizes the
 * eigenvector launches).  n <= NIF_EIGH_MAX_N. */
#define NIF_EIGH_MAX_N 6144
size_t nif_eigh_kept_workspace_bytes(int n);
int nif_eigh_kept_f64(const double* G, int64_t ldg, int n, double cutoff_rel, double* lam, double* V, int64_t ldv,
                      int32_t* kept_host, void* ws, size_t ws_bytes, void* stream);

/* Diagnostic for the parity tests (no counterpart in the reference): launches of each tridiagonalisation step kernel
 * since the process started, counts30[(mode * 5 + cls) * 2 + two] with mode 0 = update + store, 1 = update left
 * pending, 2 = two pending updates applied; cls 0..4 = 1, 2, 3, 4, 6 column chunks per wave; two = second row group
 * prefetched.  HOST array of 30 uint64. */
int nif_eigh_trd_launch_counts(uint64_t* counts30);

/* ---- least-squares refit W = pinv(X) A: hf_kf.py:213 (`torch.linalg.pinv(X)`), :233-234 and :377,:397
 * (`pinv @ acts`) -------------------------------------------------------------------------------------------
 * X: m x d f32 (ldx) design matrix of the layer (im2col rows + ones column), A: m x N f32 (lda) pre-activations of
 * all models side by side.  Outputs, all f64: W (d x N, ldw) = pinv(X) A with torch.linalg.pinv's default cutoff for
 * f32 data (singular values <= max(m, d) * eps_f32 * sigma_max dropped), R (d x N, ldr) = X^T A and na (N) = squared
 * column norms of A (what nif_cost_hf and the k-means Gram are built from), *rank_host (HOST int) = number of singular
 * values kept.  m >= d: normal equations G = X^T X; m < d: dual form through X X^T (minimum-norm solution), unless
 * NIF_LSQ_PRIMAL.  NIF_LSQ_NORMALIZE first scales A's columns to unit norm (norm_acts, fusion_utils.py:38-40; R and na
 * come back for the scaled columns).  Gram builds and products on the f64 matrix cores, factorisation by
 * nif_eigh_kept_f64; the call blocks the host twice on `stream` (live-column count, kept count). */
#define NIF_LSQ_NORMALIZE 1
#define NIF_LSQ_PRIMAL 2
size_t nif_lsq_refit_workspace_bytes(int64_t m, int d, int N, int flags);
int nif_lsq_refit_f32(const float* X, int64_t ldx, int64_t m, int d, const float* A, int64_t lda, int N, int flags,
                      double* W, int64_t ldw, double* R, int64_t ldr, double* na, int32_t* rank_host, void* ws,
                      size_t ws_bytes, void* stream);
/* The solve alone, for callers that reduce the normal equations over several GPUs first (sample-sharded fusion:
 * every rank all-reduces its partial G = X^T X (d x d) and R = X^T A (d x N), then calls this): W = G^+ R keeping
 * the eigenpairs above cutoff_rel * lambda_max (cutoff_rel = (max(m, d) eps_f32)^2 for pinv's rule).  Rows / columns
 * of G whose diagonal is zero are taken out first and give zero rows of W. */
size_t nif_lsq_solve_workspace_bytes(int d, int N);
int nif_lsq_solve_f64(const double* G, int64_t ldg, const double* R, int64_t ldr, int d, int N, double cutoff_rel,
                      double* W, int64_t ldw, int32_t* rank_host, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NIF_H_ */
