// host_check.cpp -- walks the host side of every libnif entry point that can run without a GPU (argument
// validation, workspace sizing, error strings, profile slots) so that an AddressSanitizer / UBSan build of the
// library's host code has something to chew on: `make asan-check`.  Kernels are verified by the differential
// tests under tests/ -m gpu; GPU sanitizers are not available on this pool.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nif.h"

static int failures = 0;
#define EXPECT(cond)                                                \
  do {                                                              \
    if (!(cond)) {                                                  \
      fprintf(stderr, "host_check: %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                   \
    }                                                               \
  } while (0)

int main() {
  EXPECT(nif_version() >= 100);
  for (int code = 1; code >= -8; --code) EXPECT(strlen(nif_error_string(code)) > 0);
  // sizing: sane shapes give a size, shapes outside the kernels' range give 0, nothing overflows on the way
  EXPECT(nif_xty_workspace_bytes(325000, 384, 384, 0, 0) > 0);
  EXPECT(nif_xty_workspace_bytes(INT64_MAX / 4, 24576, 24576, 1, 0) > 0);
  EXPECT(nif_lap_workspace_bytes(384) > 0);
  EXPECT(nif_lap_workspace_bytes(8192) > 0);
  EXPECT(nif_lap_workspace_bytes(100000) == 0);
  EXPECT(nif_lap_workspace_bytes(-3) == 0);
  EXPECT(nif_lap_batch_workspace_bytes(384, 16) == 16 * nif_lap_workspace_bytes(384));
  EXPECT(nif_lap_batch_workspace_bytes(384, 0) == 0);
  EXPECT(nif_lap_batch_workspace_bytes(100000, 2) == 0);
  EXPECT(nif_kmeans_workspace_bytes(24576, 3072) > (size_t)24576 * 3072 * 4);
  EXPECT(nif_kmeans_workspace_bytes(0, 5) == 0);
  EXPECT(nif_emd_workspace_bytes(2100, 1500) > 0);
  EXPECT(nif_emd_workspace_bytes(-1, 4) == 0);
  EXPECT(nif_emd_batch_workspace_bytes(384, 4) > 4 * nif_emd_workspace_bytes(384, 384));
  EXPECT(nif_emd_batch_workspace_bytes(384, 0) == 0);
  EXPECT(nif_emd_batch_workspace_bytes(-1, 4) == 0);
  EXPECT(nif_eigh_kept_workspace_bytes(4609) > 0);
  EXPECT(nif_eigh_kept_workspace_bytes(NIF_EIGH_MAX_N + 1) == 0);
  EXPECT(nif_lsq_solve_workspace_bytes(4609, 1024) > 0);
  EXPECT(nif_lsq_refit_workspace_bytes(400, 4609, 1024, 0) > 0);
  EXPECT(nif_lsq_refit_workspace_bytes(400, 4609, 1024, NIF_LSQ_PRIMAL) > 0);
  EXPECT(nif_lsq_refit_workspace_bytes(1 << 20, NIF_EIGH_MAX_N + 1, 8, 0) == 0);
  EXPECT(nif_wmse_workspace_bytes() > 0);
  // null pointers / bad sizes are rejected before any launch
  int32_t i32 = 0;
  double f64 = 0.0;
  int64_t i64 = 0;
  EXPECT(nif_xty(NULL, 4, NULL, 4, 8, 4, 4, 0, 0, 0, NULL, 4, NULL, NULL, NULL, 0, NULL) == NIF_ERR_INVALID_ARG);
  EXPECT(nif_lap_solve_f32(NULL, 4, 4, 0, NULL, NULL, NULL, NULL, NULL, 0, NULL) == NIF_ERR_INVALID_ARG);
  {
    // the batched entry points: every output is required, problems may not overlap, the batch must be positive
    float c4[64];
    int32_t i4[16];
    double d4[16];
    EXPECT(nif_lap_solve_batch_f32(NULL, 4, 16, 4, 2, 0, i4, i4, d4, d4, d4, 1 << 20, NULL) == NIF_ERR_INVALID_ARG);
    EXPECT(nif_lap_solve_batch_f32(c4, 4, 16, 4, 2, 0, i4, NULL, d4, d4, d4, 1 << 20, NULL) == NIF_ERR_INVALID_ARG);
    EXPECT(nif_lap_solve_batch_f32(c4, 4, 15, 4, 2, 0, i4, i4, d4, d4, d4, 1 << 20, NULL) == NIF_ERR_INVALID_ARG);
    EXPECT(nif_lap_solve_batch_f32(c4, 4, 16, 4, 0, 0, i4, i4, d4, d4, d4, 1 << 20, NULL) == NIF_ERR_INVALID_ARG);
    EXPECT(nif_lap_solve_batch_f32(c4, 4, 16, 4, 2, 0, i4, i4, d4, d4, d4, 8, NULL) == NIF_ERR_WORKSPACE);
    EXPECT(nif_emd_solve_batch_f32(c4, 4, 16, 4, 2, d4, d4, NULL, 0, i4, i4, d4, 16, i4, d4, d4, d4, 1 << 20, NULL) ==
           NIF_ERR_INVALID_ARG);
    EXPECT(nif_emd_solve_batch_f32(c4, 4, 15, 4, 2, d4, d4, d4, 0, i4, i4, d4, 16, i4, d4, d4, d4, 1 << 20, NULL) ==
           NIF_ERR_INVALID_ARG);
    EXPECT(nif_emd_solve_batch_f32(c4, 4, 16, 4, 2, d4, d4, d4, NIF_EMD_ARC_POOL, i4, i4, d4, 16, i4, d4, d4, d4, 1 << 20,
                                   NULL) == NIF_ERR_UNSUPPORTED);
    EXPECT(nif_emd_solve_batch_f32(c4, 4, 16, 4, 2, d4, d4, d4, 0, i4, i4, d4, 16, i4, d4, d4, d4, 8, NULL) ==
           NIF_ERR_WORKSPACE);
  }
  EXPECT(nif_kmeans_lloyd_step(NULL, 4, NULL, NULL, NULL, 4, 2, NULL, NULL, NULL, NULL, NULL, 0, NULL) ==
         NIF_ERR_INVALID_ARG);
  EXPECT(nif_kmeans_lloyd_step_gf32(NULL, 4, NULL, NULL, NULL, NULL, 4, 2, NULL, NULL, NULL, NULL, NULL, 0, NULL) ==
         NIF_ERR_INVALID_ARG);
  EXPECT(nif_kmeans_farthest_init(NULL, 4, NULL, 4, 2, 0, NULL, NULL, NULL, NULL) == NIF_ERR_INVALID_ARG);
  EXPECT(nif_kmeans_targets(NULL, 4, 4, NULL, NULL, 4, 2, NULL, 2, NULL, NULL, 0, NULL) == NIF_ERR_INVALID_ARG);
  EXPECT(nif_refit_pairs(NULL, 4, NULL, NULL, NULL, 4, 4, NULL, NULL) == NIF_ERR_INVALID_ARG);
  EXPECT(nif_lsq_refit_f32(NULL, 4, 4, 4, NULL, 4, 4, 0, NULL, 4, NULL, 4, NULL, &i32, NULL, 0, NULL) ==
         NIF_ERR_INVALID_ARG);
  // profile slots: out-of-range slots are refused, an empty slot collects nothing
  EXPECT(nif_profile_collect(-1, &f64, &i64) == NIF_ERR_INVALID_ARG);
  EXPECT(nif_profile_collect(NIF_PROFILE_SLOTS, &f64, &i64) == NIF_ERR_INVALID_ARG);
  EXPECT(nif_profile_collect(0, &f64, &i64) == NIF_OK && i64 == 0);
  EXPECT(nif_profile_work(0, &f64) == NIF_OK && f64 == 0.0);
  nif_profile_enable(0);
  if (failures) {
    fprintf(stderr, "host_check: %d failure(s)\n", failures);
    return 1;
  }
  printf("host_check: ok\n");
  return 0;
}
