// nif_common.h -- small device/host helpers shared by the libnif translation units.
#ifndef NIF_COMMON_H_
#define NIF_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace nif {

__device__ __forceinline__ double shfl_xor_f64(double x, int mask) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += shfl_xor_f64(x, o);
  return x;
}

// hipGetLastError() is per-thread state shared with every other HIP user in the process (torch's
// event polling leaves hipErrorNotReady behind), so entry points clear it first and read it after
// their own launches.
static inline void clear_err() { (void)hipGetLastError(); }
static inline int launch_ok() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) fprintf(stderr, "[libnif] HIP error after launch: %s\n", hipGetErrorString(e));
  return e == hipSuccess;
}

// Optional per-kernel timing (nif_profile_enable): HIP events recorded on the launch stream
// around the dominant kernel of an entry point.  No-ops unless enabled.
void prof_begin(int slot, hipStream_t st);
void prof_end(int slot, hipStream_t st);

}  // namespace nif

#endif  // NIF_COMMON_H_
