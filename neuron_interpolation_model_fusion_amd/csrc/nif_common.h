// nif_common.h -- small device/host helpers shared by the libnif translation units.
#ifndef NIF_COMMON_H_
#define NIF_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

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

static inline int launch_ok() { return hipGetLastError() == hipSuccess; }

}  // namespace nif

#endif  // NIF_COMMON_H_
