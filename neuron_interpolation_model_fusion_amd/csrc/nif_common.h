// nif_common.h -- small device/host helpers shared by the libnif translation units.
#ifndef NIF_COMMON_H_
#define NIF_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

namespace nif {

__device__ __forceinline__ double shfl_xor_f64(double x, int mask) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return __hiloint2double(hi, lo);
}

// Wave-wide f64 sum, the same value returned to every lane, through DPP row shifts / row broadcasts: a fixed tree
// (sums of 2, 4, 8, 16 neighbours inside each row of 16 lanes, then rows 0+1 and 2+3, then the halves).  The xor
// butterfly over __shfl_xor this replaces compiles to ds_bpermute on gfx950 -- twelve trips through the LDS
// crossbar, ~0.3 us per sum, which was a sixth of a latency-bound tridiagonalisation step.  All 64 lanes must be
// active at the call (a lane whose DPP source is inactive or outside its row adds 0).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_f64(double x) {
  x += dpp_f64<0x111, 0xf>(x);  // row_shr:1
  x += dpp_f64<0x112, 0xf>(x);  // row_shr:2
  x += dpp_f64<0x114, 0xf>(x);  // row_shr:4
  x += dpp_f64<0x118, 0xf>(x);  // row_shr:8   -> lane 15 of each row of 16 holds the row's sum
  x += dpp_f64<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  x += dpp_f64<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63),
                          __builtin_amdgcn_readlane(__double2loint(x), 63));
}

// ---- wave-wide minimum through DPP row shifts / row broadcasts (no LDS crossbar traffic): the result is
// returned to every lane.  Lanes without a source keep `old`, the identity of the reduction.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int x) {
  return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
  const int id = -1;  // 0xffffffff
#define NIF_DPP_STEP(CTRL, RM) x = min(x, (unsigned)dpp_i32<CTRL, RM>(id, (int)x))
  NIF_DPP_STEP(0x111, 0xf);  // row_shr:1
  NIF_DPP_STEP(0x112, 0xf);  // row_shr:2
  NIF_DPP_STEP(0x114, 0xf);  // row_shr:4
  NIF_DPP_STEP(0x118, 0xf);  // row_shr:8   -> lane 15 of each row of 16 holds the row's minimum
  NIF_DPP_STEP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  NIF_DPP_STEP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
#undef NIF_DPP_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// minimum over a row of 16 lanes (rows 1-3 compute their own): the value lane 15 ends up with
__device__ __forceinline__ unsigned row16_min_u32(unsigned x) {
  const int id = -1;
#define NIF_DPP_STEP(CTRL, RM) x = min(x, (unsigned)dpp_i32<CTRL, RM>(id, (int)x))
  NIF_DPP_STEP(0x111, 0xf);
  NIF_DPP_STEP(0x112, 0xf);
  NIF_DPP_STEP(0x114, 0xf);
  NIF_DPP_STEP(0x118, 0xf);
#undef NIF_DPP_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)x, 15);
}

// order-preserving map of finite / infinite doubles onto unsigned 64-bit integers (-0.0 sorts below +0.0:
// callers that care add +0.0 first) and back: lets a floating-point minimum run as integer DPP minima
__device__ __forceinline__ unsigned long long f64_to_ordered(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ordered_to_f64(unsigned long long o) {
  const unsigned long long b = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
  return __longlong_as_double((long long)b);
}

// wave-wide f64 minimum (finite or infinite inputs, no NaN) as integer DPP minima over the ordered form: the
// high word, then -- only if several lanes share it -- the low word among them
__device__ __forceinline__ double wave_min_f64(double x) {
  const unsigned long long o = f64_to_ordered(x);
  const unsigned hi = (unsigned)(o >> 32), lo = (unsigned)o;
  const unsigned whi = wave_min_u32(hi);
  const bool holder = hi == whi;
  const unsigned long long m = __ballot(holder);
  unsigned wlo;
  if (m & (m - 1)) {
    wlo = wave_min_u32(holder ? lo : 0xffffffffu);
  } else {
    wlo = (unsigned)__builtin_amdgcn_readlane((int)lo, __ffsll(m) - 1);
  }
  return ordered_to_f64(((unsigned long long)whi << 32) | wlo);
}

// Tuning aids read from the environment once per process (0 when unset): call sites keep them in a function-local
// static, so the hot path pays nothing for them.
static inline int env_int(const char* name) {
  const char* e = getenv(name);
  return e ? atoi(e) : 0;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: a process that uses the library on
// several GPUs has to raise it on each.  A call site keeps one of these (function-local static) and runs its
// attribute calls through it: once per device ordinal (devices beyond 63: every call), under a lock, so that a second
// host thread cannot launch before the first has finished setting the attribute.
struct PerDeviceOnce {
  std::mutex mu;
  unsigned long long done = 0;
  template <typename F>
  void run(F&& f) {
    int dev = -1;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    std::lock_guard<std::mutex> lk(mu);
    if (known && (done >> dev & 1ull)) return;
    f();
    if (known) done |= 1ull << dev;
  }
};

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
void prof_add_work(int slot, double units);  // algorithmic work of the spans of a slot (nif_profile_work)

// roctx range around an entry point's host side (named after the entry point), so that rocprofv3 --marker-trace
// groups the kernels of one call.  Off unless NIF_ROCTX=1: then the roctx library is opened on first use (no link
// dependency); a missing library turns the ranges off again.
struct Range {
  explicit Range(const char* name);
  ~Range();
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;

 private:
  bool open_;
};

}  // namespace nif

#endif  // NIF_COMMON_H_
