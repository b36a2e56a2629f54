// nif_core.hip -- version / error strings / device probe of libnif.so.
#include <hip/hip_runtime.h>

#include "../../include/nif.h"

extern "C" int nif_version(void) { return 100; }  // 0.1.0

extern "C" const char* nif_error_string(int code) {
  switch (code) {
    case NIF_OK: return "ok";
    case NIF_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size, bad enum or leading dimension)";
    case NIF_ERR_WORKSPACE: return "workspace missing or too small (see *_workspace_bytes)";
    case NIF_ERR_UNSUPPORTED: return "size or mode outside what the kernels cover";
    case NIF_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
    case NIF_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown error";
  }
}

extern "C" int nif_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return NIF_ERR_NO_DEVICE;
  return n;
}

// ---- optional kernel timing -------------------------------------------------------
#include <mutex>
#include <vector>

#include "nif_common.h"

namespace nif {
namespace {
struct Span {
  int slot;
  hipEvent_t a, b;
};
std::mutex g_mu;
unsigned g_enabled = 0;  // bit s: spans of slot s are recorded
std::vector<Span> g_spans;
hipEvent_t g_open[NIF_PROFILE_SLOTS];
double g_work[NIF_PROFILE_SLOTS];
}  // namespace

void prof_begin(int slot, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot < 0 || slot >= NIF_PROFILE_SLOTS || !((g_enabled >> slot) & 1)) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, st);
  g_open[slot] = e;
}

void prof_end(int slot, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot < 0 || slot >= NIF_PROFILE_SLOTS || !((g_enabled >> slot) & 1) || !g_open[slot]) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, st);
  g_spans.push_back(Span{slot, g_open[slot], e});
  g_open[slot] = nullptr;
}
}  // namespace nif

namespace nif {
void prof_add_work(int slot, double units) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot < 0 || slot >= NIF_PROFILE_SLOTS || !((g_enabled >> slot) & 1)) return;
  g_work[slot] += units;
}
}  // namespace nif

extern "C" int nif_profile_work(int slot, double* units) {
  if (!units || slot < 0 || slot >= NIF_PROFILE_SLOTS) return NIF_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(nif::g_mu);
  *units = nif::g_work[slot];
  nif::g_work[slot] = 0.0;
  return NIF_OK;
}

extern "C" void nif_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(nif::g_mu);
  nif::g_enabled = on < 0 ? ~0u : (unsigned)on;  // bit s = slot s; negative = every slot
}

// durations of a slot's finished spans in record order (the spans are consumed): the first `cap` go to ms_out
static int collect_spans(int slot, double* total_ms, int64_t* launches, float* ms_out, int64_t cap) {
  std::lock_guard<std::mutex> lk(nif::g_mu);
  double tot = 0.0;
  int64_t cnt = 0;
  std::vector<nif::Span> keep;
  for (auto& s : nif::g_spans) {
    if (s.slot != slot) {
      keep.push_back(s);
      continue;
    }
    float ms = 0.f;
    if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      if (ms_out && cnt < cap) ms_out[cnt] = ms;
      tot += ms;
      ++cnt;
    }
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  nif::g_spans.swap(keep);
  *total_ms = tot;
  *launches = cnt;
  return NIF_OK;
}

extern "C" int nif_profile_collect(int slot, double* total_ms, int64_t* launches) {
  if (!total_ms || !launches || slot < 0 || slot >= NIF_PROFILE_SLOTS) return NIF_ERR_INVALID_ARG;
  return collect_spans(slot, total_ms, launches, nullptr, 0);
}

extern "C" int nif_profile_collect_spans(int slot, float* ms_out, int64_t cap, int64_t* launches) {
  if (!launches || slot < 0 || slot >= NIF_PROFILE_SLOTS || cap < 0 || (cap > 0 && !ms_out)) return NIF_ERR_INVALID_ARG;
  double tot = 0.0;
  return collect_spans(slot, &tot, launches, ms_out, cap);
}

// ---- roctx ranges ---------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <stdlib.h>

namespace nif {
namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
roctx_push_t g_push = nullptr;
roctx_pop_t g_pop = nullptr;
std::once_flag g_roctx_once;

void roctx_load() {
  const char* e = getenv("NIF_ROCTX");
  if (!e || atoi(e) == 0) return;
  const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so",
                         "libroctx64.so.4"};
  for (const char* n : names) {
    void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    g_push = reinterpret_cast<roctx_push_t>(dlsym(h, "roctxRangePushA"));
    g_pop = reinterpret_cast<roctx_pop_t>(dlsym(h, "roctxRangePop"));
    if (g_push && g_pop) return;
    g_push = nullptr;
    g_pop = nullptr;
  }
}
}  // namespace

Range::Range(const char* name) : open_(false) {
  std::call_once(g_roctx_once, roctx_load);
  if (g_push && g_pop) {
    (void)g_push(name);
    open_ = true;
  }
}
Range::~Range() {
  if (open_) (void)g_pop();
}
}  // namespace nif
