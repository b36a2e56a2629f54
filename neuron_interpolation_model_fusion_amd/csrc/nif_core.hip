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
