// nif_im2row.hip -- design matrix of a conv level for the LS refit (hf_kf.py:196-208, 224-225: F.unfold of
// the level input, rows ordered (sample, out_y, out_x), columns (in_channel, ky, kx), plus a ones column for
// the bias), written in ONE launch straight into the row-padded layout the Gram kernel reads.
//
// HBM-bound: writes 4*ld bytes per row once (the pad columns are written as zeros, so the buffer needs no
// clearing); the input is m*d/(kh*kw*stride^2)-ish smaller and its re-reads (each pixel is used by kh*kw
// patches) come from L2.  One thread per output element, columns fastest: stores are fully coalesced, loads
// of consecutive kx are contiguous.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nif.h"
#include "nif_common.h"

namespace nif {

struct Im2RowArgs {
  const float* x;
  float* out;
  int64_t ldo;
  int C, H, W, kh, kw, sh, sw, ph, pw, OH, OW;
  int d;     // C * kh * kw
  int ones;  // 1: column d holds 1.0
};

constexpr int IM2ROW_NT = 256;

__global__ __launch_bounds__(IM2ROW_NT) void im2row_kernel(Im2RowArgs a, int64_t rows) {
  // blockIdx.y strides over rows (uniform per block: its index arithmetic is scalar); blockIdx.x * blockDim.x + tid = column (including the ones / pad columns)
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= a.ldo) return;
  int c = 0, ky = 0, kx = 0;
  const bool patch = col < a.d;
  if (patch) {
    const int kk = a.kh * a.kw;
    c = col / kk;
    const int r = col - c * kk;
    ky = r / a.kw;
    kx = r - ky * a.kw;
  }
  const float fill = (a.ones && col == a.d) ? 1.0f : 0.0f;
  const int64_t plane = (int64_t)a.H * a.W;
  const int ohw = a.OH * a.OW;
  for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
    float val = fill;
    if (patch) {
      const int64_t b = row / ohw;
      const int p = (int)(row - b * ohw);
      const int oy = p / a.OW;
      const int ox = p - oy * a.OW;
      const int iy = oy * a.sh - a.ph + ky;
      const int ix = ox * a.sw - a.pw + kx;
      val = 0.0f;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) val = a.x[(b * a.C + c) * plane + (int64_t)iy * a.W + ix];
    }
    a.out[row * a.ldo + col] = val;
  }
}

}  // namespace nif

extern "C" int nif_im2row_f32(const float* x, int64_t batch, int C, int H, int W, int kh, int kw, int sh, int sw,
                              int ph, int pw, int add_ones, float* out, int64_t ldo, void* stream) {
  nif::Range nif_range_("nif_im2row_f32");
  using namespace nif;
  if (!x || !out || batch <= 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 ||
      pw < 0)
    return NIF_ERR_INVALID_ARG;
  if (H + 2 * ph < kh || W + 2 * pw < kw) return NIF_ERR_INVALID_ARG;
  const int64_t d64 = (int64_t)C * kh * kw;
  if (d64 + (add_ones ? 1 : 0) > ldo || ldo > (int64_t)1 << 30) return NIF_ERR_INVALID_ARG;
  Im2RowArgs a;
  a.x = x;
  a.out = out;
  a.ldo = ldo;
  a.C = C; a.H = H; a.W = W; a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw;
  a.OH = (H + 2 * ph - kh) / sh + 1;
  a.OW = (W + 2 * pw - kw) / sw + 1;
  a.d = (int)d64;
  a.ones = add_ones ? 1 : 0;
  const int64_t rows = batch * a.OH * a.OW;
  clear_err();
  const int nt = ldo >= IM2ROW_NT ? IM2ROW_NT : (int)((ldo + 63) / 64 * 64);  // narrow levels: fewer idle lanes
  const unsigned gx = (unsigned)((ldo + nt - 1) / nt);
  int64_t gy = rows;
  const int64_t cap = 65535;  // rows beyond it are strided over
  if (gy > cap) gy = cap;
  hipLaunchKernelGGL(im2row_kernel, dim3(gx, (unsigned)gy), dim3(nt), 0, reinterpret_cast<hipStream_t>(stream), a,
                     rows);
  return launch_ok() ? NIF_OK : NIF_ERR_LAUNCH;
}
