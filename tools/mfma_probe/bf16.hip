// Probe: operand layout and accumulation behaviour of v_mfma_f32_32x32x16_bf16 on gfx950 (tuning aid).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline unsigned short f2bf(float x) {  // round to nearest even
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ inline float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// A: [16][32] (k, i) f32, B: [16][32] (k, j) -> C[i][j] = sum_k A[k][i] B[k][j]; terms: 1 = hi*hi only, 6 = split
__global__ void probe(const float* A, const float* B, float* C, int terms) {
  const int lane = threadIdx.x;
  const int i = lane & 31, kb = 8 * (lane >> 5);
  union { bf16x8 v; unsigned short s[8]; } a[3], b[3];
  for (int e = 0; e < 8; ++e) {
    float x = A[(kb + e) * 32 + i];
    float y = B[(kb + e) * 32 + i];
    for (int s = 0; s < 3; ++s) {
      a[s].s[e] = f2bf(x); x -= bf2f(a[s].s[e]);
      b[s].s[e] = f2bf(y); y -= bf2f(b[s].s[e]);
    }
  }
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  if (terms == 1) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
  } else {
    // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
  }
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    C[row * 32 + (lane & 31)] = acc[e];
  }
}

int main() {
  float hA[512], hB[512], hC[1024];
  srand(1);
  float *dA, *dB, *dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
  for (int mode = 0; mode < 2; ++mode) {
    for (int q = 0; q < 512; ++q) {
      if (mode == 0) { hA[q] = (float)(rand() % 17 - 8); hB[q] = (float)(rand() % 13 - 6); }  // exact in bf16
      else { hA[q] = (float)rand() / RAND_MAX * 2 - 1; hB[q] = (float)rand() / RAND_MAX * 2 - 1; }
    }
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC, mode == 0 ? 1 : 6);
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    double maxerr = 0, maxerr32 = 0, maxref = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0; float r32 = 0.f;
      for (int k = 0; k < 16; ++k) { ref += (double)hA[k * 32 + i] * hB[k * 32 + j]; r32 = fmaf(hA[k * 32 + i], hB[k * 32 + j], r32); }
      maxerr = fmax(maxerr, fabs(hC[i * 32 + j] - ref)); maxerr32 = fmax(maxerr32, fabs((double)r32 - ref)); maxref = fmax(maxref, fabs(ref));
    }
    printf("mode %d: max |mfma - exact| = %.3e (f32 fma chain: %.3e), max |ref| = %.3f\n", mode, maxerr, maxerr32, maxref);
  }
  return 0;
}
