// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (tuning aid).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[16][128];  // [row][col], value = row * 256 + col
  for (int q = threadIdx.x; q < 16 * 128; q += 64) img[q / 128][q % 128] = (unsigned short)((q / 128) * 256 + (q % 128));
  __syncthreads();
  const int lane = threadIdx.x;
  const int grp = lane >> 4, l = lane & 15, q = l >> 2, pp = l & 3;
  // group g: block rows r0 .. r0+3, columns c0 .. c0+15; lane 4q+p supplies row r0+q, columns c0+4p ..
  const int r0 = 4 * (grp >> 1), c0 = 16 * (grp & 1);
  typedef __attribute__((address_space(3))) s16x4 lds_v;
  lds_v* p = (lds_v*)(&img[r0 + q][c0 + 4 * pp]);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, 512);
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; lane += 1) {
    if (lane % 16 < 3 || lane % 16 == 15) {
      printf("lane %2d:", lane);
      for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[lane * 4 + e] >> 8, h[lane * 4 + e] & 255);
      printf("\n");
    }
  }
  return 0;
}
