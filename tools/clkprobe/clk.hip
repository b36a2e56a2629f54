// Shader clock seen by a single-workgroup latency-bound kernel vs. a chip-filling one (tuning probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chase(const int* __restrict__ next, int steps, long long* out, int* sink) {
  int p = threadIdx.x + blockIdx.x;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int s = 0; s < steps; ++s) p = next[p];
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  if (p == -12345) *sink = p;
}
__global__ void lds_chase(int steps, long long* out, int* sink) {
  __shared__ int nx[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) nx[i] = (i * 17 + 1) & 1023;
  __syncthreads();
  int p = threadIdx.x;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int s = 0; s < steps; ++s) p = nx[p];
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  if (p == -12345) *sink = p;
}
__global__ void barrier_loop(int steps, long long* out) {
  long long c0 = clock64(), w0 = wall_clock64();
  for (int s = 0; s < steps; ++s) __syncthreads();
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
int main() {
  const int N = 1 << 18;  // 1 MB of ints: L2 resident
  int* h = (int*)malloc(N * 4);
  for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 4099 + 64 * 33) % N);
  int *d, *sink; long long* out;
  hipMalloc(&d, N * 4); hipMalloc(&sink, 4); hipMalloc(&out, 16);
  hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
  long long r[2];
  int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", wc_khz);
  for (int rep = 0; rep < 2; ++rep) {
    const int steps = 20000;
    chase<<<1, 64>>>(d, steps, out, sink); hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    printf("global chase 1 wave: %.1f shader cycles/step, %.1f ns/step\n", (double)r[0] / steps, (double)r[1] / steps * 1e6 / wc_khz);
    lds_chase<<<1, 64>>>(steps, out, sink); hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    printf("lds chase: %.1f cycles/step, %.1f ns/step\n", (double)r[0] / steps, (double)r[1] / steps * 1e6 / wc_khz);
    barrier_loop<<<1, 256>>>(steps, out); hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    printf("barrier 4 waves: %.1f cycles/step, %.1f ns/step\n", (double)r[0] / steps, (double)r[1] / steps * 1e6 / wc_khz);
    chase<<<2048, 64>>>(d, steps, out, sink); hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    printf("global chase 2048 blocks: %.1f cycles/step, %.1f ns/step\n", (double)r[0] / steps, (double)r[1] / steps * 1e6 / wc_khz);
  }
  return 0;
}
