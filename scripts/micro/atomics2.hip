#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
// G hot groups whose slots are spread over a 65536-slot table (one cache line each), 2 atomics per row
__global__ void k_spread(double* f, u64* c, long n, int G, int spread) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    unsigned h = (unsigned)(i * 2654435761u) >> 7;
    unsigned g = h % G;
    unsigned slot = spread ? (g * 40503u) & 65535u : g;
    unsafeAtomicAdd(&f[slot], 1.0);
    atomicAdd(&c[slot], 1ULL);
  }
}
int main() {
  long n = 1L << 24;
  double* f; u64* c; hipMalloc(&f, 8L << 16); hipMalloc(&c, 8L << 16); hipMemset(f, 0, 8L << 16); hipMemset(c, 0, 8L << 16);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int spread = 0; spread < 2; spread++)
    for (int G : {8, 64, 1024}) {
      float ms;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a, s);
        hipLaunchKernelGGL(k_spread, 2048, 256, 0, s, f, c, n, G, spread);
        hipEventRecord(b, s); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      }
      printf("spread=%d G=%5d: %9.3f ms for %ld rows x 2 atomics (%.1f M rows/s)\n", spread, G, ms, n, n / ms / 1e3);
    }
  return 0;
}
