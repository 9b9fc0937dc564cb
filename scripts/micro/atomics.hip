// micro-benchmark: rate of global atomic adds onto G hot addresses (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__global__ void k_f64(double* t, long n, int G) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    unsigned h = (unsigned)(i * 2654435761u) >> 7;
    unsafeAtomicAdd(&t[h % G], 1.0);
  }
}
__global__ void k_u64(u64* t, long n, int G) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    unsigned h = (unsigned)(i * 2654435761u) >> 7;
    atomicAdd(&t[h % G], 1ULL);
  }
}
__global__ void k_f32(float* t, long n, int G) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    unsigned h = (unsigned)(i * 2654435761u) >> 7;
    unsafeAtomicAdd(&t[h % G], 1.0f);
  }
}
__global__ void k_load(const u64* t, long n, int G, u64* out) {
  u64 acc = 0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    unsigned h = (unsigned)(i * 2654435761u) >> 7;
    acc += __hip_atomic_load(&t[h % G], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == 12345) out[0] = acc;
}
int main() {
  long n = 1L << 24;
  void* t; hipMalloc(&t, 8L << 22); hipMemset(t, 0, 8L << 22);
  u64* out; hipMalloc(&out, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int Gs[] = {1, 64, 1024, 65536, 4194304};
  for (int G : Gs) {
    float ms[4];
    for (int k = 0; k < 4; k++) {
      hipEventRecord(a);
      if (k == 0) hipLaunchKernelGGL(k_f64, 2048, 256, 0, 0, (double*)t, n, G);
      if (k == 1) hipLaunchKernelGGL(k_u64, 2048, 256, 0, 0, (u64*)t, n, G);
      if (k == 2) hipLaunchKernelGGL(k_f32, 2048, 256, 0, 0, (float*)t, n, G);
      if (k == 3) hipLaunchKernelGGL(k_load, 2048, 256, 0, 0, (const u64*)t, n, G, out);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[k], a, b);
    }
    printf("G=%8d  f64 add %8.3f ms (%7.1f M/s)  u64 add %8.3f ms (%7.1f M/s)  f32 add %8.3f ms (%7.1f M/s)  sc1 load %8.3f ms (%7.1f M/s)\n", G,
           ms[0], n / ms[0] / 1e3, ms[1], n / ms[1] / 1e3, ms[2], n / ms[2] / 1e3, ms[3], n / ms[3] / 1e3);
  }
  return 0;
}
