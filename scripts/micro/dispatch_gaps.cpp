// Do back-to-back dispatches on one stream leave the device idle between them?  Elapsed time (HIP events) of short sequences of
// memsets / fill kernels enqueued without any host wait, against the sum of their own durations.  hipcc -O2 dispatch_gaps.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fill(unsigned long long* p, unsigned long long v, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
__global__ void k_touch(unsigned long long* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    void *big, *big2, *small;
    hipMalloc(&big, 256u << 20);
    hipMalloc(&big2, 256u << 20);
    hipMalloc(&small, 4096);
    hipEvent_t e[8];
    for (auto& x : e) hipEventCreate(&x);
    auto ms = [&](int a, int b) { float t = 0; hipEventElapsedTime(&t, e[a], e[b]); return t * 1e3f; };
    for (int round = 0; round < 3; round++) {
        hipStreamSynchronize(s);
        hipEventRecord(e[0], s);
        hipMemsetAsync(small, 0, 512, s);
        hipEventRecord(e[1], s);
        hipMemsetAsync(big, 0, 232u << 20, s);
        hipEventRecord(e[2], s);
        hipLaunchKernelGGL(k_fill, 2048, 256, 0, s, (unsigned long long*)big2, 1ULL, (long long)(116u << 20) / 8);
        hipEventRecord(e[3], s);
        hipMemsetAsync(small, 0, 512, s);
        hipEventRecord(e[4], s);
        hipLaunchKernelGGL(k_touch, 1, 64, 0, s, (unsigned long long*)small);
        hipEventRecord(e[5], s);
        hipStreamSynchronize(s);
        printf("with events between: small memset %.1f | 232 MB memset %.1f | 116 MB fill kernel %.1f | small memset %.1f | tiny kernel %.1f | total %.1f us\n", ms(0, 1), ms(1, 2),
               ms(2, 3), ms(3, 4), ms(4, 5), ms(0, 5));
        hipEventRecord(e[0], s);
        hipMemsetAsync(small, 0, 512, s);
        hipMemsetAsync(big, 0, 232u << 20, s);
        hipLaunchKernelGGL(k_fill, 2048, 256, 0, s, (unsigned long long*)big2, 1ULL, (long long)(116u << 20) / 8);
        hipMemsetAsync(small, 0, 512, s);
        hipLaunchKernelGGL(k_touch, 1, 64, 0, s, (unsigned long long*)small);
        hipEventRecord(e[1], s);
        hipStreamSynchronize(s);
        printf("the same five, no events between: total %.1f us\n", ms(0, 1));
        hipEventRecord(e[0], s);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_touch, 1, 64, 0, s, (unsigned long long*)small);
        hipEventRecord(e[1], s);
        hipStreamSynchronize(s);
        printf("five tiny kernels: total %.1f us\n", ms(0, 1));
        hipEventRecord(e[0], s);
        for (int i = 0; i < 5; i++) hipMemsetAsync(small, 0, 512, s);
        hipEventRecord(e[1], s);
        hipStreamSynchronize(s);
        printf("five small memsets: total %.1f us\n", ms(0, 1));
    }
    return 0;
}
