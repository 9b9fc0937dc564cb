// Host-side cost of the runtime calls an operator makes while it sets up (1 x MI355X): hipMemsetAsync of small / large ranges,
// a fill kernel launch, hipEventCreate, hipEventRecord, hipStreamQuery.  hipcc -O2 host_costs.cpp -o host_costs
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
using Clock = std::chrono::steady_clock;
__global__ void k_fill(unsigned long long* p, unsigned long long v, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
template <typename F> double per_call_us(F&& f, int reps)
{
    const auto t0 = Clock::now();
    for (int i = 0; i < reps; i++) f();
    return std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / reps;
}
int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    void* big;
    hipMalloc(&big, 256u << 20);
    void* small;
    hipMalloc(&small, 4096);
    hipStreamSynchronize(s);
    for (int round = 0; round < 2; round++) {
        printf("hipMemsetAsync 512 B      : %.1f us per call (host)\n", per_call_us([&] { hipMemsetAsync(small, 0, 512, s); }, 50));
        hipStreamSynchronize(s);
        printf("hipMemsetAsync 232 MB     : %.1f us per call (host)\n", per_call_us([&] { hipMemsetAsync(big, 0, 232u << 20, s); }, 10));
        hipStreamSynchronize(s);
        printf("fill kernel 232 MB        : %.1f us per call (host)\n", per_call_us([&] { hipLaunchKernelGGL(k_fill, 2048, 256, 0, s, (unsigned long long*)big, 0ULL, (long long)(232u << 20) / 8); }, 10));
        hipStreamSynchronize(s);
        std::vector<hipEvent_t> ev(64);
        int k = 0;
        printf("hipEventCreate            : %.1f us per call\n", per_call_us([&] { hipEventCreate(&ev[k++]); }, 64));
        printf("hipEventCreateWithFlags(disable timing): %.1f us per call\n", per_call_us([&] { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); hipEventDestroy(e); }, 64));
        printf("hipEventRecord            : %.1f us per call\n", per_call_us([&] { hipEventRecord(ev[0], s); }, 64));
        printf("hipStreamQuery (idle)     : %.1f us per call\n", per_call_us([&] { hipStreamQuery(s); }, 64));
        for (auto e : ev) hipEventDestroy(e);
        hipStreamSynchronize(s);
        void* p = nullptr;
        printf("hipMalloc+hipFree 64 MB   : %.1f us per pair\n", per_call_us([&] { hipMalloc(&p, 64u << 20); hipFree(p); }, 5));
    }
    return 0;
}
