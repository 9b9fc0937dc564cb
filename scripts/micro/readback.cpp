// readback.cpp -- what one "device decides, host reads the word, host launches the next kernel" round trip costs on this part.
//   A  hipMemcpyAsync(D2H, 8 B) into pinned memory + hipStreamSynchronize
//   P  the same into pageable memory (a variable on the stack: the runtime stages the copy and blocks)
//   B  a one-wave kernel stores the words + a sequence flag into host-coherent pinned memory; the host spins on the flag
//   C  as B, but the producing kernel itself publishes (no extra launch)
// Each loop: work kernel (touches `n` elements) -> read back its result word -> next iteration depends on it.
// hipcc --offload-arch=gfx950 -O2 scripts/micro/readback.cpp -o scripts/micro/readback
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_work(const int* in, int n, int* result, int add)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && in[i] == -1) atomicAdd(result, 1);
    if (i == 0) atomicAdd(result, add);
}
__global__ void k_publish(const int* src, volatile int* dst, int words, volatile unsigned* flag, unsigned seq)
{
    if ((int)threadIdx.x < words) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store((unsigned*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_work_publish(const int* in, int n, int* result, int add, volatile int* dst, volatile unsigned* flag, unsigned seq, unsigned* ticket)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && in[i] == -1) atomicAdd(result, 1);
    if (i == 0) atomicAdd(result, add);
    __threadfence();
    __shared__ unsigned last;
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u);
    __syncthreads();
    if (last == gridDim.x - 1 && threadIdx.x == 0) {   // the last workgroup to finish publishes
        dst[0] = __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *ticket = 0;
        __threadfence_system();
        __hip_atomic_store((unsigned*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1 << 20, iters = 2000;
    int *in, *result;
    unsigned* ticket;
    CK(hipMalloc(&in, (size_t)n * 4));
    CK(hipMemset(in, 0, (size_t)n * 4));
    CK(hipMalloc(&result, 64));
    CK(hipMalloc(&ticket, 4));
    CK(hipMemset(ticket, 0, 4));
    int* host;
    CK(hipHostMalloc(&host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    volatile unsigned* flag = (volatile unsigned*)(host + 512);
    *flag = 0;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int grid = (n + 255) / 256;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int mode = -1; mode < 3; mode++) {
        long long check = 0;
        unsigned seq = *flag;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemsetAsync(result, 0, 4, s));
            CK(hipStreamSynchronize(s));
            auto t0 = now();
            int add = 1;
            for (int it = 0; it < iters; it++) {
                if (mode == -1) {
                    int pageable[2];
                    hipLaunchKernelGGL(k_work, dim3(grid), dim3(256), 0, s, in, n, result, add);
                    CK(hipMemcpyAsync(pageable, result, 8, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    host[0] = pageable[0];
                }
                else if (mode == 0) {
                    hipLaunchKernelGGL(k_work, dim3(grid), dim3(256), 0, s, in, n, result, add);
                    CK(hipMemcpyAsync(host, result, 8, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                }
                else if (mode == 1) {
                    hipLaunchKernelGGL(k_work, dim3(grid), dim3(256), 0, s, in, n, result, add);
                    seq++;
                    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, result, host, 2, flag, seq);
                    while (__atomic_load_n((unsigned*)flag, __ATOMIC_ACQUIRE) != seq) {}
                }
                else {
                    seq++;
                    hipLaunchKernelGGL(k_work_publish, dim3(grid), dim3(256), 0, s, in, n, result, add, host, flag, seq, ticket);
                    while (__atomic_load_n((unsigned*)flag, __ATOMIC_ACQUIRE) != seq) {}
                }
                add = (host[0] & 1) + 1;   // the next launch depends on the word read back
                check += host[0];
            }
            CK(hipStreamSynchronize(s));
            auto t1 = now();
            if (rep == 1) printf("mode %c  n=%d  %.2f us per round trip (check %lld)\n", "PABC"[mode + 1], n, us(t0, t1) / iters, check);
        }
    }
    // the bare kernel, back to back, for reference
    CK(hipStreamSynchronize(s));
    auto t0 = now();
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(k_work, dim3(grid), dim3(256), 0, s, in, n, result, 1);
    CK(hipStreamSynchronize(s));
    printf("bare   n=%d  %.2f us per launch, back to back\n", n, us(t0, now()) / iters);
    return 0;
}
