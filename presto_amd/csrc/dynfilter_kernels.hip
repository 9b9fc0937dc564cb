// dynfilter_kernels.hip -- see dynfilter_kernels.hpp.  DynamicFilterSourceOperator.addInput
// (core/trino-main/src/main/java/io/trino/operator/DynamicFilterSourceOperator.java:225-267: every position into the
// channel's TypedSet) and updateMinMaxValues (:303-347).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "dynfilter_kernels.hpp"
#include "kernels/pa_device.h"

namespace pa {

namespace {

__device__ __forceinline__ u64 df_key(int type, const void* values, i64 r)
{
    switch (type) {
        case PA_BIGINT: return (u64)((const i64*)values)[r];
        case PA_INTEGER:
        case PA_DATE: return (u64)(i64)((const i32*)values)[r];
        case PA_BOOLEAN: return ((const u8*)values)[r] != 0 ? 1ULL : 0ULL;
        case PA_REAL: {  // the float widened (exactly) to the DOUBLE key: one NaN, +0.0 for both zeros
            const float f = ((const float*)values)[r];
            if (f != f) return 0x7ff8000000000000ULL;
            const u64 b = (u64)__double_as_longlong((double)f);
            return b == 0x8000000000000000ULL ? 0ULL : b;
        }
        default: {  // DOUBLE
            const u64 b = ((const u64*)values)[r];
            if ((b & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL) return 0x7ff8000000000000ULL;
            return b == 0x8000000000000000ULL ? 0ULL : b;
        }
    }
}

__global__ __launch_bounds__(256) void k_df_collect(int type, const void* __restrict__ values, const u8* __restrict__ nulls, i64 n, DfSet set,
                                                    bool with_set, i64* __restrict__ partials)
{
    __shared__ i64 s_min[4], s_max[4];
    __shared__ int s_any[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i64 mn = INT64_MAX, mx = INT64_MIN;
    bool any = false, saw_null = false, saw_empty = false;
    const i64 padded = (n + 255) & ~(i64)255;  // whole waves make every round: the claims are counted per wave
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < padded; r += (i64)gridDim.x * 256) {
        const bool in = r < n;
        const bool is_null = in && nulls && nulls[r];
        saw_null |= is_null;
        const bool active = in && !is_null;
        const u64 key = active ? df_key(type, values, r) : 0ULL;
        if (active) {
            any = true;
            mn = min(mn, (i64)key);
            mx = max(mx, (i64)key);
        }
        if (!with_set) continue;
        bool searching = active;
        if (searching && key == kDfEmpty) {
            saw_empty = true;
            searching = false;
        }
        u32 i = (u32)pa_murmur3_fmix(key) & set.cap_mask;
        while (__ballot(searching) != 0ULL) {
            bool claimed = false;
            if (searching) {
                u64 cur = __hip_atomic_load(&set.keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == kDfEmpty) {
                    // over the limit: the set is given up after this page, no need to fill it further
                    if (__hip_atomic_load(&set.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > set.limit) searching = false;
                    else {
                        cur = atomicCAS((unsigned long long*)&set.keys[i], (unsigned long long)kDfEmpty, (unsigned long long)key);
                        claimed = cur == kDfEmpty;
                    }
                }
                if (searching) {
                    if (claimed || cur == key) searching = false;
                    else i = (i + 1) & set.cap_mask;
                }
            }
            const u64 claims = __ballot(claimed);
            if (claims != 0ULL && lane == __ffsll((long long)claims) - 1) atomicAdd(&set.counters[0], (u32)__popcll(claims));
        }
    }
    if (with_set) {
        if (__ballot(saw_null) != 0ULL && lane == 0) set.counters[1] = 1u;
        if (__ballot(saw_empty) != 0ULL && lane == 0) set.counters[2] = 1u;
    }
    if (partials) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn = min(mn, (i64)__shfl_xor((long long)mn, d, 64));
            mx = max(mx, (i64)__shfl_xor((long long)mx, d, 64));
        }
        const bool wave_any = __ballot(any) != 0ULL;
        if (lane == 0) {
            s_min[wave] = mn;
            s_max[wave] = mx;
            s_any[wave] = wave_any ? 1 : 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; w++) {
                s_min[0] = min(s_min[0], s_min[w]);
                s_max[0] = max(s_max[0], s_max[w]);
                s_any[0] |= s_any[w];
            }
            partials[3 * blockIdx.x + 0] = s_min[0];
            partials[3 * blockIdx.x + 1] = s_max[0];
            partials[3 * blockIdx.x + 2] = s_any[0];
        }
    }
}

// the blocks' partial results into the running {min, max, any}: no atomics, the stream orders it behind the collect kernel
__global__ __launch_bounds__(64) void k_df_fold(const i64* __restrict__ partials, int blocks, i64* __restrict__ running)
{
    i64 mn = INT64_MAX, mx = INT64_MIN;
    bool any = false;
    for (int b = threadIdx.x; b < blocks; b += 64) {
        if (partials[3 * b + 2]) {
            any = true;
            mn = min(mn, partials[3 * b + 0]);
            mx = max(mx, partials[3 * b + 1]);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = min(mn, (i64)__shfl_xor((long long)mn, d, 64));
        mx = max(mx, (i64)__shfl_xor((long long)mx, d, 64));
    }
    const bool wave_any = __ballot(any) != 0ULL;
    if (threadIdx.x == 0 && wave_any) {
        if (running[2]) {
            mn = min(mn, running[0]);
            mx = max(mx, running[1]);
        }
        running[0] = mn;
        running[1] = mx;
        running[2] = 1;
    }
}

__global__ __launch_bounds__(256) void k_df_values(DfSet set, u64* __restrict__ out, u32* __restrict__ count_out)
{
    const int lane = threadIdx.x & 63;
    const i64 cap = (i64)set.cap_mask + 1;  // a power of two >= 4096: whole waves
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < cap; i += (i64)gridDim.x * 256) {
        const u64 k = set.keys[i];
        const bool used = k != kDfEmpty;
        const u64 m = __ballot(used);
        if (m == 0ULL) continue;
        u32 base = 0;
        if (lane == 0) base = atomicAdd(count_out, (u32)__popcll(m));
        base = (u32)__shfl((int)base, 0, 64);
        if (used) out[base + (u32)__popcll(m & ((1ULL << lane) - 1ULL))] = k;
    }
}

}  // namespace

size_t df_partials_bytes() { return (size_t)kDfBlocks * 3 * 8; }

void launch_df_collect(int32_t type, const void* values, const uint8_t* nulls, int64_t n, const DfSet* set, int64_t* partials, int64_t* running,
                       hipStream_t s)
{
    if (n <= 0) return;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, kDfBlocks));
    DfSet none{};
    hipLaunchKernelGGL(k_df_collect, grid, 256, 0, s, (int)type, values, nulls, (i64)n, set ? *set : none, set != nullptr, (i64*)partials);
    if (partials) hipLaunchKernelGGL(k_df_fold, 1, 64, 0, s, (const i64*)partials, grid, (i64*)running);
    PA_HIP(hipGetLastError());
}

void launch_df_values(const DfSet& set, uint64_t* out, uint32_t* count_out, hipStream_t s)
{
    PA_HIP(hipMemsetAsync(count_out, 0, 4, s));
    const int64_t cap = (int64_t)set.cap_mask + 1;
    hipLaunchKernelGGL(k_df_values, (int)std::min<int64_t>((cap + 255) / 256, 1024), 256, 0, s, set, (u64*)out, count_out);
    PA_HIP(hipGetLastError());
}

}  // namespace pa
