// topn_kernels.hip -- device side of TopNOperator (core/trino-main/src/main/java/io/trino/operator/TopNOperator.java,
// TopNProcessor.java, GroupedTopNBuilder.java): the reference keeps a heap of N rows and compares every input row with
// its root.  Here a page is reduced to the rows that can still be among the N best -- rows whose first sort key is not
// beyond the N-th best seen so far -- by an order-preserving 64-bit key, a radix selection of the N-th smallest key and
// a stable compaction; the exact multi-channel comparison of the few survivors is the host's (op_topn.cpp).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "kernels/pa_device.h"
#include "topn_kernels.hpp"

namespace pa {

namespace {

constexpr int kHistGrid = 512;

// SortOrder.java: ASC_NULLS_FIRST(0), ASC_NULLS_LAST(1), DESC_NULLS_FIRST(2), DESC_NULLS_LAST(3)
__device__ __forceinline__ u64 order_key(u64 ascending_image, bool is_null, int sort_order)
{
    const bool descending = sort_order >= 2, nulls_first = (sort_order & 1) == 0;
    if (is_null) return nulls_first ? 0ULL : ~0ULL;
    return descending ? ~ascending_image : ascending_image;
}

__global__ __launch_bounds__(256) void k_topn_keys(i32 type, const void* __restrict__ values, const i32* __restrict__ offsets,
                                                   const u8* __restrict__ nulls, i64 n, i32 sort_order, u64* __restrict__ keys)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const bool is_null = nulls && nulls[i];
        u64 img = 0;
        if (!is_null) {
            switch (type) {
                case PA_BIGINT: img = (u64)((const i64*)values)[i] ^ 0x8000000000000000ULL; break;
                case PA_INTEGER:
                case PA_DATE: img = (u64)(i64)((const i32*)values)[i] ^ 0x8000000000000000ULL; break;
                case PA_BOOLEAN: img = ((const u8*)values)[i] ? 1ULL : 0ULL; break;
                case PA_DOUBLE: {
                    // Double.compare order (DoubleType.compareTo): -0.0 < 0.0, NaN above everything, one NaN
                    double d = ((const double*)values)[i];
                    u64 b = d != d ? 0x7ff8000000000000ULL : (u64)__double_as_longlong(d);
                    img = (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
                    break;
                }
                case PA_REAL: {
                    // Float.compare order (RealType.comparisonOperator) = Double.compare order of the widened values
                    const double d = (double)((const float*)values)[i];
                    u64 b = d != d ? 0x7ff8000000000000ULL : (u64)__double_as_longlong(d);
                    img = (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
                    break;
                }
                case PA_VARCHAR: {
                    // Slice.compareTo = unsigned bytes, shorter first: the first 8 bytes big-endian are a monotone image
                    const i32 o = offsets[i], len = offsets[i + 1] - o;
                    const u8* p = (const u8*)values + o;
                    for (int b = 0; b < 8; b++) img = (img << 8) | (b < len ? (u64)p[b] : 0ULL);
                    break;
                }
                default: break;
            }
        }
        keys[i] = order_key(img, is_null, sort_order);
    }
}

// digit histogram of the keys that share the already selected prefix; one 256-bin row per workgroup
__global__ __launch_bounds__(256) void k_topn_hist(const u64* __restrict__ keys, i64 n, u64 prefix, int shift, int first, u32* __restrict__ slab)
{
    __shared__ u32 hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const u64 k = keys[i];
        if (first || (k >> (shift + 8)) == prefix) atomicAdd(&hist[(k >> shift) & 255ULL], 1u);
    }
    __syncthreads();
    slab[(u64)blockIdx.x * 256 + threadIdx.x] = hist[threadIdx.x];
}

// (a single 256-thread workgroup walking all rows one after the other took 125 us per pass -- 1 ms of Q3's 28 ms step)
__global__ __launch_bounds__(1024) void k_topn_hist_reduce(const u32* __restrict__ slab, int rows, u32* __restrict__ out)
{
    __shared__ u32 part[4][256];
    const int bin = threadIdx.x & 255, q = threadIdx.x >> 8;
    u32 sum = 0;
#pragma unroll 8
    for (int r = q; r < rows; r += 4) sum += slab[(u64)r * 256 + bin];
    part[q][bin] = sum;
    __syncthreads();
    if (q == 0) out[bin] = part[0][bin] + part[1][bin] + part[2][bin] + part[3][bin];
}

// the keys that carry the selected prefix, compacted (order irrelevant: the later passes only histogram them).  A workgroup
// owns one contiguous chunk: it counts its matches, reserves their room with ONE atomic (many atomics on one counter
// retire at ~0.15 M/s on this part: a per-wave reservation made this kernel 9 ms for 64 M keys), then walks the chunk again.
__global__ __launch_bounds__(256) void k_topn_compact(const u64* __restrict__ keys, i64 n, u64 prefix, int shift, u64* __restrict__ out,
                                                      u32* __restrict__ counter)
{
    __shared__ u32 wave_count[4];
    __shared__ u32 block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 per = ((n + gridDim.x - 1) / gridDim.x + 255) & ~(i64)255;
    const i64 c0 = (i64)blockIdx.x * per, c1 = c0 + per < n ? c0 + per : n;
    u32 mine = 0;
    for (i64 i = c0 + threadIdx.x; i < c1; i += 256) mine += ((keys[i] >> shift) == prefix) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += (u32)__shfl_xor((int)mine, d, 64);
    if (lane == 0) wave_count[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        block_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    // second walk: a wave's matches go behind the matches of the earlier waves' whole chunks shares -- order is free, so every wave
    // simply takes its own counted share: base + sum of the earlier waves' counts + running offset inside the wave
    u32 at = block_base;
    for (int w = 0; w < wave; w++) at += wave_count[w];
    for (i64 i0 = c0 + (i64)wave * 64; i0 < c1; i0 += 256) {
        const i64 i = i0 + lane;
        const u64 k = i < c1 ? keys[i] : 0ULL;
        const bool keep = i < c1 && (k >> shift) == prefix;
        const u64 m = __ballot(keep);
        if (keep) out[at + (u32)__popcll(m & ((1ULL << lane) - 1ULL))] = k;
        at += (u32)__popcll(m);
    }
}

// The remaining digits of the selection over a few candidates (compacted: at most kSelectSmall keys) in ONE workgroup: digit by
// digit histogram in LDS, the digit that holds the wanted rank, next digit -- no launch, read-back and host step per digit (six such
// rounds cost 0.19 ms of Q3's TopN, most of it launch gaps).  out[0] = the selected key.
constexpr int64_t kSelectSmall = 1 << 18;
__global__ __launch_bounds__(1024) void k_topn_select_small(const u64* __restrict__ keys, i64 n, u64 prefix, int shift, i64 remaining, u64* __restrict__ out)
{
    __shared__ u32 hist[256];
    __shared__ u64 s_prefix;
    __shared__ i64 s_remaining;
    if (threadIdx.x == 0) {
        s_prefix = prefix;
        s_remaining = remaining;
    }
    for (; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const u64 pf = s_prefix;
        for (i64 i = threadIdx.x; i < n; i += 1024) {
            const u64 k = keys[i];
            if (shift == 56 || (k >> (shift + 8)) == pf) atomicAdd(&hist[(k >> shift) & 255ULL], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int digit = 255;
            i64 before = 0;
            for (int d = 0; d < 256; d++) {
                if (before + (i64)hist[d] >= s_remaining) {
                    digit = d;
                    break;
                }
                before += (i64)hist[d];
            }
            s_remaining -= before;
            s_prefix = (pf << 8) | (u64)digit;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s_prefix;
}

__global__ __launch_bounds__(256) void k_topn_flag(const u64* __restrict__ keys, i64 n, u64 threshold, i32* __restrict__ partition)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) partition[i] = keys[i] <= threshold ? 0 : 1;
}

// ---- ties on the first sort key (see op_topn.cpp): per-row state 0 = among the best for sure, 1 = tied with the bound on every
// channel looked at so far, 2 = out ----
__global__ __launch_bounds__(256) void k_topn_state(const u64* __restrict__ keys, i64 n, u64 threshold, int first, u8* __restrict__ state)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        if (!first && state[i] != 1) continue;
        const u64 k = keys[i];
        state[i] = k < threshold ? 0 : (k == threshold ? 1 : 2);
    }
}
__global__ __launch_bounds__(256) void k_topn_mask_keys(const u8* __restrict__ state, i64 n, u64* __restrict__ keys)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        if (state[i] != 1) keys[i] = ~0ULL;
    }
}
// out[0] = rows in state 0, out[1] = rows in state 1
__global__ __launch_bounds__(256) void k_topn_count_states(const u8* __restrict__ state, i64 n, unsigned long long* __restrict__ out)
{
    i64 c0 = 0, c1 = 0;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const u8 v = state[i];
        c0 += v == 0;
        c1 += v == 1;
    }
    c0 = pa_wave_sum_i64(c0);
    c1 = pa_wave_sum_i64(c1);
    if ((threadIdx.x & 63) == 0) {
        if (c0) atomicAdd(out, (unsigned long long)c0);
        if (c1) atomicAdd(out + 1, (unsigned long long)c1);
    }
}
__global__ __launch_bounds__(256) void k_topn_tie_flags(const u8* __restrict__ state, i64 n, i32* __restrict__ flags)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) flags[i] = state[i] == 1 ? 1 : 0;
}
// partition[i] = 0 (kept) for state 0 and for the first `ties_kept` rows in state 1 (tie_rank = exclusive scan of the tie flags)
__global__ __launch_bounds__(256) void k_topn_state_partition(const u8* __restrict__ state, const i32* __restrict__ tie_rank, i64 n, i64 ties_kept,
                                                              i32* __restrict__ partition)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const u8 v = state[i];
        partition[i] = (v == 0 || (v == 1 && (i64)tie_rank[i] < ties_kept)) ? 0 : 1;
    }
}

int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048)); }

}  // namespace

void launch_topn_state(const uint64_t* keys, int64_t n, uint64_t threshold, bool first, uint8_t* state, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_topn_state, grid_of(n), 256, 0, s, (const u64*)keys, (i64)n, (u64)threshold, first ? 1 : 0, state);
    PA_HIP(hipGetLastError());
}
void launch_topn_mask_keys(const uint8_t* state, int64_t n, uint64_t* keys, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_topn_mask_keys, grid_of(n), 256, 0, s, state, (i64)n, (u64*)keys);
    PA_HIP(hipGetLastError());
}
void launch_topn_count_states(const uint8_t* state, int64_t n, int64_t* out2, hipStream_t s)
{
    PA_HIP(hipMemsetAsync(out2, 0, 16, s));
    if (n <= 0) return;
    hipLaunchKernelGGL(k_topn_count_states, grid_of(n), 256, 0, s, state, (i64)n, reinterpret_cast<unsigned long long*>(out2));
    PA_HIP(hipGetLastError());
}
void launch_topn_tie_flags(const uint8_t* state, int64_t n, int32_t* flags, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_topn_tie_flags, grid_of(n), 256, 0, s, state, (i64)n, flags);
    PA_HIP(hipGetLastError());
}
void launch_topn_state_partition(const uint8_t* state, const int32_t* tie_rank, int64_t n, int64_t ties_kept, int32_t* partition, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_topn_state_partition, grid_of(n), 256, 0, s, state, tie_rank, (i64)n, (i64)ties_kept, partition);
    PA_HIP(hipGetLastError());
}

void launch_topn_keys(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order,
                      uint64_t* keys, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_topn_keys, grid_of(n), 256, 0, s, type, values, (const i32*)offsets, (const u8*)nulls, (i64)n, sort_order, (u64*)keys);
    PA_HIP(hipGetLastError());
}

size_t topn_select_temp_bytes() { return (size_t)(kHistGrid + 1) * 256 * 4; }

uint64_t topn_select_kth(const uint64_t* keys, int64_t n, int64_t k, void* temp, uint32_t* host_hist, hipStream_t s)
{
    PA_REQUIRE(k >= 1 && k <= n, PA_ERR_INVALID_ARGUMENT, "selection rank out of range");
    u32* slab = static_cast<u32*>(temp);
    u32* total = slab + (size_t)kHistGrid * 256;
    // MSB radix selection.  After a pass only the keys with the selected digit matter: once they are few they are compacted,
    // and the remaining passes histogram thousands of keys instead of walking all n again (64 M DOUBLE keys: 8 full passes
    // became one full pass, one compaction and seven trivial ones).
    DevBuf cand[2], counter_buf;
    u32* counter = static_cast<u32*>(counter_buf.ensure(64));
    const u64* cur = reinterpret_cast<const u64*>(keys);
    int64_t cur_n = n;
    int which = 0;
    uint64_t prefix = 0;
    int64_t remaining = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (cur_n <= kSelectSmall) {  // few candidates (a compaction's worth, or a small page): the remaining digits in one launch
            u64* result = reinterpret_cast<u64*>(counter + 2);
            hipLaunchKernelGGL(k_topn_select_small, 1, 1024, 0, s, cur, (i64)cur_n, (u64)prefix, shift, (i64)remaining, result);
            PA_HIP(hipGetLastError());
            PA_HIP(hipMemcpyAsync(host_hist, result, 8, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            memcpy(&prefix, host_hist, 8);
            break;
        }
        const int grid = (int)std::min<int64_t>(kHistGrid, std::max<int64_t>(1, (cur_n + 255) / 256));
        hipLaunchKernelGGL(k_topn_hist, grid, 256, 0, s, cur, (i64)cur_n, (u64)prefix, shift, shift == 56 ? 1 : 0, slab);  // (compacted candidates only share the prefix of their compaction: keep checking)
        hipLaunchKernelGGL(k_topn_hist_reduce, 1, 1024, 0, s, (const u32*)slab, grid, total);
        PA_HIP(hipGetLastError());
        PA_HIP(hipMemcpyAsync(host_hist, total, 256 * 4, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        int digit = 255;
        int64_t before = 0;
        for (int d = 0; d < 256; d++) {
            if (before + (int64_t)host_hist[d] >= remaining) {
                digit = d;
                break;
            }
            before += host_hist[d];
        }
        remaining -= before;
        prefix = (prefix << 8) | (uint64_t)digit;
        const int64_t next_n = (int64_t)host_hist[digit];
        if (shift > 0 && next_n * 4 <= cur_n && cur_n > 4096) {
            u64* out = static_cast<u64*>(cand[which].ensure((size_t)std::max<int64_t>(next_n, 1) * 8));
            PA_HIP(hipMemsetAsync(counter, 0, 4, s));
            hipLaunchKernelGGL(k_topn_compact, (int)std::min<int64_t>(1024, std::max<int64_t>(1, (cur_n + 4095) / 4096)), 256, 0, s, cur, (i64)cur_n, (u64)prefix, shift, out, counter);
            PA_HIP(hipGetLastError());
            cur = out;
            cur_n = next_n;
            which ^= 1;
        }
    }
    PA_HIP(hipStreamSynchronize(s));  // the candidate buffers go back to the pool
    return prefix;
}

void launch_topn_flag(const uint64_t* keys, int64_t n, uint64_t threshold, int32_t* partition, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_topn_flag, grid_of(n), 256, 0, s, (const u64*)keys, (i64)n, (u64)threshold, partition);
    PA_HIP(hipGetLastError());
}

// ---- OrderByOperator (op_order_by.cpp): digits of a radix pass, taken through the current permutation --------------------
namespace {
__global__ __launch_bounds__(256) void k_sort_digits(const u64* __restrict__ keys, const i32* __restrict__ perm, i64 n, int shift, u32 mask,
                                                     i32* __restrict__ digits)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) digits[i] = (i32)((keys[perm[i]] >> shift) & mask);
}
__global__ __launch_bounds__(256) void k_sort_null_digits(const u8* __restrict__ nulls, const i32* __restrict__ perm, i64 n, int nulls_first,
                                                          i32* __restrict__ digits)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const int is_null = nulls[perm[i]] ? 1 : 0;
        digits[i] = nulls_first ? 1 - is_null : is_null;
    }
}
// chunk >= 0: bytes [8 chunk, 8 chunk + 8) big-endian, zero padded; chunk < 0: the length.  NULL rows get image 0 (their
// place is decided by the NULL digit); descending = complement.
__global__ __launch_bounds__(256) void k_varchar_chunk_keys(const u8* __restrict__ values, const i32* __restrict__ offsets, const u8* __restrict__ nulls,
                                                            i64 n, int chunk, int descending, u64* __restrict__ keys)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        u64 img = 0;
        if (!(nulls && nulls[i])) {
            const i32 o = offsets[i], len = offsets[i + 1] - o;
            if (chunk < 0) img = (u64)(u32)len;
            else {
                const u8* p = values + o + 8 * chunk;
                const i32 left = len - 8 * chunk;
                for (int b = 0; b < 8; b++) img = (img << 8) | (b < left ? (u64)p[b] : 0ULL);
            }
            if (descending) img = ~img;
        }
        keys[i] = img;
    }
}
__global__ __launch_bounds__(256) void k_iota_i32(i32* dst, i64 n)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) dst[i] = (i32)i;
}
__global__ __launch_bounds__(256) void k_varchar_max_length(const i32* __restrict__ offsets, i64 n, i32* out)
{
    i32 m = 0;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) m = max(m, offsets[i + 1] - offsets[i]);
    for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}
}  // namespace

void launch_sort_digits(const uint64_t* keys, const int32_t* perm, int64_t n, int shift, int bits, int32_t* digits, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sort_digits, grid_of(n), 256, 0, s, (const u64*)keys, perm, (i64)n, shift, (u32)((1u << bits) - 1u), digits);
    PA_HIP(hipGetLastError());
}
void launch_sort_null_digits(const uint8_t* nulls, const int32_t* perm, int64_t n, int nulls_first, int32_t* digits, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sort_null_digits, grid_of(n), 256, 0, s, nulls, perm, (i64)n, nulls_first, digits);
    PA_HIP(hipGetLastError());
}
void launch_varchar_chunk_keys(const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int chunk, int descending,
                               uint64_t* keys, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_varchar_chunk_keys, grid_of(n), 256, 0, s, (const u8*)values, offsets, nulls, (i64)n, chunk, descending, (u64*)keys);
    PA_HIP(hipGetLastError());
}
void launch_iota_i32(int32_t* dst, int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_iota_i32, grid_of(n), 256, 0, s, dst, (i64)n);
    PA_HIP(hipGetLastError());
}
int32_t varchar_max_length(const int32_t* offsets, int64_t n, void* temp_dev_8, hipStream_t s)
{
    int32_t* out = static_cast<int32_t*>(temp_dev_8);
    PA_HIP(hipMemsetAsync(out, 0, 4, s));
    if (n > 0) hipLaunchKernelGGL(k_varchar_max_length, grid_of(n), 256, 0, s, offsets, (i64)n, out);
    int32_t h = 0;
    PA_HIP(hipMemcpyAsync(&h, out, 4, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    return h;
}

}  // namespace pa
