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

// order-preserving 64-bit image of row i of a sort channel (see topn_kernels.hpp)
__device__ __forceinline__ u64 row_key(i32 type, const void* __restrict__ values, const i32* __restrict__ offsets, const u8* __restrict__ nulls, i64 i,
                                       i32 sort_order)
{
    if (type < 0) return ((const u64*)values)[i];  // PA_TOPN_KEYS: the column already holds keys
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
    return order_key(img, is_null, sort_order);
}

__global__ __launch_bounds__(256) void k_topn_keys(i32 type, const void* __restrict__ values, const i32* __restrict__ offsets,
                                                   const u8* __restrict__ nulls, i64 n, i32 sort_order, u64* __restrict__ keys)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) keys[i] = row_key(type, values, offsets, nulls, i, sort_order);
}

// the same, and OR / AND of the keys per workgroup (out[2 b], out[2 b + 1]; at most kKeysOrAndBlocks workgroups): OrderBy sorts by the
// bits in which the images differ, and finds them out without another pass over the images
constexpr int kKeysOrAndBlocks = 1024;
__global__ __launch_bounds__(256) void k_topn_keys_or_and(i32 type, const void* __restrict__ values, const i32* __restrict__ offsets,
                                                          const u8* __restrict__ nulls, i64 n, i32 sort_order, u64* __restrict__ keys, u64* __restrict__ out)
{
    __shared__ u64 s_o[4], s_a[4];
    u64 o = 0ULL, a = ~0ULL;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const u64 k = row_key(type, values, offsets, nulls, i, sort_order);
        keys[i] = k;
        o |= k;
        a &= k;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        o |= (u64)__shfl_xor((long long)o, d, 64);
        a &= (u64)__shfl_xor((long long)a, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_o[threadIdx.x >> 6] = o;
        s_a[threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = s_o[0] | s_o[1] | s_o[2] | s_o[3];
        out[2 * blockIdx.x + 1] = s_a[0] & s_a[1] & s_a[2] & s_a[3];
    }
}

// keys of every stride-th row (the sample the bound of a page is drawn from)
__global__ __launch_bounds__(256) void k_topn_sample_keys(i32 type, const void* __restrict__ values, const i32* __restrict__ offsets,
                                                          const u8* __restrict__ nulls, i64 stride, i64 count, i32 sort_order, u64* __restrict__ keys)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < count; i += (i64)gridDim.x * 256) {
        keys[i] = row_key(type, values, offsets, nulls, i * stride, sort_order);
    }
}

// The rows whose key is not beyond the bound -- min(bound, *device_bound) -- in ANY order: positions and keys of the first
// `capacity` of them, and how many there are in all (counter[0]; the host sees an overflow there).  One pass over the sort
// channel itself: no key array.  A wave reserves room for its matches with one atomic; matches are few by construction.
__global__ __launch_bounds__(256) void k_topn_filter(i32 type, const void* __restrict__ values, const i32* __restrict__ offsets, const u8* __restrict__ nulls,
                                                     i64 n, i32 sort_order, u64 bound, const u64* __restrict__ device_bound, u32 capacity,
                                                     i32* __restrict__ out_positions, u64* __restrict__ out_keys, u32* __restrict__ counter)
{
    const u64 limit = device_bound && *device_bound < bound ? *device_bound : bound;
    const u32 lane = threadIdx.x & 63u;
    const i64 step = (i64)gridDim.x * 256;
    const i64 rounds = (n + step - 1) / step;  // wave-uniform trip count (the ballots below want whole waves)
    i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    for (i64 r = 0; r < rounds; r += 2, i += 2 * step) {
        // two rows per lane and step: their loads are in flight together
        const i64 j = i + step;
        const u64 k0 = i < n ? row_key(type, values, offsets, nulls, i, sort_order) : ~0ULL;
        const u64 k1 = j < n ? row_key(type, values, offsets, nulls, j, sort_order) : ~0ULL;
        const bool keep0 = i < n && k0 <= limit, keep1 = j < n && k1 <= limit;
        const u64 m0 = __ballot(keep0), m1 = __ballot(keep1);
        if ((m0 | m1) == 0ULL) continue;
        const u32 c0 = (u32)__popcll(m0), total = c0 + (u32)__popcll(m1);
        u32 base = 0;
        if (lane == 0) base = atomicAdd(counter, total);
        base = (u32)__shfl((int)base, 0, 64);
        if (keep0) {
            const u32 at = base + (u32)__popcll(m0 & ((1ULL << lane) - 1ULL));
            if (at < capacity) {
                out_positions[at] = (i32)i;
                out_keys[at] = k0;
            }
        }
        if (keep1) {
            const u32 at = base + c0 + (u32)__popcll(m1 & ((1ULL << lane) - 1ULL));
            if (at < capacity) {
                out_positions[at] = (i32)j;
                out_keys[at] = k1;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counter[1] = 0u;  // (pads the read-back to 8 bytes)
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
    __shared__ u32 scan[256];
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
        // (the leading bytes of a page's keys are mostly the same: when all lanes of a wave hold one digit they add once, together
        // -- 64 atomics on one LDS address otherwise take their turns)
        const i64 rounds = (n + 1023) / 1024;
        for (i64 r = 0; r < rounds; r++) {
            const i64 i = r * 1024 + threadIdx.x;
            const u64 k = i < n ? keys[i] : 0ULL;
            const bool in = i < n && (shift == 56 || (k >> (shift + 8)) == pf);
            const u32 digit = (u32)((k >> shift) & 255ULL);
            const u64 members = __ballot(in);
            if (members == 0ULL) continue;
            const u32 first = (u32)__shfl((int)digit, __ffsll((long long)members) - 1, 64);
            if (__ballot(in && digit == first) == members) {
                if ((threadIdx.x & 63) == (u32)(__ffsll((long long)members) - 1)) atomicAdd(&hist[first], (u32)__popcll(members));
            }
            else if (in) atomicAdd(&hist[digit], 1u);
        }
        __syncthreads();
        // the digit that holds the wanted rank: inclusive prefix sums of the 256 bins (one bin per thread of the first four waves), then
        // the one bin whose sums straddle the rank speaks up (a single thread walking the bins took 25 us per digit)
        if (threadIdx.x < 256) scan[threadIdx.x] = hist[threadIdx.x];
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            u32 add = 0;
            if (threadIdx.x < 256 && (int)threadIdx.x >= d) add = scan[threadIdx.x - d];
            __syncthreads();
            if (threadIdx.x < 256) scan[threadIdx.x] += add;
            __syncthreads();
        }
        const i64 want = s_remaining;
        __syncthreads();
        if (threadIdx.x < 256) {
            const i64 incl = (i64)scan[threadIdx.x], before = incl - (i64)hist[threadIdx.x];
            // (the last bin answers when the rank lies beyond all keys -- it cannot: the callers pass ranks within the key count)
            if ((before < want && want <= incl) || (threadIdx.x == 255 && want > incl)) {
                s_remaining = want - before;
                s_prefix = (pf << 8) | (u64)threadIdx.x;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s_prefix;
}

// The sample of a page and the selection over it in ONE workgroup: a thread holds its 16 of the (at most) 16 384 sample keys in
// registers -- every evenly spaced row's key, computed from the column itself -- and the eight digit rounds run over registers and
// an LDS histogram (walking a 2^16-key sample array in global memory eight times took 132 us of Q3's TopN).  out[0] = the rank-th
// smallest sample key.
constexpr int kSampleRegs = 16;
__global__ __launch_bounds__(1024) void k_topn_sample_select(i32 type, const void* __restrict__ values, const i32* __restrict__ offsets,
                                                             const u8* __restrict__ nulls, i64 stride, i64 count, i32 sort_order, i64 rank, u64* __restrict__ out)
{
    __shared__ u32 hist[256];
    __shared__ u32 scan[256];
    __shared__ u64 s_prefix;
    __shared__ i64 s_remaining;
    u64 mine[kSampleRegs];
    u32 have = 0;
#pragma unroll
    for (int j = 0; j < kSampleRegs; j++) {
        const i64 i = (i64)j * 1024 + threadIdx.x;
        mine[j] = ~0ULL;
        if (i < count) {
            mine[j] = row_key(type, values, offsets, nulls, i * stride, sort_order);
            have |= 1u << j;
        }
    }
    if (threadIdx.x == 0) {
        s_prefix = 0;
        s_remaining = rank;
    }
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const u64 pf = s_prefix;
#pragma unroll
        for (int j = 0; j < kSampleRegs; j++) {
            const u64 k = mine[j];
            const bool in = ((have >> j) & 1u) && (shift == 56 || (k >> (shift + 8)) == pf);
            const u32 digit = (u32)((k >> shift) & 255ULL);
            const u64 members = __ballot(in);
            if (members == 0ULL) continue;
            const int leader = __ffsll((long long)members) - 1;
            const u32 first = (u32)__shfl((int)digit, leader, 64);
            if (__ballot(in && digit == first) == members) {
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[first], (u32)__popcll(members));
            }
            else if (in) atomicAdd(&hist[digit], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 256) scan[threadIdx.x] = hist[threadIdx.x];
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            u32 add = 0;
            if (threadIdx.x < 256 && (int)threadIdx.x >= d) add = scan[threadIdx.x - d];
            __syncthreads();
            if (threadIdx.x < 256) scan[threadIdx.x] += add;
            __syncthreads();
        }
        const i64 want = s_remaining;
        __syncthreads();
        if (threadIdx.x < 256) {
            const i64 incl = (i64)scan[threadIdx.x], before = incl - (i64)hist[threadIdx.x];
            if ((before < want && want <= incl) || (threadIdx.x == 255 && want > incl)) {
                s_remaining = want - before;
                s_prefix = (pf << 8) | (u64)threadIdx.x;
            }
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

int launch_topn_keys_or_and(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order, uint64_t* keys,
                            uint64_t* or_and, hipStream_t s)
{
    if (n <= 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, kKeysOrAndBlocks));
    hipLaunchKernelGGL(k_topn_keys_or_and, blocks, 256, 0, s, type, values, (const i32*)offsets, (const u8*)nulls, (i64)n, sort_order, (u64*)keys, (u64*)or_and);
    PA_HIP(hipGetLastError());
    return blocks;
}

// the values back from their keys (BIGINT / INTEGER / DATE channels without NULL rows: the image is the value with its sign bit flipped,
// complemented for descending orders)
__global__ __launch_bounds__(256) void k_topn_values_of_keys(i32 type, const u64* __restrict__ keys, i64 n, int descending, void* __restrict__ values)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const u64 img = descending ? ~keys[i] : keys[i];
        const i64 v = (i64)(img ^ 0x8000000000000000ULL);
        if (type == PA_BIGINT) ((i64*)values)[i] = v;
        else ((i32*)values)[i] = (i32)v;
    }
}
void launch_topn_values_of_keys(int32_t type, const uint64_t* keys, int64_t n, bool descending, void* values, hipStream_t s)
{
    if (n <= 0) return;
    PA_REQUIRE(type == PA_BIGINT || type == PA_INTEGER || type == PA_DATE, PA_ERR_DEVICE, "internal: keys of this type do not give the values back");
    hipLaunchKernelGGL(k_topn_values_of_keys, grid_of(n), 256, 0, s, type, (const u64*)keys, (i64)n, descending ? 1 : 0, values);
    PA_HIP(hipGetLastError());
}

void launch_topn_sample_bound(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order,
                              int64_t sample_rows, int64_t rank, uint64_t* sample_keys, uint64_t* bound_out, hipStream_t s)
{
    PA_REQUIRE(sample_rows >= 1 && sample_rows <= kSelectSmall && rank >= 1 && rank <= sample_rows && n >= sample_rows, PA_ERR_INVALID_ARGUMENT,
               "bad TopN sample");
    const int64_t stride = n / sample_rows;
    if (sample_rows <= (int64_t)kSampleRegs * 1024) {  // sample and selection in one workgroup, the keys in registers
        hipLaunchKernelGGL(k_topn_sample_select, 1, 1024, 0, s, type, values, (const i32*)offsets, (const u8*)nulls, (i64)stride, (i64)sample_rows, sort_order,
                           (i64)rank, (u64*)bound_out);
        PA_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(k_topn_sample_keys, grid_of(sample_rows), 256, 0, s, type, values, (const i32*)offsets, (const u8*)nulls, (i64)stride,
                       (i64)sample_rows, sort_order, (u64*)sample_keys);
    hipLaunchKernelGGL(k_topn_select_small, 1, 1024, 0, s, (const u64*)sample_keys, (i64)sample_rows, (u64)0, 56, (i64)rank, (u64*)bound_out);
    PA_HIP(hipGetLastError());
}

void launch_topn_filter(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order, uint64_t bound,
                        const uint64_t* device_bound, uint32_t capacity, int32_t* out_positions, uint64_t* out_keys, uint32_t* counter, hipStream_t s)
{
    PA_HIP(hipMemsetAsync(counter, 0, 8, s));
    if (n <= 0) return;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 511) / 512, 4096));
    hipLaunchKernelGGL(k_topn_filter, grid, 256, 0, s, type, values, (const i32*)offsets, (const u8*)nulls, (i64)n, sort_order, (u64)bound,
                       (const u64*)device_bound, capacity, out_positions, (u64*)out_keys, counter);
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
    // (one atomic per workgroup of at most 512: same-address atomics retire one after the other at the memory side)
    __shared__ i32 s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        if (m > 0) atomicMax(out, m);
    }
}
}  // namespace

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
    if (n > 0) hipLaunchKernelGGL(k_varchar_max_length, std::min(grid_of(n), 512), 256, 0, s, offsets, (i64)n, out);
    int32_t h = 0;
    PA_HIP(hipMemcpyAsync(&h, out, 4, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    return h;
}

}  // namespace pa
