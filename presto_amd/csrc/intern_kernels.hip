// intern_kernels.hip -- see intern_kernels.hpp.  MultiChannelGroupByHash compares VARCHAR keys of any length
// (core/trino-main/src/main/java/io/trino/operator/MultiChannelGroupByHash.java:441-452, positionNotDistinctFromCurrentRow);
// the packed keys of the device aggregation hold 15 bytes.  Longer strings are interned first: string -> dense id, exact.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "intern_kernels.hpp"
#include "kernels/pa_device.h"

namespace pa {

namespace {

// 8-byte word w of a string (zero padded behind its end): the unit strings are stored and compared in
__device__ __forceinline__ u64 string_word(const u8* p, i32 len, i32 w)
{
    const i32 base = 8 * w;
    if (base + 8 <= len) return pa_rd64(p + base);  // one unaligned load
    u64 v = 0;
    for (int b = 0; base + b < len; b++) v |= (u64)p[base + b] << (8 * b);
    return v;
}

__global__ __launch_bounds__(256) void k_intern(InternTable t, const u8* __restrict__ values, const i32* __restrict__ offsets,
                                                const u8* __restrict__ nulls, i64 n, i32* __restrict__ ids_out)
{
    const int lane = threadIdx.x & 63;
    const i64 padded = (n + 255) & ~(i64)255;  // every lane of a wave makes the same number of rounds (ballots below)
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < padded; r += (i64)gridDim.x * 256) {
        const bool active = r < n && !(nulls && nulls[r]);
        if (r < n && !active) ids_out[r] = 0;  // NULL key: the id is not looked at
        const i32 o = active ? offsets[r] : 0, len = active ? offsets[r + 1] - o : 0;
        const u8* p = values + o;
        const i32 words = (len + 7) >> 3;
        const u64 full_hash = active ? pa_xxh64(p, len) : 0ULL;
        const u64 h = full_hash & 0x3fffffffffffffffULL;
        const u64 busy = (h << 2) | 1ULL, ready = (h << 2) | 3ULL;
        u32 i = (u32)(h ^ (h >> 32)) & t.cap_mask;
        int result = active ? -2 : -1;  // -2 searching, >= 0 id
        int spins = 0;
        while (__ballot(result == -2) != 0ULL) {
            const bool searching = result == -2;
            u64 tg = 1ULL;
            if (searching) tg = __hip_atomic_load((u64*)&t.tag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool claimed = false;
            if (searching && tg == 0ULL) {
                claimed = atomicCAS((unsigned long long*)&t.tag[i], 0ULL, (unsigned long long)busy) == 0ULL;
            }
            // ids and arena space for the claims of this round: one atomic each per wave (prefix sums over the claiming lanes)
            const u64 claims = __ballot(claimed);
            if (claims != 0ULL) {
                u32 need = claimed ? (u32)words : 0u, incl = need;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const u32 up = (u32)__shfl_up((int)incl, d, 64);
                    if (lane >= d) incl += up;
                }
                const int leader = __ffsll((long long)claims) - 1, last = 63 - __clzll((long long)claims);
                const u32 total_words = (u32)__shfl((int)incl, last, 64);  // inclusive sum at the last claiming lane = all claims
                u32 id_base = 0, word_base = 0;
                if (lane == leader) {
                    id_base = atomicAdd(&t.counters[0], (u32)__popcll(claims));
                    word_base = atomicAdd(&t.counters[1], total_words);
                }
                id_base = (u32)__shfl((int)id_base, leader, 64);
                word_base = (u32)__shfl((int)word_base, leader, 64);
                if (claimed) {
                    const u32 id = id_base + (u32)__popcll(claims & ((1ULL << lane) - 1ULL));
                    const u32 at = word_base + incl - need;
                    for (i32 w = 0; w < words; w++) __hip_atomic_store((u64*)&t.arena[at + w], string_word(p, len, w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((u64*)&t.meta[i], ((u64)id << 32) | (u64)(u32)len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&t.off[i], at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t.id_off[id] = at;
                    t.id_len[id] = (u32)len;
                    t.id_hash[id] = full_hash;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store((u64*)&t.tag[i], ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    result = (int)id;
                }
            }
            if (searching && !claimed && tg != 0ULL) {
                if (tg == ready) {
                    const u64 m = __hip_atomic_load((u64*)&t.meta[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool eq = (u32)m == (u32)len;
                    if (eq) {
                        const u32 at = __hip_atomic_load(&t.off[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (i32 w = 0; w < words && eq; w++)
                            eq = __hip_atomic_load((u64*)&t.arena[at + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == string_word(p, len, w);
                    }
                    if (eq) result = (int)(m >> 32);
                    else i = (i + 1) & t.cap_mask;  // another string in this slot (same hash or not)
                }
                else if (tg == busy) {
                    if (++spins > (1 << 22)) result = 0;  // cannot happen: the publisher finishes inside its round
                }
                else i = (i + 1) & t.cap_mask;
            }
            // (tg == 0 and the CAS lost: look at the slot again in the next round)
        }
        if (active) ids_out[r] = result;
    }
}

__global__ __launch_bounds__(256) void k_intern_rehash(const u64* __restrict__ old_tag, const u64* __restrict__ old_meta, const u32* __restrict__ old_off,
                                                       u32 old_cap, InternTable t)
{
    // every string is in the old table exactly once: no comparison, just find an empty slot
    for (i64 s = (i64)blockIdx.x * 256 + threadIdx.x; s < (i64)old_cap; s += (i64)gridDim.x * 256) {
        const u64 tg = old_tag[s];
        if (tg == 0ULL) continue;
        const u64 h = tg >> 2;
        u32 i = (u32)(h ^ (h >> 32)) & t.cap_mask;
        for (;;) {
            if (atomicCAS((unsigned long long*)&t.tag[i], 0ULL, (unsigned long long)tg) == 0ULL) {
                t.meta[i] = old_meta[s];
                t.off[i] = old_off[s];
                break;
            }
            i = (i + 1) & t.cap_mask;
        }
    }
}

__global__ __launch_bounds__(256) void k_intern_lengths(InternTable t, const i32* __restrict__ ids, const u8* __restrict__ nulls, i64 n, i32* __restrict__ out)
{
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) out[r] = (nulls && nulls[r]) ? 0 : (i32)t.id_len[ids[r]];
}

__global__ __launch_bounds__(256) void k_intern_bytes(InternTable t, const i32* __restrict__ ids, const u8* __restrict__ nulls, i64 n,
                                                      const i32* __restrict__ out_offsets, u8* __restrict__ out_bytes)
{
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
        if (nulls && nulls[r]) continue;
        const u32 id = (u32)ids[r];
        const u8* src = (const u8*)(t.arena + t.id_off[id]);
        u8* dst = out_bytes + out_offsets[r];
        const i32 len = (i32)t.id_len[id];
        i32 b = 0;
        for (; b + 8 <= len; b += 8) {  // the arena is 8-byte aligned; the destination need not be
            const u64 w = *(const u64*)(src + b);
            __builtin_memcpy(dst + b, &w, 8);
        }
        for (; b < len; b++) dst[b] = src[b];
    }
}

int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048)); }

}  // namespace

void launch_intern(const InternTable& t, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t* ids_out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_intern, grid_of(n), 256, 0, s, t, (const u8*)values, offsets, nulls, (i64)n, ids_out);
    PA_HIP(hipGetLastError());
}
void launch_intern_rehash(const uint64_t* old_tag, const uint64_t* old_meta, const uint32_t* old_off, uint32_t old_cap, const InternTable& t, hipStream_t s)
{
    if (old_cap == 0) return;
    hipLaunchKernelGGL(k_intern_rehash, grid_of(old_cap), 256, 0, s, (const u64*)old_tag, (const u64*)old_meta, old_off, old_cap, t);
    PA_HIP(hipGetLastError());
}
void launch_intern_lengths(const InternTable& t, const int32_t* ids, const uint8_t* nulls, int64_t n, int32_t* out_lengths, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_intern_lengths, grid_of(n), 256, 0, s, t, ids, nulls, (i64)n, out_lengths);
    PA_HIP(hipGetLastError());
}
void launch_intern_bytes(const InternTable& t, const int32_t* ids, const uint8_t* nulls, int64_t n, const int32_t* out_offsets, uint8_t* out_bytes, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_intern_bytes, grid_of(n), 256, 0, s, t, ids, nulls, (i64)n, out_offsets, out_bytes);
    PA_HIP(hipGetLastError());
}

namespace {
__global__ __launch_bounds__(256) void k_rank_image(const i32* __restrict__ ids, const u8* __restrict__ nulls, const u32* __restrict__ ranks, i64 n,
                                                    i64* __restrict__ image)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        if (nulls && nulls[i]) {
            image[i] = 0;
            continue;
        }
        const u32 id = (u32)ids[i];
        image[i] = (i64)(((u64)(ranks[id] + 1u) << 32) | (u64)id);
    }
}
__global__ __launch_bounds__(256) void k_rerank_words(u64* __restrict__ words, i64 n, int is_min, const u32* __restrict__ ranks)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const u64 w = words[i];
        if (w == 0ULL) continue;
        const u64 img = is_min ? ~w : w;
        const u32 id = (u32)(img ^ 0x8000000000000000ULL);
        const u64 now = ((((u64)(ranks[id] + 1u)) << 32) | (u64)id) ^ 0x8000000000000000ULL;
        words[i] = is_min ? ~now : now;
    }
}
}  // namespace
void launch_rank_image(const int32_t* ids, const uint8_t* nulls, const uint32_t* ranks, int64_t n, int64_t* image, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_rank_image, grid_of(n), 256, 0, s, ids, nulls, ranks, (i64)n, (i64*)image);
    PA_HIP(hipGetLastError());
}
void launch_rerank_words(uint64_t* words, int64_t n, bool is_min, const uint32_t* ranks, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_rerank_words, grid_of(n), 256, 0, s, (u64*)words, (i64)n, is_min ? 1 : 0, ranks);
    PA_HIP(hipGetLastError());
}

}  // namespace pa
