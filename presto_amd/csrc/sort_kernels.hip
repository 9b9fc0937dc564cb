// sort_kernels.hip -- the pair sort under OrderByOperator's passes: (64-bit key image, row id) pairs by a range of the image's bits, stable.
// (PagesIndex.sort / PagesIndexOrdering.quickSort, core/trino-main/src/main/java/io/trino/operator/PagesIndex.java:418-426: the
// reference quick-sorts row addresses with a comparator; what the sort is asked to do here -- order-preserving images per sort channel and
// direction, 8-byte chunks + length for VARCHAR, the NULL placement as a digit of its own, constant bits left out -- is op_order_by.cpp's.)
//
// Most-significant bits first, and the pairs cross HBM three times instead of once per 8-bit digit:
//   1. two stable partition passes by the top T = bits1 + bits2 bits of the range (at most 7 bits each): a count launch (keys only), a
//      scan of the tile x digit matrix, and a scatter launch whose 4096-row tiles rank their rows with wave ballots (a row's place in its
//      (tile, digit) run = earlier rows of the tile with the same digit), stage the tile in LDS in digit order and write every run
//      as one contiguous piece.  The second pass works inside the first pass's buckets (tiles never straddle a bucket).
//      T is chosen so that a final bucket holds ~1024 pairs (n / 2^T); bits the range does not have are not partitioned by.
//   2. one launch that sorts every final bucket in LDS by the remaining bits: stable 7-bit passes over at most kCap = 2048 pairs, the
//      same ballot ranking, ping-pong between two LDS copies; the bucket is read once and written once, in place of its final rows.
// 2^24 pairs by 40 bits: 5 one-sweep passes of the library sort read and write the pairs 5 times (24 B per pair and pass); this reads
// 8 + 24 + 8 + 24 + 24 and writes 12 + 12 + 12 bytes per pair.
// Bucket bounds: bit prefixes for keys that spread over their varying bits (row ids, hashes, uniform integers: what OR / AND of the images
// leaves); for keys that crowd under few prefixes (images of doubles: the exponent; text) the caller says so and the bounds are every 16th
// key (at least) of sorted samples -- 2048 keys of the input, then 2048 of every first-pass bucket, each sample sorted by the bucket-sort
// kernel itself --, a digit is the number of bounds <= key, and a final bucket's LDS sort takes the bits in which its bounds differ.
// Payload: up to four 4- / 8-byte columns ride along with the pairs (OrderBy's output channels): staged through the same LDS tile by the
// partition passes, fetched from the bucket's window by the LDS sort -- sequential passes instead of a random gather by the sorted rows.
// When a final bucket would not fit in LDS (prefix mode: crowded keys after all; sampled mode: one value filling a bucket) or the input is
// beyond 27 M pairs (9.8 M by samples), rocPRIM's device radix sort does the whole job instead -- decided from the bucket sizes, which are
// known on the device before the last pass; the partition passes already done are then wasted (0.2 ms at 2^24 pairs).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "sort_kernels.hpp"

namespace pa {

namespace {

using u64 = unsigned long long;
using u32 = unsigned int;
using i32 = int;
using i64 = long long;

constexpr int kTile = 4096;       // rows of a partition tile: 4 waves x 16 rounds of 64
constexpr int kPartBits = 7;      // digit of a partition pass (cnt[4][128]: the kernel's LDS stays under 53 KB -> 3 workgroups per CU)
constexpr int kCap = 2048;        // pairs of a final bucket the LDS sort takes
constexpr int kSortBits = 7;      // digit of an LDS pass
constexpr int kMaxTopBits = 2 * kPartBits;
constexpr i64 kFastMaxRows = 27000000;   // 2^14 buckets of <= 1650 pairs on average: six sigma of a uniform spread stay under kCap

// columns that ride along with the pairs (OrderBy's output channels): moved by every pass to the rows their pairs go to
struct PayloadDev {
    const void* in[PA_SORT_MAX_PAYLOAD];
    void* out[PA_SORT_MAX_PAYLOAD];
    int width[PA_SORT_MAX_PAYLOAD];   // 4 or 8 bytes
    int n;
};

// Bucket bounds from sorted samples of the keys instead of bit prefixes (keys that crowd under a few prefixes: doubles, text).  A pass
// over bucket b (the whole input: b = 0) with nd digits has kSamplePer sampled keys of that bucket, sorted, at sorted[b * kSamplePer ..]:
// digit t starts at sorted[b * kSamplePer + t * (kSamplePer / nd)].  sorted == nullptr: the digit is bits of the key.
constexpr int kSamplePer = 2048;   // sampled keys per partitioned range (= one LDS bucket sort): >= 16 per digit
struct Splitters {
    const u64* sorted;
};
// ... and of the final buckets (b, t) for the LDS sort, which takes the bits its bucket's bounds leave open: first-pass bounds s1 (one
// range), second-pass bounds s2 (one range per first-pass bucket; null when there was one pass)
struct FinalBounds {
    const u64* s1;
    const u64* s2;
    i32 nd1, nd2;
};

struct SortCtl {
    i32 max_bucket;   // largest final bucket
    i32 tiles_b;      // tiles of the second partition pass
    i32 pad[14];
};

__device__ __forceinline__ i32 wave_inclusive_scan(i32 v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const i32 u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// bits [shift, shift + width) of a key as a 32-bit digit (width <= 8).  The shift is the same for every lane: one 32-bit funnel shift
// instead of a 64-bit shift (quarter rate)
__device__ __forceinline__ u32 key_digit(u64 key, int shift, u32 mask)
{
    const u32 lo = (u32)key, hi = (u32)(key >> 32);
    const u32 v = shift >= 32 ? hi >> (shift - 32) : __builtin_amdgcn_alignbit(hi, lo, (u32)shift);
    return v & mask;
}

// the bounds of bucket `bucket`'s digits in LDS: spl[t] = lower bound of digit t (spl[0] = 0: everything below the first bound)
__device__ __forceinline__ void load_splitters(u64* spl, const Splitters& sp, i32 bucket, i32 nd)
{
    if ((i32)threadIdx.x < nd) spl[threadIdx.x] = threadIdx.x == 0 ? 0ULL : sp.sorted[(i64)bucket * kSamplePer + (i64)threadIdx.x * (kSamplePer / nd)];
}
// digit of a key = number of bounds (beyond spl[0]) that are <= key; nd = 1 << bits of them in spl
__device__ __forceinline__ u32 splitter_digit(const u64* spl, int bits, u64 key)
{
    u32 pos = 0;
    for (int step = bits > 0 ? 1 << (bits - 1) : 0; step > 0; step >>= 1) {
        if (spl[pos + step] <= key) pos += (u32)step;
    }
    return pos;
}

// rows of this lane's digit among the live lanes of the wave: `peers` = their lanes, returns how many of them come before this lane.
// One ballot per bit of the digit; a lane keeps the lanes whose bit equals its own: peers &= ~(ballot ^ s) with s = the lane's bit
// spread over a word (0 or ~0).
template <int BITS>
__device__ __forceinline__ int wave_digit_peers_n(u32 d, bool live, int lane, u64* peers_out)
{
    const u64 all = __ballot(live);
    u32 lo = (u32)all, hi = (u32)(all >> 32);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const i32 s = __builtin_amdgcn_sbfe((i32)d, b, 1);   // 0 or -1
        const u64 bal = __ballot(s != 0);
        lo &= ~((u32)bal ^ (u32)s);
        hi &= ~((u32)(bal >> 32) ^ (u32)s);
    }
    *peers_out = ((u64)hi << 32) | lo;
    return (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}

__device__ __forceinline__ int wave_digit_peers(u32 d, bool live, int bits, int lane, u64* peers_out)
{
    switch (bits) {   // (uniform: the digit's width is a launch parameter)
        case 1: return wave_digit_peers_n<1>(d, live, lane, peers_out);
        case 2: return wave_digit_peers_n<2>(d, live, lane, peers_out);
        case 3: return wave_digit_peers_n<3>(d, live, lane, peers_out);
        case 4: return wave_digit_peers_n<4>(d, live, lane, peers_out);
        case 5: return wave_digit_peers_n<5>(d, live, lane, peers_out);
        case 6: return wave_digit_peers_n<6>(d, live, lane, peers_out);
        default: return wave_digit_peers_n<7>(d, live, lane, peers_out);
    }
}

// digit counts of every tile: counts[digit * tiles_cap + tile].  Tiles: 4096 consecutive rows of the input (tile_start == nullptr), or
// the rows the plan kernel listed (second pass: tiles inside the first pass's buckets; workgroups past *ntiles_dev leave).
__global__ __launch_bounds__(256) void k_sort_count(const u64* __restrict__ keys, i64 n, const i32* __restrict__ tile_start, const i32* __restrict__ tile_rows,
                                                    const i32* __restrict__ tile_bucket, const i32* __restrict__ ntiles_dev, i32 tiles_cap, int shift, int bits,
                                                    i32* __restrict__ counts, Splitters sp)
{
    __shared__ i32 hist[1 << kPartBits];
    __shared__ u64 spl[1 << kPartBits];
    const i32 tile = (i32)blockIdx.x;
    if (ntiles_dev != nullptr && tile >= *ntiles_dev) return;
    if (threadIdx.x < (1u << kPartBits)) hist[threadIdx.x] = 0;
    if (sp.sorted) load_splitters(spl, sp, tile_bucket ? tile_bucket[tile] : 0, 1 << bits);
    __syncthreads();
    const i64 start = tile_start ? (i64)tile_start[tile] : (i64)tile * kTile;
    const i32 rows = tile_rows ? tile_rows[tile] : (i32)(n - start < (i64)kTile ? n - start : (i64)kTile);
    const u32 mask = (1u << bits) - 1u;
#pragma unroll
    for (int i = 0; i < kTile / 256; i++) {
        const i32 idx = i * 256 + (i32)threadIdx.x;
        if (idx < rows) {
            const u64 k = keys[start + idx];
            atomicAdd(&hist[sp.sorted ? splitter_digit(spl, bits, k) : key_digit(k, shift, mask)], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < (1u << bits)) counts[(i64)threadIdx.x * tiles_cap + tile] = hist[threadIdx.x];
}

// counts[digit][tile] -> rows of the digit in earlier tiles of the same bucket (exclusive, in place); totals[bucket * nd + digit] = rows of
// the digit in the bucket.  One workgroup per (digit, bucket) -- 256 threads for the one bucket of the first pass, one wave for the second
// pass's buckets of a few dozen tiles -- and a run of consecutive tiles per thread.  bucket_tile_base == nullptr: one bucket of `tiles` tiles.
__global__ __launch_bounds__(256) void k_sort_scan_tiles(i32* __restrict__ counts, i32 tiles_cap, const i32* __restrict__ bucket_tile_base, i32 tiles, i32 nd,
                                                         i32* __restrict__ totals)
{
    __shared__ i32 wave_sums[4];
    const i32 d = (i32)blockIdx.x, b = (i32)blockIdx.y;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const i32 threads = (i32)blockDim.x;
    const i32 t0 = bucket_tile_base ? bucket_tile_base[b] : 0;
    const i32 t1 = bucket_tile_base ? bucket_tile_base[b + 1] : tiles;
    i32* row = counts + (i64)d * tiles_cap;
    const i32 per = (t1 - t0 + threads - 1) / threads;
    const i32 lo = t0 + (i32)threadIdx.x * per < t1 ? t0 + (i32)threadIdx.x * per : t1;
    const i32 hi = lo + per < t1 ? lo + per : t1;
    i32 sum = 0;
    for (i32 t = lo; t < hi; t++) sum += row[t];
    const i32 inc = wave_inclusive_scan(sum, lane);
    i32 before = 0, all = inc;
    if (threads > 64) {
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        all = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w < wave) before += wave_sums[w];
            all += wave_sums[w];
        }
    }
    else all = __shfl(inc, 63, 64);
    i32 run = before + inc - sum;
    for (i32 t = lo; t < hi; t++) {
        const i32 v = row[t];
        row[t] = run;
        run += v;
    }
    if (threadIdx.x == 0) totals[(i64)b * nd + d] = all;
}

// offs[0..m] = exclusive prefix sums of totals[0..m), *max_out = the largest of them.  One workgroup of 1024, a run of at most kOffsPer
// entries per thread (m <= 2^14), held in registers between the two sweeps.
constexpr int kOffsPer = (1 << kMaxTopBits) / 1024;
__global__ __launch_bounds__(1024) void k_sort_offsets(const i32* __restrict__ totals, i32 m, i32* __restrict__ offs, i32* __restrict__ max_out)
{
    __shared__ i32 wave_sums[16];
    __shared__ i32 wave_max[16];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const i32 per = (m + 1023) / 1024;
    const i32 lo = (i32)threadIdx.x * per;
    i32 vals[kOffsPer];
    i32 sum = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < kOffsPer; k++) {
        const i32 i = lo + k;
        vals[k] = (k < per && i < m) ? totals[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < kOffsPer; k++) {
        sum += vals[k];
        mx = vals[k] > mx ? vals[k] : mx;
    }
    const i32 inc = wave_inclusive_scan(sum, lane);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const i32 u = __shfl_xor(mx, o, 64);
        mx = u > mx ? u : mx;
    }
    if (lane == 63) wave_sums[wave] = inc;
    if (lane == 0) wave_max[wave] = mx;
    __syncthreads();
    i32 before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        if (w < wave) before += wave_sums[w];
        all += wave_sums[w];
    }
    i32 run = before + inc - sum;
#pragma unroll
    for (int k = 0; k < kOffsPer; k++) {
        const i32 i = lo + k;
        if (k < per && i < m) offs[i] = run;
        run += vals[k];
    }
    if (threadIdx.x == 0) {
        offs[m] = all;
        if (max_out != nullptr) {
            i32 best = wave_max[0];
            for (int w = 1; w < 16; w++) best = wave_max[w] > best ? wave_max[w] : best;
            *max_out = best;
        }
    }
}

// The tiles of the second partition pass: every bucket of the first pass (offs[b] .. offs[b + 1]) cut into pieces of kTile rows.
// One workgroup, one thread per bucket (nb <= 128).
__global__ __launch_bounds__(256) void k_sort_plan_tiles(const i32* __restrict__ offs, i32 nb, i32* __restrict__ bucket_tile_base, i32* __restrict__ tile_start,
                                                         i32* __restrict__ tile_rows, i32* __restrict__ tile_bucket, i32* __restrict__ tiles_out)
{
    __shared__ i32 wave_sums[4];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const i32 b = (i32)threadIdx.x;
    const i32 first = b < nb ? offs[b] : 0;
    const i32 size = b < nb ? offs[b + 1] - first : 0;
    const i32 nt = (size + kTile - 1) / kTile;
    const i32 inc = wave_inclusive_scan(nt, lane);
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    i32 before = 0, all = 0;
    for (int w = 0; w < 4; w++) {
        if (w < wave) before += wave_sums[w];
        all += wave_sums[w];
    }
    const i32 tb = before + inc - nt;
    if (b < nb) bucket_tile_base[b] = tb;
    if (b == 0) {
        bucket_tile_base[nb] = all;
        *tiles_out = all;
    }
    for (i32 k = 0; k < nt; k++) {
        tile_start[tb + k] = first + k * kTile;
        tile_rows[tb + k] = size - k * kTile < kTile ? size - k * kTile : kTile;
        tile_bucket[tb + k] = b;
    }
}

// One stable partition pass over a tile: the tile's rows go to out[base[bucket * nd + d] + offs[d * tiles_cap + tile] + (place of the row
// among the tile's rows of digit d)].  Wave w of the workgroup takes rows [1024 w, 1024 (w + 1)) of the tile in 16 rounds of 64 consecutive
// rows, so "earlier in the tile" = earlier wave, then earlier round, then lower lane.
__global__ __launch_bounds__(256) void k_sort_partition(const u64* __restrict__ kin, const i32* __restrict__ rin, u64* __restrict__ kout, i32* __restrict__ rout, i64 n,
                                                        const i32* __restrict__ tile_start, const i32* __restrict__ tile_rows, const i32* __restrict__ tile_bucket,
                                                        const i32* __restrict__ ntiles_dev, i32 tiles_cap, int shift, int bits, const i32* __restrict__ offs,
                                                        const i32* __restrict__ base, PayloadDev pl, Splitters sp)
{
    constexpr int kRounds = kTile / 256;
    __shared__ u64 skey[kTile];
    __shared__ i32 srow[kTile];
    __shared__ i32 cnt[4][1 << kPartBits];
    __shared__ i32 lstart[1 << kPartBits], goff[1 << kPartBits];
    __shared__ u64 spl[1 << kPartBits];
    __shared__ i32 wave_sums[4];
    const i32 tile = (i32)blockIdx.x;
    if (ntiles_dev != nullptr && tile >= *ntiles_dev) return;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const i64 start = tile_start ? (i64)tile_start[tile] : (i64)tile * kTile;
    const i32 rows = tile_rows ? tile_rows[tile] : (i32)(n - start < (i64)kTile ? n - start : (i64)kTile);
    const i32 bucket = tile_bucket ? tile_bucket[tile] : 0;
    const i32 nd = 1 << bits;
    const u32 mask = (u32)nd - 1u;
    if ((i32)threadIdx.x < nd) {
#pragma unroll
        for (int w = 0; w < 4; w++) cnt[w][threadIdx.x] = 0;
    }
    if (sp.sorted) load_splitters(spl, sp, bucket, nd);
    u64 key[kRounds];
    i32 row[kRounds];
    i32 lr[kRounds];
    u32 dg[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const i32 idx = wave * (kTile / 4) + r * 64 + lane;
        const bool live = idx < rows;
        key[r] = live ? kin[start + idx] : 0ULL;
        row[r] = live ? (rin ? rin[start + idx] : (i32)(start + idx)) : 0;
    }
    __syncthreads();
    // the wave's count of every digit grows round by round: the first lane of a digit's rows in a round adds them and gets the count of
    // the rounds before (LDS executes a wave's atomics in order); the adds of all rounds are in flight together, their results are
    // handed to the other lanes of the group afterwards
    i32 prior[kRounds];
    int leader[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const i32 idx = wave * (kTile / 4) + r * 64 + lane;
        const bool live = idx < rows;
        const u32 d = live ? (sp.sorted ? splitter_digit(spl, bits, key[r]) : key_digit(key[r], shift, mask)) : 0u;
        dg[r] = d;
        u64 peers;
        const int before = wave_digit_peers(d, live, bits, lane, &peers);
        lr[r] = before;
        leader[r] = live ? __ffsll((long long)peers) - 1 : lane;
        prior[r] = 0;
        if (live && before == 0) prior[r] = __hip_atomic_fetch_add(&cnt[wave][d], (i32)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int r = 0; r < kRounds; r++) lr[r] += __shfl(prior[r], leader[r], 64);
    __syncthreads();
    {
        i32 total = 0;
        if ((i32)threadIdx.x < nd) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const i32 c = cnt[w][threadIdx.x];
                cnt[w][threadIdx.x] = total;
                total += c;
            }
        }
        const i32 inc = wave_inclusive_scan(total, lane);
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        i32 before = 0;
        for (int w = 0; w < wave; w++) before += wave_sums[w];
        if ((i32)threadIdx.x < nd) {
            lstart[threadIdx.x] = before + inc - total;
            goff[threadIdx.x] = base[(i64)bucket * nd + threadIdx.x] + offs[(i64)threadIdx.x * tiles_cap + tile];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const i32 idx = wave * (kTile / 4) + r * 64 + lane;
        if (idx < rows) {
            const u32 d = dg[r];
            const i32 p = lstart[d] + cnt[wave][d] + lr[r];
            lr[r] = p;   // (the row's place in the staged tile: the payload columns go the same way)
            skey[p] = key[r];
            srow[p] = row[r];
        }
    }
    __syncthreads();
    i32 dest[kRounds];   // where staged row threadIdx.x + 256 i goes
#pragma unroll
    for (int i = 0; i < kRounds; i++) {
        const i32 j = (i32)threadIdx.x + 256 * i;
        dest[i] = 0;
        if (j < rows) {
            const u64 k = skey[j];
            u32 d;
            if (sp.sorted) {  // the digit whose rows of the staged tile hold position j (the last one that starts at or before j)
                d = 0;
                for (int step = nd >> 1; step > 0; step >>= 1) {
                    if (lstart[d + step] <= j) d += (u32)step;
                }
            }
            else d = key_digit(k, shift, mask);
            dest[i] = goff[d] + (j - lstart[d]);
            kout[dest[i]] = k;
            rout[dest[i]] = srow[j];
        }
    }
    for (int c = 0; c < pl.n; c++) {
        __syncthreads();   // the staged keys / rows (or the column before) have been written out
        if (pl.width[c] == 8) {
            const u64* pin = static_cast<const u64*>(pl.in[c]);
            u64* pout = static_cast<u64*>(pl.out[c]);
#pragma unroll
            for (int r = 0; r < kRounds; r++) {
                const i32 idx = wave * (kTile / 4) + r * 64 + lane;
                if (idx < rows) skey[lr[r]] = pin[start + idx];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < kRounds; i++) {
                const i32 j = (i32)threadIdx.x + 256 * i;
                if (j < rows) pout[dest[i]] = skey[j];
            }
        }
        else {
            const u32* pin = static_cast<const u32*>(pl.in[c]);
            u32* pout = static_cast<u32*>(pl.out[c]);
#pragma unroll
            for (int r = 0; r < kRounds; r++) {
                const i32 idx = wave * (kTile / 4) + r * 64 + lane;
                if (idx < rows) srow[lr[r]] = (i32)pin[start + idx];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < kRounds; i++) {
                const i32 j = (i32)threadIdx.x + 256 * i;
                if (j < rows) pout[dest[i]] = (u32)srow[j];
            }
        }
    }
}

// Every final bucket sorted in LDS by bits [begin_bit, begin_bit + rest_bits) of its keys: `passes` stable passes of `bits_per` bits.
// Bucket b = rows offs[b] .. offs[b + 1] of the input (offs == nullptr: the one bucket of rows 0 .. n_single); at most kCap rows each
// (the host checked).  The sorted bucket lands at the same rows of the output.
// A pass: every wave takes consecutive rows (the same number of 64-row rounds each), ranks them like the partition pass, and the rows go to
// their places IN the one LDS copy -- every row sits in a register of its thread between the barrier that ends the reads and the writes.
// The first pass takes its rows from HBM directly.  26.5 KB of LDS: six workgroups per CU.
__global__ __launch_bounds__(256) void k_sort_buckets(const u64* __restrict__ kin, const i32* __restrict__ rin, u64* __restrict__ kout, i32* __restrict__ rout,
                                                      const i32* __restrict__ offs, i32 n_single, int begin_bit, int rest_bits, int passes, int bits_per, PayloadDev pl,
                                                      FinalBounds fb)
{
    constexpr int kRounds = kCap / 256;
    __shared__ u64 kbuf[kCap];
    __shared__ i32 rbuf[kCap];
    __shared__ i32 cnt[2][4][1 << kSortBits];
    __shared__ i32 dstart[1 << kSortBits];
    const i32 bucket = (i32)blockIdx.x;
    const i32 start = offs ? offs[bucket] : bucket * n_single;   // (no offsets: buckets of n_single rows each, one behind the other)
    const i32 count = offs ? offs[bucket + 1] - start : n_single;
    if (count <= 0 || count > kCap) return;   // (a bucket beyond the LDS copy: the host has seen it and sorts everything another way)
    if (fb.s1) {
        // bounds from sorted samples: the bucket's keys lie between two of them, so the bits above the highest one in which the bounds
        // differ are the same in all of them (rest_bits = the whole range here: the bounds say what is left of it)
        const i32 b = bucket / fb.nd2, tt = bucket - b * fb.nd2;
        const i32 step1 = kSamplePer / fb.nd1, step2 = kSamplePer / fb.nd2;
        const u64 lo1 = b > 0 ? fb.s1[(i64)b * step1] : 0ULL;
        const u64 hi1 = b + 1 < fb.nd1 ? fb.s1[(i64)(b + 1) * step1] : ~0ULL;
        const u64 lo = (fb.s2 && tt > 0) ? fb.s2[(i64)b * kSamplePer + (i64)tt * step2] : lo1;
        const u64 hi = (fb.s2 && tt + 1 < fb.nd2) ? fb.s2[(i64)b * kSamplePer + (i64)(tt + 1) * step2] : hi1;
        const u64 x = lo ^ hi;
        const int top = x ? 64 - __builtin_clzll(x) : 0;
        const int end = top < begin_bit + rest_bits ? top : begin_bit + rest_bits;
        rest_bits = end > begin_bit ? end - begin_bit : 0;
        passes = (rest_bits + kSortBits - 1) / kSortBits;
        bits_per = passes ? (rest_bits + passes - 1) / passes : 0;
    }
    if (passes == 0) {  // every key of the bucket the same in the bits that count: the rows stay as the partition passes left them
        for (i32 j = (i32)threadIdx.x; j < count; j += 256) {
            kout[(i64)start + j] = kin[(i64)start + j];
            rout[(i64)start + j] = rin ? rin[(i64)start + j] : start + j;
            for (int c = 0; c < pl.n; c++) {
                if (pl.width[c] == 8) static_cast<u64*>(pl.out[c])[(i64)start + j] = static_cast<const u64*>(pl.in[c])[(i64)start + j];
                else static_cast<u32*>(pl.out[c])[(i64)start + j] = static_cast<const u32*>(pl.in[c])[(i64)start + j];
            }
        }
        return;
    }
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const i32 chunk = ((count + 255) / 256) * 64;
    const int rounds = chunk / 64;
    u64 key[kRounds];
    i32 row[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const i32 idx = wave * chunk + r * 64 + lane;
        const bool live = r < rounds && idx < count;
        key[r] = live ? kin[(i64)start + idx] : 0ULL;
        row[r] = idx;   // where the pair lies in the input bucket: its row id and its payload are fetched from there at the end
    }
    if (threadIdx.x < (1u << kSortBits)) {
#pragma unroll
        for (int w = 0; w < 4; w++) cnt[0][w][threadIdx.x] = 0;
    }
    for (int p = 0; p < passes; p++) {
        const int lo = begin_bit + p * bits_per;
        const int bits = (begin_bit + rest_bits - lo) < bits_per ? (begin_bit + rest_bits - lo) : bits_per;
        const u32 mask = (1u << bits) - 1u;
        i32(*c)[1 << kSortBits] = cnt[p & 1];
        __syncthreads();   // the counters are zero; the previous pass's rows are in their places
        if (p > 0) {
#pragma unroll
            for (int r = 0; r < kRounds; r++) {
                const i32 idx = wave * chunk + r * 64 + lane;
                if (r < rounds && idx < count) {
                    key[r] = kbuf[idx];
                    row[r] = rbuf[idx];
                }
            }
        }
        i32 lr[kRounds], prior[kRounds];
        int leader[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            lr[r] = 0;
            prior[r] = 0;
            leader[r] = lane;
            if (r < rounds) {
                const i32 idx = wave * chunk + r * 64 + lane;
                const bool live = idx < count;
                const u32 d = key_digit(key[r], lo, mask);
                u64 peers;
                const int before = wave_digit_peers(d, live, bits, lane, &peers);
                lr[r] = before;
                if (live) leader[r] = __ffsll((long long)peers) - 1;
                if (live && before == 0) prior[r] = __hip_atomic_fetch_add(&c[wave][d], (i32)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            if (r < rounds) lr[r] += __shfl(prior[r], leader[r], 64);
        }
        __syncthreads();   // every wave has counted, and has read its rows of the LDS copy
        if (wave == 0) {
            // two digits per lane: counts of the waves -> rows of the digit in the waves before; start of the digit's rows in the bucket
            i32 t0 = 0, t1 = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const i32 c0 = c[w][2 * lane], c1 = c[w][2 * lane + 1];
                c[w][2 * lane] = t0;
                c[w][2 * lane + 1] = t1;
                t0 += c0;
                t1 += c1;
            }
            const i32 inc = wave_inclusive_scan(t0 + t1, lane);
            dstart[2 * lane] = inc - t0 - t1;
            dstart[2 * lane + 1] = inc - t1;
        }
        else if (wave == 1) {
            // the other set of counters for the next pass
            i32(*z)[1 << kSortBits] = cnt[(p + 1) & 1];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                z[w][2 * lane] = 0;
                z[w][2 * lane + 1] = 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const i32 idx = wave * chunk + r * 64 + lane;
            if (r < rounds && idx < count) {
                const u32 d = key_digit(key[r], lo, mask);
                const i32 at = dstart[d] + c[wave][d] + lr[r];
                kbuf[at] = key[r];
                rbuf[at] = row[r];
            }
        }
    }
    __syncthreads();
    // the sorted keys from LDS; row ids and payload columns from the input bucket (a window of <= 2048 rows that was just read)
    for (i32 j = (i32)threadIdx.x; j < count; j += 256) {
        const i32 from = rbuf[j];
        kout[(i64)start + j] = kbuf[j];
        rout[(i64)start + j] = rin ? rin[(i64)start + from] : start + from;
        for (int c = 0; c < pl.n; c++) {
            if (pl.width[c] == 8) static_cast<u64*>(pl.out[c])[(i64)start + j] = static_cast<const u64*>(pl.in[c])[(i64)start + from];
            else static_cast<u32*>(pl.out[c])[(i64)start + j] = static_cast<const u32*>(pl.in[c])[(i64)start + from];
        }
    }
}

// kSamplePer keys of every bucket (offs == nullptr: of the one range 0 .. n), evenly spaced: sample[b * kSamplePer + i]
__global__ __launch_bounds__(256) void k_sort_sample(const u64* __restrict__ keys, const i32* __restrict__ offs, i64 n, u64* __restrict__ sample)
{
    const i32 b = (i32)blockIdx.y;
    const i32 i = (i32)(blockIdx.x * 256 + threadIdx.x);
    const i64 first = offs ? (i64)offs[b] : 0;
    const i64 size = offs ? (i64)offs[b + 1] - first : n;
    if (i < kSamplePer) sample[(i64)b * kSamplePer + i] = size > 0 ? keys[first + (size * i + (size >> 1)) / kSamplePer] : 0ULL;
}

__global__ __launch_bounds__(256) void k_sort_iota(i32* __restrict__ rows, i64 n)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rows[i] = (i32)i;
}

constexpr int64_t kSampleMax = (int64_t)(1 << kPartBits) * kSamplePer;   // the second pass's samples: one range per first-pass bucket
// A bucket between two sampled bounds (every 16th key of a sorted sample, at least) holds n / buckets rows on average and a
// Gamma(16)-distributed multiple of 1/16 of that: at 512 rows on average the 2048 of the LDS copy are 64 / 16 -- never reached (10^-14 per
// bucket); at 1024 on average they are 32 / 16: one bucket in 2000.
constexpr int64_t kSampledMeanRows = 512;
constexpr int64_t kSampledMaxRows = (int64_t)(1 << kMaxTopBits) * 600;   // beyond: the library sort

struct FastLayout {
    size_t payload = 0;   // scratch copies of the payload columns, 8 n bytes each, behind each other
    size_t sample_in = 0, sample1 = 0, sample2 = 0, sample_rows = 0;   // bounds from samples: as drawn, sorted (first pass, second pass), row ids nobody reads
    size_t keys = 0, rows = 0, counts_a = 0, totals_a = 0, offs_a = 0, bucket_tiles = 0, tile_start = 0, tile_rows = 0, tile_bucket = 0, counts_b = 0, totals_b = 0,
           offs_b = 0, ctl = 0, end = 0;
    i64 tiles_a = 0, tiles_cap_b = 0;
};

FastLayout fast_layout(int64_t n, int payload_columns)
{
    FastLayout l;
    l.tiles_a = (n + kTile - 1) / kTile;
    l.tiles_cap_b = l.tiles_a + (1 << kPartBits);
    size_t at = 0;
    auto take = [&at](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) & ~(size_t)255;
        return here;
    };
    l.keys = take((size_t)n * 8);
    l.rows = take((size_t)n * 4);
    l.counts_a = take((size_t)l.tiles_a * (1 << kPartBits) * 4);
    l.totals_a = take((size_t)(1 << kPartBits) * 4);
    l.offs_a = take((size_t)((1 << kPartBits) + 1) * 4);
    l.bucket_tiles = take((size_t)((1 << kPartBits) + 1) * 4);
    l.tile_start = take((size_t)l.tiles_cap_b * 4);
    l.tile_rows = take((size_t)l.tiles_cap_b * 4);
    l.tile_bucket = take((size_t)l.tiles_cap_b * 4);
    l.counts_b = take((size_t)l.tiles_cap_b * (1 << kPartBits) * 4);
    l.totals_b = take((size_t)(1 << kMaxTopBits) * 4);
    l.offs_b = take((size_t)((1 << kMaxTopBits) + 1) * 4);
    l.ctl = take(sizeof(SortCtl));
    l.sample_in = take((size_t)kSampleMax * 8);
    l.sample1 = take((size_t)kSamplePer * 8);
    l.sample2 = take((size_t)kSampleMax * 8);
    l.sample_rows = take((size_t)kSampleMax * 4);
    l.payload = at;
    for (int c = 0; c < payload_columns; c++) take((size_t)n * 8);
    l.end = at;
    return l;
}

size_t library_temp_bytes(int64_t n)
{
    size_t bytes = 0;
    PA_HIP(rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                     (size_t)std::max<int64_t>(n, 1), 0u, 64u, (hipStream_t) nullptr));
    return bytes;
}

void library_sort(const uint64_t* keys_in, const int32_t* rows_in, int32_t* rows_scratch, uint64_t* keys_out, int32_t* rows_out, int64_t n, int begin_bit,
                  int end_bit, void* temp, size_t temp_bytes, hipStream_t s)
{
    if (rows_in == nullptr) {  // rows 0, 1, 2, ... were implied: the library wants them in memory
        PA_REQUIRE(rows_scratch != nullptr, PA_ERR_DEVICE, "internal: pair sort without row ids and without room for them");
        hipLaunchKernelGGL(k_sort_iota, (int)((n + 255) / 256), 256, 0, s, rows_scratch, (i64)n);
        rows_in = rows_scratch;
    }
    // the scratch was sized for the whole key (sort_pairs_temp_bytes): rocPRIM does not promise that a narrower bit range needs no more
    // (its merge-sort and one-sweep paths size differently) -- ask for THIS range and refuse to run short
    size_t need = 0;
    PA_HIP(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, rows_in, rows_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, s));
    PA_REQUIRE(need <= temp_bytes, PA_ERR_DEVICE, "internal: pair sort scratch smaller than this bit range needs");
    PA_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rows_in, rows_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, s));
}

int ceil_log2(int64_t v)
{
    int b = 0;
    while (((int64_t)1 << b) < v) b++;
    return b;
}

}  // namespace

size_t sort_pairs_temp_bytes(int64_t n, int payload_columns)
{
    n = std::max<int64_t>(n, 1);
    const size_t fast = n <= kFastMaxRows ? fast_layout(n, payload_columns).end : 0;
    return std::max(library_temp_bytes(n), fast) + 256;
}

int launch_sort_pairs(const uint64_t* keys_in, const int32_t* rows_in, int32_t* rows_scratch, uint64_t* keys_out, int32_t* rows_out, int64_t n, int begin_bit,
                      int end_bit, void* temp, size_t temp_bytes, hipStream_t s, const SortPayload* payload, int hint)
{
    const int pcols = payload ? payload->count : 0;
    PA_REQUIRE(pcols >= 0 && pcols <= PA_SORT_MAX_PAYLOAD, PA_ERR_DEVICE, "internal: payload columns of a pair sort");
    for (int c = 0; c < pcols; c++) PA_REQUIRE(payload->width[c] == 4 || payload->width[c] == 8, PA_ERR_DEVICE, "internal: payload width of a pair sort");
    // the payload columns of one pass: from the caller's arrays / the output / the scratch copies to the next of them
    auto stage = [&](int from, int to) {   // 0 = the caller's input, 1 = the caller's output, 2 = scratch
        PayloadDev d;
        memset(&d, 0, sizeof d);
        d.n = pcols;
        for (int c = 0; c < pcols; c++) {
            char* scratch = static_cast<char*>(temp) + fast_layout(n, pcols).payload + (size_t)c * (((size_t)n * 8 + 255) & ~(size_t)255);
            d.in[c] = from == 0 ? payload->in[c] : (from == 1 ? payload->out[c] : scratch);
            d.out[c] = to == 1 ? payload->out[c] : scratch;
            d.width[c] = payload->width[c];
        }
        return d;
    };
    if (n <= 0) return PA_SORT_NONE;
    PA_REQUIRE(begin_bit >= 0 && end_bit > begin_bit && end_bit <= 64, PA_ERR_DEVICE, "internal: bit range of a pair sort");
    static const bool library_only = getenv("PRESTO_AMD_SORT_LIBRARY") != nullptr;
    const bool by_sample = hint == PA_SORT_HINT_CROWDED && n > kCap;
    if (library_only || n > kFastMaxRows || (by_sample && n > kSampledMaxRows)) {
        library_sort(keys_in, rows_in, rows_scratch, keys_out, rows_out, n, begin_bit, end_bit, temp, temp_bytes, s);
        return PA_SORT_LIBRARY;
    }
    const int width = end_bit - begin_bit;
    const u64* kin = reinterpret_cast<const u64*>(keys_in);
    u64* kout = reinterpret_cast<u64*>(keys_out);
    auto lds_passes = [](int rest, int* bits_per) {
        const int passes = (rest + kSortBits - 1) / kSortBits;
        *bits_per = passes ? (rest + passes - 1) / passes : 0;
        return passes;
    };
    const Splitters none{nullptr};
    if (n <= kCap) {  // one bucket: the LDS sort alone
        int bits_per = 0;
        const int passes = lds_passes(width, &bits_per);
        hipLaunchKernelGGL(k_sort_buckets, 1, 256, 0, s, kin, rows_in, kout, rows_out, (const i32*)nullptr, (i32)n, begin_bit, width, passes, bits_per, stage(0, 1),
                           FinalBounds{nullptr, nullptr, 1, 1});
        PA_HIP(hipGetLastError());
        return PA_SORT_BUCKETS;
    }
    const FastLayout l = fast_layout(n, pcols);
    PA_REQUIRE(l.end <= temp_bytes, PA_ERR_DEVICE, "internal: pair sort scratch smaller than the partition passes need");
    char* t = static_cast<char*>(temp);
    u64* tkeys = reinterpret_cast<u64*>(t + l.keys);
    i32* trows = reinterpret_cast<i32*>(t + l.rows);
    i32* counts_a = reinterpret_cast<i32*>(t + l.counts_a);
    i32* totals_a = reinterpret_cast<i32*>(t + l.totals_a);
    i32* offs_a = reinterpret_cast<i32*>(t + l.offs_a);
    i32* bucket_tiles = reinterpret_cast<i32*>(t + l.bucket_tiles);
    i32* tile_start = reinterpret_cast<i32*>(t + l.tile_start);
    i32* tile_rows = reinterpret_cast<i32*>(t + l.tile_rows);
    i32* tile_bucket = reinterpret_cast<i32*>(t + l.tile_bucket);
    i32* counts_b = reinterpret_cast<i32*>(t + l.counts_b);
    i32* totals_b = reinterpret_cast<i32*>(t + l.totals_b);
    i32* offs_b = reinterpret_cast<i32*>(t + l.offs_b);
    SortCtl* ctl = reinterpret_cast<SortCtl*>(t + l.ctl);

    // ~1024 pairs per final bucket, at most 14 bits of partitioning, never more than the range has
    // (bounded by a sample: ~512 per bucket -- the buckets differ in size)
    const int64_t mean = by_sample ? kSampledMeanRows : 1024;
    const int top = std::min(width, std::min(kMaxTopBits, std::max(1, ceil_log2((n + mean - 1) / mean))));
    const int bits1 = std::min(top, kPartBits), bits2 = top - bits1;
    const int rest = by_sample ? width : width - top;   // (by sample: what is left of the range differs bucket by bucket, the bounds tell)
    const i32 nd1 = 1 << bits1, nd2 = 1 << bits2;
    const i32 tiles_a = (i32)l.tiles_a, cap_b = (i32)l.tiles_cap_b;
    // bucket bounds from samples: 2048 evenly spaced keys of the input now, 2048 of every first-pass bucket once the first pass has placed
    // them; each sample is sorted by the LDS bucket sort itself (one workgroup per sample, the whole bit range)
    Splitters sp1 = none, sp2 = none;
    FinalBounds fb{nullptr, nullptr, nd1, nd2};
    u64* sample = reinterpret_cast<u64*>(t + l.sample_in);
    u64* sorted1 = reinterpret_cast<u64*>(t + l.sample1);
    u64* sorted2 = reinterpret_cast<u64*>(t + l.sample2);
    i32* sample_rows = reinterpret_cast<i32*>(t + l.sample_rows);
    auto sort_samples = [&](const u64* keys, const i32* offs, i32 ranges, u64* sorted) {
        hipLaunchKernelGGL(k_sort_sample, dim3(kSamplePer / 256, ranges), 256, 0, s, keys, offs, (i64)n, sample);
        int per = 0;
        const int p = lds_passes(width, &per);
        PayloadDev nothing;
        memset(&nothing, 0, sizeof nothing);
        hipLaunchKernelGGL(k_sort_buckets, ranges, 256, 0, s, (const u64*)sample, (const i32*)nullptr, sorted, sample_rows, (const i32*)nullptr, (i32)kSamplePer, begin_bit,
                           width, p, per, nothing, FinalBounds{nullptr, nullptr, 1, 1});
    };
    if (by_sample) {
        sort_samples(kin, nullptr, 1, sorted1);
        sp1 = Splitters{sorted1};
        fb.s1 = sorted1;
    }
    // where the passes land: the last one (partition or LDS sort) in the output, the ones before alternate with the scratch pairs
    const bool two = bits2 > 0, sorting = rest > 0;
    const int moves = 1 + (two ? 1 : 0) + (sorting ? 1 : 0);
    u64* dst1_k = (moves % 2) ? kout : tkeys;   // (1 move: output; 2: scratch, output; 3: output, scratch, output)
    i32* dst1_r = (moves % 2) ? rows_out : trows;
    u64* dst2_k = (moves % 2) ? tkeys : kout;
    i32* dst2_r = (moves % 2) ? trows : rows_out;

    const int shift1 = end_bit - bits1, shift2 = end_bit - top;
    hipLaunchKernelGGL(k_sort_count, tiles_a, 256, 0, s, kin, (i64)n, (const i32*)nullptr, (const i32*)nullptr, (const i32*)nullptr, (const i32*)nullptr, tiles_a, shift1,
                       bits1, counts_a, sp1);
    hipLaunchKernelGGL(k_sort_scan_tiles, dim3(nd1, 1), 256, 0, s, counts_a, tiles_a, (const i32*)nullptr, tiles_a, nd1, totals_a);
    hipLaunchKernelGGL(k_sort_offsets, 1, 1024, 0, s, (const i32*)totals_a, nd1, offs_a, &ctl->max_bucket);
    if (two) hipLaunchKernelGGL(k_sort_plan_tiles, 1, 256, 0, s, (const i32*)offs_a, nd1, bucket_tiles, tile_start, tile_rows, tile_bucket, &ctl->tiles_b);
    const int place1 = (moves % 2) ? 1 : 2, place2 = (moves % 2) ? 2 : 1;   // where the first / second pass leave their rows (as dst1 / dst2)
    hipLaunchKernelGGL(k_sort_partition, tiles_a, 256, 0, s, kin, rows_in, dst1_k, dst1_r, (i64)n, (const i32*)nullptr, (const i32*)nullptr, (const i32*)nullptr,
                       (const i32*)nullptr, tiles_a, shift1, bits1, (const i32*)counts_a, (const i32*)offs_a, stage(0, place1), sp1);
    const i32* final_offs = offs_a;
    const u64* last_k = dst1_k;
    const i32* last_r = dst1_r;
    if (two) {
        if (by_sample) {
            sort_samples(dst1_k, offs_a, nd1, sorted2);
            sp2 = Splitters{sorted2};
            fb.s2 = sorted2;
        }
        hipLaunchKernelGGL(k_sort_count, cap_b, 256, 0, s, (const u64*)dst1_k, (i64)n, (const i32*)tile_start, (const i32*)tile_rows, (const i32*)tile_bucket,
                           (const i32*)&ctl->tiles_b, cap_b, shift2, bits2, counts_b, sp2);
        hipLaunchKernelGGL(k_sort_scan_tiles, dim3(nd2, nd1), 64, 0, s, counts_b, cap_b, (const i32*)bucket_tiles, 0, nd2, totals_b);
        hipLaunchKernelGGL(k_sort_offsets, 1, 1024, 0, s, (const i32*)totals_b, nd1 * nd2, offs_b, &ctl->max_bucket);
        hipLaunchKernelGGL(k_sort_partition, cap_b, 256, 0, s, (const u64*)dst1_k, (const i32*)dst1_r, dst2_k, dst2_r, (i64)n, (const i32*)tile_start, (const i32*)tile_rows,
                           (const i32*)tile_bucket, (const i32*)&ctl->tiles_b, cap_b, shift2, bits2, (const i32*)counts_b, (const i32*)offs_b, stage(place1, place2), sp2);
        final_offs = offs_b;
        last_k = dst2_k;
        last_r = dst2_r;
    }
    const int last_place = two ? place2 : place1;
    PA_HIP(hipGetLastError());
    if (!sorting) return PA_SORT_BUCKETS;   // the range had no more bits than the partition passes took: done, nothing to wait for
    i32 max_bucket = 0;
    read_back(&max_bucket, &ctl->max_bucket, 4, s);
    if (max_bucket > kCap) {
        // keys crowd in a few bit prefixes -- or, bounded by a sample, one key value fills more than a bucket: the library's passes over the
        // whole range, from the untouched input
        library_sort(keys_in, rows_in, rows_scratch, keys_out, rows_out, n, begin_bit, end_bit, temp, temp_bytes, s);
        return PA_SORT_LIBRARY;
    }
    int bits_per = 0;
    const int passes = lds_passes(rest, &bits_per);
    hipLaunchKernelGGL(k_sort_buckets, nd1 * nd2, 256, 0, s, last_k, last_r, kout, rows_out, final_offs, 0, begin_bit, rest, passes, bits_per, stage(last_place, 1),
                       fb);
    PA_HIP(hipGetLastError());
    return PA_SORT_BUCKETS;
}

}  // namespace pa
