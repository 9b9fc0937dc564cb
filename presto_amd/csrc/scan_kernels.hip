// scan_kernels.hip -- exclusive prefix sums, variable-width gathers and the stable partition used by
// the compaction paths (FilterAndProject output, join output, exchange partitioning).  HBM-bound integer
// work: wave64 shuffles + LDS inside a workgroup, two-level block sums across workgroups.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "kernels/pa_device.h"
#include "scan_kernels.hpp"

namespace pa {

static inline int blocks_for(int64_t n, int per_block)
{
    int64_t b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : b);
}

constexpr int kScanBlock = 256;
constexpr int kScanItems = 4;                        // items per thread
constexpr int kScanTile = kScanBlock * kScanItems;   // 1024 items per workgroup

// inclusive wave scan with shuffles
__device__ __forceinline__ i32 wave_inclusive_scan(i32 v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        i32 o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// exclusive prefix of `v` over the 256 threads of the workgroup; *total = workgroup sum
__device__ __forceinline__ i32 block_exclusive_scan(i32 v, i32* total)
{
    __shared__ i32 wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32 inc = wave_inclusive_scan(v);
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    i32 base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        i32 s = wave_sums[w];
        if (w < wave) base += s;
        all += s;
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_tile_sums(const i32* __restrict__ in, i64 n, i32* __restrict__ sums)
{
    i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
    i32 s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        if (base + i < n) s += in[base + i];
    }
    i32 total;
    (void)block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single workgroup: in-place exclusive scan of up to any length (sequential over 1024-item chunks)
__global__ __launch_bounds__(kScanBlock) void k_scan_single(i32* __restrict__ data, i64 n, i32* __restrict__ total_out)
{
    i32 carry = 0;
    for (i64 chunk = 0; chunk < n; chunk += kScanTile) {
        i64 base = chunk + (i64)threadIdx.x * kScanItems;
        i32 v[kScanItems];
        i32 s = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; i++) {
            v[i] = (base + i < n) ? data[base + i] : 0;
            s += v[i];
        }
        i32 total;
        i32 ex = block_exclusive_scan(s, &total) + carry;
#pragma unroll
        for (int i = 0; i < kScanItems; i++) {
            if (base + i < n) data[base + i] = ex;
            ex += v[i];
        }
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const i32* __restrict__ in, i64 n, const i32* __restrict__ tile_offsets,
                                                           i32* __restrict__ out)
{
    i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
    i32 v[kScanItems];
    i32 s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    i32 total;
    i32 ex = block_exclusive_scan(s, &total) + tile_offsets[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
}

// one workgroup of 1024, up to kSmallScanItems consecutive entries per thread held in registers: the scan of a page's tile counts (a
// 1.4 M-row page has 1368 - 5472 of them) in one launch instead of three
constexpr int kSmallScanItems = 16;
constexpr int64_t kSmallScanMax = 1024 * kSmallScanItems;
__global__ __launch_bounds__(1024) void k_scan_small(const i32* in, i32 n, i32* out, i32* __restrict__ total_out)
{
    __shared__ i32 wave_sums[16];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const i32 per = (n + 1023) / 1024;
    const i32 lo = (i32)threadIdx.x * per;
    i32 v[kSmallScanItems];
    i32 sum = 0;
#pragma unroll
    for (int k = 0; k < kSmallScanItems; k++) {
        v[k] = (k < per && lo + k < n) ? in[lo + k] : 0;
        sum += v[k];
    }
    i32 inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const i32 u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();   // (also: every entry has been read -- `out` may be `in`)
    i32 before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        if (w < wave) before += wave_sums[w];
        all += wave_sums[w];
    }
    i32 run = before + inc - sum;
#pragma unroll
    for (int k = 0; k < kSmallScanItems; k++) {
        if (k < per && lo + k < n) out[lo + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0 && total_out) *total_out = all;
}

size_t scan_temp_bytes(int64_t n)
{
    int64_t tiles = (n + kScanTile - 1) / kScanTile;
    int64_t tiles2 = (tiles + kScanTile - 1) / kScanTile;
    return (size_t)(tiles + tiles2 + 8) * 4;
}

// out[i] = sum(in[0..i)), *total_out = sum(in[0..n)); out may alias in; out has room for n entries.
// The caller guarantees the sum fits int32 (positions / byte offsets of one Block).
void launch_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total_out, void* temp, hipStream_t s)
{
    if (n <= 0) {
        if (total_out) PA_HIP(hipMemsetAsync(total_out, 0, 4, s));
        return;
    }
    int64_t tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles > 1 && n <= kSmallScanMax) {
        hipLaunchKernelGGL(k_scan_small, 1, 1024, 0, s, in, (i32)n, out, total_out);
        PA_HIP(hipGetLastError());
        return;
    }
    if (tiles == 1) {
        if (out != in) PA_HIP(hipMemcpyAsync(out, in, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_scan_single, 1, kScanBlock, 0, s, out, (i64)n, total_out);
        PA_HIP(hipGetLastError());
        return;
    }
    i32* sums = static_cast<i32*>(temp);
    hipLaunchKernelGGL(k_scan_tile_sums, (int)tiles, kScanBlock, 0, s, in, (i64)n, sums);
    hipLaunchKernelGGL(k_scan_single, 1, kScanBlock, 0, s, sums, (i64)tiles, total_out);
    hipLaunchKernelGGL(k_scan_apply, (int)tiles, kScanBlock, 0, s, in, (i64)n, (const i32*)sums, out);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// dictionary-aware filter: selection of the rows from the selection of the dictionary entries
// ---------------------------------------------------------------------------------------------
// kFpTileQuads = kTileQuads of op_filter_project.cpp: one workgroup covers that many x 256 row quads
__global__ __launch_bounds__(256) void k_dict_filter_sel(const i32* __restrict__ ids, const u8* __restrict__ dict_sel4, i64 n, u8* __restrict__ sel4,
                                                         i32* __restrict__ tile_counts, int kFpTileQuads)
{
    __shared__ i32 wave_sum[4];
    i32 c = 0;
    for (int j = 0; j < kFpTileQuads; j++) {
        const i64 q = ((i64)blockIdx.x * kFpTileQuads + j) * 256 + threadIdx.x, row0 = q << 2;
        u32 bits = 0;
        for (int i = 0; i < 4; i++) {
            const i64 r = row0 + i;
            if (r < n) {
                const u32 id = ids ? (u32)ids[r] : 0u;
                if ((dict_sel4[id >> 2] >> (id & 3u)) & 1u) bits |= 1u << i;
            }
        }
        if (row0 < n) sel4[q] = (u8)bits;
        c += (i32)__popc(bits);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

void launch_dict_filter_sel(const int32_t* ids, const uint8_t* dict_sel4, int64_t n, uint8_t* sel4, int32_t* tile_counts, int tile_quads, hipStream_t s)
{
    if (n <= 0) return;
    const int64_t tile_rows = 1024 * (int64_t)tile_quads;
    hipLaunchKernelGGL(k_dict_filter_sel, (int)((n + tile_rows - 1) / tile_rows), 256, 0, s, ids, dict_sel4, (i64)n, sel4, tile_counts, tile_quads);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// VariableWidthBlock.copyPositions: lengths -> (scan) -> bytes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_varwidth_lengths(const i32* __restrict__ positions, i64 count, const i32* __restrict__ offsets,
                                                          const u8* __restrict__ nulls, i32* __restrict__ out_len)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < count; i += (i64)gridDim.x * 256) {
        i32 p = positions ? positions[i] : (i32)i;
        out_len[i] = (p < 0 || (nulls && nulls[p])) ? 0 : offsets[p + 1] - offsets[p];  // p < 0: NULL-extended row of an outer join
    }
}
__global__ __launch_bounds__(256) void k_varwidth_copy(const i32* __restrict__ positions, i64 count, const i32* __restrict__ offsets,
                                                       const u8* __restrict__ bytes, const u8* __restrict__ nulls,
                                                       i32* __restrict__ out_offsets, u8* __restrict__ out_bytes, i32* total)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < count; i += (i64)gridDim.x * 256) {
        i32 p = positions ? positions[i] : (i32)i;
        i32 len = (p < 0 || (nulls && nulls[p])) ? 0 : offsets[p + 1] - offsets[p];
        const u8* src = bytes + (p < 0 ? 0 : offsets[p]);
        u8* dst = out_bytes + out_offsets[i];
        i32 b = 0;
        for (; b + 8 <= len; b += 8) {  // unaligned 8-byte moves (gfx950 allows them), then the tail
            u64 w;
            __builtin_memcpy(&w, src + b, 8);
            __builtin_memcpy(dst + b, &w, 8);
        }
        for (; b < len; b++) dst[b] = src[b];
        if (i == count - 1) out_offsets[count] = *total;
    }
}

void launch_varwidth_lengths(const int32_t* positions, int64_t count, const int32_t* offsets, const uint8_t* nulls, int32_t* out_len,
                             hipStream_t s)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_varwidth_lengths, blocks_for(count, 256) > 2048 ? 2048 : blocks_for(count, 256), 256, 0, s, positions, (i64)count, offsets,
                       nulls, out_len);
    PA_HIP(hipGetLastError());
}
void launch_varwidth_copy(const int32_t* positions, int64_t count, const int32_t* offsets, const uint8_t* bytes, const uint8_t* nulls,
                          int32_t* out_offsets, uint8_t* out_bytes, int32_t* total, hipStream_t s)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_varwidth_copy, blocks_for(count, 256) > 2048 ? 2048 : blocks_for(count, 256), 256, 0, s, positions, (i64)count, offsets,
                       bytes, nulls, out_offsets, out_bytes, total);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Stable partition of positions by partition id (PartitioningExchanger.java:59-82): positions of
// partition p keep ascending order; partitions are laid out one after the other.
// counts[p * tiles + t] = rows of partition p in tile t  ->  exclusive scan  ->  scatter
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_partition_count(const i32* __restrict__ part, i64 n, i32 P, i64 tiles, i32* __restrict__ counts)
{
    __shared__ i32 hist[1024];
    for (int i = threadIdx.x; i < P; i += 256) hist[i] = 0;
    __syncthreads();
    i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * kScanItems;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        if (base + i < n) atomicAdd(&hist[part[base + i]], 1);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += 256) counts[(i64)p * tiles + blockIdx.x] = hist[p];
}

__global__ __launch_bounds__(256) void k_partition_scatter(const i32* __restrict__ part, i64 n, i32 P, i64 tiles,
                                                           const i32* __restrict__ offsets, i32* __restrict__ out_positions)
{
    // Stable rank of a row inside its (tile, partition) = rows of the same partition before it in the tile.  Cost independent
    // of P (the radix passes of OrderBy and the hash-partitioned aggregation use P = 256 .. 1024): the tile is walked in four
    // slots of 256 consecutive rows; inside a slot a lane finds the lanes of its wave with the same partition by matching the
    // id bit by bit (ballots), the first of them publishes the wave's count for that partition in LDS, and the counts of the
    // earlier waves plus the running total of the earlier slots give the rank.
    __shared__ i32 running[1024];      // rows of partition p in the slots done so far
    __shared__ i32 wave_count[4][1024];
    for (int p = threadIdx.x; p < P; p += 256) {
        running[p] = 0;
        wave_count[0][p] = wave_count[1][p] = wave_count[2][p] = wave_count[3][p] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 tile0 = (i64)blockIdx.x * kScanTile;
    for (int slot = 0; slot < kScanItems; slot++) {
        const i64 row = tile0 + (i64)slot * 256 + threadIdx.x;
        const bool live = row < n;
        const i32 p = live ? part[row] : -1;
        // lanes of this wave with the same partition id
        u64 peers = __ballot(live);
        for (int bit = 0; bit < 10; bit++) {
            const u64 b = __ballot(live && ((p >> bit) & 1));
            peers &= ((p >> bit) & 1) ? b : ~b;
        }
        const int before = __popcll(peers & ((1ULL << lane) - 1ULL));
        const bool leader = live && before == 0;
        if (leader) wave_count[wave][p] = (i32)__popcll(peers);
        __syncthreads();
        if (live) {
            i32 rank = running[p] + before;
            for (int w = 0; w < wave; w++) rank += wave_count[w][p];
            out_positions[offsets[(i64)p * tiles + blockIdx.x] + rank] = (i32)row;
        }
        __syncthreads();
        if (leader) {
            atomicAdd(&running[p], wave_count[wave][p]);
            wave_count[wave][p] = 0;
        }
        __syncthreads();
    }
}

__global__ void k_partition_totals(const i32* __restrict__ offsets, i64 tiles, i32 P, i64 n, i64* __restrict__ out_counts)
{
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        i64 start = offsets[(i64)p * tiles];
        i64 end = (p + 1 < P) ? (i64)offsets[(i64)(p + 1) * tiles] : n;
        out_counts[p] = end - start;
    }
}

constexpr int kPart2Blocks = 2048;  // chunks of the two-partition form below
size_t partition_temp_bytes(int64_t n, int32_t partition_count)
{
    int64_t tiles = (n + kScanTile - 1) / kScanTile;
    return std::max<size_t>((size_t)(tiles * partition_count) * 4 + scan_temp_bytes(tiles * partition_count) + 64, (size_t)kPart2Blocks * 4 + 64);
}

// Two partitions (selected / not selected: TopN's survivors, a join's unvisited rows, NULL / non-NULL positions of the page serde): the
// general kernels histogram every tile in LDS and scan a (tile x partition) matrix -- 0.1 ms over 11 M rows in Q3's TopN.  Here a
// workgroup owns a contiguous chunk: zeros counted with ballots, one count per chunk, and the second pass ranks 256 rows at a time.
__global__ __launch_bounds__(256) void k_part2_count(const i32* __restrict__ part, i64 n, i64 chunk, i32* __restrict__ block_zeros)
{
    __shared__ i32 wz[4];
    const i64 c0 = (i64)blockIdx.x * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
    i32 z = 0;
    for (i64 i = c0 + threadIdx.x; i < c1; i += 256) z += part[i] == 0 ? 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) z += __shfl_xor(z, d, 64);
    if ((threadIdx.x & 63) == 0) wz[threadIdx.x >> 6] = z;
    __syncthreads();
    if (threadIdx.x == 0) block_zeros[blockIdx.x] = wz[0] + wz[1] + wz[2] + wz[3];
}
__global__ __launch_bounds__(256) void k_part2_scatter(const i32* __restrict__ part, i64 n, i64 chunk, const i32* __restrict__ block_zeros, int blocks,
                                                       i32* __restrict__ out, i64* __restrict__ counts)
{
    __shared__ i64 red[2][4];
    __shared__ i32 wz[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i64 before = 0, total = 0;
    for (int b = threadIdx.x; b < blocks; b += 256) {
        const i64 z = (i64)block_zeros[b];
        total += z;
        if (b < (int)blockIdx.x) before += z;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        before += (i64)__shfl_xor((long long)before, d, 64);
        total += (i64)__shfl_xor((long long)total, d, 64);
    }
    if (lane == 0) {
        red[0][wave] = before;
        red[1][wave] = total;
    }
    __syncthreads();
    before = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counts[0] = total;
        counts[1] = n - total;
    }
    const i64 c0 = (i64)blockIdx.x * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
    i64 zpos = before, opos = total + (c0 - before);
    for (i64 t = c0; t < c1; t += 256) {
        const i64 i = t + threadIdx.x;
        const bool valid = i < c1;
        const bool zero = valid && part[i] == 0;
        const u64 m = __ballot(zero);
        if (lane == 0) wz[wave] = (i32)__popcll(m);
        __syncthreads();
        i32 zeros_before = (i32)__popcll(m & ((1ULL << lane) - 1ULL)), tile_zeros = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w < wave) zeros_before += wz[w];
            tile_zeros += wz[w];
        }
        if (zero) out[zpos + zeros_before] = (i32)i;
        else if (valid) out[opos + ((i32)threadIdx.x - zeros_before)] = (i32)i;
        const i64 rows = c1 - t < 256 ? c1 - t : 256;
        zpos += tile_zeros;
        opos += rows - tile_zeros;
        __syncthreads();
    }
}

void launch_partition_positions(const int32_t* partition, int64_t n, int32_t partition_count, int32_t* out_positions,
                                int64_t* out_counts_dev, void* temp, hipStream_t s)
{
    PA_REQUIRE(partition_count >= 1 && partition_count <= 1024, PA_ERR_NOT_SUPPORTED, "1..1024 partitions");
    if (n <= 0) {
        PA_HIP(hipMemsetAsync(out_counts_dev, 0, (size_t)partition_count * 8, s));
        return;
    }
    if (partition_count == 2) {
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(kPart2Blocks, (n + 1023) / 1024));
        const int64_t chunk = (((n + blocks - 1) / blocks) + 255) & ~(int64_t)255;
        i32* block_zeros = static_cast<i32*>(temp);
        hipLaunchKernelGGL(k_part2_count, blocks, 256, 0, s, partition, (i64)n, (i64)chunk, block_zeros);
        hipLaunchKernelGGL(k_part2_scatter, blocks, 256, 0, s, partition, (i64)n, (i64)chunk, (const i32*)block_zeros, blocks, out_positions,
                           (i64*)out_counts_dev);
        PA_HIP(hipGetLastError());
        return;
    }
    int64_t tiles = (n + kScanTile - 1) / kScanTile;
    i32* counts = static_cast<i32*>(temp);
    void* scan_temp = counts + tiles * partition_count;
    hipLaunchKernelGGL(k_partition_count, (int)tiles, 256, 0, s, partition, (i64)n, partition_count, (i64)tiles, counts);
    launch_exclusive_scan_i32(counts, counts, tiles * partition_count, nullptr, scan_temp, s);
    hipLaunchKernelGGL(k_partition_scatter, (int)tiles, 256, 0, s, partition, (i64)n, partition_count, (i64)tiles, (const i32*)counts, out_positions);
    hipLaunchKernelGGL(k_partition_totals, 1, 1024, 0, s, (const i32*)counts, (i64)tiles, partition_count, (i64)n, (i64*)out_counts_dev);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// multisplit of columns (see scan_kernels.hpp)
// ---------------------------------------------------------------------------------------------
constexpr int kMsThreads = 1024, kMsItems = 8, kMsTile = kMsThreads * kMsItems;
struct MsplitArgs {
    const i32* part;
    i64 n, tiles;
    const i32* offsets;  // exclusive scan of the (partition, tile) counts, partition-major
    i32 P, ncols;
    MsplitCol col[kMsplitMaxCols];
};

constexpr int kMsMaxParts = 4096;  // unstable form; the stable form takes 256

// ---- the unstable form (round 3) ---------------------------------------------------------------------------------------------
// Counts are TILE-major (counts[tile][partition]): a tile's workgroup writes and later reads its P values as one coalesced run --
// with the partition-major layout every (tile, partition) count was a 64-byte line request of its own, as many requests per tile
// as the tile's data (2049 partitions: 4098 strided loads per scatter tile next to 2560 data lines).  The offsets are the
// column-wise exclusive scan of that matrix plus the partitions' bases: three small kernels, coalesced across the partitions.
// The scatter ranks a tile's rows with returning LDS atomics on its own histogram (so it needs no counts, only where each
// partition's run of this tile starts) and has the ids and the first columns in registers before the first barrier: one HBM round
// trip per tile instead of one per column.
template <int MAXP>
__global__ __launch_bounds__(kMsThreads) void k_msplit_count_tm(const i32* __restrict__ part, i64 n, i32 P, i32* __restrict__ counts)
{
    __shared__ i32 hist[MAXP + 1];
    for (int i = threadIdx.x; i < P; i += kMsThreads) hist[i] = 0;
    __syncthreads();
    const i64 tile0 = (i64)blockIdx.x * kMsTile;
    i32 pid[kMsItems];
#pragma unroll
    for (int i = 0; i < kMsItems; i++) {
        const i64 row = tile0 + (i64)i * kMsThreads + threadIdx.x;
        pid[i] = row < n ? part[row] : -1;
    }
#pragma unroll
    for (int i = 0; i < kMsItems; i++) {
        if (pid[i] >= 0) atomicAdd(&hist[pid[i]], 1);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += kMsThreads) counts[(i64)blockIdx.x * P + p] = hist[p];
}

constexpr int kMsChunkTiles = 64;  // tiles per chunk of the column-wise scan
// per (chunk of tiles, partition): the rows of the chunk's tiles
__global__ __launch_bounds__(256) void k_msplit_chunk_sums(const i32* __restrict__ counts, i64 tiles, i32 P, i32* __restrict__ chunk_sums)
{
    const i32 p = (i32)(blockIdx.x * 256 + threadIdx.x);
    if (p >= P) return;
    const i64 t0 = (i64)blockIdx.y * kMsChunkTiles, t1 = t0 + kMsChunkTiles < tiles ? t0 + kMsChunkTiles : tiles;
    i32 sum = 0;
    for (i64 t = t0; t < t1; t++) sum += counts[t * P + p];
    chunk_sums[(i64)blockIdx.y * P + p] = sum;
}
// per partition: the exclusive scan over the chunks (in place) and the partition's total
__global__ __launch_bounds__(256) void k_msplit_chunk_scan(i32* __restrict__ chunk_sums, i64 chunks, i32 P, i32* __restrict__ totals, i64* __restrict__ out_counts)
{
    const i32 p = (i32)(blockIdx.x * 256 + threadIdx.x);
    if (p >= P) return;
    i32 run = 0;
#pragma unroll 8
    for (i64 c = 0; c < chunks; c++) {
        const i32 v = chunk_sums[c * P + p];
        chunk_sums[c * P + p] = run;
        run += v;
    }
    totals[p] = run;
    if (out_counts) out_counts[p] = run;
}
// one workgroup: exclusive scan of the partitions' totals -> where each partition starts in the output
__global__ __launch_bounds__(kMsThreads) void k_msplit_part_base(const i32* __restrict__ totals, i32 P, i32* __restrict__ base)
{
    __shared__ i32 wave_sums[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int PER = (P + kMsThreads - 1) / kMsThreads;
    i32 t[4] = {0, 0, 0, 0};
    i32 mine = 0;
    for (int j = 0; j < PER; j++) {
        const i32 p = (i32)threadIdx.x * PER + j;
        if (p < P) {
            t[j] = totals[p];
            mine += t[j];
        }
    }
    i32 inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const i32 o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    i32 run = inc - mine;
    for (int w = 0; w < wave; w++) run += wave_sums[w];
    for (int j = 0; j < PER; j++) {
        const i32 p = (i32)threadIdx.x * PER + j;
        if (p < P) {
            base[p] = run;
            run += t[j];
        }
    }
}
// counts[tile][p] -> where the tile's rows of partition p start in the output
__global__ __launch_bounds__(256) void k_msplit_offsets(i32* __restrict__ counts, i64 tiles, i32 P, const i32* __restrict__ chunk_base, const i32* __restrict__ base)
{
    const i32 p = (i32)(blockIdx.x * 256 + threadIdx.x);
    if (p >= P) return;
    const i64 t0 = (i64)blockIdx.y * kMsChunkTiles, t1 = t0 + kMsChunkTiles < tiles ? t0 + kMsChunkTiles : tiles;
    i32 run = base[p] + chunk_base[(i64)blockIdx.y * P + p];
#pragma unroll 8
    for (i64 t = t0; t < t1; t++) {
        const i32 v = counts[t * P + p];
        counts[t * P + p] = run;
        run += v;
    }
}

constexpr int kMsRegCols = 3;  // columns whose tile values are loaded before the first barrier
template <int MAXP>
__global__ __launch_bounds__(kMsThreads) void k_msplit_scatter_tm(MsplitArgs a)
{
    __shared__ i32 goff[MAXP + 1], lstart[MAXP + 1];
    __shared__ i32 wave_sums[16];
    __shared__ unsigned short lpart[kMsTile];
    __shared__ u64 buf[kMsTile];
    // Consecutive workgroup ids go round-robin to the 8 XCDs: give the workgroups of one XCD CONSECUTIVE tiles.  The runs two
    // neighbouring tiles write for a partition are neighbours in the output (a run of a 2048-way split is ~4 rows, half a 64-byte
    // line): tiles in flight on one XCD then complete each other's lines in that XCD's L2 before they are written back.
    const i64 tile = (gridDim.x & 7u) == 0u ? (i64)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (i64)blockIdx.x;
    const i64 tile0 = tile * kMsTile;
    const i32 tile_rows = (i32)(a.n - tile0 < (i64)kMsTile ? a.n - tile0 : (i64)kMsTile);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // everything the tile reads from HBM, issued together: the ids, the first columns, the tile's offsets
    i32 pid[kMsItems];
    u64 val[kMsRegCols][kMsItems];
#pragma unroll
    for (int i = 0; i < kMsItems; i++) {
        const i64 row = tile0 + (i64)i * kMsThreads + threadIdx.x;
        pid[i] = row < a.n ? a.part[row] : -1;
    }
#pragma unroll
    for (int c = 0; c < kMsRegCols; c++) {
        if (c < a.ncols) {
            const MsplitCol col = a.col[c];
#pragma unroll
            for (int i = 0; i < kMsItems; i++) {
                const i64 row = tile0 + (i64)i * kMsThreads + threadIdx.x;
                u64 v = 0;
                if (row < a.n) {
                    if (col.width == 8) v = ((const u64*)col.in)[row];
                    else if (col.width == 4) v = ((const u32*)col.in)[row];
                    else v = ((const u8*)col.in)[row];
                }
                val[c][i] = v;
            }
        }
    }
    for (int p = threadIdx.x; p < a.P; p += kMsThreads) {
        goff[p] = a.offsets[tile * a.P + p];
        lstart[p] = 0;
    }
    __syncthreads();
    // rank of every row inside its (tile, partition); lstart counts meanwhile
    unsigned short li[kMsItems];
#pragma unroll
    for (int i = 0; i < kMsItems; i++) li[i] = pid[i] >= 0 ? (unsigned short)atomicAdd(&lstart[pid[i]], 1) : (unsigned short)0;
    __syncthreads();
    // counts -> where each partition starts inside the tile-sorted order; a thread owns PER consecutive partitions
    const int PER = (a.P + kMsThreads - 1) / kMsThreads;
    i32 cnt[4] = {0, 0, 0, 0};
    i32 mine = 0;
    for (int j = 0; j < PER; j++) {
        const i32 p = (i32)threadIdx.x * PER + j;
        if (p < a.P) {
            cnt[j] = lstart[p];
            mine += cnt[j];
        }
    }
    i32 inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const i32 o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    i32 run = inc - mine;
    for (int w = 0; w < wave; w++) run += wave_sums[w];
    for (int j = 0; j < PER; j++) {
        const i32 p = (i32)threadIdx.x * PER + j;
        if (p < a.P) {
            lstart[p] = run;
            run += cnt[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kMsItems; i++) {
        if (pid[i] >= 0) {
            li[i] = (unsigned short)(lstart[pid[i]] + li[i]);
            lpart[li[i]] = (unsigned short)pid[i];
        }
    }
    __syncthreads();
    for (int c = 0; c < a.ncols; c++) {
        const MsplitCol col = a.col[c];
        if (c < kMsRegCols) {
#pragma unroll
            for (int i = 0; i < kMsItems; i++) {
                if (pid[i] < 0) continue;
                // (c < kMsRegCols is uniform: the compiler keeps val in registers only when the index is a literal)
                u64 v = 0;
#pragma unroll
                for (int cc = 0; cc < kMsRegCols; cc++) v = cc == c ? val[cc][i] : v;
                if (col.width == 8) buf[li[i]] = v;
                else if (col.width == 4) ((u32*)buf)[li[i]] = (u32)v;
                else ((u8*)buf)[li[i]] = (u8)v;
            }
        }
        else {
#pragma unroll
            for (int i = 0; i < kMsItems; i++) {
                const i64 row = tile0 + (i64)i * kMsThreads + threadIdx.x;
                if (pid[i] < 0) continue;
                if (col.width == 8) buf[li[i]] = ((const u64*)col.in)[row];
                else if (col.width == 4) ((u32*)buf)[li[i]] = ((const u32*)col.in)[row];
                else ((u8*)buf)[li[i]] = ((const u8*)col.in)[row];
            }
        }
        __syncthreads();
        for (i32 j = threadIdx.x; j < tile_rows; j += kMsThreads) {
            const i32 p = lpart[j];
            const i64 dest = (i64)goff[p] + (j - lstart[p]);
            if (col.width == 8) ((u64*)col.out)[dest] = buf[j];
            else if (col.width == 4) ((u32*)col.out)[dest] = ((const u32*)buf)[j];
            else ((u8*)col.out)[dest] = ((const u8*)buf)[j];
        }
        __syncthreads();
    }
}

// partition-major counts of the stable form (at most 256 partitions)
__global__ __launch_bounds__(kMsThreads) void k_msplit_count_pm(const i32* __restrict__ part, i64 n, i32 P, i64 tiles, i32* __restrict__ counts)
{
    __shared__ i32 hist[256];
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    const i64 tile0 = (i64)blockIdx.x * kMsTile;
#pragma unroll
    for (int i = 0; i < kMsItems; i++) {
        const i64 row = tile0 + (i64)i * kMsThreads + threadIdx.x;
        if (row < n) atomicAdd(&hist[part[row]], 1);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += kMsThreads) counts[(i64)p * tiles + blockIdx.x] = hist[p];
}

// The stable form (at most 256 partitions): a row's place inside its (tile, partition) is the number of earlier rows of the tile
// with the same partition -- wave ballots and per-wave counts -- so every partition keeps ascending row order,
// what the exchange needs (PartitioningExchanger appends positions in order).
__global__ __launch_bounds__(kMsThreads) void k_msplit_scatter_stable(MsplitArgs a)
{
    __shared__ i32 goff[256], lstart[256], running[256];
    __shared__ i32 cnt[16][256], off[16][256];
    __shared__ i32 wave_sums[16];
    __shared__ u8 ldigit[kMsTile];
    __shared__ u64 buf[kMsTile];
    const i64 tile0 = (i64)blockIdx.x * kMsTile;
    const i32 tile_rows = (i32)(a.n - tile0 < (i64)kMsTile ? a.n - tile0 : (i64)kMsTile);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32 mine = 0;
    if (threadIdx.x < 256) {
        goff[threadIdx.x] = 0;
        running[threadIdx.x] = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) cnt[w][threadIdx.x] = 0;
        if ((i32)threadIdx.x < a.P) {
            const i64 idx = (i64)threadIdx.x * a.tiles + blockIdx.x;
            const i32 o = a.offsets[idx];
            const i32 nx = idx + 1 < (i64)a.P * a.tiles ? a.offsets[idx + 1] : (i32)a.n;
            goff[threadIdx.x] = o;
            mine = nx - o;
        }
    }
    i32 inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const i32 v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    if (threadIdx.x < 256) {
        i32 base = 0;
        for (int w = 0; w < wave; w++) base += wave_sums[w];
        lstart[threadIdx.x] = base + inc - mine;
    }
    __syncthreads();
    unsigned short li[kMsItems];
    for (int i = 0; i < kMsItems; i++) {
        const i64 row = tile0 + (i64)i * kMsThreads + threadIdx.x;
        const bool live = row < a.n;
        const u32 d = live ? (u32)a.part[row] & 255u : 0u;
        u64 peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const u64 b = __ballot(live && ((d >> bit) & 1u));
            peers &= ((d >> bit) & 1u) ? b : ~b;
        }
        const int before = __popcll(peers & ((1ULL << lane) - 1ULL));
        if (live && before == 0) cnt[wave][d] = (i32)__popcll(peers);
        __syncthreads();
        if (threadIdx.x < 256) {
            i32 acc = running[threadIdx.x];
#pragma unroll
            for (int w = 0; w < 16; w++) {
                const i32 c = cnt[w][threadIdx.x];
                cnt[w][threadIdx.x] = 0;
                off[w][threadIdx.x] = acc;
                acc += c;
            }
            running[threadIdx.x] = acc;
        }
        __syncthreads();
        li[i] = 0;
        if (live) {
            const i32 at = lstart[d] + off[wave][d] + before;
            li[i] = (unsigned short)at;
            ldigit[at] = (u8)d;
        }
    }
    __syncthreads();
    for (int c = 0; c < a.ncols; c++) {
        const MsplitCol col = a.col[c];
        for (int i = 0; i < kMsItems; i++) {
            const i64 row = tile0 + (i64)i * kMsThreads + threadIdx.x;
            if (row < a.n) {
                if (col.width == 8) buf[li[i]] = ((const u64*)col.in)[row];
                else if (col.width == 4) ((u32*)buf)[li[i]] = ((const u32*)col.in)[row];
                else ((u8*)buf)[li[i]] = ((const u8*)col.in)[row];
            }
        }
        __syncthreads();
        for (i32 j = threadIdx.x; j < tile_rows; j += kMsThreads) {
            const i32 d = ldigit[j];
            const i64 dest = (i64)goff[d] + (j - lstart[d]);
            if (col.width == 8) ((u64*)col.out)[dest] = buf[j];
            else if (col.width == 4) ((u32*)col.out)[dest] = ((const u32*)buf)[j];
            else ((u8*)col.out)[dest] = ((const u8*)buf)[j];
        }
        __syncthreads();
    }
}

size_t msplit_temp_bytes(int64_t n, int32_t partition_count)
{
    const int64_t tiles = (n + kMsTile - 1) / kMsTile;
    const int64_t chunks = (tiles + kMsChunkTiles - 1) / kMsChunkTiles;
    // (the stable form scans the whole matrix; the unstable one keeps per-chunk sums behind it)
    return (size_t)(tiles * partition_count) * 4 + std::max(scan_temp_bytes(tiles * partition_count), (size_t)((chunks + 2) * partition_count) * 4) + 64;
}
int64_t msplit_tiles(int64_t n) { return (n + kMsTile - 1) / kMsTile; }
int32_t* msplit_counts(void* temp) { return static_cast<i32*>(temp); }

void launch_msplit(const int32_t* partition, int64_t n, int32_t partition_count, const MsplitCol* cols, int32_t ncols, int64_t* out_counts_dev,
                   void* temp, hipStream_t s, bool stable, bool counts_ready)
{
    PA_REQUIRE(partition_count >= 1 && partition_count <= (stable ? 256 : kMsMaxParts), PA_ERR_NOT_SUPPORTED, "1..4096 partitions (stable: 1..256)");
    PA_REQUIRE(ncols >= 0 && ncols <= kMsplitMaxCols, PA_ERR_NOT_SUPPORTED, "too many columns for one multisplit");
    PA_REQUIRE(!(stable && counts_ready), PA_ERR_INVALID_ARGUMENT, "internal: precomputed counts are tile-major, the stable form's are not");
    if (n <= 0) {
        PA_HIP(hipMemsetAsync(out_counts_dev, 0, (size_t)partition_count * 8, s));
        return;
    }
    const int64_t tiles = (n + kMsTile - 1) / kMsTile;
    i32* counts = static_cast<i32*>(temp);
    MsplitArgs a;
    memset(&a, 0, sizeof a);
    a.part = partition;
    a.n = n;
    a.tiles = tiles;
    a.offsets = counts;
    a.P = partition_count;
    a.ncols = ncols;
    for (int c = 0; c < ncols; c++) {
        PA_REQUIRE(cols[c].width == 1 || cols[c].width == 4 || cols[c].width == 8, PA_ERR_NOT_SUPPORTED, "multisplit moves 1, 4 or 8 byte elements");
        a.col[c] = cols[c];
    }
    if (stable) {
        void* scan_temp = counts + tiles * partition_count;
        hipLaunchKernelGGL(k_msplit_count_pm, (int)tiles, kMsThreads, 0, s, partition, (i64)n, partition_count, (i64)tiles, counts);
        launch_exclusive_scan_i32(counts, counts, tiles * partition_count, nullptr, scan_temp, s);
        hipLaunchKernelGGL(k_msplit_scatter_stable, (int)tiles, kMsThreads, 0, s, a);
        hipLaunchKernelGGL(k_partition_totals, 1, 1024, 0, s, (const i32*)counts, (i64)tiles, partition_count, (i64)n, (i64*)out_counts_dev);
        PA_HIP(hipGetLastError());
        return;
    }
    // Up to 1024 partitions the per-partition LDS arrays are a quarter the size
    const bool big = partition_count > 1024;
    if (!counts_ready) {
        if (big) hipLaunchKernelGGL((k_msplit_count_tm<kMsMaxParts>), (int)tiles, kMsThreads, 0, s, partition, (i64)n, partition_count, counts);
        else hipLaunchKernelGGL((k_msplit_count_tm<1024>), (int)tiles, kMsThreads, 0, s, partition, (i64)n, partition_count, counts);
    }
    const int64_t chunks = (tiles + kMsChunkTiles - 1) / kMsChunkTiles;
    i32* chunk_sums = counts + tiles * partition_count;
    const dim3 grid((unsigned)((partition_count + 255) / 256), (unsigned)chunks);
    hipLaunchKernelGGL(k_msplit_chunk_sums, grid, 256, 0, s, (const i32*)counts, (i64)tiles, partition_count, chunk_sums);
    i32* totals = chunk_sums + chunks * partition_count;
    i32* base = totals + partition_count;
    hipLaunchKernelGGL(k_msplit_chunk_scan, (partition_count + 255) / 256, 256, 0, s, chunk_sums, (i64)chunks, partition_count, totals, (i64*)out_counts_dev);
    hipLaunchKernelGGL(k_msplit_part_base, 1, kMsThreads, 0, s, (const i32*)totals, partition_count, base);
    hipLaunchKernelGGL(k_msplit_offsets, grid, 256, 0, s, counts, (i64)tiles, partition_count, (const i32*)chunk_sums, (const i32*)base);
    if (big) hipLaunchKernelGGL((k_msplit_scatter_tm<kMsMaxParts>), (int)tiles, kMsThreads, 0, s, a);
    else hipLaunchKernelGGL((k_msplit_scatter_tm<1024>), (int)tiles, kMsThreads, 0, s, a);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// OrderBy helpers
// ---------------------------------------------------------------------------------------------
// OR and AND of the keys, one pair per workgroup (out[2 b], out[2 b + 1]); the host folds the pairs it reads back anyway -- a pair of
// same-address atomics per wave (8192 of them) took 120 us over 2^24 keys, four times the pass itself
constexpr int kOrAndBlocks = 1024;
__global__ __launch_bounds__(256) void k_key_or_and(const u64* __restrict__ keys, i64 n, u64* __restrict__ out)
{
    __shared__ u64 s_o[4], s_a[4];
    u64 o = 0ULL, a = ~0ULL;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const u64 k = keys[i];
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

size_t key_or_and_bytes() { return (size_t)kOrAndBlocks * 16; }
int launch_key_or_and(const uint64_t* keys, int64_t n, uint64_t* out, hipStream_t s)
{
    if (n <= 0) return 0;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, kOrAndBlocks));
    hipLaunchKernelGGL(k_key_or_and, blocks, 256, 0, s, (const u64*)keys, (i64)n, (u64*)out);
    PA_HIP(hipGetLastError());
    return blocks;
}

__global__ __launch_bounds__(256) void k_offsets_append(const i32* __restrict__ src, i64 count, i32 dst_base, i32* __restrict__ dst, int write_first)
{
    const i32 first = src[0];
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < count; i += (i64)gridDim.x * 256) dst[i + 1] = dst_base + (src[i + 1] - first);
    if (write_first && blockIdx.x == 0 && threadIdx.x == 0) dst[0] = dst_base;
}
void launch_offsets_append(const int32_t* src, int64_t count, int32_t dst_base, int32_t* dst, bool write_first, hipStream_t s)
{
    if (count <= 0) return;
    int grid = (int)std::min<int64_t>((count + 255) / 256, 2048);
    hipLaunchKernelGGL(k_offsets_append, grid, 256, 0, s, (const i32*)src, (i64)count, (i32)dst_base, (i32*)dst, write_first ? 1 : 0);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// page serde helpers (PagesSerde block encodings)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_null_bits(const u8* __restrict__ nulls, i64 n, u8* __restrict__ packed)
{
    const i64 bytes = (n + 7) >> 3;
    for (i64 b = (i64)blockIdx.x * 256 + threadIdx.x; b < bytes; b += (i64)gridDim.x * 256) {
        u32 v = 0;
        for (int k = 0; k < 8; k++) {
            const i64 p = b * 8 + k;
            if (p < n && nulls[p]) v |= 0x80u >> k;
        }
        packed[b] = (u8)v;
    }
}
__global__ __launch_bounds__(256) void k_unpack_null_bits(const u8* __restrict__ packed, i64 n, u8* __restrict__ nulls)
{
    for (i64 p = (i64)blockIdx.x * 256 + threadIdx.x; p < n; p += (i64)gridDim.x * 256) nulls[p] = (packed[p >> 3] >> (7 - (p & 7))) & 1u;
}
__global__ __launch_bounds__(256) void k_null_flag(const u8* __restrict__ nulls, i64 n, i32* __restrict__ partition)
{
    for (i64 p = (i64)blockIdx.x * 256 + threadIdx.x; p < n; p += (i64)gridDim.x * 256) partition[p] = nulls[p] ? 1 : 0;
}
template <typename T>
__global__ __launch_bounds__(256) void k_scatter(const T* __restrict__ src, const i32* __restrict__ pos, i64 n, T* __restrict__ dst)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) dst[pos[i]] = src[i];
}
__global__ __launch_bounds__(256) void k_varwidth_ends(const i32* __restrict__ offsets, i64 n, i32* __restrict__ ends)
{
    const i32 first = offsets[0];
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) ends[i] = offsets[i + 1] - first;
}
__global__ __launch_bounds__(256) void k_varwidth_from_ends(const i32* __restrict__ ends, i64 n, i32* __restrict__ offsets)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) offsets[i + 1] = ends[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[0] = 0;
}
static int serde_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048)); }
void launch_pack_null_bits(const uint8_t* nulls, int64_t n, uint8_t* packed, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pack_null_bits, serde_grid((n + 7) / 8), 256, 0, s, nulls, (i64)n, packed);
    PA_HIP(hipGetLastError());
}
void launch_unpack_null_bits(const uint8_t* packed, int64_t n, uint8_t* nulls, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unpack_null_bits, serde_grid(n), 256, 0, s, packed, (i64)n, nulls);
    PA_HIP(hipGetLastError());
}
void launch_null_flag(const uint8_t* nulls, int64_t n, int32_t* partition, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_null_flag, serde_grid(n), 256, 0, s, nulls, (i64)n, partition);
    PA_HIP(hipGetLastError());
}
void launch_scatter_flat(const void* src, int elem_bytes, const int32_t* positions, int64_t count, void* dst, hipStream_t s)
{
    if (count <= 0) return;
    const int g = serde_grid(count);
    switch (elem_bytes) {
        case 8: hipLaunchKernelGGL(k_scatter<u64>, g, 256, 0, s, (const u64*)src, positions, (i64)count, (u64*)dst); break;
        case 4: hipLaunchKernelGGL(k_scatter<u32>, g, 256, 0, s, (const u32*)src, positions, (i64)count, (u32*)dst); break;
        case 1: hipLaunchKernelGGL(k_scatter<u8>, g, 256, 0, s, (const u8*)src, positions, (i64)count, (u8*)dst); break;
        default: throw Error(PA_ERR_INVALID_ARGUMENT, "unsupported element width");
    }
    PA_HIP(hipGetLastError());
}
void launch_varwidth_ends(const int32_t* offsets, int64_t n, int32_t* ends, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_varwidth_ends, serde_grid(n), 256, 0, s, offsets, (i64)n, ends);
    PA_HIP(hipGetLastError());
}
void launch_varwidth_from_ends(const int32_t* ends, int64_t n, int32_t* offsets, hipStream_t s)
{
    hipLaunchKernelGGL(k_varwidth_from_ends, serde_grid(std::max<int64_t>(n, 1)), 256, 0, s, ends, (i64)n, offsets);
    PA_HIP(hipGetLastError());
}

}  // namespace pa
