// static_kernels.hip -- query-independent gfx950 kernels of the page-processing path: dictionary decode,
// partial-state merges, group-table rehash, row hashing / partitioning, synthetic TPC-H column
// generation.  Expression-bearing kernels are generated per query (op_fused.cpp, op_filter_project.cpp).
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "kernels/pa_device.h"
#include "static_kernels.hpp"

namespace pa {

static inline int grid_for(int64_t work, int block, int max_blocks = 256 * 8)
{
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------
// Block.copyPositions / dictionary decode
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_gather(const T* __restrict__ src, const i32* __restrict__ pos, i64 n, T* __restrict__ dst)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) dst[i] = src[pos[i]];
}
template <typename T>
__global__ __launch_bounds__(256) void k_fill(T* __restrict__ dst, const T* __restrict__ one, i64 n)
{
    T v = *one;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) dst[i] = v;
}

// Block.copyPositions where position -1 stands for a NULL row (the build side of an outer join's unmatched probe rows):
// value 0 and NULL flag 1 there; elsewhere the source value and the source's NULL flag (or 0)
template <typename T>
__global__ __launch_bounds__(256) void k_gather_or_null(const T* __restrict__ src, const u8* __restrict__ src_nulls, const i32* __restrict__ pos, i64 n,
                                                        T* __restrict__ dst, u8* __restrict__ dst_nulls)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const i32 p = pos[i];
        if (dst) dst[i] = p < 0 ? (T)0 : src[p];
        dst_nulls[i] = p < 0 ? (u8)1 : (src_nulls ? src_nulls[p] : (u8)0);
    }
}
void launch_gather_or_null(const void* src, int elem_bytes, const uint8_t* src_nulls, const int32_t* positions, int64_t count, void* dst,
                           uint8_t* dst_nulls, hipStream_t s)
{
    if (count <= 0) return;
    int g = grid_for(count, 256);
    switch (elem_bytes) {
        case 8: hipLaunchKernelGGL(k_gather_or_null<u64>, g, 256, 0, s, (const u64*)src, src_nulls, positions, count, (u64*)dst, dst_nulls); break;
        case 4: hipLaunchKernelGGL(k_gather_or_null<u32>, g, 256, 0, s, (const u32*)src, src_nulls, positions, count, (u32*)dst, dst_nulls); break;
        case 1: hipLaunchKernelGGL(k_gather_or_null<u8>, g, 256, 0, s, (const u8*)src, src_nulls, positions, count, (u8*)dst, dst_nulls); break;
        case 0: hipLaunchKernelGGL(k_gather_or_null<u8>, g, 256, 0, s, (const u8*)nullptr, src_nulls, positions, count, (u8*)nullptr, dst_nulls); break;
        default: throw Error(PA_ERR_INVALID_ARGUMENT, "unsupported element width");
    }
    PA_HIP(hipGetLastError());
}

void launch_gather_flat(const void* src, int elem_bytes, const int32_t* positions, int64_t count, void* dst, hipStream_t s)
{
    if (count <= 0) return;
    int g = grid_for(count, 256);
    switch (elem_bytes) {
        case 8: hipLaunchKernelGGL(k_gather<u64>, g, 256, 0, s, (const u64*)src, positions, count, (u64*)dst); break;
        case 4: hipLaunchKernelGGL(k_gather<u32>, g, 256, 0, s, (const u32*)src, positions, count, (u32*)dst); break;
        case 1: hipLaunchKernelGGL(k_gather<u8>, g, 256, 0, s, (const u8*)src, positions, count, (u8*)dst); break;
        default: throw Error(PA_ERR_INVALID_ARGUMENT, "gather: unsupported element width");
    }
    PA_HIP(hipGetLastError());
}
// Several flat columns gathered by ONE launch (LookupJoinPageBuilder.build: probe output channels by probe index, build output
// channels by build position): a join output page of a few million rows is a handful of small gathers, each dominated by its
// launch; a position of -1 (the build side of an unmatched probe row of an outer join) gives a zero value and, where the column
// has NULL flags to write, a NULL
__global__ __launch_bounds__(256) void k_gather_multi(GatherMultiArgs a)
{
    const i64 count = a.count_dev ? (i64)*a.count_dev : a.count;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < count; i += (i64)gridDim.x * 256) {
        const i32 p0 = a.positions[0] ? a.positions[0][i] : 0, p1 = a.positions[1] ? a.positions[1][i] : 0;
        for (int c = 0; c < a.ncols; c++) {
            const GatherMultiCol col = a.col[c];
            const i32 p = col.which ? p1 : p0;
            if (col.dst) {
                if (col.width == 8) ((u64*)col.dst)[i] = p < 0 ? 0ULL : ((const u64*)col.src)[p];
                else if (col.width == 4) ((u32*)col.dst)[i] = p < 0 ? 0u : ((const u32*)col.src)[p];
                else ((u8*)col.dst)[i] = p < 0 ? (u8)0 : ((const u8*)col.src)[p];
            }
            if (col.dst_nulls) col.dst_nulls[i] = p < 0 ? (u8)1 : (col.src_nulls ? col.src_nulls[p] : (u8)0);
        }
    }
}
void launch_gather_multi(const GatherMultiArgs& args, hipStream_t s)
{
    if (args.count <= 0 || args.ncols <= 0) return;
    hipLaunchKernelGGL(k_gather_multi, grid_for(args.count, 256), 256, 0, s, args);
    PA_HIP(hipGetLastError());
}

void launch_gather_nulls(const uint8_t* src, const int32_t* positions, int64_t count, uint8_t* dst, hipStream_t s)
{
    launch_gather_flat(src, 1, positions, count, dst, s);
}
void launch_fill_flat(void* dst, int elem_bytes, const void* src_one, int64_t count, hipStream_t s)
{
    if (count <= 0) return;
    int g = grid_for(count, 256);
    switch (elem_bytes) {
        case 8: hipLaunchKernelGGL(k_fill<u64>, g, 256, 0, s, (u64*)dst, (const u64*)src_one, count); break;
        case 4: hipLaunchKernelGGL(k_fill<u32>, g, 256, 0, s, (u32*)dst, (const u32*)src_one, count); break;
        case 1: hipLaunchKernelGGL(k_fill<u8>, g, 256, 0, s, (u8*)dst, (const u8*)src_one, count); break;
        default: throw Error(PA_ERR_INVALID_ARGUMENT, "fill: unsupported element width");
    }
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// merge of the per-workgroup partial states of the GLOBAL variant: one workgroup, word w handled by
// wave-strided lanes in a fixed order => the result does not depend on scheduling.
// state[w] (+)= sum over blocks b (ascending) of slab[b][w]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_merge_global_slab(const u64* __restrict__ slab, int blocks, int nw,
                                                           const i32* __restrict__ kinds, u64* __restrict__ state, i32* err)
{
    __shared__ u64 part[256];
    for (int w = 0; w < nw; w++) {
        const int kind = kinds[w];
        // lane t accumulates blocks t, t+256, ... in ascending order
        if (kind == PA_W_SUMF) {
            double acc = 0.0;
            for (int b = threadIdx.x; b < blocks; b += 256) acc = acc + __longlong_as_double((i64)slab[(u64)b * nw + w]);
            part[threadIdx.x] = (u64)__double_as_longlong(acc);
        }
        else if (kind == PA_W_MAXU) {
            u64 acc = 0;
            for (int b = threadIdx.x; b < blocks; b += 256) {
                const u64 v = slab[(u64)b * nw + w];
                acc = v > acc ? v : acc;
            }
            part[threadIdx.x] = acc;
        }
        else {
            i64 acc = 0;
            for (int b = threadIdx.x; b < blocks; b += 256) {
                i64 v = (i64)slab[(u64)b * nw + w];
                if (kind == PA_W_SUMI) acc = pa_add_exact(acc, v, err);
                else acc += v;
            }
            part[threadIdx.x] = (u64)acc;
        }
        __syncthreads();
        for (int stride = 128; stride >= 1; stride >>= 1) {
            if ((int)threadIdx.x < stride) {
                if (kind == PA_W_SUMF) {
                    double a = __longlong_as_double((i64)part[threadIdx.x]) + __longlong_as_double((i64)part[threadIdx.x + stride]);
                    part[threadIdx.x] = (u64)__double_as_longlong(a);
                }
                else if (kind == PA_W_SUMI) {
                    part[threadIdx.x] = (u64)pa_add_exact((i64)part[threadIdx.x], (i64)part[threadIdx.x + stride], err);
                }
                else if (kind == PA_W_MAXU) {
                    part[threadIdx.x] = part[threadIdx.x + stride] > part[threadIdx.x] ? part[threadIdx.x + stride] : part[threadIdx.x];
                }
                else {
                    part[threadIdx.x] += part[threadIdx.x + stride];
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (kind == PA_W_SUMF) state[w] = (u64)__double_as_longlong(__longlong_as_double((i64)state[w]) + __longlong_as_double((i64)part[0]));
            else if (kind == PA_W_SUMI) state[w] = (u64)pa_add_exact((i64)state[w], (i64)part[0], err);
            else if (kind == PA_W_MAXU) state[w] = part[0] > state[w] ? part[0] : state[w];
            else state[w] += part[0];
        }
        __syncthreads();
    }
}

void launch_merge_global_slab(const uint64_t* slab, int blocks, int nw, const int32_t* kinds_dev, uint64_t* state, int32_t* err,
                              hipStream_t s)
{
    hipLaunchKernelGGL(k_merge_global_slab, 1, 256, 0, s, (const u64*)slab, blocks, nw, kinds_dev, (u64*)state, err);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// merge of the per-wave partial tables of the LDS variant into the HBM group table, in two launches:
//   1. one lane per (wave, slot) entry upserts the entry's key and records its table slot;
//   2. one workgroup per occupied table slot adds the words of its entries in a fixed (lane-strided,
//      then tree) order -- no atomics on the accumulators, bitwise reproducible for a fixed grid.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_merge_lds_keys(const u64* __restrict__ slab, i64 entries, int W, int NW, u64* tag, u64* keys,
                                                        u32 mask, i32 max_fill, i32* count, i32* err, const u64* overflow_rows,
                                                        i32* __restrict__ entry_slot)
{
    // slab is field-major: field f of entry e at slab[f * entries + e]; fields = occupied, W key words, NW words
    // a launch whose register tables overflowed is discarded as a whole: the page is redone on the HBM table
    const bool discard = *overflow_rows != 0ULL;
    PaGtCtr ctr = pa_gt_ctr_init(count, false);
    for (i64 e = (i64)blockIdx.x * 256 + threadIdx.x; e < entries; e += (i64)gridDim.x * 256) {
        i32 g = -1;
        if (!discard && slab[e] != 0ULL) {
            u64 k[8];
            for (int w = 0; w < W; w++) k[w] = slab[(i64)(1 + w) * entries + e];
            g = pa_gt_upsert_n(tag, keys, mask, pa_key_hash(k, W), k, W, ctr, max_fill, err);
            if (g < 0) pa_raise(err, PA_DEV_ERR_RESOURCES);  // the host sized the table for every entry of the slab
        }
        entry_slot[e] = g;
    }
    pa_gt_ctr_flush(ctr, count);
}

__global__ __launch_bounds__(256) void k_merge_lds_words(const u64* __restrict__ slab, i64 entries, int W, int NW,
                                                         const i32* __restrict__ kinds, const u64* __restrict__ tag, u64* words, u32 cap,
                                                         const i32* __restrict__ entry_slot, i32* err)
{
    __shared__ u64 part[256];
    __shared__ u32 occupied[256];
    __shared__ u32 n_occupied;
    // one coalesced look at this workgroup's 256 table slots; the (few) occupied ones are then handled in turn
    if (threadIdx.x == 0) n_occupied = 0;
    __syncthreads();
    {
        const u32 s = blockIdx.x * 256 + threadIdx.x;
        if (s < cap && tag[s] != 0ULL) occupied[atomicAdd(&n_occupied, 1u)] = s;
    }
    __syncthreads();
    const u32 n_occ = n_occupied;
    for (u32 oi = 0; oi < n_occ; oi++) {
        // ascending slot order keeps the workgroup's work list independent of the atomicAdd arrival order
        u32 slot = 0xffffffffu;
        {
            // k-th smallest occupied slot: n_occ is tiny, a linear selection is enough
            u32 best = 0xffffffffu;
            for (u32 j = 0; j < n_occ; j++) {
                u32 cand = occupied[j];
                u32 rank = 0;
                for (u32 i2 = 0; i2 < n_occ; i2++) rank += occupied[i2] < cand ? 1u : 0u;
                if (rank == oi) best = cand;
            }
            slot = best;
        }
        // blockIdx.y = the accumulator word: the NW passes over the entries run side by side (Q1: 15 words -- one pass's latency
        // instead of fifteen); each (slot, word) is still reduced by one workgroup in one fixed order
        for (int w = blockIdx.y; w < NW; w += gridDim.y) {
            const int kind = kinds[w];
            const u64* col = slab + (i64)(1 + W + w) * entries;
            double accf = 0.0;
            i64 acci = 0;
            u64 accm = 0;
            // coalesced and branch-free: entries of other slots contribute an exact zero.  8 entries per lane are
            // loaded before the first use so that the loop is not one HBM round trip per entry.
            for (i64 e0 = threadIdx.x; e0 < entries; e0 += 256 * 8) {
                u64 v[8];
                i32 es[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const i64 e = e0 + (i64)j * 256;
                    const bool in = e < entries;
                    es[j] = in ? entry_slot[e] : -1;
                    v[j] = in ? col[e] : 0ULL;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const bool mine = es[j] == (i32)slot;
                    if (kind == PA_W_SUMF) accf = accf + (mine ? __longlong_as_double((i64)v[j]) : 0.0);
                    else if (kind == PA_W_SUMI) acci = pa_add_exact(acci, mine ? (i64)v[j] : 0, err);
                    else if (kind == PA_W_MAXU) accm = (mine && v[j] > accm) ? v[j] : accm;
                    else acci += mine ? (i64)v[j] : 0;
                }
            }
            part[threadIdx.x] = kind == PA_W_SUMF ? (u64)__double_as_longlong(accf) : (kind == PA_W_MAXU ? accm : (u64)acci);
            __syncthreads();
            for (int s = 128; s >= 1; s >>= 1) {
                if ((int)threadIdx.x < s) {
                    if (kind == PA_W_SUMF) part[threadIdx.x] = (u64)__double_as_longlong(__longlong_as_double((i64)part[threadIdx.x]) + __longlong_as_double((i64)part[threadIdx.x + s]));
                    else if (kind == PA_W_SUMI) part[threadIdx.x] = (u64)pa_add_exact((i64)part[threadIdx.x], (i64)part[threadIdx.x + s], err);
                    else if (kind == PA_W_MAXU) part[threadIdx.x] = part[threadIdx.x + s] > part[threadIdx.x] ? part[threadIdx.x + s] : part[threadIdx.x];
                    else part[threadIdx.x] += part[threadIdx.x + s];
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                u64* dst = &words[(u64)w * cap + slot];
                if (kind == PA_W_SUMF) *dst = (u64)__double_as_longlong(__longlong_as_double((i64)*dst) + __longlong_as_double((i64)part[0]));
                else if (kind == PA_W_SUMI) *dst = (u64)pa_add_exact((i64)*dst, (i64)part[0], err);
                else if (kind == PA_W_MAXU) *dst = part[0] > *dst ? part[0] : *dst;
                else *dst += part[0];
            }
            __syncthreads();
        }
    }
}

void launch_merge_lds_slab(const uint64_t* slab, int waves, int c, int w, int nw, const int32_t* kinds_dev, uint64_t* gt_tag,
                           uint64_t* gt_keys, uint64_t* gt_words, uint32_t gt_mask, int32_t gt_max_fill, int32_t* gt_count,
                           int32_t* err, const uint64_t* overflow_rows, int32_t* entry_slot, hipStream_t s)
{
    int64_t entries = (int64_t)waves * c;
    hipLaunchKernelGGL(k_merge_lds_keys, grid_for(entries, 256), 256, 0, s, (const u64*)slab, entries, w, nw, (u64*)gt_tag, (u64*)gt_keys,
                       gt_mask, gt_max_fill, gt_count, err, (const u64*)overflow_rows, entry_slot);
    hipLaunchKernelGGL(k_merge_lds_words, dim3((gt_mask + 256) / 256, (unsigned)std::max(1, nw), 1), dim3(256, 1, 1), 0, s, (const u64*)slab, entries, w, nw, kinds_dev, (const u64*)gt_tag,
                       (u64*)gt_words, gt_mask + 1, (const i32*)entry_slot, err);
    PA_HIP(hipGetLastError());
}

// Re-insert every group of the old table(s) into new ones (GroupByHash.tryRehash counterpart), folding replicas:
// blockIdx.y = old replica r, its groups go to new replica r & new_rep_mask; states of one key arriving from several old
// replicas are combined with the aggregates' combine functions (adds), so the same kernel grows a table, changes the
// number of replicas and folds all replicas into one before the result is read.
__global__ __launch_bounds__(256) void k_gt_fold(const u64* __restrict__ old_tag, const u64* __restrict__ old_keys,
                                                 const u64* __restrict__ old_words, u32 old_cap, int W, int NW, const i32* __restrict__ kinds,
                                                 u64* tag, u64* keys, u64* words, u32 mask, u32 new_rep_mask, i32* count0, i32* rep_count,
                                                 i32* err)
{
    const u64 cap = (u64)mask + 1ULL;
    const u32 r_old = blockIdx.y, r_new = r_old & new_rep_mask;
    old_tag += (u64)r_old * old_cap;
    old_keys += (u64)r_old * old_cap * (u64)W;
    old_words += (u64)r_old * old_cap * (u64)NW;
    tag += (u64)r_new * cap;
    keys += (u64)r_new * cap * (u64)W;
    words += (u64)r_new * cap * (u64)NW;
    i32* count = r_new ? rep_count + r_new : count0;
    PaGtCtr ctr = pa_gt_ctr_init(count, false);
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < (i64)old_cap; i += (i64)gridDim.x * 256) {
        if (old_tag[i] == 0ULL) continue;
        u64 k[8];
        for (int w = 0; w < W; w++) k[w] = old_keys[(u64)i * W + w];
        u32 h = pa_key_hash(k, W);
        int g = pa_gt_upsert_n(tag, keys, mask, h, k, W, ctr, 0x7fffffff, err);
        if (g < 0) {
            pa_raise(err, PA_DEV_ERR_RESOURCES);  // cannot happen: the host sized the new tables for all groups
            continue;
        }
        for (int w = 0; w < NW; w++) {
            const u64 v = old_words[(u64)w * old_cap + i];
            const u64 idx = (u64)w * cap + (u64)g;
            const int kind = kinds[w];
            if (kind == PA_W_SUMF) pa_gt_add_f64(words, idx, __longlong_as_double((i64)v));
            else if (kind == PA_W_SUMI) pa_gt_add_i64_exact(words, idx, (i64)v, err);
            else if (kind == PA_W_MAXU) pa_gt_max_u64(words, idx, v);
            else pa_gt_add_u64(words, idx, v);
        }
    }
    pa_gt_ctr_flush(ctr, count);
}

void launch_gt_fold(const uint64_t* old_tag, const uint64_t* old_keys, const uint64_t* old_words, uint32_t old_cap, uint32_t old_reps, int w,
                    int nw, const int32_t* kinds_dev, uint64_t* tag, uint64_t* keys, uint64_t* words, uint32_t mask, uint32_t new_reps,
                    int32_t* count0, int32_t* rep_count, int32_t* err, hipStream_t s)
{
    dim3 grid((unsigned)std::max(1, std::min(grid_for(old_cap, 256), 2048 / (int)std::max(1u, std::min(old_reps, 8u)))), old_reps, 1);
    hipLaunchKernelGGL(k_gt_fold, grid, dim3(256, 1, 1), 0, s, (const u64*)old_tag, (const u64*)old_keys, (const u64*)old_words, old_cap, w, nw,
                       kinds_dev, (u64*)tag, (u64*)keys, (u64*)words, mask, new_reps - 1, count0, rep_count, err);
    PA_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_fill_u64(u64* __restrict__ dst, u64 v, i64 n)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) dst[i] = v;
}
void launch_fill_u64(uint64_t* dst, uint64_t value, int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fill_u64, grid_for(n, 256), 256, 0, s, (u64*)dst, (u64)value, (i64)n);
    PA_HIP(hipGetLastError());
}

// dense (keys, words) rows of the occupied slots, for the output blocks (order = arrival at the counter)
__global__ __launch_bounds__(256) void k_gt_compact(const u64* __restrict__ tag, const u64* __restrict__ keys, const u64* __restrict__ words,
                                                    u32 cap, int W, int NW, GtStrides st, u64* __restrict__ out_keys, u64* __restrict__ out_words,
                                                    u32* counter)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < (i64)cap; i += (i64)gridDim.x * 256) {
        if (tag[(u64)i * st.tag] == st.empty) continue;
        u32 idx = atomicAdd(counter, 1u);
        for (int w = 0; w < W; w++) out_keys[(u64)idx * W + w] = keys[(u64)i * W + w];
        for (int w = 0; w < NW; w++) out_words[(u64)idx * NW + w] = words[(u64)w * st.word + (u64)i * st.slot];
    }
}
void launch_gt_compact(const uint64_t* tag, const uint64_t* keys, const uint64_t* words, uint32_t cap, int w, int nw, uint64_t* out_keys,
                       uint64_t* out_words, uint32_t* counter, hipStream_t s, const GtStrides* strides)
{
    const GtStrides st = strides ? *strides : GtStrides{1, cap, 1, 0, 0};
    hipLaunchKernelGGL(k_gt_compact, grid_for(cap, 256), 256, 0, s, (const u64*)tag, (const u64*)keys, (const u64*)words, cap, w, nw, st,
                       (u64*)out_keys, (u64*)out_words, (u32*)counter);
    PA_HIP(hipGetLastError());
}

// keys -> output blocks, states -> final values, in the pass that compacts the table (see static_kernels.hpp).
// A workgroup takes 4096 consecutive slots: occupancy bits -> workgroup scan -> ONE counter atomic per tile (atomics
// on a single address retire at only ~0.25 G/s on this part, so per-wave or per-slot counting would bound the kernel).
// one output column of the group in slot i: its value bits and NULL flag (KEY columns also feed the row's $hashvalue)
__device__ __forceinline__ void gt_emit_value(const GtEmitArgs& a, const GtEmitCol& col, u64 i, const u64* kw, const u64* wd, i64& row_hash, u64& bits, bool& is_null)
{
    bits = 0;
    is_null = false;
    switch (col.kind) {
        case GT_EMIT_KEY: {
            is_null = col.null_word >= 0 && ((kw[col.null_word] >> col.null_shift) & 1ULL);
            const u64 mask = col.bits >= 64 ? ~0ULL : ((1ULL << col.bits) - 1ULL);
            const u64 w0 = (kw[col.word] >> col.shift) & mask;
            i64 h = 0;  // NULL hashes to 0
            if (!is_null) {
                switch (col.type) {
                    case PA_BIGINT: bits = w0; h = pa_hash_bigint((i64)w0); break;
                    case PA_INTEGER:
                    case PA_DATE:
                    case PA_REAL:  // canonical float bits, hashed as the int they are (RealType.hashCodeOperator)
                        bits = (u64)(u32)w0;
                        h = col.dict_hash ? (i64)col.dict_hash[(u32)w0] : pa_hash_bigint((i64)(i32)(u32)w0);
                        break;
                    case PA_BOOLEAN: bits = w0 != 0 ? 1ULL : 0ULL; h = (i64)pa_xxh64_long(bits); break;
                    default: bits = w0; h = pa_hash_bigint((i64)w0); break;  // DOUBLE: canonical bits
                }
            }
            row_hash = pa_combine_hash(row_hash, h);
            break;
        }
        case GT_EMIT_COLUMN: {  // group key read where it lives (slot = build position); DOUBLE in the group key's canonical form
            is_null = col.src_nulls != nullptr && col.src_nulls[i] != 0;
            if (!is_null) {
                if (col.width == 8) bits = ((const u64*)col.src)[i];
                else if (col.width == 4) bits = (u64)((const u32*)col.src)[i];
                else bits = ((const u8*)col.src)[i] != 0 ? 1ULL : 0ULL;
                if (col.type == PA_DOUBLE) {
                    const double v = __longlong_as_double((i64)bits);
                    bits = v == 0.0 ? 0ULL : (v != v ? 0x7ff8000000000000ULL : bits);
                }
                else if (col.type == PA_REAL) bits = (u64)pa_real_key_bits(__uint_as_float((u32)bits));
            }
            break;
        }
        case GT_EMIT_HASH: bits = (u64)row_hash; break;
        // (a count word of -1 is the implicit count of op_fused.cpp: it counts as 1)
        case GT_EMIT_STATE: bits = col.word >= 0 ? wd[(u64)col.word * a.st.word + i * a.st.slot] : 1ULL; break;
        case GT_EMIT_COUNT: bits = wd[(u64)col.cw * a.st.word + i * a.st.slot]; break;
        case GT_EMIT_SUM:
            if (col.cw >= 0 && wd[(u64)col.cw * a.st.word + i * a.st.slot] == 0ULL) is_null = true;
            else {
                bits = wd[(u64)col.vw * a.st.word + i * a.st.slot];
                if (col.type == PA_REAL) bits = (u64)__float_as_uint((float)__longlong_as_double((i64)bits));  // RealSumAggregation.output: (float) sum
            }
            break;
        case GT_EMIT_MINMAX: {
            if (col.cw >= 0 && wd[(u64)col.cw * a.st.word + i * a.st.slot] == 0ULL) is_null = true;
            else {
                u64 img = wd[(u64)col.vw * a.st.word + i * a.st.slot];
                if (col.shift) img = ~img;  // min is kept as the maximum of the complement
                if (col.type == PA_DOUBLE) bits = pa_unimg_f64_bits(img);
                else if (col.type == PA_REAL) bits = (u64)__float_as_uint((float)__longlong_as_double((i64)pa_unimg_f64_bits(img)));
                else if (col.type == PA_BOOLEAN) bits = img;
                else bits = (u64)pa_unimg_i64(img);
            }
            break;
        }
        case GT_EMIT_AVG: {
            const i64 count = (i64)wd[(u64)col.cw * a.st.word + i * a.st.slot];
            if (count == 0) is_null = true;
            else {
                double avg = __longlong_as_double((i64)wd[(u64)col.vw * a.st.word + i * a.st.slot]) / (double)count;
                bits = col.type == PA_REAL ? (u64)__float_as_uint((float)avg) : (u64)__double_as_longlong(avg);
            }
            break;
        }
        default: break;
    }
}

__device__ __forceinline__ void gt_emit_slot(const GtEmitArgs& a, u64 i, u64 g)
{
    const u64* kw = (const u64*)a.keys + i * (u64)a.W;
    const u64* wd = (const u64*)a.words;
    i64 row_hash = 0;
    for (int c = 0; c < a.ncols; c++) {
        const GtEmitCol& col = a.col[c];
        u64 bits;
        bool is_null;
        gt_emit_value(a, col, i, kw, wd, row_hash, bits, is_null);
        if (col.width == 8) ((u64*)col.values)[g] = bits;
        else if (col.width == 4) ((u32*)col.values)[g] = (u32)bits;
        else ((u8*)col.values)[g] = (u8)bits;
        if (col.nulls) {
            col.nulls[g] = is_null ? 1 : 0;
            if (is_null) a.null_flags[c] = 1u;
        }
    }
}

__device__ __forceinline__ u64 pa_sort_key(i32 type, u64 bits, bool is_null, i32 sort_order);
// the order-preserving key of output column `column` of the group in slot i
__device__ __forceinline__ u64 gt_emit_sort_key(const GtEmitArgs& a, int column, i32 sort_order, u64 i)
{
    const GtEmitCol& col = a.col[column];
    const u64* kw = (const u64*)a.keys + i * (u64)a.W;
    i64 row_hash = 0;
    u64 bits;
    bool is_null;
    gt_emit_value(a, col, i, kw, (const u64*)a.words, row_hash, bits, is_null);
    return pa_sort_key(col.type, bits, is_null, sort_order);
}

__global__ __launch_bounds__(256) void k_gt_emit(GtEmitArgs a)
{
    // Output positions are handed out so that one store instruction of a wave writes CONSECUTIVE rows of every output block:
    // pass j of wave w takes slots first + j * 256 + w * 64 + lane; its occupied lanes get consecutive positions (ballot rank)
    // behind the tile's (pass, wave) prefix.  (Ranking a thread's 16 slots together instead leaves ~13 rows between the stores
    // of neighbouring lanes: every 8-byte store its own memory transaction -- 0.95 ms for 12 M groups, 4 x the time of this form.)
    __shared__ u32 part[64];   // occupied slots of (pass j, wave w) at [j * 4 + w], then their exclusive prefix
    __shared__ u32 tile_base;
    const i64 cap = (i64)a.cap;
    const i64 tiles = (cap + 4095) >> 12;
    const u64* tag = (const u64*)a.tag;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool filtered = a.filter_bound != nullptr;  // (a consumer's TopN: groups beyond the bound are left out)
    const u64 bound = filtered ? *a.filter_bound : ~0ULL;
    for (i64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const i64 first = (tile << 12) + threadIdx.x;
        u32 occ = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const i64 i = first + (i64)j * 256;
            if (i < cap && tag[(u64)i * a.st.tag] != a.st.empty && (!filtered || gt_emit_sort_key(a, a.filter_col, a.filter_order, (u64)i) <= bound)) occ |= 1u << j;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const u64 m = __ballot((occ >> j) & 1u);
            if (lane == 0) part[j * 4 + wave] = (u32)__popcll(m);
        }
        __syncthreads();
        if (wave == 0) {  // exclusive prefix of the 64 counts, in one wave
            const u32 mine = part[lane];
            u32 incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 o = (u32)__shfl_up((int)incl, off, 64);
                if (lane >= (u32)off) incl += o;
            }
            part[lane] = incl - mine;
            if (lane == 63) tile_base = incl != 0u ? atomicAdd(a.counter, incl) : 0u;  // ONE counter atomic per tile
        }
        __syncthreads();
        const u64 base = (u64)tile_base;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const bool mine = (occ >> j) & 1u;
            const u64 m = __ballot(mine);
            if (mine) gt_emit_slot(a, (u64)(first + (i64)j * 256), base + part[j * 4 + wave] + (u64)__popcll(m & ((1ULL << lane) - 1ULL)));
        }
        __syncthreads();  // `part` is rewritten by the next tile
    }
}
// Order-preserving 64-bit key of a value under a SortOrder (ASC_NULLS_FIRST 0, ASC_NULLS_LAST 1, DESC_NULLS_FIRST 2, DESC_NULLS_LAST 3):
// a <= b in the requested order implies key(a) <= key(b).  `bits` = the value as an output block holds it (4-byte types in the low
// half).  Double.compare / Float.compare order: -0.0 < 0.0, one NaN above everything.
__device__ __forceinline__ u64 pa_sort_key(i32 type, u64 bits, bool is_null, i32 sort_order)
{
    const bool descending = sort_order >= 2, nulls_first = (sort_order & 1) == 0;
    if (is_null) return nulls_first ? 0ULL : ~0ULL;
    u64 img;
    switch (type) {
        case PA_INTEGER:
        case PA_DATE: img = (u64)(i64)(i32)(u32)bits ^ 0x8000000000000000ULL; break;
        case PA_BOOLEAN: img = bits != 0ULL ? 1ULL : 0ULL; break;
        case PA_DOUBLE: {
            const double d = __longlong_as_double((i64)bits);
            const u64 b = d != d ? 0x7ff8000000000000ULL : bits;
            img = (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
            break;
        }
        case PA_REAL: {
            const double d = (double)__uint_as_float((u32)bits);
            const u64 b = d != d ? 0x7ff8000000000000ULL : (u64)__double_as_longlong(d);
            img = (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
            break;
        }
        default: img = bits ^ 0x8000000000000000ULL; break;  // BIGINT, short DECIMAL
    }
    return descending ? ~img : img;
}

// The order-preserving key (topn_kernels.hpp) of output column `column` for every slot of the table; ~0 for slots without a group.
__global__ __launch_bounds__(256) void k_gt_emit_keys(GtEmitArgs a, int column, int sort_order, u64* __restrict__ keys)
{
    const u64* tag = (const u64*)a.tag;
    const u64* wd = (const u64*)a.words;
    const GtEmitCol& col = a.col[column];
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < (i64)a.cap; i += (i64)gridDim.x * 256) {
        u64 key = ~0ULL;
        if (tag[(u64)i * a.st.tag] != a.st.empty) {
            const u64* kw = (const u64*)a.keys + (u64)i * (u64)a.W;
            i64 row_hash = 0;
            u64 bits;
            bool is_null;
            gt_emit_value(a, col, (u64)i, kw, wd, row_hash, bits, is_null);
            key = pa_sort_key(col.type, bits, is_null, sort_order);
        }
        keys[i] = key;
    }
}
__global__ __launch_bounds__(256) void k_gt_emit_keys_strided(GtEmitArgs a, int column, int sort_order, i64 stride, i64 count, u64* __restrict__ keys)
{
    const u64* tag = (const u64*)a.tag;
    for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < count; j += (i64)gridDim.x * 256) {
        const i64 i = j * stride;
        keys[j] = tag[(u64)i * a.st.tag] != a.st.empty ? gt_emit_sort_key(a, column, sort_order, (u64)i) : ~0ULL;
    }
}
void launch_gt_emit_keys_strided(const GtEmitArgs& args, int column, int sort_order, int64_t stride, int64_t count, uint64_t* keys, hipStream_t s)
{
    hipLaunchKernelGGL(k_gt_emit_keys_strided, grid_for(count, 256), 256, 0, s, args, column, sort_order, (i64)stride, (i64)count, (u64*)keys);
    PA_HIP(hipGetLastError());
}
void launch_gt_emit_keys(const GtEmitArgs& args, int column, int sort_order, uint64_t* keys, hipStream_t s)
{
    hipLaunchKernelGGL(k_gt_emit_keys, grid_for((int64_t)args.cap, 256), 256, 0, s, args, column, sort_order, (u64*)keys);
    PA_HIP(hipGetLastError());
}

void launch_gt_emit(const GtEmitArgs& args, hipStream_t s)
{
    const int64_t tiles = ((int64_t)args.cap + 4095) >> 12;
    hipLaunchKernelGGL(k_gt_emit, (int)std::max<int64_t>(1, std::min<int64_t>(tiles, 256 * 8)), 256, 0, s, args);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// InterpretedHashGenerator.hashPosition over typed columns (…/operator/InterpretedHashGenerator.java:62-70)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hash_page(HashPageArgs a)
{
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < a.n; r += (i64)gridDim.x * 256) {
        i64 result = 0;
        for (int c = 0; c < a.ncols; c++) {
            const HashCol& col = a.col[c];
            i64 h = 0;  // NULL -> 0 (BlockTypeOperators.java:102-108)
            if (!(col.nulls && col.nulls[r])) {
                switch (col.type) {
                    case PA_BIGINT: h = pa_hash_bigint(((const i64*)col.values)[r]); break;
                    case PA_INTEGER:
                    case PA_DATE: h = pa_hash_bigint((i64)((const i32*)col.values)[r]); break;
                    case PA_DOUBLE: h = pa_hash_double(((const double*)col.values)[r]); break;
                    case PA_REAL: h = pa_hash_real(((const float*)col.values)[r]); break;
                    case PA_BOOLEAN: h = (i64)pa_xxh64_long(((const u8*)col.values)[r] ? 1ULL : 0ULL); break;
                    case PA_VARCHAR: {
                        i32 o = col.offsets[r];
                        h = (i64)pa_xxh64((const u8*)col.values + o, col.offsets[r + 1] - o);
                        break;
                    }
                    default: break;
                }
            }
            result = pa_combine_hash(result, h);
        }
        a.out[r] = result;
    }
}

void launch_hash_page(const HashPageArgs& args, hipStream_t s)
{
    if (args.n <= 0) return;
    hipLaunchKernelGGL(k_hash_page, grid_for(args.n, 256), 256, 0, s, args);
    PA_HIP(hipGetLastError());
}

// LocalPartitionGenerator.getPartition (…/operator/exchange/LocalPartitionGenerator.java:45-65) /
// HashGenerator.getPartition (…/operator/HashGenerator.java:24-35)
__global__ __launch_bounds__(256) void k_partition_ids(const i64* __restrict__ raw_hash, i64 n, i32 partition_count, i32 local,
                                                       i32* __restrict__ out)
{
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
        u64 h = (u64)raw_hash[r];
        i32 p;
        if (local) {
            u64 rev = __brevll(h);  // Long.reverse
            p = (i32)(u32)pa_xxh64_long(rev) & (partition_count - 1);
        }
        else {
            p = (i32)((i64)(h & 0x7fffffffffffffffULL) % (i64)partition_count);
        }
        out[r] = p;
    }
}

void launch_partition_ids(const int64_t* raw_hash, int64_t n, int32_t partition_count, int32_t local, int32_t* out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_partition_ids, grid_for(n, 256), 256, 0, s, (const i64*)raw_hash, (i64)n, partition_count, local, out);
    PA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// synthetic TPC-H-shaped columns: row r of a column depends only on (seed, column, r, sf)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 g_mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ u64 g_rnd(u64 seed, u32 stream, u64 row)
{
    return g_mix64(seed + (u64)stream * 0xD1342543DE82EF95ULL + (row + 1) * 0x9E3779B97F4A7C15ULL);
}
enum { S_QTY = 1, S_PRICE = 2, S_DISC = 3, S_TAX = 4, S_SHIP = 5, S_RECEIPT = 6, S_FLAG = 7, S_CUST = 8, S_ODATE = 9, S_SEG = 10 };

__device__ __forceinline__ i64 g_sparse_orderkey(i64 o) { return (o >> 3) * 32 + (o & 7) + 1; }
__device__ __forceinline__ i32 g_shipdate(u64 seed, i64 r) { return 8036 + (i32)(g_rnd(seed, S_SHIP, (u64)r) % 2526ULL); }
__device__ __forceinline__ i32 g_segment(u64 seed, i64 r)
{
    u64 u = g_rnd(seed, S_SEG, (u64)(r / 5));
    i32 start = (i32)(u % 5ULL);
    i32 stride = 1 + (i32)((u >> 8) % 4ULL);
    return (start + stride * (i32)(r % 5)) % 5;
}
__constant__ char g_segments[5][11] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"};
__constant__ i32 g_segment_len[5] = {10, 8, 9, 9, 9};
__constant__ signed char g_order_in_block[28] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 6};

__global__ __launch_bounds__(256) void k_tpch(i32 column, i64 orders, u64 customers, i64 first_row, i64 n, u64 seed, void* values,
                                              i32* offsets)
{
    for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) {
        const i64 r = first_row + k;
        switch (column) {
            case PA_L_ORDERKEY: {
                i64 o = (r / 28) * 7 + g_order_in_block[r % 28];
                if (o >= orders) o = orders - 1;
                ((i64*)values)[k] = g_sparse_orderkey(o);
                break;
            }
            case PA_L_QUANTITY: ((double*)values)[k] = (double)(1 + g_rnd(seed, S_QTY, (u64)r) % 50ULL); break;
            case PA_L_EXTENDEDPRICE: ((double*)values)[k] = (double)(90100ULL + g_rnd(seed, S_PRICE, (u64)r) % 10404851ULL) / 100.0; break;
            case PA_L_DISCOUNT: ((double*)values)[k] = (double)(g_rnd(seed, S_DISC, (u64)r) % 11ULL) / 100.0; break;
            case PA_L_TAX: ((double*)values)[k] = (double)(g_rnd(seed, S_TAX, (u64)r) % 9ULL) / 100.0; break;
            case PA_L_SHIPDATE: ((i32*)values)[k] = g_shipdate(seed, r); break;
            case PA_L_RETURNFLAG: {
                i32 receipt = g_shipdate(seed, r) + 1 + (i32)(g_rnd(seed, S_RECEIPT, (u64)r) % 30ULL);
                char f = 'N';
                if (receipt <= 9298) f = (g_rnd(seed, S_FLAG, (u64)r) & 1ULL) ? 'R' : 'A';
                ((u8*)values)[k] = (u8)f;
                offsets[k] = (i32)k;
                if (k == n - 1) offsets[n] = (i32)n;
                break;
            }
            case PA_L_LINESTATUS:
                ((u8*)values)[k] = (u8)(g_shipdate(seed, r) > 9298 ? 'O' : 'F');
                offsets[k] = (i32)k;
                if (k == n - 1) offsets[n] = (i32)n;
                break;
            case PA_O_ORDERKEY: ((i64*)values)[k] = g_sparse_orderkey(r); break;
            case PA_O_CUSTKEY: {
                u64 usable = customers - customers / 3ULL;
                u64 u = g_rnd(seed, S_CUST, (u64)r) % usable;
                ((i64*)values)[k] = (i64)((u / 2ULL) * 3ULL + (u % 2ULL) + 1ULL);
                break;
            }
            case PA_O_ORDERDATE: ((i32*)values)[k] = 8035 + (i32)(g_rnd(seed, S_ODATE, (u64)r) % 2406ULL); break;
            case PA_O_SHIPPRIORITY: ((i32*)values)[k] = 0; break;
            case PA_C_CUSTKEY: ((i64*)values)[k] = r + 1; break;
            case PA_C_MKTSEGMENT: {
                i64 off = (r / 5 - first_row / 5) * 45;
                for (i64 q = (r / 5) * 5; q < r; q++) off += g_segment_len[g_segment(seed, q)];
                i32 sgm = g_segment(seed, r);
                i32 len = g_segment_len[sgm];
                for (i32 b = 0; b < len; b++) ((u8*)values)[off + b] = (u8)g_segments[sgm][b];
                offsets[k] = (i32)off;
                if (k == n - 1) offsets[n] = (i32)(off + len);
                break;
            }
            default: break;
        }
    }
}

void launch_tpch(int32_t column, double sf, int64_t first_row, int64_t n, uint64_t seed, void* values, int32_t* offsets, hipStream_t s)
{
    if (n <= 0) return;
    PA_REQUIRE(column >= PA_L_ORDERKEY && column <= PA_C_MKTSEGMENT, PA_ERR_INVALID_ARGUMENT, "unknown tpch column");
    PA_REQUIRE(column != PA_C_MKTSEGMENT || first_row % 5 == 0, PA_ERR_INVALID_ARGUMENT, "mktsegment first_row must be a multiple of 5");
    int64_t orders = (int64_t)(1500000.0 * sf);
    uint64_t customers = (uint64_t)(150000.0 * sf);
    hipLaunchKernelGGL(k_tpch, grid_for(n, 256), 256, 0, s, column, (i64)orders, (u64)customers, (i64)first_row, (i64)n, (u64)seed, values, offsets);
    PA_HIP(hipGetLastError());
}

// ---- partition-owned tables of the fused aggregation (V_LDSP) ----
// out[i] = in[0] + ... + in[i - 1] for i in [0, n]  (n <= 4097: the partitions of one multisplit)
__global__ __launch_bounds__(1024) void k_exclusive_prefix_i64(const i64* __restrict__ in, i32 n, i64* __restrict__ out)
{
    __shared__ i64 part[1024];
    // thread t owns the elements [t * 5, t * 5 + 5): 1024 x 5 >= 4097
    i64 local[5];
    i64 sum = 0;
    for (int j = 0; j < 5; j++) {
        const int i = (int)threadIdx.x * 5 + j;
        local[j] = sum;
        if (i < n) sum += in[i];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        i64 v = 0;
        if ((int)threadIdx.x >= off) v = part[threadIdx.x - off];
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const i64 base = part[threadIdx.x] - sum;
    for (int j = 0; j < 5; j++) {
        const int i = (int)threadIdx.x * 5 + j;
        if (i <= n) out[i] = base + local[j];
    }
}
void launch_exclusive_prefix_i64(const int64_t* in, int32_t n, int64_t* out, hipStream_t s)
{
    PA_REQUIRE(n >= 0 && n <= 5119, PA_ERR_NOT_SUPPORTED, "prefix of at most 5119 values");
    hipLaunchKernelGGL(k_exclusive_prefix_i64, 1, 1024, 0, s, (const i64*)in, n, (i64*)out);
    PA_HIP(hipGetLastError());
}

// ---- reference-format aggregation states (op_states.cpp) ----
__global__ __launch_bounds__(256) void k_widen_i32_i64(const i32* __restrict__ in, i64 n, i64* __restrict__ out)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) out[i] = (i64)in[i];
}
__global__ __launch_bounds__(256) void k_narrow_i64_i32(const i64* __restrict__ in, i64 n, i32* __restrict__ out)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) out[i] = (i32)in[i];
}
__global__ __launch_bounds__(256) void k_count_from_nulls(const u8* __restrict__ nulls, i64 n, i64* __restrict__ out)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) out[i] = (nulls && nulls[i]) ? 0 : 1;
}
void launch_widen_i32_i64(const int32_t* in, int64_t n, int64_t* out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_widen_i32_i64, grid_for(n, 256), 256, 0, s, in, (i64)n, (i64*)out);
    PA_HIP(hipGetLastError());
}
void launch_narrow_i64_i32(const int64_t* in, int64_t n, int32_t* out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_narrow_i64_i32, grid_for(n, 256), 256, 0, s, (const i64*)in, (i64)n, out);
    PA_HIP(hipGetLastError());
}
void launch_count_from_nulls(const uint8_t* nulls, int64_t n, int64_t* out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_count_from_nulls, grid_for(n, 256), 256, 0, s, nulls, (i64)n, (i64*)out);
    PA_HIP(hipGetLastError());
}

}  // namespace pa
