// static_kernels.hpp -- host-callable launchers of the query-independent kernels (static_kernels.hip).
#pragma once

#include "common.hpp"

namespace pa {

struct HashCol {
    const void* values;
    const int32_t* offsets;
    const uint8_t* nulls;
    int32_t type;
    int32_t pad;
};
struct HashPageArgs {
    HashCol col[16];
    int32_t ncols;
    int32_t pad;
    int64_t n;
    int64_t* out;
};

void launch_hash_page(const HashPageArgs& args, hipStream_t s);
void launch_partition_ids(const int64_t* raw_hash, int64_t n, int32_t partition_count, int32_t local, int32_t* out, hipStream_t s);
void launch_tpch(int32_t column, double sf, int64_t first_row, int64_t n, uint64_t seed, void* values, int32_t* offsets, hipStream_t s);

// Grouped output assembled on device (InMemoryHashAggregationBuilder.buildResult): one pass over the table slots
// compacts the occupied ones (one counter atomic per wave) and writes every output block -- unpacked key columns,
// $hashvalue, final aggregate values or PARTIAL states -- directly.
enum GtEmitKind { GT_EMIT_KEY = 0, GT_EMIT_HASH = 1, GT_EMIT_COUNT = 2, GT_EMIT_SUM = 3, GT_EMIT_AVG = 4, GT_EMIT_STATE = 5, GT_EMIT_MINMAX = 6,
                  GT_EMIT_COLUMN = 7 };  // COLUMN: the key is element [slot] of a column (a build-row table's keys are build columns)
struct GtEmitCol {
    int32_t kind, type;          // GtEmitKind, pa_type of the output block
    int32_t word, shift, bits;   // KEY: packed position; STATE: accumulator word; MINMAX: shift = 1 for min
    int32_t null_word, null_shift;
    int32_t cw, vw;              // aggregates: count word, value word
    int32_t width;               // bytes per output element
    void* values;
    uint8_t* nulls;              // may be null when the column cannot hold NULLs
    const uint64_t* dict_hash;   // KEY of an interned VARCHAR channel: per id the hash of the string ($hashvalue), else null
    const void* src;             // COLUMN: the column's values (element width = width) and NULL flags (may be null)
    const uint8_t* src_nulls;
};
// how a group table lays out its tags and accumulator words, in 8-byte words: tag of slot i at tag[i * tag], word w of slot i at
// words[w * word + i * slot].  The hashed tables are word-major ({1, capacity, 1}); a build-row table keeps one record
// [tag, word 0, ...] per build position ({1 + NW, 1, 1 + NW} with words = tag + 1)
struct GtStrides {
    uint32_t tag, word, slot;
    uint32_t pad;
    uint64_t empty;  // tag value of a slot without a group (0 for real tags; a build-row table may read an accumulator word as its tag)
};
// launch_gather_multi: up to 24 flat columns by one launch; which = 0 / 1 names the position list (probe index / build position)
constexpr int GATHER_MULTI_MAX_COLS = 24;
struct GatherMultiCol {
    const void* src;
    const uint8_t* src_nulls;
    void* dst;
    uint8_t* dst_nulls;   // written when non-null (a -1 position gives 1)
    int32_t width;        // 1, 4 or 8
    int32_t which;
};
struct GatherMultiArgs {
    const int32_t* positions[2];
    int64_t count;             // rows to gather -- or, with count_dev, the most there can be (the grid is sized for it)
    const int32_t* count_dev;  // non-null: the row count is read on the device (a join's match total that the host has not seen yet)
    int32_t ncols, pad;
    GatherMultiCol col[GATHER_MULTI_MAX_COLS];
};
void launch_gather_multi(const GatherMultiArgs& args, hipStream_t s);
constexpr int GT_EMIT_MAX_COLS = 32;
struct GtEmitArgs {
    const uint64_t* tag;
    const uint64_t* keys;
    const uint64_t* words;
    GtStrides st;
    uint32_t cap;
    int32_t W, NW, ncols;
    uint32_t* counter;
    uint32_t* null_flags;        // [ncols]: set to 1 when the column wrote a NULL
    // filter_bound set: only the groups whose order-preserving key of output column filter_col under filter_order (launch_gt_emit_keys'
    // key, computed from the table on the way) is <= *filter_bound are emitted -- the consumer is a TopN
    const uint64_t* filter_bound;
    int32_t filter_col, filter_order;
    GtEmitCol col[GT_EMIT_MAX_COLS];
};
void launch_gt_emit(const GtEmitArgs& args, hipStream_t s);
// keys[slot] = the order-preserving 64-bit key (as TopN's, topn_kernels.hpp) of output column `column` under `sort_order`, ~0 for
// slots without a group; kinds KEY / COLUMN / STATE / COUNT / SUM / AVG / MINMAX
void launch_gt_emit_keys(const GtEmitArgs& args, int column, int sort_order, uint64_t* keys, hipStream_t s);
// the same for `count` evenly spaced slots only: keys[j] = the key of slot j * stride (the sample a TopN bound is drawn from)
void launch_gt_emit_keys_strided(const GtEmitArgs& args, int column, int sort_order, int64_t stride, int64_t count, uint64_t* keys, hipStream_t s);
// out[i] = in[0] + ... + in[i - 1] for i in [0, n], n <= 5119 (static_kernels.hip)
void launch_exclusive_prefix_i64(const int64_t* in, int32_t n, int64_t* out, hipStream_t s);

}  // namespace pa
