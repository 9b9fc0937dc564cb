// dynfilter_kernels.hpp -- device side of DynamicFilterSourceOperator (dynfilter_kernels.hip): the distinct build-side
// values of a channel (TypedSet) and its min / max.
#pragma once

#include "common.hpp"

namespace pa {

// Distinct values of one fixed-width channel as canonical 64-bit keys: open addressing, one CAS per probe, no payload.
// A slot holds kDfEmpty or a key; the key that equals kDfEmpty itself is tracked by a flag.
constexpr uint64_t kDfEmpty = 0x8000000000000001ULL;
constexpr int kDfBlocks = 1024;  // grid of the collect kernel: bounds the inserts still in flight when the limit is reached
struct DfSet {
    uint64_t* keys;       // [cap]
    uint32_t cap_mask;
    uint32_t limit;       // stop inserting once more than `limit` distinct non-null values were seen
    uint32_t* counters;   // [0] distinct non-null keys, [1] a NULL was seen, [2] kDfEmpty was seen as a value
};

// One pass over a page's column: (a) keys into the set (set != null), (b) the page's min / max as signed 64-bit values
// (partials != null; types whose order is the signed integer order: BIGINT, INTEGER, DATE, BOOLEAN) folded into
// running[0..2] = {min, max, any non-null seen}.  DOUBLE keys are canonical: one NaN, +0.0 for -0.0 (TypedSet's
// IS DISTINCT FROM: NaN is not distinct from NaN, -0.0 not from 0.0).
void launch_df_collect(int32_t type, const void* values, const uint8_t* nulls, int64_t n, const DfSet* set, int64_t* partials,
                       int64_t* running, hipStream_t s);
size_t df_partials_bytes();
// the occupied slots, compacted: out[0..count) (order unspecified), *count_out on the device
void launch_df_values(const DfSet& set, uint64_t* out, uint32_t* count_out, hipStream_t s);

}  // namespace pa
