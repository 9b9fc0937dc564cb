// sort_kernels.hpp -- launchers of sort_kernels.hip.
#pragma once

#include "common.hpp"

namespace pa {

// Stable sort of (key, row) pairs by bits [begin_bit, end_bit) of the keys, ascending.  temp: sort_pairs_temp_bytes(n) bytes; the inputs
// are left as they are, the outputs must not overlap them.  rows_in == nullptr: the rows are 0, 1, 2, ... (no array is read for them;
// rows_scratch, n entries, is where the library path writes them out first).
// Returns which sort ran: the partition passes + LDS bucket sort of sort_kernels.hip, or the library's radix sort (keys that crowd in
// a few bit prefixes, inputs beyond 27 M pairs, PRESTO_AMD_SORT_LIBRARY set).
enum { PA_SORT_NONE = 0, PA_SORT_BUCKETS = 1, PA_SORT_LIBRARY = 2 };
size_t sort_pairs_temp_bytes(int64_t n);
int launch_sort_pairs(const uint64_t* keys_in, const int32_t* rows_in, int32_t* rows_scratch, uint64_t* keys_out, int32_t* rows_out, int64_t n, int begin_bit,
                      int end_bit, void* temp, size_t temp_bytes, hipStream_t s);

}  // namespace pa
