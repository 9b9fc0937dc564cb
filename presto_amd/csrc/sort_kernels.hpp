// sort_kernels.hpp -- launchers of sort_kernels.hip.
#pragma once

#include "common.hpp"

namespace pa {

// Stable sort of (key, row) pairs by bits [begin_bit, end_bit) of the keys, ascending.  temp: sort_pairs_temp_bytes(n, payload columns) bytes;
// the inputs are left as they are, the outputs must not overlap them.  rows_in == nullptr: the rows are 0, 1, 2, ... (no array is read for
// them; rows_scratch, n entries, is where the library path writes them out first).
// payload (may be null): up to PA_SORT_MAX_PAYLOAD columns of 4- or 8-byte values, one per pair in the order of the input pairs; the
// hand-written sort moves them along, so out[c][i] belongs to output pair i -- a sequential pass per move instead of a random gather by the
// sorted row ids afterwards.  The library path does not move them: the caller gathers by rows_out then.
// Returns which sort ran: the partition passes + LDS bucket sort of sort_kernels.hip (PA_SORT_BUCKETS: payload moved), or the library's
// radix sort (keys that crowd in a few bit prefixes, inputs beyond 27 M pairs, PRESTO_AMD_SORT_LIBRARY set).
enum { PA_SORT_NONE = 0, PA_SORT_BUCKETS = 1, PA_SORT_LIBRARY = 2 };
// hint: what the caller knows about the keys.  PA_SORT_HINT_SPREAD: they spread over the bits of the range (integers, dates, hashes) --
// buckets are bit prefixes.  PA_SORT_HINT_CROWDED: they crowd under few prefixes (images of doubles: the exponent bits; text) -- the bucket
// bounds come from a sorted sample of the keys (16 per bucket); asks that the keys differ only inside [begin_bit, end_bit).
enum { PA_SORT_HINT_SPREAD = 0, PA_SORT_HINT_CROWDED = 1 };
#define PA_SORT_MAX_PAYLOAD 4
struct SortPayload {
    int count;
    const void* in[PA_SORT_MAX_PAYLOAD];
    void* out[PA_SORT_MAX_PAYLOAD];
    int width[PA_SORT_MAX_PAYLOAD];
};
size_t sort_pairs_temp_bytes(int64_t n, int payload_columns = 0);
int launch_sort_pairs(const uint64_t* keys_in, const int32_t* rows_in, int32_t* rows_scratch, uint64_t* keys_out, int32_t* rows_out, int64_t n, int begin_bit,
                      int end_bit, void* temp, size_t temp_bytes, hipStream_t s, const SortPayload* payload = nullptr, int hint = PA_SORT_HINT_SPREAD);

}  // namespace pa
