// topn_kernels.hpp -- launchers of topn_kernels.hip (TopNOperator's device side).
#pragma once

#include "common.hpp"

namespace pa {

// key[i] = order-preserving 64-bit image of the first sort channel of row i: a <= b in the requested order implies
// key(a) <= key(b) (monotone, not injective: VARCHAR uses its first 8 bytes, NULLs share the extreme value of their side).
void launch_topn_keys(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order,
                      uint64_t* keys, hipStream_t s);
// ... and OR / AND of the keys, one pair per workgroup in or_and (key_or_and_bytes() bytes); returns the number of pairs
int launch_topn_keys_or_and(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order, uint64_t* keys,
                            uint64_t* or_and, hipStream_t s);
// BIGINT / INTEGER / DATE values back from keys made without NULL rows (an OrderBy whose first sort channel is also an output channel
// writes that column from the sorted keys: a sequential pass instead of a gather)
void launch_topn_values_of_keys(int32_t type, const uint64_t* keys, int64_t n, bool descending, void* values, hipStream_t s);
// type code of a "column" that already holds keys (values = uint64 keys, no offsets / nulls): launch_topn_sample_bound / _filter
constexpr int32_t PA_TOPN_KEYS = -1;
// A bound for a page drawn from a sample: the keys of `sample_rows` rows (every (n / sample_rows)-th; sample_rows <= 2^18) go to
// sample_keys, the rank-th smallest of them (1-based) to *bound_out -- both device memory; nothing is waited for.
void launch_topn_sample_bound(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order,
                              int64_t sample_rows, int64_t rank, uint64_t* sample_keys, uint64_t* bound_out, hipStream_t s);
// One pass over the sort channel: positions and keys (in no particular order) of the rows whose key is <= min(bound, *device_bound)
// (device_bound may be null), at most `capacity` of them; counter[0] = how many such rows there are (8 bytes of device memory).
void launch_topn_filter(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order, uint64_t bound,
                        const uint64_t* device_bound, uint32_t capacity, int32_t* out_positions, uint64_t* out_keys, uint32_t* counter, hipStream_t s);
// The k-th smallest key (1-based) by MSB radix selection; temp >= topn_select_temp_bytes(); synchronises the stream.
size_t topn_select_temp_bytes();
uint64_t topn_select_kth(const uint64_t* keys, int64_t n, int64_t k, void* temp, uint32_t* host_hist_pinned, hipStream_t s);
// partition[i] = key[i] <= threshold ? 0 : 1
void launch_topn_flag(const uint64_t* keys, int64_t n, uint64_t threshold, int32_t* partition, hipStream_t s);

// Ties with the bound (op_topn.cpp): state[i] = 0 key < threshold, 1 key == threshold, 2 beyond; first = false refines only the
// rows in state 1 (the next sort channel's keys)
void launch_topn_state(const uint64_t* keys, int64_t n, uint64_t threshold, bool first, uint8_t* state, hipStream_t s);
// keys[i] = ~0 unless state[i] == 1 (the selection then only sees the tied rows)
void launch_topn_mask_keys(const uint8_t* state, int64_t n, uint64_t* keys, hipStream_t s);
// out2[0] = rows in state 0, out2[1] = rows in state 1 (device memory, 16 bytes)
void launch_topn_count_states(const uint8_t* state, int64_t n, int64_t* out2, hipStream_t s);
void launch_topn_tie_flags(const uint8_t* state, int64_t n, int32_t* flags, hipStream_t s);
// partition[i] = 0 for state 0 and for the first ties_kept rows in state 1 (tie_rank = exclusive scan of the tie flags), else 1
void launch_topn_state_partition(const uint8_t* state, const int32_t* tie_rank, int64_t n, int64_t ties_kept, int32_t* partition, hipStream_t s);

}  // namespace pa
