// topn_kernels.hpp -- launchers of topn_kernels.hip (TopNOperator's device side).
#pragma once

#include "common.hpp"

namespace pa {

// key[i] = order-preserving 64-bit image of the first sort channel of row i: a <= b in the requested order implies
// key(a) <= key(b) (monotone, not injective: VARCHAR uses its first 8 bytes, NULLs share the extreme value of their side).
void launch_topn_keys(int32_t type, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t sort_order,
                      uint64_t* keys, hipStream_t s);
// The k-th smallest key (1-based) by MSB radix selection; temp >= topn_select_temp_bytes(); synchronises the stream.
size_t topn_select_temp_bytes();
uint64_t topn_select_kth(const uint64_t* keys, int64_t n, int64_t k, void* temp, uint32_t* host_hist_pinned, hipStream_t s);
// partition[i] = key[i] <= threshold ? 0 : 1
void launch_topn_flag(const uint64_t* keys, int64_t n, uint64_t threshold, int32_t* partition, hipStream_t s);

}  // namespace pa
