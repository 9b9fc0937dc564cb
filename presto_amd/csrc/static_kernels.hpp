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

}  // namespace pa
