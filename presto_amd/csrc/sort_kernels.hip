// sort_kernels.hip -- the pair sort under OrderByOperator's passes: (64-bit key image, row id) pairs by a range of the image's bits, stable.
// rocPRIM's device radix sort (one-sweep passes: every pass reads and writes the pairs once) is the library primitive for exactly this
// and replaces the operator's own LDS-staged 8-bit passes (count + scan + scatter: the pairs were read twice per pass) -- 2^24 pairs by
// 64 bits 2.3 ms -> see DESIGN.md section 6.  What the sort is asked to do stays the operator's: order-preserving images per sort channel
// and direction, 8-byte chunks + length for VARCHAR, the NULL placement as a digit of its own, constant bits left out (op_order_by.cpp;
// PagesIndex.sort / PagesIndexOrdering.quickSort, core/trino-main/src/main/java/io/trino/operator/PagesIndex.java:418-426).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "sort_kernels.hpp"

namespace pa {

size_t sort_pairs_temp_bytes(int64_t n)
{
    size_t bytes = 0;
    PA_HIP(rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                     (size_t)std::max<int64_t>(n, 1), 0u, 64u, (hipStream_t) nullptr));
    return bytes + 256;
}

void launch_sort_pairs(const uint64_t* keys_in, const int32_t* rows_in, uint64_t* keys_out, int32_t* rows_out, int64_t n, int begin_bit, int end_bit,
                       void* temp, size_t temp_bytes, hipStream_t s)
{
    if (n <= 0) return;
    PA_REQUIRE(begin_bit >= 0 && end_bit > begin_bit && end_bit <= 64, PA_ERR_DEVICE, "internal: bit range of a pair sort");
    // the scratch was sized for the whole key (sort_pairs_temp_bytes): rocPRIM does not promise that a narrower bit range needs no more
    // (its merge-sort and one-sweep paths size differently) -- ask for THIS range and refuse to run short
    size_t need = 0;
    PA_HIP(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, rows_in, rows_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, s));
    PA_REQUIRE(need <= temp_bytes, PA_ERR_DEVICE, "internal: pair sort scratch smaller than this bit range needs");
    PA_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rows_in, rows_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, s));
}

}  // namespace pa
