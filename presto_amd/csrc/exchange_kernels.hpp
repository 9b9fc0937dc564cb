// exchange_kernels.hpp -- launchers of exchange_kernels.hip: moving column segments between the per-destination buffers of
// a partitioned exchange, its send / receive blobs and the consumer's page.
#pragma once

#include "common.hpp"

namespace pa {

// One contiguous run of bytes; src == nullptr writes zeros (the NULL flags of a source rank that had none).
struct CopySeg {
    const void* src;
    void* dst;
    int64_t bytes;
    int64_t first_chunk;  // filled by launch_copy_segments
    int32_t add_i32 = 0;  // != 0: the segment is int32 elements (4-byte aligned) and this is added to each while it is copied
                          // (the offsets of a VariableWidthBlock appended behind other blocks' bytes)
    int32_t pad = 0;
};
// device-side table size for n segments
size_t copy_segments_table_bytes(size_t n);
// Copies every segment in ONE launch (a workgroup per 64 KB chunk, 16-byte accesses where both sides are aligned).
// `segs` is host memory (first_chunk is filled in), `host_table` pinned staging and `dev_table` HBM of
// copy_segments_table_bytes(n) each, both untouched by the caller until the stream has passed this launch.
void launch_copy_segments(CopySeg* segs, size_t n, void* host_table, void* dev_table, hipStream_t s);
// The same for at most kInlineSegs segments, the table travelling in the kernel arguments: no staging buffer, one launch.
constexpr int kInlineSegs = 64;
void launch_copy_segments_inline(CopySeg* segs, int n, hipStream_t s);
// out[i] = i (topn_kernels.hip)
void launch_iota_i32(int32_t* out, int64_t n, hipStream_t s);
// dst[i] |= src[r * words + i] for r in [0, reps)
void launch_or_words(uint64_t* dst, const uint64_t* src, int64_t words, int32_t reps, hipStream_t s);

}  // namespace pa
