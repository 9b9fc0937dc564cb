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

// One VariableWidthBlock of a DEVICE page to be appended behind the bytes an arena channel already holds.  Where the block's
// bytes start (offsets[0]) and how many there are is known on the device only, so the position they land at comes from a
// byte cursor kept in HBM: the segment starts at *cursor_in (0 when `fresh`) and leaves the cursor behind its bytes in
// *cursor_out -- two slots used in turn, so that a launch never reads the slot it writes.  The offsets are copied rebased
// (offsets[k] - offsets[0] + start), rows + 1 entries: the last one is the next block's first.
struct VarSeg {
    const char* values;
    const int32_t* offsets;
    char* dst_bytes;        // the arena channel's byte buffer ...
    int64_t capacity;       // ... and its size: a block that does not fit raises PA_ERR_INVALID_ARGUMENT in *err, nothing is written
    int32_t* dst_offsets;   // the arena channel's offsets + the arena row the block starts at
    const int64_t* cursor_in;
    int64_t* cursor_out;
    int64_t first_wg;       // filled by the launchers
    int64_t start;          // filled on the device (k_var_plan): byte position of the block in dst_bytes, -1 = refused
    int32_t rows;
    int32_t byte_wgs;       // workgroups the block's bytes are strided over (the host's guess of bytes / 64 KB, at least 1)
    int32_t first;          // filled on the device: offsets[0]
    int32_t len;            // filled on the device: offsets[rows] - offsets[0]
    int32_t slot;           // segments of one arena channel share a slot (< kVarSlots) and are appended in table order
    int32_t fresh;          // the arena channel is empty: the cursor reads as 0
};
constexpr int kVarSlots = 16;
constexpr int kInlineVarSegs = 16;
// One page's blocks (n <= kInlineVarSegs, distinct slots), the table travelling in the kernel arguments: one launch.
void launch_var_append_inline(VarSeg* segs, int n, int32_t* err, hipStream_t s);
// Many pages' blocks in table order: one planning launch (a workgroup: the cursor positions by prefix sums per slot) and
// one copy launch.  Tables as for launch_copy_segments (copy_var_table_bytes(n) each).
size_t copy_var_table_bytes(size_t n);
void launch_var_append(VarSeg* segs, size_t n, void* host_table, void* dev_table, int32_t* err, hipStream_t s);
// out[i] = i (topn_kernels.hip)
void launch_iota_i32(int32_t* out, int64_t n, hipStream_t s);
// dst[i] |= src[r * words + i] for r in [0, reps)
void launch_or_words(uint64_t* dst, const uint64_t* src, int64_t words, int32_t reps, hipStream_t s);

}  // namespace pa
