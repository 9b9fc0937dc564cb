// scan_kernels.hpp -- launchers of scan_kernels.hip.
#pragma once

#include "common.hpp"

namespace pa {

size_t scan_temp_bytes(int64_t n);
void launch_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total_out, void* temp, hipStream_t s);
void launch_varwidth_lengths(const int32_t* positions, int64_t count, const int32_t* offsets, const uint8_t* nulls, int32_t* out_len,
                             hipStream_t s);
void launch_varwidth_copy(const int32_t* positions, int64_t count, const int32_t* offsets, const uint8_t* bytes, const uint8_t* nulls,
                          int32_t* out_offsets, uint8_t* out_bytes, int32_t* total, hipStream_t s);
// dst[i + 1] = dst_base + src[i + 1] - src[0] for i in [0, count); dst[0] = dst_base when write_first (appending a
// VariableWidthBlock's offsets behind others, MergePages' PageBuilder)
void launch_offsets_append(const int32_t* src, int64_t count, int32_t dst_base, int32_t* dst, bool write_first, hipStream_t s);
// EncoderUtil.encodeNullsAsBits / decodeNullBits (core/trino-spi/.../block/EncoderUtil.java:35-110): 1 B / position <-> bits,
// most significant bit first
void launch_pack_null_bits(const uint8_t* nulls, int64_t n, uint8_t* packed, hipStream_t s);
void launch_unpack_null_bits(const uint8_t* packed, int64_t n, uint8_t* nulls, hipStream_t s);
// partition[i] = nulls[i] ? 1 : 0 (non-null positions first under the stable partition)
void launch_null_flag(const uint8_t* nulls, int64_t n, int32_t* partition, hipStream_t s);
// dst[positions[i]] = src[i] (inverse of Block.copyPositions: the values of the non-null positions back in place)
void launch_scatter_flat(const void* src, int elem_bytes, const int32_t* positions, int64_t count, void* dst, hipStream_t s);
// VariableWidthBlockEncoding offsets: ends[i] = offsets[i + 1] - offsets[0]; and back: offsets[0] = 0, offsets[i + 1] = ends[i]
void launch_varwidth_ends(const int32_t* offsets, int64_t n, int32_t* ends, hipStream_t s);
void launch_varwidth_from_ends(const int32_t* ends, int64_t n, int32_t* offsets, hipStream_t s);
// DictionaryAwarePageFilter.selectDictionaryPositions (DictionaryAwarePageFilter.java:117-140): the filter's verdict on the
// dictionary entries (dict_sel4: 4 selection bits per byte, the layout of pa_fp_count) looked up through the ids (null ids =
// RunLengthEncodedBlock: entry 0 for every row) -> the page's sel4 + selected rows per 1024-row tile
// (tile_quads x 1024 rows per tile_counts entry, as the generated kernels of the operator count them)
void launch_dict_filter_sel(const int32_t* ids, const uint8_t* dict_sel4, int64_t n, uint8_t* sel4, int32_t* tile_counts, int tile_quads, hipStream_t s);
// Multisplit of fixed-width columns: the rows of a chunk physically reordered so that every partition is contiguous (order
// inside a partition unspecified).  A 1024-thread workgroup takes 8192 consecutive rows, sorts them by partition in LDS and
// writes every column out in runs -- both the reads and the writes are coalesced, unlike a gather through the position list
// of launch_partition_positions, which costs a cache line per row and column once the partitions are many.
constexpr int kMsplitMaxCols = 48;
struct MsplitCol {
    const void* in;
    void* out;
    int32_t width;  // 1, 4 or 8 bytes
    int32_t pad;
};
size_t msplit_temp_bytes(int64_t n, int32_t partition_count);
// stable = ascending row order inside every partition (at most 256 partitions)
// counts_ready (unstable form only): the tile x partition counts are already in `temp` -- msplit_counts(temp)[tile * P + p] for the
// msplit_tiles(n) tiles of 8192 consecutive rows -- written by whoever made the partition ids (the generated partition pass of
// the fused aggregation histograms its tiles while it writes the ids), so the ids are not read a second time
void launch_msplit(const int32_t* partition, int64_t n, int32_t partition_count, const MsplitCol* cols, int32_t ncols, int64_t* out_counts_dev,
                   void* temp, hipStream_t s, bool stable = false, bool counts_ready = false);
int64_t msplit_tiles(int64_t n);
int32_t* msplit_counts(void* temp);
constexpr int kMsplitTileRows = 8192;
// OR and AND of keys: a bit in which they agree is constant (OrderBy leaves such bits out of its sort)
// per workgroup b: out[2 b] = OR, out[2 b + 1] = AND of its keys; returns the workgroups launched (out: key_or_and_bytes())
size_t key_or_and_bytes();
int launch_key_or_and(const uint64_t* keys, int64_t n, uint64_t* out, hipStream_t s);
size_t partition_temp_bytes(int64_t n, int32_t partition_count);
void launch_partition_positions(const int32_t* partition, int64_t n, int32_t partition_count, int32_t* out_positions,
                                int64_t* out_counts_dev, void* temp, hipStream_t s);

}  // namespace pa
