// intern_kernels.hpp -- device-side string interning (intern_kernels.hip): VARCHAR group keys of any length become dense
// 32-bit ids that the packed-key machinery of the aggregation handles like an INTEGER column.
#pragma once

#include <string>
#include <vector>

#include "common.hpp"

namespace pa {

// One dictionary per interned channel of an operator: an open-addressing table hash -> (id, arena offset, length) with the
// claim / publish protocol of the group table, the strings themselves in an arena (8-byte padded), and per id the
// (offset, length) needed to turn ids back into strings.  Equality is decided on the bytes (hash and length first), so the
// ids are exact.  The host sizes table, id arrays and arena before every page for the worst case (every row a new
// string), so the kernel never runs out of room.
struct InternTable {
    uint64_t* tag;        // [cap]  0 empty | hash<<2|1 busy | hash<<2|3 ready
    uint64_t* meta;       // [cap]  id << 32 | length
    uint32_t* off;        // [cap]  arena offset / 8
    uint32_t cap_mask;
    uint32_t* id_off;     // [ids]  arena offset / 8
    uint32_t* id_len;     // [ids]
    uint64_t* id_hash;    // [ids]  XxHash64 of the bytes (VarcharType.hash: the key's share of $hashvalue)
    uint64_t* arena;      // 8-byte words
    uint32_t* counters;   // [0] next id, [1] arena words used
};

void launch_intern(const InternTable& t, const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int32_t* ids_out, hipStream_t s);
void launch_intern_rehash(const uint64_t* old_tag, const uint64_t* old_meta, const uint32_t* old_off, uint32_t old_cap, const InternTable& t, hipStream_t s);
// ids -> strings: out_lengths[i] (0 for NULL rows), then after an exclusive scan into out_offsets the bytes
void launch_intern_lengths(const InternTable& t, const int32_t* ids, const uint8_t* nulls, int64_t n, int32_t* out_lengths, hipStream_t s);
void launch_intern_bytes(const InternTable& t, const int32_t* ids, const uint8_t* nulls, int64_t n, const int32_t* out_offsets, uint8_t* out_bytes, hipStream_t s);

// min / max over strings by rank (op_fused_intern.cpp: rank_values).  image[i] = (ranks[ids[i]] + 1) << 32 | ids[i] -- a positive BIGINT whose
// order is the order of the strings, and whose low half names the string; 0 for the rows flagged in `nulls` (may be null)
void launch_rank_image(const int32_t* ids, const uint8_t* nulls, const uint32_t* ranks, int64_t n, int64_t* image, hipStream_t s);
// accumulator words holding such images -- as pa_img_i64 leaves them, complemented for min; 0 = no value yet -- brought up to date with
// new ranks: the id stays, the rank is looked up again
void launch_rerank_words(uint64_t* words, int64_t n, bool is_min, const uint32_t* ranks, hipStream_t s);

// The dictionary of one channel, owned by the operator that interns it.  Not thread safe (one operator = one driver thread).
class StringInterner {
public:
    // VARCHAR column of a staged page -> its ids (n x i32 on the device, valid until the next call); NULL rows get id 0
    // bytes_hint >= 0: an upper bound of the column's bytes the caller knows (a page gathered from host pages), which saves the
    // round trip for offsets[0] / offsets[n].  The dictionary's counters come back behind the launch without a wait: whoever
    // needs them next (size(), the next intern) waits then -- by which time the stream has usually passed the copy.
    const int32_t* intern(const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, hipStream_t s, int64_t bytes_hint = -1);
    // ids -> VariableWidthBlock arrays (offsets: n + 1 entries); rows flagged in `nulls` (may be null) become empty
    void decode(const int32_t* ids, const uint8_t* nulls, int64_t n, DevBuf* values, DevBuf* offsets, hipStream_t s);
    const uint64_t* hashes() const { return id_hash_.as<uint64_t>(); }
    uint32_t size() { settle(); return ids_; }
    // the strings of the ids from->size() onwards appended to `out` (a copy for the host: ranks, small results)
    void fetch_strings(uint32_t from, std::vector<std::string>* out, hipStream_t s);
    uint64_t bytes() const { return (uint64_t)words_ * 8; }  // arena in use (strings padded to 8 bytes), as of the last settled page

private:
    InternTable view() const;
    void reserve(int64_t rows, int64_t bytes, hipStream_t s);
    void settle();   // ids_ / words_ as the last launch left them
    DevBuf tag_, meta_, off_, id_off_, id_len_, id_hash_, arena_, counters_, ids_out_;
    PinnedBuf h_, h_fetch_;
    uint32_t cap_ = 0, ids_ = 0, words_ = 0;
    bool pending_ = false;             // the counters' copy of the last launch is in pending_stream_, not looked at yet
    hipStream_t pending_stream_ = nullptr;
};

}  // namespace pa
