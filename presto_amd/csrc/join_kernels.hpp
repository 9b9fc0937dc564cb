// join_kernels.hpp -- launchers of join_kernels.hip.
#pragma once

#include "common.hpp"

namespace pa {

constexpr int kMaxJoinChannels = 8;

struct JoinCol {
    const void* values;
    const int32_t* offsets;
    const uint8_t* nulls;
    int32_t type;
    int32_t pad;
};
struct JoinKeys {
    JoinCol col[kMaxJoinChannels];
    int32_t ncols;
    int32_t pad;
};

// PagesHash constructor (…/operator/join/PagesHash.java:54-126) + ArrayPositionLinks (…/ArrayPositionLinks.java:38-50)
void launch_join_build(const JoinKeys& build, const int64_t* raw_hash, int32_t n, int32_t* key, uint32_t mask, int32_t* slot_of,
                       int32_t* links, int32_t* err, hipStream_t s);
// PagesHash.getAddressIndex + chain length per probe row (…/PagesHash.java:158-170, JoinProbe.java:87-117)
// tagged_mask + 1 = size of the probe-side table (a power of two >= 2 x build rows, >= 8)
void launch_join_tag_slots(const int32_t* key, int64_t hash_size, const int64_t* raw_hash, uint64_t* tagged, uint32_t tagged_mask, hipStream_t s);
void launch_join_probe_count(const JoinKeys& build, const JoinKeys& probe, const int64_t* probe_hash, int32_t n_probe, const uint64_t* tagged,
                             uint32_t mask, const int32_t* links, int32_t* head, int32_t* counts, int flags, hipStream_t s);  // flags: 1 probe-outer, 2 outputSingleMatch
// Joins on one BIGINT / INTEGER / DATE key: the probe-side table holds the key itself next to the chain head and the chain's
// length (16 B per slot), so a probe row costs ONE random access -- no tag, no visit to the build key column, and no visit to
// positionLinks unless the key really has duplicates.  probe_hash may be null: the raw hash (AbstractLongType.hash of the
// value, combined as InterpretedHashGenerator does for one channel) is then computed in the kernel.
struct JoinKeySlot {
    uint64_t key;
    int32_t head;   // build position, -1 = empty slot
    int32_t count;  // rows of the key = entries of the chain head -> positionLinks[head] -> ...: a probe row knows how many output rows
                    // it gets without walking the chain (until round 4 the slot held positionLinks[head] and the counting pass walked)
};
// slots_mask + 1 = size of the probe-side table: a power of two >= 2 x the distinct keys (the caller uses 2 x build rows)
void launch_join_key_slots(const int32_t* key, int64_t hash_size, const JoinCol& build_key, const int64_t* raw_hash, const int32_t* links,
                           JoinKeySlot* slots, uint32_t slots_mask, hipStream_t s);
// Existence bitmap over [min key, max key] of the build side (when that range is small enough to be worth it): a probe row
// whose key has no bit set cannot match and never touches the slot table.  Keys of fact tables are usually dense and the
// probe side is often clustered by them (lineitem by orderkey), so the bitmap is read almost sequentially, while the slot
// table is a random access per row -- and most probe rows of a selective join are misses.
struct JoinKeyBitmap {
    const uint64_t* bits;   // null = no bitmap
    int64_t min_key;
    uint64_t range;         // max - min
};
// The keyed probe-side table straight from the build rows (raw_hash may be null: computed from the key); links are left at -1.
// err[0] = device error word, err[1] = set to 1 when some key occurs on more than one row.
void launch_join_keyed_build(const JoinCol& build_key, const int64_t* raw_hash, int32_t n, JoinKeySlot* slots, uint32_t slots_mask, int32_t* slot_of,
                             int32_t* links, int32_t* err, hipStream_t s);
// Partitioned build (join_kernels.hip): partition = home slot >> kJoinPartSlotsLog2; part / keybits / rowpos: n entries each, to be
// regrouped by partition (launch_msplit) before launch_join_part_build, which takes first[p] = offset of partition p's rows
// (partitions + 1 entries) and writes the chains of keys with several rows into `links` (pre-filled with -1) and the slots' `next`
// fields itself.  err[1]: some key has several rows; err[2]: an overfull partition, or a key with hundreds of rows -- the caller
// then builds with launch_join_keyed_build instead.
constexpr int kJoinPartSlotsLog2 = 13;
constexpr int kJoinPartSlots = 1 << kJoinPartSlotsLog2;  // 8192 slots = 128 KB of LDS
void launch_join_part_ids(const JoinCol& build_key, int32_t n, uint32_t slots_mask, int32_t* part, uint64_t* keybits, int32_t* rowpos, hipStream_t s);
void launch_join_part_build(const uint64_t* keys, const int32_t* rows, const int64_t* first, int32_t partitions, uint32_t slots_mask, JoinKeySlot* slots,
                            int32_t* links, int32_t* err, hipStream_t s);
// positionLinks and the slots' `next` fields: only needed when some key has several rows (err[1] of the build; every link and
// every `next` is -1 otherwise, which is what the build leaves)
void launch_join_keyed_links(int32_t n, JoinKeySlot* slots, uint32_t slots_mask, const int32_t* slot_of, int32_t* links, hipStream_t s);
void launch_join_key_bitmap(const JoinCol& build_key, int32_t n, int64_t min_key, uint64_t range, uint64_t* bits, hipStream_t s);
// Key rank index: the bitmap's words, each with the number of build keys below it.  Over a build side WITHOUT duplicate keys the rank
// of a key among the build keys -- below + popcount(bits under the key's bit) -- names its build row: directly when the rows arrived
// in key order (rows == null), through rows[rank] otherwise.  One 16-byte load answers a probe (does the key exist, and where), where
// the slot table takes the bitmap word plus one or two 64-byte lines at a random place; and nothing has to be hashed, regrouped or
// inserted to build it.
struct JoinRankWord {
    uint64_t bits;
    uint32_t below;
    uint32_t pad;
};
struct JoinRankIndex {
    const JoinRankWord* words;  // null = none
    const int32_t* rows;        // rank -> build position, or null when rank == build position
    int64_t min_key;
    uint64_t range;
};
// words[i] = {bits[i], number of set bits in bits[0 .. i)}; *total_out (device) = number of set bits = distinct build keys.
// counts: join_rank_tiles(nwords) int32 of scratch; temp: scan_temp_bytes(join_rank_tiles(nwords))
int64_t join_rank_tiles(int64_t nwords);
void launch_join_rank_words(const uint64_t* bits, int64_t nwords, JoinRankWord* words, int32_t* counts, void* temp, int32_t* total_out, hipStream_t s);
// rows[rank(key of build row i)] = i for all rows (no NULL keys); *unordered (device, zeroed by the caller) = 1 when some rank != i
// (distinct: device word holding the number of distinct build keys, or null -- when it is not n the index cannot hold and the pass does nothing)
void launch_join_rank_rows(const JoinCol& build_key, int32_t n, const JoinRankWord* words, int64_t min_key, int32_t* rows, int32_t* unordered, hipStream_t s,
                           const int32_t* distinct = nullptr);
// wrap = mask, or kJoinPartSlots - 1 for a table built in partitions (the probe sequence of a key then stays inside the
// kJoinPartSlots-slot partition of its home slot)
// tile_totals (optional, join_probe_tiles(n_probe) entries): filled -- the function then returns true -- when the lookup source has no
// key on several rows and the four-rows-per-step kernel runs: the pairs can then be emitted by launch_join_probe_emit_tiles after an
// exclusive scan of the tile totals alone
int64_t join_probe_tiles(int32_t n_probe);
bool launch_join_probe_count_keyed(const JoinCol& probe_key, const int64_t* probe_hash, int32_t n_probe, const JoinKeySlot* slots, uint32_t mask,
                                   uint32_t wrap, const int32_t* links, const JoinKeyBitmap& bitmap, const JoinRankIndex& rank, int32_t* head, int32_t* counts,
                                   int flags, hipStream_t s, int64_t* total = nullptr, bool unique_keys = false, int32_t* tile_totals = nullptr);   // total: 16 counters (zeroed by the caller) whose sum is the page's output rows
void launch_join_probe_emit_tiles(const int32_t* head, const int32_t* tile_offsets, int32_t n_probe, int flags, int32_t* probe_idx, int32_t* build_pos,
                                  uint8_t* visited, hipStream_t s);
// build rows out of key order: the bitmap and the rank -> row array over (key, row) pairs regrouped by key range (join_kernels.hip)
// min / max / presence / disorder of the build keys in one pass.  out: 4 x u64, see join_key_stats_decode; temp: join_key_stats_temp_bytes()
size_t join_key_stats_temp_bytes();
void launch_join_key_stats(const JoinCol& key, int32_t n, uint64_t* out, void* temp, hipStream_t s);
struct JoinKeyStats {
    int64_t min_key = 0, max_key = 0;
    bool any = false;        // some key is not NULL
    bool descending = false; // some key is smaller than the key of the row before it
    uint64_t descents = 0;   // ... how many
};
inline JoinKeyStats join_key_stats_decode(const uint64_t (&h)[4])
{
    JoinKeyStats st;
    st.any = h[2] != 0;
    st.descending = h[3] != 0;
    st.descents = h[3];
    if (st.any) {
        st.min_key = (int64_t)(~h[0] ^ 0x8000000000000000ULL);
        st.max_key = (int64_t)(h[1] ^ 0x8000000000000000ULL);
    }
    return st;
}
int join_range_shift(uint64_t range);   // log2 of the key values per partition (16..19), -1: the range is too wide
void launch_join_range_ids(const JoinCol& build_key, int32_t n, int64_t min_key, int shift, int32_t partitions, int32_t* part, uint64_t* keybits,
                           int32_t* rowpos, hipStream_t s);
void launch_join_range_bitmap(const uint64_t* keys, const int64_t* first, int32_t partitions, int64_t min_key, int shift, uint64_t range, uint64_t* bits,
                              hipStream_t s);
void launch_join_rank_rows_pairs(const uint64_t* keys, const int32_t* rowpos, int64_t n, const JoinRankWord* words, int64_t min_key, int32_t* rows,
                                 hipStream_t s, const int64_t* first = nullptr, int32_t partitions = 0, int shift = 0, const int32_t* distinct = nullptr);
void launch_join_unvisited_flag(const uint8_t* visited, int64_t n, int32_t* partition, hipStream_t s);
// DefaultPageJoiner.joinCurrentPosition: (probe position, build position) pairs in emission order
void launch_join_probe_emit(const int32_t* head, const int32_t* offsets, int32_t n_probe, int32_t total, const int32_t* links, int32_t* probe_idx,
                            int32_t* build_pos, int flags, uint8_t* visited, hipStream_t s);
void launch_fill_i32(int32_t* dst, int32_t value, int64_t n, hipStream_t s);
void launch_rebase_offsets(const int32_t* in, int32_t in_base, int32_t out_base, int64_t n_plus_1, int32_t* out, hipStream_t s);

// JoinFilterFunction: from the indices of the candidates the filter kept (`eligible`, ascending = emission order) to output pairs
void launch_jf_first_of_row(const int32_t* eligible, int32_t ne, const int32_t* cand_probe, int32_t* keep, hipStream_t s);
void launch_jf_compact(const int32_t* eligible, int32_t ne, const int32_t* keep_scan, int32_t kept, int32_t* out, hipStream_t s);
void launch_jf_count_rows(const int32_t* eligible, int32_t ne, const int32_t* cand_probe, int32_t* per_row, hipStream_t s);
void launch_jf_outer(const int32_t* eligible, int32_t ne, const int32_t* cand_probe, const int32_t* cand_build, const int32_t* per_row, const int32_t* first,
                     const int32_t* at, int32_t rows, int32_t* out_probe, int32_t* out_build, hipStream_t s);
void launch_jf_max1(const int32_t* in, int64_t n, int32_t* out, hipStream_t s);
void launch_jf_mark_visited(const int32_t* build_pos, int64_t n, uint8_t* visited, hipStream_t s);

// *out = sum of v[0..n) in 64 bits (out: device memory, 8-byte aligned)
void launch_sum_i32_i64(const int32_t* v, int64_t n, int64_t* out, hipStream_t s);

}  // namespace pa
