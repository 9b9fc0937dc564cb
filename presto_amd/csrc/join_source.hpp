// join_source.hpp -- the lookup source a build operator publishes and its probe-side operators read (op_join.cpp builds it;
// LookupJoinOperator and the fused probe of op_fused.hpp read it).
// (LookupSourceFactory / JoinBridge: core/trino-main/src/main/java/io/trino/operator/join/PartitionedLookupSourceFactory.java:179-206;
//  PagesHash.java:54-126 for what the table holds)
#pragma once

#include <algorithm>
#include <mutex>

#include <atomic>
#include <memory>
#include <vector>

#include "common.hpp"
#include "join_kernels.hpp"

namespace pa {

struct BuildColumn {
    int32_t type = PA_BIGINT;
    bool varwidth = false;
    DevBuf values, offsets, nulls;
    bool has_nulls = false;
    int64_t bytes = 0;  // VARWIDTH: bytes used
};

// The lookup source shared between the build operator and its probe operators
// (LookupSourceFactory / JoinBridge; PartitionedLookupSourceFactory.java:179-206).
struct LookupSourceImpl {
    // PA_PAGE_RETAINED build pages whose block arrays the build columns read in place: released when the lookup source goes
    // (the last probe operator or the bridge, whichever holds it longest)
    struct Release {
        void (*fn)(void*);
        void* ctx;
    };
    std::vector<Release> releases;
    ~LookupSourceImpl()
    {
        for (const Release& r : releases) r.fn(r.ctx);
    }
    std::vector<BuildColumn> cols;
    int32_t n = 0;
    std::vector<int> join_channels, output_channels;
    int hash_channel = -1;
    DevBuf key, links, raw_hash, slot_of, tagged;
    DevBuf key_slots;           // JoinKeySlot[hash_size] when the join key is one BIGINT / INTEGER / DATE column (else `tagged`)
    bool keyed = false;
    uint32_t probe_mask = 0;    // size - 1 of key_slots
    uint32_t probe_wrap = 0;    // probe sequences wrap inside (slot & ~probe_wrap): probe_mask, or kJoinPartSlots - 1 after a partitioned build
    DevBuf key_bits;            // existence bitmap over [key_min, key_min + key_range] (keyed joins with a dense enough key range)
    JoinKeyBitmap bitmap{nullptr, 0, 0};
    // key rank index (join_kernels.hpp): built instead of key_slots over a keyed build side with a bitmap, no NULL key and no duplicate
    // key -- every probe (LookupJoinOperator's, the fused ones) then goes through it and key_slots stays empty
    DevBuf rank_words, rank_rows;
    JoinRankIndex rank{nullptr, nullptr, 0, 0};
    bool reference_built = false;  // PagesHash.key[] (the reference's layout) exists; keyed joins build it on demand
    bool key_range_valid = false;  // keyed join with at least one non-NULL build key: [key_min, key_max]
    int64_t key_min = 0, key_max = 0;
    DevBuf shared_bits;  // the bitmap of pa_lookup_source_shared_key_bitmap (union key range of all ranks)
    // OuterPositionTracker.visitedPositions: 1 B per build position, written by LOOKUP_OUTER / FULL_OUTER probes and read by the
    // LookupOuterOperator -- created (and cleared) when the first of them asks: an inner or probe-outer join never pays for it
    DevBuf visited;
    std::once_flag visited_once;
    uint8_t* visited_positions(hipStream_t s)
    {
        std::call_once(visited_once, [&] {
            const size_t bytes = (size_t)std::max(n, 1);
            PA_HIP(hipMemsetAsync(visited.ensure(bytes), 0, bytes, s));
            PA_HIP(hipStreamSynchronize(s));  // (probe operators on other streams may be the next to touch it)
        });
        return visited.as<uint8_t>();
    }
    uint32_t mask = 0;
    // keyed joins: some key occurs on more than one build row (positionLinks chains exist).  Set before `built`.
    bool has_duplicates = true;
    std::atomic<bool> built{false};
    std::atomic<int32_t> error{0};

    JoinKeys build_keys() const
    {
        JoinKeys k;
        memset(&k, 0, sizeof k);
        k.ncols = (int32_t)join_channels.size();
        for (int i = 0; i < k.ncols; i++) {
            const BuildColumn& c = cols[join_channels[i]];
            k.col[i].values = c.values.ptr();
            k.col[i].offsets = c.offsets.as<int32_t>();
            k.col[i].nulls = c.has_nulls ? c.nulls.as<uint8_t>() : nullptr;
            k.col[i].type = c.type;
        }
        return k;
    }
};

}  // namespace pa

struct pa_lookup_source {
    std::shared_ptr<pa::LookupSourceImpl> impl;
};

