// fused_plan.hpp -- what the fused scan-filter-project(-probe)-aggregate operator is asked to do, in the form its code generator
// (fused_codegen.cpp + one fused_tier_*.cpp per kernel tier) and its host side (op_fused.hpp) share: the plan (Spec), the kernel
// argument block's host mirror (FusedArgs), the tiers (Variant) and what a generated kernel tells the host about itself (KernelInfo).
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "exprgen.hpp"
#include "join_source.hpp"
#include "operator.hpp"

namespace pa {
namespace fused {

constexpr int kMaxChannels = 32;  // PA_MAX_CHANNELS in pa_device.h
constexpr int kMaxBuildChannels = 8;  // PA_MAX_BUILD_CHANNELS

// host mirror of PaFusedArgs (pa_device.h)
struct FusedArgs {
    const void* v[kMaxChannels];
    const int32_t* o[kMaxChannels];
    const uint8_t* nl[kMaxChannels];
    int64_t n;
    int32_t vec;
    int32_t pad;
    uint64_t* slab;
    uint64_t* gt_tag;
    uint64_t* gt_keys;
    uint64_t* gt_words;
    uint32_t gt_mask;
    int32_t gt_max_fill;
    int32_t* gt_count;
    int32_t* err;
    uint64_t* overflow_rows;
    const int32_t* row_list;
    int64_t n_list;
    int32_t* spill_rows;
    uint32_t* spill_count;
    uint32_t gt_rep_mask;
    uint32_t part_mask;
    int32_t* gt_rep_count;
    int32_t* part_ids;
    int32_t list_blocked;
    int32_t pad3;
    uint64_t* sub_tag;
    uint64_t* sub_keys;
    uint64_t* sub_words;
    int32_t* sub_count;
    const int64_t* part_first;
    const void* jslots;
    const uint64_t* jbits;
    int64_t jmin;
    uint64_t jrange;
    uint32_t jmask;
    int32_t jrows;
    uint32_t jwrap;
    uint32_t jpad;
    const void* bv[kMaxBuildChannels];
    const uint8_t* bn[kMaxBuildChannels];
    const void* jrank;
    const int32_t* jrank_rows;
    const uint64_t* ranges;
    int64_t n_ranges;
};

// V_LDSP: the LDS-table variant with partition-owned tables (see PaFusedArgs::sub_tag)
// V_BROW: probe stage whose group keys are functions of the build row: the table slot is the build position
// V_GLOBAL_R / V_LDS_R: the ungrouped / few-groups kernels over a TABLE of row ranges (stable device pages that do not continue
// each other in memory, taken in place by one launch: see ranges_)
enum Variant { V_GLOBAL = 0, V_LDS = 1, V_GT = 2, V_LDSH = 3, V_HASH = 4, V_LDSP = 5, V_BROW = 6, V_GLOBAL_R = 7, V_LDS_R = 8 };
// rows per entry of a range table: one workgroup takes an entry at a time
constexpr int64_t kRangeRows = 8192, kRangeRowsLds = 4096;
enum WordKind { W_CNT = 0, W_SUMF = 1, W_SUMI = 2, W_MAXU = 3 };

constexpr int kLdsSlots = 8;  // C of the LDS variant: 8 groups x NW words x 64 lanes x 8 B of LDS per wave

// Group keys are bit-packed into as few 64-bit words as possible (Q1: two VARCHAR(1) keys -> one word).
struct KeyPart {
    int32_t type = PA_BIGINT;
    int word = 0;        // word holding the value (first of two for long VARCHAR)
    int shift = 0;       // bit offset inside the word
    int bits = 64;       // value bits (long VARCHAR: 128 = two dedicated words)
    int bound = 0;       // short VARCHAR: declared length bound (1..7)
    int null_word = -1;  // position of the IS NULL flag, or -1
    int null_shift = 0;
};

// The probe stage between the projections and the aggregation (see the head of the file).
struct JoinStage {
    std::shared_ptr<LookupSourceImpl> ls;
    int key_proj = -1;                 // projection that is the probe join key
    std::vector<int> build_cols;       // virtual channel n_in + v reads ls->cols[build_cols[v]] at the build position
    std::vector<int32_t> build_types;
    // per group key: the projection to take it from when the slot is the build position (build columns only: the probe join key
    // is replaced by the build key column, equal on every match); empty = the group keys do not determine / are not determined
    // by the build row, no BROW variant
    std::vector<int> brow_group_proj;
};

struct Spec {
    std::shared_ptr<JoinStage> join;   // null: no probe stage
    // per channel: read inside the selected-rows block only (probe stage: everything the filter and the probe key do not need
    // is loaded for the rows that found a match, not for the whole page)
    std::vector<bool> lazy_channel;
    int n_in = 0;
    std::vector<int32_t> in_types, in_params;
    bool has_filter = false;
    OwnedExpr filter;
    std::vector<OwnedExpr> proj;
    std::vector<int> group_proj;
    int hash_channel = -1;
    std::vector<pa_aggregate> aggs;
    int expected_groups = 0;
    int64_t max_partial_memory = 0;
    int step = PA_STEP_SINGLE;
    int output_mem = PA_MEM_HOST;
    std::vector<bool> used_channel;
    std::vector<int> short_bound;  // per channel: > 0 when the channel is a short VARCHAR group key (packed bytes passed as cs<c>)
    // per channel: VARCHAR group key of unknown or long (> 15 bytes) bound, replaced by its interned id before the kernels
    // see the page (intern_kernels.hpp); in_types / the projection's type say INTEGER for such a channel
    std::vector<bool> interned;
    // per channel: VARCHAR argument of min / max without a short bound.  The page's strings are interned and the kernels see a BIGINT
    // column in their place -- (rank of the string among the dictionary's strings) << 32 | id -- whose integer order is the strings'
    // order; the operator re-ranks what it accumulated whenever a page brings new strings (op_fused_intern.cpp: rank_values)
    std::vector<bool> ranked;
    bool any_ranked() const
    {
        for (bool r : ranked) if (r) return true;
        return false;
    }
    // channels the kernels do not read as they arrive
    bool derived(int c) const { return interned[c] || ranked[c]; }
};

struct KernelInfo {
    std::string source, entry;
    int variant = V_GLOBAL;
    bool ranged = false;  // the kernel walks a table of row ranges (V_GLOBAL_R / V_LDS_R; `variant` names the base variant)
    int nw = 0, w = 0, c = 0, block = 256;
    int lc = 0;  // V_LDSH: slots of the workgroup's LDS table
    // V_BROW: the accumulator word every row of a group updates -- "this build row has a group" is read off it (its value differs
    // from occ_empty), and the kernel stores no tags -- or -1: tags are stored
    int occ_word = -1;
    uint64_t occ_empty = 0;
    std::vector<int32_t> word_kind;
    std::vector<std::pair<int, int>> agg_words;  // per aggregate: (count word, value word or -1)
    // per aggregate: > 0 for sum / avg over a DECIMAL -- the value is kept as that many limb words from agg_words[k].second on
    // (pa_dec_limb: independent integer sums, put together at output: decimal_host.hpp)
    std::vector<int> agg_limbs;
    std::vector<KeyPart> keys;
    // identity of the state layout (key packing + meaning of every accumulator word): states are only ever merged,
    // folded or emitted under the layout they were accumulated with
    std::string layout_id;
};

// thrown by adopt_layout before anything of the page was launched: the page's signature needs another state layout
struct LayoutChange {};

OwnedExpr input_ref_expr(int32_t channel, int32_t type);
// jd / bridge: the probe stage between the projections and the aggregation (null: none); the aggregation's channels then index
// the join's output page = [probe output channels, build output channels] (LookupJoinPageBuilder.java:76-139)
Spec make_spec(const pa_filter_project_desc& fp, const pa_hash_aggregation_desc& ag, const pa_lookup_join_desc* jd = nullptr,
               pa_lookup_source* bridge = nullptr);
Spec make_spec(const pa_fused_aggregation_desc* d);
// channels read, short / interned VARCHAR keys: everything of a Spec that follows from its expressions and aggregates
void finalize_spec(Spec& s);

// Words of a range-table entry, in this order: per used channel its values pointer, its offsets pointer when it is a VARCHAR
// channel, its NULL flags pointer when the layout calls it nullable; then rows | (vec << 32).
int range_entry_words(const Spec& s, const std::vector<ChannelLayout>& layout);
// the translation unit of one tier (`variant`) for pages of one layout signature (fused_codegen.cpp)
KernelInfo generate(const Spec& s, const std::vector<ChannelLayout>& layout, int variant);

}  // namespace fused
}  // namespace pa
