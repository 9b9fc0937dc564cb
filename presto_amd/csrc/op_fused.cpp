// op_fused.cpp -- fused scan-filter-project-aggregate operator.
//
// Replaces the operator chain  FilterAndProjectOperator -> AggregationOperator   (Q6 shape) or
//                              FilterAndProjectOperator -> HashAggregationOperator (Q1 shape)
// of the reference with ONE pass over the page's columns in HBM:
//   PageProcessor.createWorkProcessor / ProjectSelectedPositions (…/operator/project/PageProcessor.java:
//     111-137, 180-263), generated PageFilter / PageProjection loops (…/sql/gen/PageFunctionCompiler.java:
//     283-320, 477-499), AggregationOperator.addInput (…/operator/AggregationOperator.java:145-160),
//   InMemoryHashAggregationBuilder.processPage (…/aggregation/builder/InMemoryHashAggregationBuilder.java:
//     139-155), accumulator input functions.
// With an empty filter and identity projections it is the stand-alone (Hash)AggregationOperator.
//
// Kernel variants (generated per (expressions, column-layout signature), compiled by jit.cpp):
//   GLOBAL  no group keys: per-lane register accumulators -> wave shuffle -> LDS -> one partial state per
//           workgroup in a slab -> fixed-order merge kernel (bitwise reproducible).
//   LDS     <= C groups: one wave per workgroup, wave-private LDS key table and lane-private LDS
//           accumulators (no atomics, no barriers), per-wave partial tables in a slab -> merge kernel.
//   GT      any cardinality: open-addressing table in HBM, agent-scope atomics.
//   BROW    probe stage only (below): the group IS the build row -- accumulators indexed by build position, no hashing.
// All are HBM-read bound: algorithmic bytes per row = sum of the widths of the referenced columns.
//
// Probe stage (pa_fused_join_aggregation_desc): FilterAndProjectOperator -> LookupJoinOperator (INNER) -> (Hash)AggregationOperator
// as ONE pass -- filter, key-bitmap test, keyed probe (JoinProbe.getCurrentJoinPosition, JoinProbe.java:87-117;
// DefaultPageJoiner.joinCurrentPosition, DefaultPageJoiner.java:236-320) and accumulation in the generated row function; no
// compacted probe page, no (probe, build) position lists, no gathered join output.  Taken when the lookup source has one integer
// key and no duplicate keys (each probe row then has at most one match); op_fused_join.cpp runs the three operators one after
// the other otherwise.  The channels of the join's output page become projections: probe outputs are the FilterAndProject
// projections, build outputs are "virtual" input channels n_in + v read from the lookup source's columns at the build position.
#include <algorithm>
#include <cmath>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>

#include "decimal_host.hpp"
#include "exchange_kernels.hpp"
#include "exprgen.hpp"
#include "host_hash.hpp"
#include "jit.hpp"
#include "join_source.hpp"
#include "operator.hpp"
#include "rowgen.hpp"
#include "scan_kernels.hpp"
#include "static_kernels.hpp"
#include "intern_kernels.hpp"

namespace pa {

// static kernels (static_kernels.hip)
void launch_merge_global_slab(const uint64_t* slab, int blocks, int nw, const int32_t* kinds_dev, uint64_t* state, int32_t* err,
                              hipStream_t s);
void launch_merge_lds_slab(const uint64_t* slab, int waves, int c, int w, int nw, const int32_t* kinds_dev, uint64_t* gt_tag,
                           uint64_t* gt_keys, uint64_t* gt_words, uint32_t gt_mask, int32_t gt_max_fill, int32_t* gt_count,
                           int32_t* err, const uint64_t* overflow_rows, int32_t* entry_slot, hipStream_t s);
void launch_gt_fold(const uint64_t* old_tag, const uint64_t* old_keys, const uint64_t* old_words, uint32_t old_cap, uint32_t old_reps, int w,
                    int nw, const int32_t* kinds_dev, uint64_t* tag, uint64_t* keys, uint64_t* words, uint32_t mask, uint32_t new_reps,
                    int32_t* count0, int32_t* rep_count, int32_t* err, hipStream_t s);

void launch_exclusive_prefix_i64(const int64_t* in, int32_t n, int64_t* out, hipStream_t s);
void launch_fill_u64(uint64_t* dst, uint64_t value, int64_t n, hipStream_t s);
void launch_gt_compact(const uint64_t* tag, const uint64_t* keys, const uint64_t* words, uint32_t cap, int w, int nw, uint64_t* out_keys,
                       uint64_t* out_words, uint32_t* counter, hipStream_t s, const GtStrides* strides = nullptr);

namespace {

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

void finalize_spec(Spec& s);

OwnedExpr input_ref_expr(int32_t channel, int32_t type)
{
    OwnedExpr e;
    pa_expr_node node{};
    node.kind = PA_EXPR_INPUT_REF;
    node.type = type;
    node.channel = channel;
    e.nodes.push_back(node);
    e.strings.emplace_back();
    e.root = 0;
    return e;
}

// jd / bridge: the probe stage between the projections and the aggregation (null: none); the aggregation's channels then index
// the join's output page = [probe output channels, build output channels] (LookupJoinPageBuilder.java:76-139)
Spec make_spec(const pa_filter_project_desc& fp, const pa_hash_aggregation_desc& ag, const pa_lookup_join_desc* jd = nullptr,
               pa_lookup_source* bridge = nullptr)
{
    Spec s;
    PA_REQUIRE(fp.input_channel_count > 0 && fp.input_channel_count <= kMaxChannels, PA_ERR_NOT_SUPPORTED,
               "fused aggregation supports 1..32 input channels");
    s.n_in = fp.input_channel_count;
    s.in_types.assign(fp.input_types, fp.input_types + s.n_in);
    s.in_params.assign(s.n_in, 0);
    if (fp.input_type_params) s.in_params.assign(fp.input_type_params, fp.input_type_params + s.n_in);
    s.has_filter = fp.filter != nullptr;
    if (s.has_filter) {
        s.filter = OwnedExpr::copy(*fp.filter);
        PA_REQUIRE(s.filter.root_type() == PA_BOOLEAN, PA_ERR_INVALID_ARGUMENT, "filter must be BOOLEAN");
    }
    for (int32_t j = 0; j < fp.projection_count; j++) s.proj.push_back(OwnedExpr::copy(fp.projections[j]));
    // channel of the aggregation's input page -> projection
    std::vector<int> to_proj;
    if (!jd) {
        for (int32_t j = 0; j < fp.projection_count; j++) to_proj.push_back(j);
    }
    else {
        PA_REQUIRE(bridge != nullptr && bridge->impl != nullptr, PA_ERR_ILLEGAL_STATE, "lookup source has no build operator yet");
        PA_REQUIRE(ag.step == PA_STEP_SINGLE || ag.step == PA_STEP_PARTIAL, PA_ERR_NOT_SUPPORTED, "an aggregation over a join output is SINGLE or PARTIAL");
        PA_REQUIRE(jd->join_type == PA_JOIN_INNER, PA_ERR_NOT_SUPPORTED, "the fused probe is an inner join");
        PA_REQUIRE(jd->filter == nullptr, PA_ERR_NOT_SUPPORTED, "a join filter function runs in the LookupJoinOperator, not in the fused probe");
        auto js = std::make_shared<JoinStage>();
        js->ls = bridge->impl;
        const LookupSourceImpl& ls = *js->ls;
        PA_REQUIRE(jd->probe_channel_count == fp.projection_count, PA_ERR_INVALID_ARGUMENT, "the probe page is the projection output");
        PA_REQUIRE(jd->join_channel_count == 1 && ls.join_channels.size() == 1, PA_ERR_NOT_SUPPORTED, "the fused probe takes one join key");
        const int kc = jd->probe_join_channels[0];
        PA_REQUIRE(kc >= 0 && kc < fp.projection_count, PA_ERR_INVALID_ARGUMENT, "probe join channel out of range");
        const int build_key_col = ls.join_channels[0];
        const int32_t kt = s.proj[kc].root_type();
        PA_REQUIRE(kt == PA_BIGINT || kt == PA_INTEGER || kt == PA_DATE, PA_ERR_NOT_SUPPORTED, "the fused probe takes a BIGINT / INTEGER / DATE key");
        PA_REQUIRE(kt == ls.cols[build_key_col].type, PA_ERR_INVALID_ARGUMENT, "probe / build join key types differ");
        js->key_proj = kc;
        // virtual channels are made for the build columns the aggregation reads
        std::set<int> read;
        for (int32_t g = 0; g < ag.group_by_count; g++) read.insert(ag.group_by_channels[g]);
        for (int32_t k = 0; k < ag.aggregate_count; k++) {
            if (ag.aggregates[k].fn != PA_AGG_COUNT_STAR) read.insert(ag.aggregates[k].input_channel);
            if (ag.aggregates[k].mask_channel >= 0) read.insert(ag.aggregates[k].mask_channel);
        }
        std::map<int, int> proj_of_build_col;
        auto build_proj = [&](int col) {
            auto it = proj_of_build_col.find(col);
            if (it != proj_of_build_col.end()) return it->second;
            const int32_t t = ls.cols[col].type;
            PA_REQUIRE(t != PA_VARCHAR, PA_ERR_NOT_SUPPORTED, "VARCHAR build columns are not read by the fused probe");
            PA_REQUIRE((int)js->build_cols.size() < kMaxBuildChannels, PA_ERR_NOT_SUPPORTED, "the fused probe reads at most 8 build columns");
            const int v = (int)js->build_cols.size();
            js->build_cols.push_back(col);
            js->build_types.push_back(t);
            s.proj.push_back(input_ref_expr(s.n_in + v, t));
            return proj_of_build_col[col] = (int)s.proj.size() - 1;
        };
        std::vector<int> build_col_of;  // per channel of the joined page: the build column, or -1 for a probe output
        for (int32_t i = 0; i < jd->probe_output_channel_count; i++) {
            const int c = jd->probe_output_channels[i];
            PA_REQUIRE(c >= 0 && c < fp.projection_count, PA_ERR_INVALID_ARGUMENT, "probe output channel out of range");
            to_proj.push_back(c);
            build_col_of.push_back(-1);
        }
        for (int col : ls.output_channels) {
            const int j = (int)to_proj.size();
            to_proj.push_back(read.count(j) ? build_proj(col) : -1);
            build_col_of.push_back(col);
        }
        // build-row tables: every group key is the join key or a build output, and the join key is among them (without it,
        // two build rows could hold the same group)
        bool eligible = ag.group_by_count > 0, has_key = false;
        std::vector<int> brow;
        for (int32_t g = 0; g < ag.group_by_count && eligible; g++) {
            const int ch = ag.group_by_channels[g];
            if (ch < 0 || ch >= (int)to_proj.size()) break;  // refused below
            if (build_col_of[ch] >= 0) {
                brow.push_back(to_proj[ch]);
                has_key = has_key || build_col_of[ch] == build_key_col;
            }
            else if (to_proj[ch] == kc) {
                brow.push_back(-2);  // the build key column: made a virtual channel below, when the variant is possible at all
                has_key = true;
            }
            else eligible = false;
        }
        if (eligible && has_key && (int)brow.size() == ag.group_by_count) {
            for (int& j : brow) {
                if (j == -2) j = build_proj(build_key_col);
            }
            js->brow_group_proj = brow;
        }
        s.join = js;
    }
    PA_REQUIRE(ag.input_channel_count == (int32_t)to_proj.size(), PA_ERR_INVALID_ARGUMENT,
               jd ? "aggregation input channels must be the join output" : "aggregation input channels must be the projection outputs");
    const int32_t n_agg_in = (int32_t)to_proj.size();
    PA_REQUIRE(ag.step == PA_STEP_SINGLE || ag.step == PA_STEP_PARTIAL || ag.step == PA_STEP_FINAL, PA_ERR_INVALID_ARGUMENT, "unknown aggregation step");
    s.step = ag.step;
    for (int32_t g = 0; g < ag.group_by_count; g++) {
        int ch = ag.group_by_channels[g];
        PA_REQUIRE(ch >= 0 && ch < n_agg_in, PA_ERR_INVALID_ARGUMENT, "group-by channel out of range");
        s.group_proj.push_back(to_proj[ch]);
    }
    s.hash_channel = ag.hash_channel;
    for (int32_t k = 0; k < ag.aggregate_count; k++) {
        pa_aggregate a = ag.aggregates[k];
        PA_REQUIRE(a.fn == PA_AGG_COUNT_STAR || (a.input_channel >= 0 && a.input_channel < n_agg_in), PA_ERR_INVALID_ARGUMENT,
                   "aggregate input channel out of range");
        PA_REQUIRE(a.mask_channel < n_agg_in, PA_ERR_INVALID_ARGUMENT, "aggregate mask channel out of range");
        if (a.fn != PA_AGG_COUNT_STAR) a.input_channel = to_proj[a.input_channel];
        if (a.mask_channel >= 0) a.mask_channel = to_proj[a.mask_channel];
        // min / max over VARCHAR: strings of a declared length of at most 7 bytes have an order-preserving 64-bit image (bytes
        // big-endian, then the length: Slice.compareTo = unsigned bytes, then length) and ride the integer max machinery; longer or
        // unbounded strings stay with the Java operator
        auto short_varchar = [&](int proj) {
            const OwnedExpr& pe = s.proj[proj];
            if (!pe.is_input_ref()) return false;
            const int ch = pe.node(pe.root).channel;
            return ch >= 0 && ch < s.n_in && s.in_params[ch] >= 1 && s.in_params[ch] <= 7;
        };
        if ((a.fn == PA_AGG_MIN || a.fn == PA_AGG_MAX) && ag.step != PA_STEP_FINAL && a.input_channel >= 0 && s.proj[a.input_channel].root_type() == PA_VARCHAR) {
            PA_REQUIRE(short_varchar(a.input_channel), PA_ERR_NOT_SUPPORTED, "min/max over VARCHAR: only channels declared VARCHAR(n), n <= 7, are on the device path");
        }
        if (s.step == PA_STEP_FINAL) {
            // intermediate input: [count BIGINT] for count / count(*), [count BIGINT, sum] for sum / avg
            PA_REQUIRE(a.input_channel >= 0 && s.proj[a.input_channel].root_type() == PA_BIGINT, PA_ERR_INVALID_ARGUMENT,
                       "FINAL step: the aggregate's first state channel must be the BIGINT count");
            PA_REQUIRE(a.mask_channel < 0, PA_ERR_INVALID_ARGUMENT, "FINAL step takes no mask");
            if (a.fn == PA_AGG_MIN || a.fn == PA_AGG_MAX) {
                PA_REQUIRE(a.input_channel + 1 < fp.projection_count, PA_ERR_INVALID_ARGUMENT, "FINAL step: missing value state channel");
                PA_REQUIRE(s.proj[a.input_channel + 1].root_type() != PA_VARCHAR || short_varchar(a.input_channel + 1), PA_ERR_NOT_SUPPORTED,
                           "min/max over VARCHAR: only state channels declared VARCHAR(n), n <= 7, are on the device path");
            }
            if (a.fn == PA_AGG_SUM || a.fn == PA_AGG_AVG) {
                PA_REQUIRE(a.input_channel + 1 < fp.projection_count, PA_ERR_INVALID_ARGUMENT, "FINAL step: missing sum state channel");
                int32_t t = s.proj[a.input_channel + 1].root_type();
                PA_REQUIRE(t == PA_DOUBLE || (a.fn == PA_AGG_SUM && t == PA_BIGINT) || t == PA_LONG_DECIMAL, PA_ERR_INVALID_ARGUMENT, "FINAL step: bad sum state type");
                // (sum / avg over DECIMAL: the state's sum is a DECIMAL(38, s); pa_aggregate.input_type names the aggregate's RESULT type)
                PA_REQUIRE(t != PA_LONG_DECIMAL || a.input_type == PA_DECIMAL || a.input_type == PA_LONG_DECIMAL, PA_ERR_INVALID_ARGUMENT,
                           "FINAL step over a DECIMAL sum state: input_type is the aggregate's result type");
            }
        }
        else if (a.fn != PA_AGG_COUNT_STAR) {
            int32_t t = s.proj[a.input_channel].root_type();
            const bool min_max = a.fn == PA_AGG_MIN || a.fn == PA_AGG_MAX;
            PA_REQUIRE(a.fn == PA_AGG_COUNT || t == PA_DOUBLE || t == PA_REAL || t == PA_BIGINT || t == PA_INTEGER || t == PA_DECIMAL ||
                           (!min_max && t == PA_LONG_DECIMAL) || (min_max && (t == PA_DATE || t == PA_BOOLEAN || t == PA_VARCHAR)),
                       PA_ERR_NOT_SUPPORTED, "aggregate input type not supported on device");
        }
        s.aggs.push_back(a);
    }
    s.expected_groups = ag.expected_groups;
    PA_REQUIRE(ag.max_partial_memory >= 0, PA_ERR_INVALID_ARGUMENT, "maxPartialMemory must not be negative");
    s.max_partial_memory = ag.step == PA_STEP_PARTIAL ? ag.max_partial_memory : 0;
    s.output_mem = ag.output_mem;
    finalize_spec(s);
    return s;
}
Spec make_spec(const pa_fused_aggregation_desc* d) { return make_spec(d->filter_project, d->aggregation); }

// channels read, short / interned VARCHAR keys: everything of a Spec that follows from its expressions and aggregates
void finalize_spec(Spec& s)
{
    // channels actually read
    std::set<int32_t> used;
    if (s.has_filter) s.filter.collect_channels(&used);
    std::set<int> used_proj(s.group_proj.begin(), s.group_proj.end());
    if (s.join) {
        used_proj.insert(s.join->key_proj);
        used_proj.insert(s.join->brow_group_proj.begin(), s.join->brow_group_proj.end());
    }
    for (const auto& a : s.aggs) {
        if (a.fn != PA_AGG_COUNT_STAR || s.step == PA_STEP_FINAL) used_proj.insert(a.input_channel);
        if (s.step == PA_STEP_FINAL && a.fn != PA_AGG_COUNT && a.fn != PA_AGG_COUNT_STAR) used_proj.insert(a.input_channel + 1);  // the value state
        if (a.mask_channel >= 0) used_proj.insert(a.mask_channel);
    }
    for (int j : used_proj) s.proj[j].collect_channels(&used);
    s.used_channel.assign(s.n_in, false);
    const int n_virtual = s.join ? (int)s.join->build_cols.size() : 0;
    for (int32_t c : used) {
        PA_REQUIRE(c >= 0 && c < s.n_in + n_virtual, PA_ERR_INVALID_ARGUMENT, "expression references a channel outside the page");
        if (c < s.n_in) s.used_channel[c] = true;
    }
    s.lazy_channel.assign(s.n_in, false);
    if (s.join) {
        // the filter and the probe key run for every row; what only the aggregation reads is loaded for the matches
        std::set<int32_t> eager;
        if (s.has_filter) s.filter.collect_channels(&eager);
        s.proj[s.join->key_proj].collect_channels(&eager);
        for (int c = 0; c < s.n_in; c++) s.lazy_channel[c] = s.used_channel[c] && !eager.count(c);
    }
    s.short_bound.assign(s.n_in, 0);
    s.interned.assign(s.n_in, false);
    for (int j : s.group_proj) {
        const OwnedExpr& pe = s.proj[j];
        if (pe.is_input_ref() && pe.root_type() == PA_VARCHAR) {
            int c = pe.node(pe.root).channel;
            if (s.in_params[c] >= 1 && s.in_params[c] <= 7) s.short_bound[c] = s.in_params[c];
            if (s.in_params[c] >= 1 && s.in_params[c] <= 15) continue;  // fits the packed key: bytes in the key words
            // unknown or long bound: the key is compared through its interned id -- possible when nothing but the grouping
            // (and count(), which only looks at the NULL flag) reads the channel
            bool only_key = !(s.has_filter && [&] { std::set<int32_t> f; s.filter.collect_channels(&f); return f.count(c) != 0; }());
            for (size_t q = 0; q < s.proj.size() && only_key; q++) {
                std::set<int32_t> ch;
                s.proj[q].collect_channels(&ch);
                if (!ch.count(c)) continue;
                only_key = s.proj[q].is_input_ref();
                for (const pa_aggregate& a : s.aggs) {
                    if (a.mask_channel == (int32_t)q) only_key = false;
                    if (a.input_channel == (int32_t)q && a.fn != PA_AGG_COUNT && a.fn != PA_AGG_COUNT_STAR) only_key = false;
                    if (s.step == PA_STEP_FINAL && (a.input_channel == (int32_t)q || a.input_channel + 1 == (int32_t)q)) only_key = false;
                }
            }
            if (only_key) {
                s.interned[c] = true;
                continue;
            }
            // a declared bound the packed key cannot hold is refused now, so that the planner keeps the Java operator; for an
            // undeclared bound (0) the operator is optimistic: a key longer than 15 bytes fails the query at run time
            PA_REQUIRE(s.in_params[c] <= 15, PA_ERR_NOT_SUPPORTED,
                       "VARCHAR group keys longer than 15 bytes that other expressions read are not on the device path");
        }
    }
    for (int c = 0; c < s.n_in; c++) {
        if (!s.interned[c]) continue;
        s.in_types[c] = PA_INTEGER;
        for (OwnedExpr& pe : s.proj) {
            if (pe.is_input_ref() && pe.node(pe.root).channel == c) pe.nodes[pe.root].type = PA_INTEGER;
        }
    }
    for (int c = 0; c < s.n_in; c++) {
        if (s.interned[c] || s.in_types[c] == PA_VARCHAR) s.lazy_channel[c] = false;  // strings are handed to the row function whole
    }
}

// ---- source generation -------------------------------------------------------------------------

// first-fit bit packing of the key parts into 64-bit words
struct KeyPacker {
    std::vector<int> used;  // bits used per word
    int place(int bits, int* shift)
    {
        for (size_t w = 0; w < used.size(); w++) {
            if (used[w] + bits <= 64) {
                *shift = used[w];
                used[w] += bits;
                return (int)w;
            }
        }
        used.push_back(bits);
        *shift = 0;
        return (int)used.size() - 1;
    }
};

// Words of a range-table entry, in this order: per used channel its values pointer, its offsets pointer when it is a VARCHAR
// channel, its NULL flags pointer when the layout calls it nullable; then rows | (vec << 32).
int range_entry_words(const Spec& s, const std::vector<ChannelLayout>& layout)
{
    int words = 1;
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used_channel[c]) continue;
        words += 1 + (layout[c].type == PA_VARCHAR ? 1 : 0) + (layout[c].nullable ? 1 : 0);
    }
    return words;
}

KernelInfo generate(const Spec& s, const std::vector<ChannelLayout>& layout, int variant)
{
    KernelInfo k;
    const bool ranged = variant == V_GLOBAL_R || variant == V_LDS_R;
    if (ranged) {
        PA_REQUIRE(!s.join, PA_ERR_NOT_SUPPORTED, "no range-table variant behind a probe stage");
        variant = variant == V_GLOBAL_R ? V_GLOBAL : V_LDS;
    }
    k.variant = variant;
    k.ranged = ranged;
    k.entry = ranged ? "pa_fused_ranges" : "pa_fused";
    k.block = variant == V_LDS ? 64 : 256;
    k.c = variant == V_LDS ? kLdsSlots : 0;

    PA_REQUIRE(variant != V_BROW || (s.join && !s.join->brow_group_proj.empty()), PA_ERR_NOT_SUPPORTED, "no build-row variant for this plan");
    PA_REQUIRE(!s.join || (variant != V_HASH && variant != V_LDSP), PA_ERR_NOT_SUPPORTED, "no hash-partitioned variants behind a probe stage");
    RowInputs ri;
    ri.n_in = s.n_in;
    ri.used = s.used_channel;
    for (int c = 0; c < s.n_in; c++) ri.used[c] = s.used_channel[c] && !s.lazy_channel[c];
    ri.short_bound = s.short_bound;
    std::ostringstream body;  // inside pa_row
    // the page's channels, then the build columns of the probe stage as channels n_in + v (`layout` may already hold them:
    // their nullability is the lookup source's)
    std::vector<ChannelLayout> ext(layout.begin(), layout.begin() + s.n_in);
    if (s.join) {
        for (size_t v = 0; v < s.join->build_cols.size(); v++) {
            ChannelLayout cl;
            cl.type = s.join->build_types[v];
            cl.nullable = (size_t)s.n_in + v < layout.size() ? layout[(size_t)s.n_in + v].nullable : true;
            ext.push_back(cl);
        }
    }
    RowCodegen gen(ext, "a.err");

    // 1. filter
    std::string sel = "true";
    if (s.has_filter) {
        GenValue f = gen.emit(s.filter, body);
        sel = f.nullable() ? "(!" + f.n + " && " + f.v + ")" : f.v;  // PageFunctionCompiler.java:539-542
    }
    std::ostringstream pre;  // probe stage: body of pa_pre (filter, then the probe key of the rows it keeps)
    if (s.join) {
        // 1b. rows the filter keeps look their key up; a NULL key matches nothing (JoinProbe.java:89-91)
        pre << body.str() << "sel0 = live && " << sel << ";\njk = 0ULL;\nif (sel0) {\n";
        GenValue pk = gen.emit(s.proj[s.join->key_proj], pre);
        if (pk.nullable()) pre << "if (" << pk.n << ") sel0 = false; else ";
        pre << "jk = (u64)(i64)" << pk.v << ";\n}\n";
        body.str("");
        // what follows (pa_post) runs per row with `jb`, the build position of the match -- the lookup source has no duplicate
        // keys, so it is the only one -- or -1
        body << "const bool sel = jb >= 0;\n";
    }
    else {
        body << "const bool sel = live && " << sel << ";\n";
    }

    // 2. projections used downstream, evaluated once, only for selected rows
    std::ostringstream inner;
    std::map<int, GenValue> pv;
    auto proj_value = [&](int j) -> const GenValue& {
        auto it = pv.find(j);
        if (it == pv.end()) it = pv.emplace(j, gen.emit(s.proj[j], inner)).first;
        return it->second;
    };

    // 3. group keys -> bit-packed words
    KeyPacker packer;
    std::vector<std::vector<std::string>> word_terms;  // per word: OR-ed terms
    auto add_term = [&](int word, const std::string& term) {
        if ((int)word_terms.size() <= word) word_terms.resize(word + 1);
        word_terms[word].push_back(term);
    };
    // BROW: the key words are not computed per row -- the slot is the build position -- but once per group, by pa_brow_keys, from
    // the build columns alone
    const bool brow = variant == V_BROW;
    const std::vector<int>& gp = brow ? s.join->brow_group_proj : s.group_proj;
    std::ostringstream key_os;
    std::map<int, GenValue> kpv;
    auto key_value = [&](int j) -> const GenValue& {
        if (!brow) return proj_value(j);
        auto it = kpv.find(j);
        if (it == kpv.end()) it = kpv.emplace(j, gen.emit(s.proj[j], key_os)).first;
        return it->second;
    };
    std::ostringstream& kinner = brow ? key_os : inner;
    for (size_t gi = 0; gi < gp.size(); gi++) {
        const GenValue& kv = key_value(gp[gi]);
        const OwnedExpr& pe = s.proj[gp[gi]];
        KeyPart part;
        part.type = kv.type;
        std::string value;  // u64 expression already confined to `bits` bits
        switch (kv.type) {
            case PA_BIGINT:
            case PA_DECIMAL:  // ShortDecimalType: equal values are equal longs
                part.bits = 64;
                value = "(u64)" + kv.v;
                break;
            case PA_INTEGER:
            case PA_DATE:
                part.bits = 32;
                value = "(u64)(u32)(i32)" + kv.v;
                break;
            case PA_REAL:
                part.bits = 32;  // the key's canonical bits: -0 == +0, NaN == NaN (RealType.java:127-140), hashed as RealType hashes them
                value = "(u64)pa_real_key_bits(" + kv.v + ")";
                break;
            case PA_BOOLEAN:
                part.bits = 1;
                value = "(" + kv.v + " ? 1ULL : 0ULL)";
                break;
            case PA_DOUBLE:
                // IS NOT DISTINCT semantics of the group key: -0 == +0, NaN == NaN (DoubleType.java:163-184)
                part.bits = 64;
                value = "((" + kv.v + " == 0.0) ? 0ULL : ((" + kv.v + " != " + kv.v + ") ? 0x7ff8000000000000ULL : (u64)__double_as_longlong(" +
                        kv.v + ")))";
                break;
            case PA_VARCHAR: {
                int ch = pe.is_input_ref() ? pe.node(pe.root).channel : -1;
                if (ch >= 0 && ch < s.n_in && s.short_bound[ch] > 0) {
                    part.bound = s.short_bound[ch];
                    part.bits = 8 * part.bound + 4;
                    value = "(cs" + std::to_string(ch) + " | ((u64)" + kv.len + " << " + std::to_string(8 * part.bound) + "))";
                }
                else {
                    part.bits = 128;
                }
                break;
            }
            default:
                throw Error(PA_ERR_NOT_SUPPORTED, "group key type not supported on device");
        }
        std::string guard = kv.nullable() ? "(" + kv.n + ") ? 0ULL : " : "";
        if (part.bits == 128) {
            // up to 15 bytes in two dedicated words, length in the top byte of the second
            int sh;
            part.word = packer.place(64, &sh);
            int w2 = packer.place(64, &sh);
            PA_REQUIRE(w2 == part.word + 1, PA_ERR_NOT_SUPPORTED, "internal: long VARCHAR key words not adjacent");
            std::string id = "ks" + std::to_string(gi);
            kinner << "u64 " << id << "a = 0, " << id << "b = 0;\n";
            kinner << "if (" << (kv.nullable() ? "!" + kv.n : "true") << ") {\n";
            kinner << "  if (" << kv.len << " > 15) pa_raise(a.err, -3);\n";
            kinner << "  for (i32 b = 0; b < " << kv.len << " && b < 15; b++) {\n";
            kinner << "    if (b < 8) " << id << "a |= (u64)" << kv.v << "[b] << (8 * b); else " << id << "b |= (u64)" << kv.v
                  << "[b] << (8 * (b - 8));\n  }\n";
            kinner << "  " << id << "b |= (u64)" << kv.len << " << 56;\n}\n";
            add_term(part.word, id + "a");
            add_term(part.word + 1, id + "b");
        }
        else {
            part.word = packer.place(part.bits, &part.shift);
            add_term(part.word, "((" + guard + value + ") << " + std::to_string(part.shift) + ")");
        }
        if (kv.nullable()) {
            part.null_word = packer.place(1, &part.null_shift);
            add_term(part.null_word, "((" + kv.n + ") ? " + std::to_string(1ULL << part.null_shift) + "ULL : 0ULL)");
        }
        k.keys.push_back(part);
    }
    k.w = (int)packer.used.size();
    PA_REQUIRE(k.w <= 8, PA_ERR_NOT_SUPPORTED, "group key wider than 8 words");

    // 4. accumulator words, shared between aggregates over the same (input, mask)
    struct WordDef { int kind; std::string cond; std::string val; };
    std::vector<WordDef> words;
    std::map<std::string, int> word_index;
    auto word = [&](int kind, const std::string& cond, const std::string& val, const std::string& key) {
        auto it = word_index.find(key);
        if (it != word_index.end()) return it->second;
        words.push_back({kind, cond, val});
        word_index[key] = (int)words.size() - 1;
        return (int)words.size() - 1;
    };
    // min / max: u64 maximum of an order-preserving image (pa_img_*; min takes the complement), see pa_device.h
    auto minmax_image = [&](const GenValue& x, bool is_min) {
        std::string img;
        switch (x.type) {
            case PA_BIGINT:
            case PA_INTEGER:
            case PA_DECIMAL:  // ShortDecimalType's comparison is the longs' (one scale)
            case PA_DATE: img = "pa_img_i64((i64)" + x.v + ")"; break;
            case PA_DOUBLE: img = "pa_img_f64(" + x.v + ")"; break;
            case PA_REAL: img = "pa_img_f64((double)" + x.v + ")"; break;  // (float order = order of the widened values)
            case PA_BOOLEAN: img = "(" + x.v + " ? 1ULL : 0ULL)"; break;
            case PA_VARCHAR: img = "pa_img_str7(" + x.v + ", " + x.len + ", a.err)"; break;
            default: throw Error(PA_ERR_NOT_SUPPORTED, "min/max input type not supported on device");
        }
        return is_min ? "(~" + img + ")" : img;
    };
    for (const auto& ag : s.aggs) {
        if (s.step == PA_STEP_FINAL) {
            // combine functions (DoubleSumAggregation.combine, AverageAggregations.combine, CountAggregation.combine,
            // LongSumAggregation.combine: SURVEY a15): counts and sums of the partial states add up
            const GenValue& c = proj_value(ag.input_channel);
            std::string ch = std::to_string(ag.input_channel);
            std::string ccond = c.nullable() ? "(!" + c.n + ")" : "true";
            int cw = word(W_CNT, ccond, c.v, "fcnt|" + ch);
            int vw = -1;
            if (ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_AVG) {
                const GenValue& v = proj_value(ag.input_channel + 1);
                std::string vcond = v.nullable() ? "(" + ccond + " && !" + v.n + ")" : ccond;
                if (v.type == PA_LONG_DECIMAL) {  // the sum half of a DECIMAL state: limbs again (combine = add)
                    const int limbs = decimal_limbs_for_bits(128);
                    for (int l = 0; l < limbs; l++) {
                        const int w = word(W_SUMI, vcond, "pa_dec_limb(" + v.v + ", " + std::to_string(l) + ", " + std::to_string(limbs - 1) + ")", "fdec" + std::to_string(l) + "|" + ch);
                        if (l == 0) vw = w;
                        PA_REQUIRE(w == vw + l, PA_ERR_NOT_SUPPORTED, "internal: the limb words of a DECIMAL sum are not adjacent");
                    }
                    k.agg_words.emplace_back(cw, vw);
                    k.agg_limbs.resize(k.agg_words.size(), 0);
                    k.agg_limbs.back() = limbs;
                    continue;
                }
                vw = word(v.type == PA_DOUBLE ? W_SUMF : W_SUMI, vcond, v.v, "fsum|" + ch);
            }
            else if (ag.fn == PA_AGG_MIN || ag.fn == PA_AGG_MAX) {
                // AbstractMinMaxAggregationFunction.combine: compare-and-update with the other state's value
                const GenValue& v = proj_value(ag.input_channel + 1);
                std::string vcond = v.nullable() ? "(" + ccond + " && !" + v.n + ")" : ccond;
                vw = word(W_MAXU, vcond, minmax_image(v, ag.fn == PA_AGG_MIN), std::string(ag.fn == PA_AGG_MIN ? "fmin|" : "fmax|") + ch);
            }
            k.agg_words.emplace_back(cw, vw);
            continue;
        }
        std::string cond = "true", ckey = "m" + std::to_string(ag.mask_channel);
        if (ag.mask_channel >= 0) {
            const GenValue& m = proj_value(ag.mask_channel);
            PA_REQUIRE(m.type == PA_BOOLEAN, PA_ERR_INVALID_ARGUMENT, "mask channel must be BOOLEAN");
            cond = m.nullable() ? "(!" + m.n + " && " + m.v + ")" : "(" + m.v + ")";  // CompilerOperations.java:65-74
        }
        if (ag.fn == PA_AGG_COUNT_STAR) {
            k.agg_words.emplace_back(word(W_CNT, cond, "1", "cnt|*|" + ckey), -1);
            continue;
        }
        const GenValue& x = proj_value(ag.input_channel);
        std::string xkey = s.proj[ag.input_channel].fingerprint();
        std::string ccond = cond, cntkey = "cnt|*|" + ckey;
        if (x.nullable()) {
            ccond = "(" + cond + " && !" + x.n + ")";
            cntkey = "cnt|" + xkey + "|" + ckey;
        }
        // sum / min / max only ask "was there any input?" -- and a GROUP exists because a row created it: with a non-null
        // input and no mask the answer is always yes, so the group needs no count word (one HBM atomic less per row on the
        // table tier: Q3's sum(revenue) keeps ONE accumulator word).  -1 = "counts as 1" for every consumer of agg_words.
        // (Step.PARTIAL keeps the real count: its [count, value] state channels are part of the boundary, include/presto_amd.h)
        const bool implicit_count = s.step == PA_STEP_SINGLE && !s.group_proj.empty() && ag.mask_channel < 0 && !x.nullable() &&
                                    (ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_MIN || ag.fn == PA_AGG_MAX);
        int cw = implicit_count ? -1 : word(W_CNT, ccond, "1", cntkey);
        int vw = -1;
        if ((ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_AVG) && (x.type == PA_DECIMAL || x.type == PA_LONG_DECIMAL)) {
            // DecimalSumAggregation / DecimalAverageAggregation: the exact sum as limb words, shared between sum(x) and avg(x)
            // (as many limbs as the TYPE's precision needs: |x| < 10^p by the planner's type derivation; the top limb is signed and
            // takes whatever is left of a value that breaks it, up to 63 bits)
            const int limbs = decimal_limbs_for_bits(std::min(x.type == PA_DECIMAL ? 64 : 128, decimal_bits_for_precision(PA_DECIMAL_PRECISION(x.param))));
            for (int l = 0; l < limbs; l++) {
                const int w = word(W_SUMI, ccond, "pa_dec_limb((i128)" + x.v + ", " + std::to_string(l) + ", " + std::to_string(limbs - 1) + ")",
                                   "dec" + std::to_string(l) + "/" + std::to_string(limbs) + "|" + xkey + "|" + ckey);
                if (l == 0) vw = w;
                PA_REQUIRE(w == vw + l, PA_ERR_NOT_SUPPORTED, "internal: the limb words of a DECIMAL sum are not adjacent");
            }
            k.agg_words.emplace_back(cw, vw);
            k.agg_limbs.resize(k.agg_words.size(), 0);
            k.agg_limbs.back() = limbs;
            continue;
        }
        // (REAL inputs: RealSumAggregation / RealAverageAggregation keep a DOUBLE sum of the widened floats -- the same accumulator
        // words as for DOUBLE; the output functions narrow the result)
        if (ag.fn == PA_AGG_SUM && x.type != PA_DOUBLE && x.type != PA_REAL) {
            vw = word(W_SUMI, ccond, x.v, "sumi|" + xkey + "|" + ckey);
        }
        else if (ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_AVG) {
            std::string v = x.type == PA_DOUBLE ? x.v : "((double)" + x.v + ")";  // AverageAggregations.java:34-39
            vw = word(W_SUMF, ccond, v, std::string("sumf|") + (x.type == PA_DOUBLE ? "d|" : "i|") + xkey + "|" + ckey);
        }
        else if (ag.fn == PA_AGG_MIN || ag.fn == PA_AGG_MAX) {
            vw = word(W_MAXU, ccond, minmax_image(x, ag.fn == PA_AGG_MIN), std::string(ag.fn == PA_AGG_MIN ? "min|" : "max|") + xkey + "|" + ckey);
        }
        k.agg_words.emplace_back(cw, vw);
    }
    k.nw = (int)words.size();
    k.agg_limbs.resize(k.agg_words.size(), 0);
    PA_REQUIRE(k.nw > 0 || k.w > 0, PA_ERR_INVALID_ARGUMENT, "aggregation without aggregates and keys");
    if (k.nw == 0) {  // DISTINCT-style group by without aggregates: keep a row count so the kernels stay uniform
        words.push_back({W_CNT, "true", "1"});
        k.nw = 1;
    }
    for (const auto& w : words) k.word_kind.push_back(w.kind);
    if (variant == V_BROW) {
        // a word every row of a group updates tells whether the build row has a group: a count is then > 0; a DOUBLE sum that
        // starts at -0.0 and only ever takes values canonicalised by + 0.0 (x + 0.0 is x, except that -0.0 becomes +0.0) is then
        // anything but -0.0 -- and equals what the reference computes, whose sum starts at +0.0 (0.0 + -0.0 = 0.0)
        for (size_t w = 0; w < words.size() && k.occ_word < 0; w++) {
            if (words[w].cond != "true" || (words[w].kind != W_CNT && words[w].kind != W_SUMF)) continue;
            k.occ_word = (int)w;
            k.occ_empty = words[w].kind == W_SUMF ? 0x8000000000000000ULL : 0ULL;
        }
        if (getenv("PRESTO_AMD_BROW_TAGS")) k.occ_word = -1;  // (test switch: the tag-storing form)
    }
    {
        std::vector<std::string> names(words.size(), "rows");
        for (const auto& kv : word_index) names[(size_t)kv.second] = kv.first;
        std::ostringstream id;
        for (const KeyPart& kp : k.keys) {
            id << kp.type << ',' << kp.word << ',' << kp.shift << ',' << kp.bits << ',' << kp.bound << ',' << kp.null_word << ',' << kp.null_shift << ';';
        }
        id << '#';
        for (size_t w = 0; w < words.size(); w++) id << words[w].kind << ':' << names[w] << ';';
        k.layout_id = id.str();
    }
    if (variant == V_LDS) {
        PA_REQUIRE((size_t)k.nw * kLdsSlots * 64 * 8 <= 64 * 1024, PA_ERR_NOT_SUPPORTED, "too many accumulator words for the LDS variant");
    }
    if (variant == V_LDSH || variant == V_LDSP) {
        // one table per workgroup: tag + key words + accumulator words per slot.  A 1024-thread workgroup with 150 of the CU's
        // 160 KB of LDS (4096 slots for a one-word key and two accumulator words) against two 512-thread workgroups with 62 KB
        // each, measured over 64 M rows: 300 groups 64 -> 77 G rows/s, 1 K 26 -> 61 G, 2 K 21 -> 45 G (these now fit the
        // table without the hash-partitioning passes), 20 K 15 -> 18 G, 100 K 13 -> 15 G (fewer, denser partitions)
        const size_t slot_bytes = 8 * (size_t)(1 + std::max(k.w, 1) + k.nw);
        static const size_t budget = [] {
            const char* e = getenv("PRESTO_AMD_LDSH_KB");
            return (size_t)(e ? atoi(e) : 150) * 1024;
        }();
        int lc = 4096;
        while (lc > 32 && (size_t)lc * slot_bytes > budget) lc >>= 1;
        PA_REQUIRE((size_t)lc * slot_bytes <= budget, PA_ERR_NOT_SUPPORTED, "group state too wide for the LDS-table variant");
        k.lc = lc;
        k.block = budget > 64 * 1024 ? 1024 : 512;
    }

    // ---- assemble the translation unit ----
    std::ostringstream src;
    // (BROW: the row loop's "key" is the build position, one word; PA_TW = key words of the table, written by pa_brow_keys)
    src << "#define PA_NW " << k.nw << "\n#define PA_KW " << (brow ? 1 : (k.w > 0 ? k.w : 1)) << "\n#define PA_TW " << (k.w > 0 ? k.w : 1) << "\n#define PA_C "
        << (k.c > 0 ? k.c : 1) << "\n";
    if (variant == V_GLOBAL) {
        src << "struct PaAcc {";
        for (int w = 0; w < k.nw; w++) src << (words[w].kind == W_SUMF ? " double" : (words[w].kind == W_MAXU ? " u64" : " i64")) << " w" << w << ";";
        src << " };\n";
    }
    else if (variant == V_LDS) {
        // key table of the wave in (scalar) registers; accumulators lane-private in LDS: word w of group g
        // of lane l lives at pa_accw[(w * C + g) * 64 + l], so no two lanes ever share an address
        src << "__shared__ u64 pa_accw[PA_NW * PA_C * 64];\n";
        src << "struct PaAcc { u64 tk[PA_C][PA_KW]; int tcount; u32 lane; };\n";
    }
    else if (variant == V_LDSH || variant == V_LDSP) {
        // Medium cardinality: the workgroup aggregates into an open-addressing table in LDS (ds_cmpst / ds_add: no HBM
        // atomics in the row loop -- atomics of many rows on a few HBM addresses retire at ~16 M/s per address on this
        // part), and adds its table to the HBM table once, at the end of the kernel.  A row whose group finds no room
        // in the LDS table (more than PA_LC / 2 groups seen by the workgroup) goes to the HBM table directly.
        int lc_bits = 0;
        while ((1 << lc_bits) < k.lc) lc_bits++;
        // the LDS table is indexed by the TOP bits of the 32-bit key hash: the low bits choose the partition (hash-partitioned
        // path) and the HBM-table slot, so rows of one partition would otherwise share their home slots
        // rows whose group finds the table this full go to the HBM table: half of it when the table is the workgroup's own for one
        // launch (its groups are flushed into HBM, which must have room), three quarters when it is a partition's table for good
        src << "#define PA_LC " << k.lc << "\n#define PA_LC_SHIFT " << (32 - lc_bits) << "\n#define PA_LT_LIMIT " << (variant == V_LDSP ? "(PA_LC * 3 / 4)" : "(PA_LC / 2)") << "\n";
        src << "__shared__ u64 pa_lt_tag[PA_LC];\n__shared__ u64 pa_lt_key[PA_LC * PA_KW];\n__shared__ u64 pa_lt_acc[PA_LC * PA_NW];\n"
               "__shared__ i32 pa_lt_count;\n";
        src << "struct PaAcc { PaGtView tv; PaGtCtr gt; PaGtCtr flush; i64 fell; };\n";
        // same claim / publish protocol as pa_gt_upsert_n, on LDS: tag 0 -> busy -> ready, wave-uniform loop so that a
        // lane waiting for a slot another lane of its wave is publishing cannot starve it
        src << "__device__ __forceinline__ int pa_lt_upsert(const u32 h, const u64 (&k)[PA_KW])\n{\n"
               "    const u64 busy = ((u64)h << 2) | 1ULL, ready = ((u64)h << 2) | 3ULL;\n"
               "    u32 i = h >> PA_LC_SHIFT;\n    u32 probes = 0;\n    int spins = 0;\n    int result = -2;\n"
               "    while (__ballot(result == -2) != 0ULL) {\n        if (result == -2) {\n"
               "            const u64 t = __hip_atomic_load(&pa_lt_tag[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
               "            bool advance = false;\n"
               "            if (t == 0ULL) {\n"
               "                if (__hip_atomic_load(&pa_lt_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= PA_LT_LIMIT) result = -1;\n"
               "                else {\n"
               "                    u64 expected = 0ULL;\n"
               "                    if (__hip_atomic_compare_exchange_strong(&pa_lt_tag[i], &expected, busy, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {\n"
               "#pragma unroll\n                        for (int w = 0; w < PA_KW; w++) pa_lt_key[i * PA_KW + w] = k[w];\n"
               "                        __hip_atomic_store(&pa_lt_tag[i], ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
               "                        __hip_atomic_fetch_add(&pa_lt_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
               "                        result = (int)i;\n                    }\n                }\n            }\n"
               "            else if ((t | 2ULL) == ready) {\n"
               "                if (t == busy) { if (++spins > (1 << 20)) result = -1; }\n"
               "                else {\n                    bool eq = true;\n#pragma unroll\n"
               "                    for (int w = 0; w < PA_KW; w++) eq = eq && (pa_lt_key[i * PA_KW + w] == k[w]);\n"
               "                    if (eq) result = (int)i; else advance = true;\n                }\n            }\n"
               "            else advance = true;\n"
               "            if (advance) { i = (i + 1) & (PA_LC - 1); if (++probes >= PA_LC) result = -1; }\n"
               "        }\n    }\n    return result;\n}\n";
    }
    else if (variant == V_HASH) {
        // Hash-partitioning pass in front of the LDS-table variant at medium cardinality (hundreds to ~10^5 groups): it only
        // computes every row's partition = hash(key) mod P (P + 1 for rows the filter drops).  The rows are then taken in
        // partition order, a contiguous slice per workgroup, so that a workgroup's LDS table meets a few partitions' groups only.
        // (round 3) ... and histograms every 8192-row tile on the way (the tiles of the multisplit behind it, scan_kernels.hpp): the
        // multisplit's own counting pass read the ids a second time
        src << "struct PaAcc { int unused; };\n__shared__ i32 pa_hist[4097];\n";
    }
    else if (brow) {
        // Build-row table.  Every scattered store / atomic INSTRUCTION of a wave costs the CU on the order of 100 ns whatever the
        // number of active lanes (measured on Q3's lineitem pages: a flush wherever a thread's key changes -- up to five divergent
        // tag-store + atomic sequences per quad -- 1.77 ms per 2^28-row page; one sequence per quad 1.15 ms).  So the rows of a quad
        // are only NOTED (slot r of the thread: build position, flags, values; a row continuing its predecessor's build position
        // takes that one's values over), and at the end of the quad the wave's noted rows -- a dozen of its 256 when 5 % match --
        // are compacted through LDS and go out together: one tag store and one atomic per accumulator word for up to 64 of them.
        // The slot is the build position: nothing to search, nothing to claim.  The tag only says "this build row has a group"
        // -- a plain store into its own array: every writer stores the same value, so the XCD L2s need not agree on the line before
        // the kernel ends; the key words are written once per group by pa_brow_keys.  (One record [tag, words] per build position
        // instead of word-major arrays was 3 x slower: the memory-side atomics of neighbouring build rows share 64-byte requests
        // only while the words of a kind lie side by side.)
        auto wtype = [&](int w) { return std::string(words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")); };
        src << "struct PaAcc { PaGtView tv; PaGtCtr gt; bool ev[4]; u32 eg[4];";
        for (int w = 0; w < k.nw; w++) src << " bool eu" << w << "[4]; " << wtype(w) << " ex" << w << "[4];";
        src << " };\n";
        // ... and they do not go out one by one (round 3).  Every wave walks ONE contiguous row range of the page, so when the probe
        // side is clustered by the join key -- a fact table ordered by the key of its dimension, lineitem by orderkey -- the build
        // positions a wave meets rise with its rows.  The wave keeps a WINDOW of PA_WIN consecutive build positions in LDS
        // (accumulator words + one touched bit per position): a noted row inside the window is an LDS atomic (ds_add_f64 / ds_add_u64
        // / ds_max_u64) -- no HBM traffic, no waiting --, a row beyond it first flushes the window and moves it there.  A flush
        // hands the touched positions to the table 64 consecutive positions per instruction: the memory-side atomics of one
        // instruction share a 64-byte request when their addresses are neighbours, so eight build rows go out per request where the
        // sorted drains of round 2 (128 noted rows, bitonic sort, one atomic per distinct position) reached about two -- and the
        // sort is gone.  Windows of different waves overlap only where their row ranges meet, and the flush is atomic, so nothing
        // here depends on the clustering for correctness: rows in no particular order move the window at most PA_WIN_MOVES times
        // per quad and then go to the table directly, one atomic each.
        int win = 256;
        while (win > 64 && (size_t)win * 8 * (size_t)k.nw * 4 > 48 * 1024) win >>= 1;
        src << "#define PA_WIN " << win << "u\n#define PA_WIN_MOVES 2\n";
        src << "__shared__ u64 pa_win[4][PA_NW][PA_WIN];\n__shared__ u64 pa_wtouch[4][PA_WIN / 64u];\n__shared__ u32 pa_sfill[4];\n__shared__ u32 pa_wbase[4];\n";
        src << "#define PA_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); "
               "__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\"); } while (0)\n";
        // noting row `slot` of the quad (a literal at every call site: the arrays stay in registers)
        src << "__device__ __forceinline__ void pa_acc(const PaFusedArgs& a, PaAcc& acc, const int slot, const bool sel, const u64 (&key)[PA_KW]";
        for (int w = 0; w < k.nw; w++) src << ", const bool u" << w << ", const " << wtype(w) << " x" << w;
        src << ")\n{\n  acc.ev[slot] = sel;\n  acc.eg[slot] = (u32)key[0];\n";
        for (int w = 0; w < k.nw; w++) src << "  acc.eu" << w << "[slot] = u" << w << "; acc.ex" << w << "[slot] = x" << w << ";\n";
        src << "  if (slot > 0 && sel && acc.ev[slot > 0 ? slot - 1 : 0] && acc.eg[slot > 0 ? slot - 1 : 0] == acc.eg[slot]) {\n    const int p = slot > 0 ? slot - 1 : 0;\n";
        for (int w = 0; w < k.nw; w++) {
            const std::string P = "acc.ex" + std::to_string(w) + "[p]", X = "acc.ex" + std::to_string(w) + "[slot]";
            std::string comb;
            if (words[w].kind == W_SUMF || words[w].kind == W_CNT) comb = P + " + " + X;
            else if (words[w].kind == W_SUMI) comb = "pa_add_exact(" + P + ", " + X + ", a.err)";
            else comb = "(" + X + " > " + P + " ? " + X + " : " + P + ")";
            src << "    if (acc.eu" << w << "[p]) { " << X << " = acc.eu" << w << "[slot] ? " << comb << " : " << P << "; acc.eu" << w << "[slot] = true; }\n";
        }
        src << "    acc.ev[p] = false;\n  }\n}\n";
        // one value for build position g -> table (the direct route, and the window's flush)
        auto emit_issue = [&](const std::string& ind, const std::string& g, const std::function<std::string(int)>& cond, const std::function<std::string(int)>& val) {
            if (k.occ_word < 0) src << ind << "acc.tv.tag[" << g << "] = 3ULL;\n";
            for (int w = 0; w < k.nw; w++) {
                const std::string W = std::to_string(w), idx = W + "ULL * cap + " + g, v = val(w), c = cond(w);
                src << ind << (c.empty() ? std::string() : "if (" + c + ") ");
                if (words[w].kind == W_SUMF) src << "pa_gt_add_f64(acc.tv.words, " << idx << ", " << v << (w == k.occ_word ? " + 0.0" : "") << ");\n";
                else if (words[w].kind == W_SUMI) src << "pa_gt_add_i64_exact(acc.tv.words, " << idx << ", " << v << ", a.err);\n";
                else if (words[w].kind == W_MAXU) src << "pa_gt_max_u64(acc.tv.words, " << idx << ", " << v << ");\n";
                else src << "pa_gt_add_u64(acc.tv.words, " << idx << ", (u64)" << v << ");\n";
            }
        };
        // the window -> table: lane l takes positions base + 64 k + l; only touched positions issue (and are reset)
        // (the issuing lanes are the ACTIVE ones, by rank: lanes that have left the row loop issue nothing, and the window is
        // complete all the same)
        src << "__device__ __forceinline__ void pa_window_flush(const PaFusedArgs& a, PaAcc& acc)\n{\n"
               "  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;\n  const u64 cap = (u64)a.gt_mask + 1ULL;\n"
               "  const u64 act = __ballot(true);\n  const u32 nact = (u32)__popcll(act), rank = (u32)__popcll(act & ((1ULL << lane) - 1ULL));\n"
               "  PA_WAVE_SYNC();\n"
               "  const u32 wbase = pa_wbase[wave];  // (in LDS: a lane that was not active when the window moved must see where it is)\n"
               "  for (u32 at = rank; at < PA_WIN; at += nact) {\n    const u64 touch = pa_wtouch[wave][at >> 6];\n"
               "    if ((touch >> (at & 63u)) & 1ULL) {\n      const u64 g = (u64)wbase + at;\n";
        emit_issue("      ", "g", [](int) { return std::string(); }, [&](int w) {
            const std::string X = "pa_win[wave][" + std::to_string(w) + "][at]";
            return words[w].kind == W_SUMF ? "__longlong_as_double((i64)" + X + ")" : (words[w].kind == W_MAXU ? X : "(i64)" + X);
        });
        for (int w = 0; w < k.nw; w++) src << "      pa_win[wave][" << w << "][at] = 0ULL;\n";
        src << "    }\n  }\n  PA_WAVE_SYNC();\n  for (u32 i = rank; i < PA_WIN / 64u; i += nact) pa_wtouch[wave][i] = 0ULL;\n  PA_WAVE_SYNC();\n}\n";
        src << "__device__ __forceinline__ void pa_drain(const PaFusedArgs& a, PaAcc& acc, const u32)\n{\n  pa_window_flush(a, acc);\n}\n";
        // end of a quad: the wave's noted rows go into the window, which moves on when they lie beyond it
        src << "__device__ __forceinline__ void pa_flush(const PaFusedArgs& a, PaAcc& acc, const bool)\n{\n"
               "  const u32 wave = threadIdx.x >> 6;\n  const u64 cap = (u64)a.gt_mask + 1ULL;\n"
               "  if (__ballot(acc.ev[0] || acc.ev[1] || acc.ev[2] || acc.ev[3]) == 0ULL) return;\n"
               "  u32 wbase = pa_wbase[wave];\n"
               "  for (int moves = 0;; moves++) {\n"
               "#pragma unroll\n    for (int e = 0; e < 4; e++) {\n      const u32 at = acc.eg[e] - wbase;\n      if (acc.ev[e] && at < PA_WIN) {\n";
        for (int w = 0; w < k.nw; w++) {
            const std::string W = std::to_string(w), L = "pa_win[wave][" + W + "][at]", X = "acc.ex" + W + "[e]";
            src << "        if (acc.eu" << W << "[e]) ";
            if (words[w].kind == W_SUMF) src << "__hip_atomic_fetch_add((double*)&" << L << ", " << X << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
            else if (words[w].kind == W_SUMI) {
                src << "{ const i64 o = (i64)__hip_atomic_fetch_add(&" << L << ", (u64)" << X << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); i64 r; "
                       "if (__builtin_add_overflow(o, " << X << ", &r)) pa_raise(a.err, PA_DEV_ERR_OUT_OF_RANGE); }\n";
            }
            else if (words[w].kind == W_MAXU) src << "__hip_atomic_fetch_max(&" << L << ", " << X << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
            else src << "__hip_atomic_fetch_add(&" << L << ", (u64)" << X << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
        }
        src << "        __hip_atomic_fetch_or(&pa_wtouch[wave][at >> 6], 1ULL << (at & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
               "        acc.ev[e] = false;\n      }\n    }\n"
               "    const bool left = acc.ev[0] || acc.ev[1] || acc.ev[2] || acc.ev[3];\n"
               "    if (__ballot(left) == 0ULL) break;\n"
               "    if (moves >= PA_WIN_MOVES) {\n"
               // rows in no particular order: the rest of the quad goes to the table directly
               "#pragma unroll\n      for (int e = 0; e < 4; e++) {\n        if (!acc.ev[e]) continue;\n        const u64 g = (u64)acc.eg[e];\n";
        emit_issue("        ", "g", [](int w) { return "acc.eu" + std::to_string(w) + "[e]"; }, [&](int w) { return "acc.ex" + std::to_string(w) + "[e]"; });
        src << "        acc.ev[e] = false;\n      }\n      break;\n    }\n"
               // the window moves to the smallest position still waiting (down to a 64-byte line of the table's word arrays)
               // (through LDS: a shuffle would read the registers of lanes that have left the loop)
               "    u32 gmin = 0xffffffffu;\n"
               "#pragma unroll\n    for (int e = 0; e < 4; e++) { if (acc.ev[e] && acc.eg[e] < gmin) gmin = acc.eg[e]; }\n"
               "    pa_sfill[wave] = 0xffffffffu;\n    PA_WAVE_SYNC();\n"
               "    if (left) __hip_atomic_fetch_min(&pa_sfill[wave], gmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
               "    PA_WAVE_SYNC();\n    gmin = pa_sfill[wave];\n"
               "    pa_window_flush(a, acc);\n    wbase = gmin & ~7u;\n    pa_wbase[wave] = wbase;\n    PA_WAVE_SYNC();\n  }\n}\n\n";
    }
    else {
        // pending run of the thread: consecutive selected rows with equal keys are combined before they touch the table
        src << "struct PaAcc { PaGtView tv; PaGtCtr gt; i32 pn; i32 prow; u64 pk[PA_KW];";
        for (int w = 0; w < k.nw; w++) src << " bool pu" << w << "; " << (words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")) << " px" << w << ";";
        src << " };\n";
    }
    const bool lds_table = variant == V_LDSH || variant == V_LDSP;
    const bool gt_like = variant == V_GT;  // a thread-private pending run in front of the table
    if (gt_like || lds_table) {
        // accumulation of one row into the workgroup's LDS table / the HBM table
        src << "__device__ __forceinline__ void " << (gt_like ? "pa_acc_now" : "pa_acc")
            << "(const PaFusedArgs& a, PaAcc& acc, const bool sel, const i32 row, " << (gt_like ? "const i32 nrows, " : "") << "const u64 (&key)[PA_KW]";
        for (int w = 0; w < k.nw; w++) src << ", const bool u" << w << ", const " << (words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")) << " x" << w;
        src << ")\n{\n";
        {
            src << "if (sel) {\n  const u32 h = pa_key_hash(key, PA_KW);\n";
            if (lds_table) {
                src << "  const int ls = pa_lt_upsert(h, key);\n  if (ls >= 0) {\n";
                for (int w = 0; w < k.nw; w++) {
                    std::string idx = "pa_lt_acc[ls * PA_NW + " + std::to_string(w) + "]";
                    if (words[w].kind == W_SUMF) {
                        src << "    if (u" << w << ") __hip_atomic_fetch_add((double*)&" << idx << ", x" << w << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
                    }
                    else if (words[w].kind == W_SUMI) {
                        src << "    if (u" << w << ") { i64 o = (i64)__hip_atomic_fetch_add(&" << idx << ", (u64)x" << w
                            << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); i64 r; if (__builtin_add_overflow(o, x" << w
                            << ", &r)) pa_raise(a.err, PA_DEV_ERR_OUT_OF_RANGE); }\n";
                    }
                    else if (words[w].kind == W_MAXU) {
                        src << "    if (u" << w << ") __hip_atomic_fetch_max(&" << idx << ", x" << w << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
                    }
                    else {
                        src << "    if (u" << w << ") __hip_atomic_fetch_add(&" << idx << ", " << ("(u64)x" + std::to_string(w))
                            << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
                    }
                }
                src << "  } else {\n  acc.fell++;\n";
            }
            src << "  int g = pa_gt_upsert<PA_KW>(acc.tv.tag, acc.tv.keys, a.gt_mask, h, key, acc.gt, a.gt_max_fill, a.err);\n";
            src << "  if (g >= 0) {\n    const u64 cap = (u64)a.gt_mask + 1ULL;\n";
            for (int w = 0; w < k.nw; w++) {
                std::string idx = std::to_string(w) + "ULL * cap + (u64)g";
                if (words[w].kind == W_SUMF) src << "    if (u" << w << ") pa_gt_add_f64(acc.tv.words, " << idx << ", x" << w << ");\n";
                else if (words[w].kind == W_SUMI) src << "    if (u" << w << ") pa_gt_add_i64_exact(acc.tv.words, " << idx << ", x" << w << ", a.err);\n";
                else if (words[w].kind == W_MAXU) src << "    if (u" << w << ") pa_gt_max_u64(acc.tv.words, " << idx << ", x" << w << ");\n";
                else src << "    if (u" << w << ") pa_gt_add_u64(acc.tv.words, " << idx << ", (u64)x" << w << ");\n";
            }
            // no room for this row's group: spill the row; the host rehashes and replays the spilled rows
            if (variant == V_GT) {
                src << "  } else {\n    const u32 sb = atomicAdd(a.spill_count, (u32)nrows);\n    for (i32 i = 0; i < nrows; i++) a.spill_rows[sb + (u32)i] = row + i;\n  }\n";
            }
            else {
                src << "  } else {\n    a.spill_rows[atomicAdd(a.spill_count, 1u)] = row;\n  }\n";
            }
            if (lds_table) src << "  }\n";
            src << "}\n";
            src << "}\n\n";
        }
        if (gt_like) {
            // Run combining.  A thread of the vector loop walks 4 consecutive rows; when their keys repeat (clustered inputs: a
            // fact table joined on its own key order) the rows are combined in registers and reach the table once -- one probe and
            // one atomic per word for the run.  pa_flush ends the pending run; the loops call it after every quad (every row
            // in the scalar / list loops), so a run is always a range of consecutive rows, which is what a spill records.
            src << "__device__ __forceinline__ void pa_flush(const PaFusedArgs& a, PaAcc& acc, const bool doit)\n{\n"
                   "  pa_acc_now(a, acc, doit && acc.pn > 0, acc.prow, acc.pn, acc.pk";
            for (int w = 0; w < k.nw; w++) src << ", acc.pu" << w << ", acc.px" << w;
            src << ");\n  if (doit) acc.pn = 0;\n}\n";
            src << "__device__ __forceinline__ void pa_acc(const PaFusedArgs& a, PaAcc& acc, const bool sel, const i32 row, const u64 (&key)[PA_KW]";
            for (int w = 0; w < k.nw; w++) src << ", const bool u" << w << ", const " << (words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")) << " x" << w;
            src << ")\n{\n  bool same = sel && acc.pn > 0;\n#pragma unroll\n  for (int w = 0; w < PA_KW; w++) same = same && key[w] == acc.pk[w];\n"
                   "  pa_flush(a, acc, !same);\n  if (sel) {\n    if (acc.pn == 0) {\n#pragma unroll\n      for (int w = 0; w < PA_KW; w++) acc.pk[w] = key[w];\n"
                   "      acc.prow = row;\n      acc.pn = 1;\n";
            for (int w = 0; w < k.nw; w++) src << "      acc.pu" << w << " = u" << w << "; acc.px" << w << " = x" << w << ";\n";
            src << "    } else {\n      acc.pn = row - acc.prow + 1;\n";
            for (int w = 0; w < k.nw; w++) {
                const std::string P = "acc.px" + std::to_string(w), U = "acc.pu" + std::to_string(w), X = "x" + std::to_string(w);
                std::string comb;
                if (words[w].kind == W_SUMF || words[w].kind == W_CNT) comb = P + " + " + X;
                else if (words[w].kind == W_SUMI) comb = "pa_add_exact(" + P + ", " + X + ", a.err)";
                else comb = "(" + X + " > " + P + " ? " + X + " : " + P + ")";
                src << "      if (u" << w << ") { " << P << " = " << U << " ? " << comb << " : " << X << "; " << U << " = true; }\n";
            }
            src << "    }\n  }\n}\n\n";
        }
    }
    // probe stage: the build columns at the match (channels n_in + v), and the page's channels only the aggregation reads
    std::ostringstream build_loads;
    if (s.join) {
        for (size_t v = 0; v < s.join->build_cols.size(); v++) {
            const std::string id = std::to_string(s.n_in + (int)v), V = std::to_string(v);
            const int32_t t = s.join->build_types[v];
            const std::string ct = RowCodegen::ctype(t);
            build_loads << "const " << ct << " c" << id << " = ";
            if (t == PA_BIGINT) build_loads << "((const i64*)a.bv[" << V << "])[jb];\n";
            else if (t == PA_INTEGER || t == PA_DATE) build_loads << "(i64)((const i32*)a.bv[" << V << "])[jb];\n";
            else if (t == PA_DOUBLE) build_loads << "((const double*)a.bv[" << V << "])[jb];\n";
            else if (t == PA_BOOLEAN) build_loads << "((const u8*)a.bv[" << V << "])[jb] != 0;\n";
            else throw Error(PA_ERR_NOT_SUPPORTED, "build column type not read by the fused probe");
            if (ext[(size_t)s.n_in + v].nullable) build_loads << "const bool cn" << id << " = a.bn[" << V << "] != nullptr && a.bn[" << V << "][jb] != 0;\n";
        }
    }
    // loads of the lazy channels of one row into the variables c<C><suffix> (cn<C><suffix>)
    auto lazy_assign = [&](const std::string& suffix, const std::string& row) {
        std::ostringstream o;
        for (int c = 0; c < s.n_in && s.join; c++) {
            if (!s.lazy_channel[c]) continue;
            const std::string C = std::to_string(c);
            const int32_t t = layout[c].type;
            o << "c" << C << suffix << " = ";
            if (t == PA_BIGINT) o << "((const i64*)a.v[" << C << "])[" << row << "]; ";
            else if (t == PA_INTEGER || t == PA_DATE) o << "(i64)((const i32*)a.v[" << C << "])[" << row << "]; ";
            else if (t == PA_DOUBLE) o << "((const double*)a.v[" << C << "])[" << row << "]; ";
            else if (t == PA_BOOLEAN) o << "((const u8*)a.v[" << C << "])[" << row << "] != 0; ";
            else throw Error(PA_ERR_NOT_SUPPORTED, "column type not supported on device");
            if (layout[c].nullable) o << "cn" << C << suffix << " = a.nl[" << C << "] != nullptr && a.nl[" << C << "][" << row << "] != 0; ";
        }
        return o.str();
    };
    // probe stage: pa_pre (filter + key) and pa_post (everything behind the probe) are separate functions, so that the vector loops
    // can probe the four rows of a quad together (pa_join_probe4); pa_row, their row-by-row composition, serves the scalar loops
    std::string lazy_params, lazy_names;
    auto lazy_declare = [&](const std::string& suffix) {
        std::string d;
        for (int c = 0; c < s.n_in && s.join; c++) {
            if (!s.lazy_channel[c]) continue;
            const std::string C = std::to_string(c);
            d += RowCodegen::ctype(layout[c].type) + " c" + C + suffix + " = 0; ";
            if (layout[c].nullable) d += "bool cn" + C + suffix + " = false; ";
        }
        return d;
    };
    if (s.join) {
        for (int c = 0; c < s.n_in; c++) {
            if (!s.lazy_channel[c]) continue;
            const std::string C = std::to_string(c), ct = RowCodegen::ctype(layout[c].type);
            lazy_params += ", const " + ct + " c" + C;
            lazy_names += ", c" + C;
            if (layout[c].nullable) {
                lazy_params += ", const bool cn" + C;
                lazy_names += ", cn" + C;
            }
        }
        src << "__device__ __forceinline__ void pa_pre(const PaFusedArgs& a, const bool live, const i32 row" << row_params(ri, layout)
            << ", bool& sel0, u64& jk)\n{\n" << pre.str() << "}\n\n";
        src << "__device__ __forceinline__ void pa_post(const PaFusedArgs& a, PaAcc& acc, const int slot, const i32 row, const i32 jb" << row_params(ri, layout)
            << lazy_params << ")\n{\n";
    }
    else {
        src << "__device__ __forceinline__ void pa_row(const PaFusedArgs& a, PaAcc& acc, const bool live, const i32 row" << row_params(ri, layout) << ")\n{\n";
    }
    src << body.str();
    // values needed after the selected-only block are declared up front
    for (int w = 0; w < k.nw; w++) {
        src << "bool u" << w << " = false; " << (words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")) << " x" << w << " = 0;\n";
    }
    if (k.w > 0) src << "u64 key[PA_KW];\n#pragma unroll\nfor (int i = 0; i < PA_KW; i++) key[i] = 0;\n";
    src << "if (sel) {\n" << build_loads.str() << inner.str();
    for (int w = 0; w < k.nw; w++) src << "u" << w << " = " << words[w].cond << "; x" << w << " = " << words[w].val << ";\n";
    if (brow) {
        src << "key[0] = (u64)(u32)jb;\n";
    }
    else {
        for (int i = 0; i < k.w; i++) {
            src << "key[" << i << "] = ";
            for (size_t t = 0; t < word_terms[i].size(); t++) src << (t ? " | " : "") << word_terms[i][t];
            src << ";\n";
        }
    }
    src << "}\n";
    if (variant == V_HASH) {
        src << "if (live) { const i32 pid = sel ? (i32)(pa_key_hash(key, PA_KW) & a.part_mask) : (i32)(a.part_mask + 1u); a.part_ids[row] = pid; "
               "__hip_atomic_fetch_add(&pa_hist[pid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }\n";
    }
    else if (variant == V_GLOBAL) {
        src << "if (sel) {\n";
        for (int w = 0; w < k.nw; w++) {
            if (words[w].kind == W_SUMF) src << "if (u" << w << ") acc.w" << w << " = acc.w" << w << " + x" << w << ";\n";
            else if (words[w].kind == W_SUMI) src << "if (u" << w << ") acc.w" << w << " = pa_add_exact(acc.w" << w << ", x" << w << ", a.err);\n";
            else if (words[w].kind == W_MAXU) src << "if (u" << w << ") acc.w" << w << " = x" << w << " > acc.w" << w << " ? x" << w << " : acc.w" << w << ";\n";
            else src << "if (u" << w << ") acc.w" << w << " += x" << w << ";\n";
        }
        src << "}\n";
    }
    else if (variant == V_LDS) {
        src << "int g = -1;\nif (sel) {\n#pragma unroll\n  for (int s = 0; s < PA_C; s++) {\n    bool eq = s < acc.tcount;\n#pragma unroll\n"
               "    for (int w = 0; w < PA_KW; w++) eq = eq && (key[w] == acc.tk[s][w]);\n    if (eq) g = s;\n  }\n}\n";
        // first occurrences: append the missing keys to the wave's table one at a time (wave-uniform loop)
        src << "u64 miss = __ballot(sel && g < 0);\nwhile (miss != 0ULL) {\n  const int src_lane = __builtin_amdgcn_readfirstlane(__ffsll((long long)miss) - 1);\n"
               "  u64 nk[PA_KW];\n#pragma unroll\n  for (int w = 0; w < PA_KW; w++) nk[w] = pa_readlane_u64(key[w], src_lane);\n"
               "  const int slot = acc.tcount;\n  if (slot < PA_C) {\n#pragma unroll\n    for (int s = 0; s < PA_C; s++) {\n      if (s == slot) {\n#pragma unroll\n"
               "        for (int w = 0; w < PA_KW; w++) acc.tk[s][w] = nk[w];\n      }\n    }\n    acc.tcount = slot + 1;\n  } else {\n    acc.tcount = PA_C + 1;\n  }\n"
               "  bool mine = sel && g == -1;\n#pragma unroll\n  for (int w = 0; w < PA_KW; w++) mine = mine && (key[w] == nk[w]);\n"
               "  if (mine) g = slot < PA_C ? slot : -2;\n  miss = __ballot(sel && g == -1);\n}\n";
        src << "if (sel) {\n  if (g >= 0) {\n";
        for (int w = 0; w < k.nw; w++) {
            std::string idx = "pa_accw[(" + std::to_string(w) + " * PA_C + g) * 64 + acc.lane]";
            if (words[w].kind == W_SUMF) {
                src << "    if (u" << w << ") __hip_atomic_fetch_add((double*)&" << idx << ", x" << w << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
            }
            else if (words[w].kind == W_SUMI) {
                src << "    if (u" << w << ") { i64* p = (i64*)&" << idx << "; *p = pa_add_exact(*p, x" << w << ", a.err); }\n";
            }
            else if (words[w].kind == W_MAXU) {
                src << "    if (u" << w << ") { u64* p = &" << idx << "; if (x" << w << " > *p) *p = x" << w << "; }\n";
            }
            else {
                src << "    if (u" << w << ") __hip_atomic_fetch_add(&" << idx << ", " << (words[w].val == "1" ? std::string("1ULL") : "(u64)x" + std::to_string(w))
                    << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
            }
        }
        // a row whose group found no slot: the wave has marked its table as overflowed (tcount = PA_C + 1, set in the loop
        // above) and reports once at the end of the kernel; the launch is discarded as a whole.  (A per-row atomic on the
        // one overflow counter would serialise the useless pass on a single address: 3 ms instead of 0.25 ms per 64 M rows.)
        src << "  }\n}\n";
    }
    else if (brow) {
        src << "pa_acc(a, acc, slot, sel, key";
        for (int w = 0; w < k.nw; w++) src << ", u" << w << ", x" << w;
        src << ");\n";
    }
    else {
        src << "pa_acc(a, acc, sel, row, key";
        for (int w = 0; w < k.nw; w++) src << ", u" << w << ", x" << w;
        src << ");\n";
    }
    src << "}\n\n";
    if (s.join) {
        src << "__device__ __forceinline__ void pa_row(const PaFusedArgs& a, PaAcc& acc, const bool live, const i32 row" << row_params(ri, layout) << ")\n{\n"
            << "bool sel0; u64 jk;\npa_pre(a, live, row" << row_param_names(ri, layout) << ", sel0, jk);\n"
            << "i32 jb = -1;\nif (sel0) jb = pa_join_probe_keyed(a, jk);\n"
            << lazy_declare("") << "\nif (jb >= 0) { " << lazy_assign("", "row") << "}\n"
            << "pa_post(a, acc, 0, row, jb" << row_param_names(ri, layout) << lazy_names << ");\n}\n\n";
    }

    // kernels.  mode 0: one kernel walks the page (GLOBAL / GT).  LDS variant: the wave's key table is wave-uniform
    // state, so every lane must take part in every pa_row call; the host splits the page and `pa_fused` (mode 1) takes the
    // leading multiple of 256 rows -- whole groups of 64 quads per wave, all lanes live, 16-byte loads -- while
    // `pa_fused_tail` (mode 2) takes the remaining < 256 rows (or everything when a buffer is unaligned) row by row with a
    // wave-uniform trip count, finished lanes riding along with live == false on a clamped row.  Two entry points keep the
    // tail's code out of the hot loop's register allocation.
    const int B = k.block;
    // the four rows 4q .. 4q + 3 of a thread of the vector loops
    auto emit_quad = [&](const std::string (&args)[4]) {
        if (!s.join) {
            for (int r = 0; r < 4; r++) src << "        pa_row(a, acc, true, (i32)(4 * q + " << r << ")" << args[r] << ");\n";
            return;
        }
        // probe stage: filter and key of the four rows, ONE staged probe for all of them, the lazy channels of the matches
        // (again four loads in flight), then the rows one by one
        src << "        bool js[4]; u64 jk[4]; i32 jb[4];\n";
        for (int r = 0; r < 4; r++) src << "        pa_pre(a, true, (i32)(4 * q + " << r << ")" << args[r] << ", js[" << r << "], jk[" << r << "]);\n";
        src << "        pa_join_probe4(a, js, jk, jb);\n";
        for (int r = 0; r < 4; r++) src << "        " << lazy_declare("_" + std::to_string(r)) << "\n";
        for (int r = 0; r < 4; r++) {
            const std::string R = std::to_string(r);
            src << "        if (jb[" << R << "] >= 0) { " << lazy_assign("_" + R, "4 * q + " + R) << "}\n";
        }
        for (int r = 0; r < 4; r++) {
            const std::string R = std::to_string(r);
            std::string names;
            for (int c = 0; c < s.n_in; c++) {
                if (!s.lazy_channel[c]) continue;
                names += ", c" + std::to_string(c) + "_" + R;
                if (layout[c].nullable) names += ", cn" + std::to_string(c) + "_" + R;
            }
            src << "        pa_post(a, acc, " << R << ", (i32)(4 * q + " << R << "), jb[" << R << "]" << args[r] << names << ");\n";
        }
    };
    auto emit_kernel = [&](const char* name, int mode) {
        src << "extern \"C\" __global__ __launch_bounds__(" << B << ") void " << name << "(PaFusedArgs a)\n{\n";
        if (variant == V_GLOBAL) {
            src << "    PaAcc acc;\n";
            for (int w = 0; w < k.nw; w++) src << "    acc.w" << w << " = 0;\n";
        }
        else if (variant == V_LDS) {
            src << "    for (int i = threadIdx.x; i < PA_NW * PA_C * 64; i += 64) pa_accw[i] = 0ULL;\n";
            src << "    __syncthreads();\n";
            src << "    PaAcc acc; acc.tcount = 0; acc.lane = threadIdx.x;\n";
            src << "#pragma unroll\n    for (int s = 0; s < PA_C; s++) {\n#pragma unroll\n        for (int w = 0; w < PA_KW; w++) acc.tk[s][w] = 0ULL;\n    }\n";
        }
        else if (variant == V_LDSP) {
            // the partition's table comes from HBM as the last launch left it (zeroes at first) ...
            src << "    const u64 sp = (u64)blockIdx.x * PA_LC;\n";
            // (accumulator words are word-major in HBM, [word][slot over all partitions] -- the layout of the HBM group table, so
            // that the partitions' tables can be emitted, or folded, as one table of gridDim.x * PA_LC slots)
            src << "    const u64 ts = (u64)gridDim.x * PA_LC;\n";
            // a.pad3: the first launch on these tables -- they are empty by definition, nothing to load (and the host cleared nothing)
            src << "    if (a.pad3) {\n";
            src << "      for (int i = threadIdx.x; i < PA_LC; i += " << B << ") pa_lt_tag[i] = 0ULL;\n";
            src << "      for (int i = threadIdx.x; i < PA_LC * PA_NW; i += " << B << ") pa_lt_acc[i] = 0ULL;\n";
            src << "      if (threadIdx.x == 0) pa_lt_count = 0;\n";
            src << "    } else {\n";
            src << "      for (int i = threadIdx.x; i < PA_LC; i += " << B << ") pa_lt_tag[i] = a.sub_tag[sp + i];\n";
            src << "      for (int i = threadIdx.x; i < PA_LC * PA_KW; i += " << B << ") pa_lt_key[i] = a.sub_keys[sp * PA_KW + i];\n";
            src << "      for (int i = threadIdx.x; i < PA_LC * PA_NW; i += " << B << ") { const int w = i / PA_LC, sl = i % PA_LC; pa_lt_acc[sl * PA_NW + w] = a.sub_words[(u64)w * ts + sp + sl]; }\n";
            src << "      if (threadIdx.x == 0) pa_lt_count = a.sub_count[blockIdx.x];\n";
            src << "    }\n    __syncthreads();\n";
            src << "    PaAcc acc; acc.tv = pa_gt_view(a, PA_KW, PA_NW); acc.gt = pa_gt_ctr_init(acc.tv.count, true, a.gt_rep_mask + 1u);\n"
                   "    acc.flush = pa_gt_ctr_init(acc.tv.count, false); acc.fell = 0;\n";
        }
        else if (variant == V_LDSH) {
            src << "    for (int i = threadIdx.x; i < PA_LC; i += " << B << ") pa_lt_tag[i] = 0ULL;\n";
            src << "    for (int i = threadIdx.x; i < PA_LC * PA_NW; i += " << B << ") pa_lt_acc[i] = 0ULL;\n";
            src << "    if (threadIdx.x == 0) pa_lt_count = 0;\n    __syncthreads();\n";
            src << "    PaAcc acc; acc.tv = pa_gt_view(a, PA_KW, PA_NW); acc.gt = pa_gt_ctr_init(acc.tv.count, true, a.gt_rep_mask + 1u);\n"
                   "    acc.flush = pa_gt_ctr_init(acc.tv.count, false); acc.fell = 0;\n";
        }
        else if (variant == V_HASH) {
            src << "    PaAcc acc; acc.unused = 0;\n";
        }
        else if (brow) {
            src << "    PaAcc acc; acc.tv = pa_gt_view(a, PA_KW, PA_NW); acc.gt = pa_gt_ctr_init(acc.tv.count, true, a.gt_rep_mask + 1u);\n"
                   "#pragma unroll\n    for (int e = 0; e < 4; e++) acc.ev[e] = false;\n"
                   "    if ((threadIdx.x & 63u) == 0u) pa_wbase[threadIdx.x >> 6] = 0u;\n"
                   "    for (u32 i = threadIdx.x & 63u; i < PA_NW * PA_WIN; i += 64u) (&pa_win[threadIdx.x >> 6][0][0])[i] = 0ULL;\n"
                   "    if ((threadIdx.x & 63u) < PA_WIN / 64u) pa_wtouch[threadIdx.x >> 6][threadIdx.x & 63u] = 0ULL;\n"
                   "    if ((threadIdx.x & 63u) == 0u) pa_sfill[threadIdx.x >> 6] = 0u;\n    PA_WAVE_SYNC();\n";
        }
        else {
            src << "    PaAcc acc; acc.tv = pa_gt_view(a, PA_KW, PA_NW); acc.gt = pa_gt_ctr_init(acc.tv.count, true, a.gt_rep_mask + 1u); acc.pn = 0;\n";
        }
        // V_GT: the pending run of every thread ends after a quad of consecutive rows / after every row of the other loops
        const std::string flush = gt_like || brow ? " pa_flush(a, acc, true);" : "";
        if (mode != 2 && mode != 3) emit_prologue(ri, layout, src);
        if (mode == 3) {
            // a table of row ranges, one workgroup per entry at a time (entries of one XCD's workgroups next to each other, as below)
            ColumnNames rn;
            rn.ranged = true;
            const int rw = range_entry_words(s, layout);
            src << "    const u32 bsw = (gridDim.x & 7u) == 0u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;\n"
                   "    for (i64 ri = bsw; ri < a.n_ranges; ri += gridDim.x) {\n"
                   "      const u64* RT = a.ranges + (u64)ri * " << rw << "ULL;\n";
            int word = 0;
            for (int c = 0; c < s.n_in; c++) {
                if (!ri.used[c]) continue;
                src << "      const void* const RV" << c << " = (const void*)RT[" << word++ << "];\n";
                if (layout[c].type == PA_VARCHAR) src << "      const i32* const RO" << c << " = (const i32*)RT[" << word++ << "];\n";
                if (layout[c].nullable) src << "      const u8* const RNL" << c << " = (const u8*)RT[" << word++ << "];\n";
            }
            src << "      const i64 RN = (i64)(RT[" << word << "] & 0xffffffffULL);\n"
                   "      const bool rvec = (RT[" << word << "] >> 32) != 0ULL;\n";
            {
                std::ostringstream pro;
                emit_prologue(ri, layout, pro, rn);
                src << pro.str();
            }
            // whole groups of 64 quads for the wave-level table of the LDS variant (every lane takes part in every row call)
            if (variant == V_LDS) src << "      const i64 nq = rvec ? (RN >> 8) << 6 : 0;\n";
            else src << "      const i64 nq = rvec ? RN >> 2 : 0;\n";
            src << "      for (i64 q = threadIdx.x; q < nq; q += " << B << ") {\n";
            std::string rargs[4];
            emit_vector_loads(ri, layout, src, rargs, rn);
            emit_quad(rargs);
            src << "      }\n";
            if (variant == V_LDS) {
                src << "      for (i64 rb = nq << 2; rb < RN; rb += 64) {\n        const bool live = rb + threadIdx.x < RN;\n"
                       "        const i64 r = live ? rb + threadIdx.x : RN - 1;\n        pa_row(a, acc, live, (i32)r" << scalar_args(ri, layout, rn) << ");\n      }\n";
            }
            else {
                src << "      for (i64 r = (nq << 2) + threadIdx.x; r < RN; r += " << B << ") {\n        pa_row(a, acc, true, (i32)r" << scalar_args(ri, layout, rn) << ");\n      }\n";
            }
            src << "    }\n";
        }
        else if (variant == V_GLOBAL) {
            // XCD-aware block -> tile mapping: consecutive workgroup ids go round-robin to the 8 XCDs, so give the
            // workgroups of one XCD consecutive tiles (each XCD's L2 / TLB then walks one contiguous eighth of every grid
            // stride).  Measured on Q6: 0.75 -> 0.79 of the HBM peak; neutral for the one-wave workgroups of the LDS variant.
            src << "    const u32 bsw = (gridDim.x & 7u) == 0u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;\n"
                   "    const i64 t = (i64)bsw * " << B << " + threadIdx.x, T = (i64)gridDim.x * " << B << ";\n";
        }
        else {
            src << "    const i64 t = (i64)blockIdx.x * " << B << " + threadIdx.x, T = (i64)gridDim.x * " << B << ";\n";
        }
        std::string args[4];
        if (mode == 3) {
            // (the loop above)
        }
        else if (mode == 1) {
            src << "    const i64 nq = a.n >> 2;  // the host passes a multiple of 256 rows\n";
            src << "    for (i64 q = t; q < nq; q += T) {\n";
            emit_vector_loads(ri, layout, src, args);
            emit_quad(args);
            src << "    }\n";
        }
        else if (mode == 2) {
            src << "    for (i64 rb = (i64)blockIdx.x * 64; rb < a.n; rb += T) {\n        const bool live = rb + threadIdx.x < a.n;\n"
                   "        const i64 r = live ? rb + threadIdx.x : a.n - 1;\n        pa_row(a, acc, live, (i32)r" << scalar_args(ri, layout) << ");\n    }\n";
        }
        else if (variant == V_HASH) {
            // tile by tile (a.sub_count: the tile x partition counts of the multisplit, tile-major)
            src << "    const i64 nq = a.vec ? (a.n >> 2) : 0;\n    const i64 tiles = (a.n + " << (kMsplitTileRows - 1) << ") / " << kMsplitTileRows << ";\n"
                   "    const i32 hp = (i32)a.part_mask + 2;\n"
                   "    for (i64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {\n"
                   "      for (i32 i = threadIdx.x; i < hp; i += " << B << ") pa_hist[i] = 0;\n      __syncthreads();\n"
                   "      const i64 r0 = tile * " << kMsplitTileRows << ", r1 = r0 + " << kMsplitTileRows << " < a.n ? r0 + " << kMsplitTileRows << " : a.n;\n"
                   "      const i64 q1 = (r1 >> 2) < nq ? (r1 >> 2) : nq;\n"
                   "      for (i64 q = (r0 >> 2) + threadIdx.x; q < q1; q += " << B << ") {\n";
            emit_vector_loads(ri, layout, src, args);
            emit_quad(args);
            src << "      }\n"
                   "      for (i64 r = ((q1 << 2) > r0 ? (q1 << 2) : r0) + threadIdx.x; r < r1; r += " << B << ") {\n        pa_row(a, acc, true, (i32)r" << scalar_args(ri, layout) << ");\n      }\n"
                   "      __syncthreads();\n"
                   "      for (i32 i = threadIdx.x; i < hp; i += " << B << ") a.sub_count[tile * hp + i] = pa_hist[i];\n      __syncthreads();\n"
                   "    }\n";
        }
        else {
            src << "    const i64 nq = a.vec ? (a.n >> 2) : 0;\n";
            if (brow) {
                // every wave walks ONE contiguous range of the page (its loads stay coalesced: 64 lanes x 16 B per instruction).
                // When the probe side is clustered by the join key, the rows a wave notes then belong to neighbouring build rows,
                // and a drained buffer reaches the table as a few dense 64-byte requests -- tags as whole lines, eight adds per atomic
                // request -- instead of one read-modify-write in HBM per group
                src << "    const i64 gw = (i64)blockIdx.x * " << (B / 64) << " + (threadIdx.x >> 6), nwv = (i64)gridDim.x * " << (B / 64) << ";\n"
                       "    const i64 per = (((nq + nwv - 1) / nwv) + 63) & ~(i64)63;\n"
                       "    const i64 q1 = (gw + 1) * per < nq ? (gw + 1) * per : nq;\n"
                       "    for (i64 q = gw * per + (threadIdx.x & 63); q < q1; q += 64) {\n";
            }
            else {
                src << "    for (i64 q = t; q < nq; q += T) {\n";
            }
            emit_vector_loads(ri, layout, src, args);
            emit_quad(args);
            src << "       " << flush << "\n    }\n";
            src << "    for (i64 r = (nq << 2) + t; r < a.n; r += T) {\n        pa_row(a, acc, true, (i32)r" << scalar_args(ri, layout) << ");" << flush << "\n    }\n";
        }
        if (variant == V_LDSP) {
            // ... the workgroup walks the rows of its partition (the columns are partition-ordered) ...
            src << "    {\n        const i64 b0 = a.part_first[blockIdx.x], b1 = a.part_first[blockIdx.x + 1];\n"
                   "        for (i64 r = b0 + threadIdx.x; r < b1; r += " << B << ") {\n            pa_row(a, acc, true, (i32)r" << scalar_args(ri, layout) << ");\n        }\n    }\n";
            // ... and the table goes back (plain coalesced stores: nobody else touches this partition)
            src << "    __syncthreads();\n";
            src << "    for (int i = threadIdx.x; i < PA_LC; i += " << B << ") a.sub_tag[sp + i] = pa_lt_tag[i];\n";
            src << "    for (int i = threadIdx.x; i < PA_LC * PA_KW; i += " << B << ") a.sub_keys[sp * PA_KW + i] = pa_lt_key[i];\n";
            src << "    for (int i = threadIdx.x; i < PA_LC * PA_NW; i += " << B << ") { const int w = i / PA_LC, sl = i % PA_LC; a.sub_words[(u64)w * ts + sp + sl] = pa_lt_acc[sl * PA_NW + w]; }\n";
            src << "    if (threadIdx.x == 0) a.sub_count[blockIdx.x] = pa_lt_count;\n";
            src << "    { const i64 f = pa_wave_sum_i64(acc.fell); if ((threadIdx.x & 63) == 0 && f != 0) atomicAdd((unsigned long long*)a.overflow_rows, (unsigned long long)f); }\n";
        }
        if (variant == V_GT || variant == V_LDSH) {
            // rows given by a list: grid-stride (spill replays), or one contiguous slice per workgroup (partition-ordered lists:
            // the workgroup's LDS table then meets the groups of a few partitions only)
            src << "    if (a.list_blocked) {\n        const i64 per = (a.n_list + gridDim.x - 1) / gridDim.x;\n"
                   "        const i64 b0 = (i64)blockIdx.x * per, b1 = b0 + per < a.n_list ? b0 + per : a.n_list;\n"
                   "        for (i64 i = b0 + threadIdx.x; i < b1; i += " << B << ") {\n            const i64 r = a.list_blocked == 2 ? i : (i64)a.row_list[i];\n            pa_row(a, acc, true, (i32)r"
                << scalar_args(ri, layout) << ");" << flush << "\n        }\n    } else {\n";
            src << "    for (i64 i = t; i < a.n_list; i += T) {\n        const i64 r = a.row_list[i];\n        pa_row(a, acc, true, (i32)r" << scalar_args(ri, layout)
                << ");" << flush << "\n    }\n    }\n";
        }
        if (variant == V_LDSH) {
            // the workgroup's table -> HBM table: one upsert and PA_NW atomics per group and workgroup
            src << "    __syncthreads();\n    const u64 cap = (u64)a.gt_mask + 1ULL;\n";
            src << "    for (int sl = threadIdx.x; sl < PA_LC; sl += " << B << ") {\n        if (pa_lt_tag[sl] == 0ULL) continue;\n"
                   "        u64 fk[PA_KW];\n#pragma unroll\n        for (int w = 0; w < PA_KW; w++) fk[w] = pa_lt_key[sl * PA_KW + w];\n"
                   "        const int g = pa_gt_upsert<PA_KW>(acc.tv.tag, acc.tv.keys, a.gt_mask, pa_key_hash(fk, PA_KW), fk, acc.flush, 0x7fffffff, a.err);\n"
                   "        if (g < 0) { pa_raise(a.err, PA_DEV_ERR_RESOURCES); continue; }\n";
            for (int w = 0; w < k.nw; w++) {
                std::string idx = std::to_string(w) + "ULL * cap + (u64)g";
                std::string v = "pa_lt_acc[sl * PA_NW + " + std::to_string(w) + "]";
                if (words[w].kind == W_SUMF) src << "        pa_gt_add_f64(acc.tv.words, " << idx << ", __longlong_as_double((i64)" << v << "));\n";
                else if (words[w].kind == W_SUMI) src << "        pa_gt_add_i64_exact(acc.tv.words, " << idx << ", (i64)" << v << ", a.err);\n";
                else if (words[w].kind == W_MAXU) src << "        pa_gt_max_u64(acc.tv.words, " << idx << ", " << v << ");\n";
                else src << "        pa_gt_add_u64(acc.tv.words, " << idx << ", " << v << ");\n";
            }
            src << "    }\n    pa_gt_ctr_flush(acc.flush, acc.tv.count);\n";
            src << "    { const i64 f = pa_wave_sum_i64(acc.fell); if ((threadIdx.x & 63) == 0 && f != 0) atomicAdd((unsigned long long*)a.overflow_rows, (unsigned long long)f); }\n";
        }
        if (brow) src << "    pa_drain(a, acc, pa_sfill[threadIdx.x >> 6]);\n";  // (all lanes are back together behind the row loops)
        if (gt_like || brow || lds_table) src << "    pa_gt_ctr_flush(acc.gt, acc.tv.count);\n";
        if (variant == V_GLOBAL) {
            src << "    __shared__ u64 red[" << (B / 64) << " * PA_NW];\n    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;\n";
            for (int w = 0; w < k.nw; w++) {
                if (words[w].kind == W_SUMF) src << "    { double v = pa_wave_sum_f64(acc.w" << w << "); if (lane == 0) red[wave * PA_NW + " << w << "] = (u64)__double_as_longlong(v); }\n";
                else if (words[w].kind == W_SUMI) src << "    { i64 v = pa_wave_sum_i64_exact(acc.w" << w << ", a.err); if (lane == 0) red[wave * PA_NW + " << w << "] = (u64)v; }\n";
                else if (words[w].kind == W_MAXU) src << "    { u64 v = pa_wave_max_u64(acc.w" << w << "); if (lane == 0) red[wave * PA_NW + " << w << "] = v; }\n";
                else src << "    { i64 v = pa_wave_sum_i64(acc.w" << w << "); if (lane == 0) red[wave * PA_NW + " << w << "] = (u64)v; }\n";
            }
            src << "    __syncthreads();\n    if (threadIdx.x < PA_NW) {\n        const int w = threadIdx.x;\n        u64 r = red[w];\n";
            src << "        for (int i = 1; i < " << (B / 64) << "; i++) {\n            u64 o = red[i * PA_NW + w];\n";
            src << "            switch (w) {\n";
            for (int w = 0; w < k.nw; w++) {
                src << "                case " << w << ": ";
                if (words[w].kind == W_SUMF) src << "r = (u64)__double_as_longlong(__longlong_as_double((i64)r) + __longlong_as_double((i64)o)); break;\n";
                else if (words[w].kind == W_SUMI) src << "r = (u64)pa_add_exact((i64)r, (i64)o, a.err); break;\n";
                else if (words[w].kind == W_MAXU) src << "r = o > r ? o : r; break;\n";
                else src << "r = r + o; break;\n";
            }
            src << "            }\n        }\n        a.slab[(u64)blockIdx.x * PA_NW + w] = r;\n    }\n";
        }
        else if (variant == V_LDS) {
            // per-wave partial table -> slab, field-major: field f of entry e = blockIdx.x * C + i at slab[f * E + e]
            src << "    __syncthreads();\n";
            src << "    if (acc.tcount > PA_C && threadIdx.x == 0) atomicAdd((unsigned long long*)a.overflow_rows, 1ULL);\n";
            src << "    const u64 E = (u64)gridDim.x * PA_C;\n";
            src << "#pragma unroll\n    for (int i = 0; i < PA_C; i++) {\n        u64* e = a.slab + (u64)blockIdx.x * PA_C + i;\n";
            src << "        const bool occ = i < acc.tcount;\n        if (threadIdx.x == 0) e[0] = occ ? 1ULL : 0ULL;\n        if (occ) {\n";
            src << "#pragma unroll\n            for (int w = 0; w < PA_KW; w++) { if (threadIdx.x == 0) e[(u64)(1 + w) * E] = acc.tk[i][w]; }\n";
            for (int w = 0; w < k.nw; w++) {
                std::string idx = "pa_accw[(" + std::to_string(w) + " * PA_C + i) * 64 + threadIdx.x]";
                std::string dst = "e[(u64)(1 + PA_KW + " + std::to_string(w) + ") * E]";
                if (words[w].kind == W_SUMF) src << "            { double v = pa_wave_sum_f64(__longlong_as_double((i64)" << idx << ")); if (threadIdx.x == 0) " << dst << " = (u64)__double_as_longlong(v); }\n";
                else if (words[w].kind == W_SUMI) src << "            { i64 v = pa_wave_sum_i64_exact((i64)" << idx << ", a.err); if (threadIdx.x == 0) " << dst << " = (u64)v; }\n";
                else if (words[w].kind == W_MAXU) src << "            { u64 v = pa_wave_max_u64(" << idx << "); if (threadIdx.x == 0) " << dst << " = v; }\n";
                else src << "            { i64 v = pa_wave_sum_i64((i64)" << idx << "); if (threadIdx.x == 0) " << dst << " = (u64)v; }\n";
            }
            src << "        }\n    }\n";
        }
        src << "}\n\n";
    };
    if (ranged) {
        emit_kernel("pa_fused_ranges", 3);
    }
    else if (variant == V_LDS) {
        emit_kernel("pa_fused", 1);
        emit_kernel("pa_fused_tail", 2);
    }
    else {
        emit_kernel("pa_fused", 0);
    }
    if (brow) {
        // key words of the groups, once per group: build row b has a group when its tag is set; its key is a function of the
        // build columns (the probe join key equals the build key column on every match)
        src << "extern \"C\" __global__ __launch_bounds__(256) void pa_brow_keys(PaFusedArgs a)\n{\n"
               "    const i64 cap = (i64)a.gt_mask + 1;\n"
               "    i64 found = 0;\n"
               "    for (i64 b = (i64)blockIdx.x * 256 + threadIdx.x; b < cap; b += (i64)gridDim.x * 256) {\n"
            << (k.occ_word < 0 ? std::string("        if (a.gt_tag[b] == 0ULL) continue;\n")
                               : "        if (a.gt_words[" + std::to_string(k.occ_word) + "ULL * (u64)cap + (u64)b] == " + std::to_string(k.occ_empty) + "ULL) continue;\n")
            << "        found++;\n        const i32 jb = (i32)b;\n";
        src << build_loads.str() << key_os.str();
        for (int i = 0; i < k.w; i++) {
            src << "        a.gt_keys[(u64)b * PA_TW + " << i << "] = ";
            for (size_t t = 0; t < word_terms[i].size(); t++) src << (t ? " | " : "") << word_terms[i][t];
            src << ";\n";
        }
        // (the groups are counted on the way: one atomic per wave on the table's group counter)
        src << "    }\n    found = pa_wave_sum_i64(found);\n    if ((threadIdx.x & 63) == 0 && found != 0) atomicAdd(a.gt_count, (i32)found);\n}\n\n";
    }
    k.source = src.str();
    return k;
}

uint32_t next_pow2(uint64_t v)
{
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return (uint32_t)p;
}

// ---- the operator ------------------------------------------------------------------------------

class FusedAggregationOperator : public pa_operator {
public:
    explicit FusedAggregationOperator(const pa_fused_aggregation_desc* d)
        : FusedAggregationOperator(make_spec(d), d->aggregation.stream ? d->aggregation.stream : d->filter_project.stream)
    {
    }
    FusedAggregationOperator(Spec spec, void* stream) : spec_(std::move(spec)), stream_(stream)
    {
        require_device();
        nullable_seen_.assign(spec_.n_in, false);
        out_partial_ = spec_.step == PA_STEP_PARTIAL;
        {
            std::ostringstream f;
            for (int c = 0; c < spec_.n_in; c++) f << spec_.in_types[c] << ',' << spec_.in_params[c] << ';';
            f << '|' << (spec_.has_filter ? spec_.filter.fingerprint() : std::string("-")) << '|';
            for (const auto& p : spec_.proj) f << p.fingerprint() << '#';
            f << '|';
            for (int g : spec_.group_proj) f << g << ',';
            f << '|' << spec_.hash_channel << '|' << spec_.step << '|';
            for (const auto& a : spec_.aggs) f << a.fn << ',' << a.input_channel << ',' << a.mask_channel << ',' << a.input_type << ';';
            if (spec_.join) {
                f << "|join:" << spec_.join->key_proj << ':';
                for (int32_t t : spec_.join->build_types) f << t << ',';
                f << ':';
                for (int j : spec_.join->brow_group_proj) f << j << ',';
            }
            plan_fingerprint_ = f.str();
        }
        grouped_ = !spec_.group_proj.empty();
        // the planner's estimate decides where a grouped aggregation starts: the few-groups register/LDS variant (a page
        // that overflows it is redone on the HBM table), or directly the HBM table when many groups are expected
        mode_ = grouped_ ? (spec_.expected_groups > (1 << 20) ? V_GT : V_LDS) : V_GLOBAL;
        cus_ = device_cu_count();
        ctl_ = static_cast<int32_t*>(ctl_buf_.ensure(64));  // [0] err  [1] gt_count  [2..3] overflow rows
        PA_HIP(hipMemsetAsync(ctl_, 0, 64, stream_.get()));
        h_ctl_ = static_cast<int32_t*>(h_ctl_buf_.ensure(256));  // [0..15] control block, [16..31] one copy per slab of the LDS variant
        memset(h_ctl_, 0, 256);
    }
    ~FusedAggregationOperator() override
    {
        // pooled buffers go back to the caches in the member destructors: all device work must be done first
        (void)hipStreamSynchronize(stream_.get());
        if (merge_stream_) pool_stream_release(merge_stream_);  // synchronises it
        for (int b = 0; b < 2; b++) {
            if (arena_[b].table_event) (void)hipEventDestroy(arena_[b].table_event);
            if (arena_[b].vtable_event) (void)hipEventDestroy(arena_[b].vtable_event);
            if (ev_main_[b]) (void)hipEventDestroy(ev_main_[b]);
            if (ev_merge_[b]) (void)hipEventDestroy(ev_merge_[b]);
        }
        for (RangeTable& t : range_table_) {
            if (t.event) (void)hipEventDestroy(t.event);
        }
    }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    // ---- Operator protocol with device work in flight -------------------------------------------------------------------
    // add_input only enqueues.  Small pages are gathered first (see "small pages" below); the fused launches of the few-groups
    // variant are confirmed one launch late (its overflow word decides whether a launch must be redone on the next tier), so
    // the host is one launch ahead of the device and never waits inside add_input for a stable page.  needs_input() turns
    // false -- and is_blocked() true -- while two launches are unconfirmed (Operator.isBlocked, Operator.java:69-80): the
    // Driver polls, as it does for a future, instead of parking a thread in the native call.
    // probe stage: no page is taken before the build side has published its lookup source (LookupJoinOperator.needsInput /
    // isBlocked on the lookup source future, LookupJoinOperator.java:63, 100)
    bool lookup_source_ready() const { return !spec_.join || spec_.join->ls->built.load(); }

    bool needs_input() override
    {
        if (finishing_) return false;
        if (!lookup_source_ready()) return false;
        if (!retry_parked()) return false;
        if (next_) return next_->needs_input();
        poll_inflight();
        return inflight_.size() < kMaxInflight;
    }
    bool is_blocked() override
    {
        if (!lookup_source_ready()) return !finishing_;
        if (!retry_parked()) return true;  // waiting for HBM (Operator.isBlocked on a memory future, Operator.java:69-80)
        if (next_) return next_->is_blocked();
        poll_inflight();
        return inflight_.size() >= kMaxInflight;
    }

    // A stable page the HBM budget had no room for is put aside whole (HashAggregationOperator's unfinishedWork,
    // HashAggregationOperator.java:435-438, 476-484) and taken up again once the pool can grant the request that failed.
    // true: nothing is parked (any more)
    bool retry_parked()
    {
        if (!parked_) return true;
        if (!pool_has_room(parked_need_)) return false;
        parked_ = false;
        pa_page page = parked_page_;
        page.columns = parked_cols_.data();
        take_page(&page);
        return !parked_;
    }

    void take_page(const pa_page* page)
    {
        const uint64_t launched = timer.begun();
        try {
            if (next_) {
                next_->add_input(page);
                return;
            }
            if (gather_small_page(page)) return;
            flush_pending();
            if (next_) {  // the flush met a layout change and started the next generation: the page belongs there
                next_->add_input(page);
                return;
            }
            process_page(page, (page->flags & PA_PAGE_STABLE) != 0 && page->mem == PA_MEM_DEVICE);
        }
        catch (const PoolExhausted& e) {
            // only a page that can be read again later, and of which nothing has been launched, can wait
            if ((page->flags & PA_PAGE_STABLE) == 0 || timer.begun() != launched || is_combiner_) throw;
            parked_page_ = *page;
            parked_cols_.assign(page->columns, page->columns + page->channel_count);
            parked_need_ = e.bytes;
            parked_ = true;
        }
    }

    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(page != nullptr, PA_ERR_INVALID_ARGUMENT, "page is null");
        PA_REQUIRE(page->channel_count == spec_.n_in, PA_ERR_INVALID_ARGUMENT, "page channel count does not match the operator's input types");
        if (page->position_count == 0) return;
        PA_REQUIRE(!parked_, PA_ERR_ILLEGAL_STATE, "Operator has unfinished work");  // HashAggregationOperator.java:384
        take_page(page);
    }

    // One page (or gathered range of pages) through the kernels; `retained`: its buffers stay valid until the operator has
    // confirmed the launches, so they may be confirmed late.
    void process_page(const pa_page* page, bool retained)
    {
        if (next_) {  // a later generation takes every page (its layout is the more general one)
            next_->process_page(page, retained);
            return;
        }
        retained_ = retained;
        try {
            add_page(page);
        }
        catch (const LayoutChange&) {
            start_next_generation();
            next_->process_page(page, retained);
        }
        retained_ = false;
    }

    void start_next_generation()
    {
        // nullability only grows, so a state sees at most one change per channel; the combiner none at all
        PA_REQUIRE(!is_combiner_ && generation_ <= spec_.n_in, PA_ERR_DEVICE, "internal: state layout changed more often than channels exist");
        // a channel turned nullable in a way that needs more count words / NULL flags: this state stays as it is, the
        // page and everything after it go to a new generation; get_output combines the generations' states
        confirm_all();
        next_ = std::make_unique<FusedAggregationOperator>(spec_, stream_.get());
        next_->nullable_seen_ = nullable_seen_;
        next_->generation_ = generation_ + 1;
    }

    // ---- small pages ----------------------------------------------------------------------------------------------------
    // An unmodified Driver hands over pages of <= 1 MB / 8192 rows (PageProcessor.java:56-58); one launch per such page
    // would leave the device idle between launches.  Two ways out, both keeping add_input a plain enqueue:
    //  * consecutive STABLE device pages that continue each other in memory (row ranges of resident columns: Page.getRegion
    //    views, pages over one pinned / HBM staging area) are merged into one range -- no copy, only pointer compares -- and
    //    launched once the range holds kGatherRows rows (or at finish);
    //  * other small pages with fixed-width used channels are copied behind each other into one of two arenas (one H2D copy
    //    per column for host pages; one segment-copy launch per page for device pages, whose buffers may be recycled by their
    //    producer after add_input returns) and the arena is launched when it is full.
    static constexpr int64_t kSmallPageRows = (int64_t)1 << 21;
    static int64_t gather_rows()
    {
        const char* e = getenv("PRESTO_AMD_GATHER_ROWS");  // tests and sweeps move the launch threshold
        return e ? std::max<int64_t>(strtoll(e, nullptr, 10), 1) : (int64_t)1 << 26;
    }
    static constexpr int64_t kArenaRows = (int64_t)1 << 22;

    bool channel_plain(const pa_column& col, int c) const
    {
        if (col.encoding == PA_FLAT) return col.type == spec_.in_types[c] || spec_.interned[c];
        return col.encoding == PA_VARWIDTH;
    }

    // true: the page was taken (merged into the pending range / copied into the arena)
    bool gather_small_page(const pa_page* page)
    {
        const int64_t n = page->position_count;
        const bool stable_dev = (page->flags & PA_PAGE_STABLE) != 0 && page->mem == PA_MEM_DEVICE;
        if (run_.rows > 0) {
            // does the page continue the pending range?
            bool cont = stable_dev && run_.rows + n <= ((int64_t)1 << 30);
            for (int c = 0; c < spec_.n_in && cont; c++) {
                if (!spec_.used_channel[c]) continue;
                const pa_column& a = run_.cols[c];
                const pa_column& b = page->columns[c];
                cont = a.encoding == b.encoding && a.type == b.type && (a.nulls == nullptr) == (b.nulls == nullptr);
                if (!cont) break;
                if (a.nulls) cont = b.nulls == a.nulls + run_.rows;
                if (a.encoding == PA_FLAT) {
                    cont = cont && b.values == static_cast<const char*>(a.values) + run_.rows * type_width(a.type);
                }
                else {
                    cont = cont && b.values == a.values && b.offsets == a.offsets + run_.rows;
                }
            }
            if (cont) {
                run_.rows += n;
                if (run_.rows >= gather_rows()) flush_pending();
                return true;
            }
            retire_run();
            if (next_) return false;  // (add_input hands the page to the generation a flush started)
        }
        bool plain = true, flat = true;
        for (int c = 0; c < spec_.n_in && plain; c++) {
            if (!spec_.used_channel[c]) continue;
            plain = channel_plain(page->columns[c], c) && !spec_.interned[c];
            flat = flat && page->columns[c].encoding == PA_FLAT;
        }
        if (!plain) return false;
        if (stable_dev && n < gather_rows()) {
            // a range starts here: whatever its size, the next page may continue it
            run_.rows = n;
            run_.flat = flat;
            run_.cols.assign(page->columns, page->columns + page->channel_count);
            return true;
        }
        if (n >= kSmallPageRows) return false;
        // VariableWidthBlocks join the arena too: the bytes are appended and the offsets rebased on the way -- by the host's
        // arithmetic when the offsets can be read here (host pages), by a byte cursor in HBM for device pages, whose first
        // offset and byte count only the device knows (launch_var_append)
        if (!flat && page->mem != PA_MEM_HOST && !device_var_gatherable()) return false;
        append_to_arena(page);
        return true;
    }

    // A device page's VariableWidthBlocks are appended without the host knowing how many bytes they hold: the arena's byte
    // buffers are sized for the most the declared types allow -- VARCHAR(n), n code points of at most 4 bytes -- which is kept
    // to 64 bytes per row (n <= 16); unbounded or longer channels get a launch per page as before.
    static constexpr int64_t kDeviceVarMaxLength = 16;
    bool device_var_gatherable() const
    {
        int slots = 0;
        for (int c = 0; c < spec_.n_in; c++) {
            if (!spec_.used_channel[c] || spec_.in_types[c] != PA_VARCHAR) continue;
            if (spec_.in_params[c] < 1 || spec_.in_params[c] > kDeviceVarMaxLength) return false;
            slots++;
        }
        return slots <= kInlineVarSegs;
    }

    // The pending range ends (the next page does not continue it): a small one joins the arena -- one segment-copy launch --
    // instead of getting a fused launch and its merges of its own; a large one is launched as it is.
    void retire_run()
    {
        if (run_.rows == 0) return;
        if (run_.rows < kSmallPageRows && ranges_possible()) {
            // the ungrouped / few-groups kernels take such ranges in place, as a table: no copy at all.  (The first launch of
            // the few-groups tier decides whether it is the right one: nothing is collected before it is confirmed)
            if (mode_ == V_LDS && !lds_probed_) {
                flush_run();
                return;
            }
            DevPage r;
            r.n = (int32_t)run_.rows;
            r.cols.resize((size_t)spec_.n_in);
            for (int c = 0; c < spec_.n_in; c++) {
                if (!spec_.used_channel[c]) continue;
                const pa_column& col = run_.cols[c];
                r.cols[c].type = col.type;
                r.cols[c].varwidth = col.encoding == PA_VARWIDTH;
                r.cols[c].values = col.values;
                r.cols[c].offsets = col.offsets;
                r.cols[c].nulls = col.nulls;
            }
            if (!ranges_) ranges_ = std::make_shared<std::vector<DevPage>>();
            ranges_->push_back(std::move(r));
            range_rows_ += run_.rows;
            run_.rows = 0;
            if (range_rows_ >= std::min<int64_t>(gather_rows(), (int64_t)1 << 30) || ranges_->size() >= kMaxRanges) flush_ranges();
            return;
        }
        if (run_.rows >= kSmallPageRows || (!run_.flat && !device_var_gatherable())) {
            flush_run();
            return;
        }
        pa_page sp{};
        sp.position_count = (int32_t)run_.rows;
        sp.channel_count = spec_.n_in;
        sp.columns = run_.cols.data();
        sp.mem = PA_MEM_DEVICE;
        sp.flags = PA_PAGE_STABLE;  // its copy can wait for the arena's launch
        run_.rows = 0;
        append_to_arena(&sp);
    }

    // stable device ranges can be handed over as a table when the tier in charge has a kernel for it
    bool ranges_possible() const
    {
        if (spec_.join || getenv("PRESTO_AMD_NO_RANGES")) return false;
        if (mode_ != V_GLOBAL && mode_ != V_LDS) return false;
        for (int c = 0; c < spec_.n_in; c++) {
            if (spec_.used_channel[c] && spec_.interned[c]) return false;
        }
        return true;
    }

    void flush_ranges()
    {
        if (!ranges_ || ranges_->empty()) return;
        std::shared_ptr<const std::vector<DevPage>> set = std::move(ranges_);
        ranges_.reset();
        const int64_t rows = range_rows_;
        range_rows_ = 0;
        const uint64_t launched = timer.begun();
        try {
            process_ranges(set, rows);
        }
        catch (const PoolExhausted&) {
            // nothing of the table was launched: it waits with the page that is being parked (take_page)
            if (timer.begun() == launched && !next_) {
                ranges_ = std::make_shared<std::vector<DevPage>>(*set);
                range_rows_ = rows;
            }
            throw;
        }
    }

    void process_ranges(const std::shared_ptr<const std::vector<DevPage>>& set, int64_t rows)
    {
        if (next_) {
            next_->process_ranges(set, rows);
            return;
        }
        retained_ = true;
        try {
            DevPage dp;
            dp.n = (int32_t)rows;
            dp.cols = set->front().cols;
            dp.ranges = set;
            std::vector<ChannelLayout> layout(spec_.n_in);
            std::string sig;
            for (int c = 0; c < spec_.n_in; c++) {
                layout[c].type = spec_.in_types[c];
                if (spec_.used_channel[c]) {
                    for (const DevPage& r : *set) {
                        PA_REQUIRE(r.cols[c].type == spec_.in_types[c], PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared input type");
                        if (r.cols[c].nulls != nullptr) nullable_seen_[c] = true;
                    }
                }
                layout[c].nullable = nullable_seen_[c];
                sig += layout[c].nullable ? 'n' : '-';
            }
            run_tiers(sig, layout, dp, true, 0);
        }
        catch (const LayoutChange&) {
            start_next_generation();
            next_->process_ranges(set, rows);
        }
        retained_ = false;
    }

    static bool range_aligned(const DevPage& r, const std::vector<bool>& used)
    {
        bool vec = true;
        for (size_t c = 0; c < r.cols.size(); c++) {
            if (!used[c]) continue;
            vec = vec && ((uintptr_t)r.cols[c].values % 16 == 0) && ((uintptr_t)r.cols[c].offsets % 16 == 0) && ((uintptr_t)r.cols[c].nulls % 4 == 0);
        }
        return vec;
    }

    void flush_run()
    {
        if (run_.rows == 0) return;
        pa_page sp{};
        sp.position_count = (int32_t)run_.rows;
        sp.channel_count = spec_.n_in;
        sp.columns = run_.cols.data();
        sp.mem = PA_MEM_DEVICE;
        sp.flags = PA_PAGE_STABLE;
        run_.rows = 0;
        std::vector<pa_column> cols;
        cols.swap(run_.cols);  // (process_page may come back here through a generation change)
        sp.columns = cols.data();
        process_page(&sp, true);
    }

    void append_to_arena(const pa_page* page)
    {
        hipStream_t s = stream_.get();
        const int64_t n = page->position_count;
        Arena& a = arena_[arena_cur_];
        // the nullability of the arena's channels is fixed by its first page: a page that differs starts the next arena
        bool fits = a.rows + n <= kArenaRows && a.segs.size() + 3 * (size_t)spec_.n_in <= kArenaMaxSegs && a.vsegs.size() + (size_t)spec_.n_in <= kArenaMaxSegs;
        // the byte cursor of a VARCHAR channel is either the host's (a.bytes) or the device's: pages of the other kind start the next arena
        const bool dev_var = page->mem != PA_MEM_HOST;
        for (int c = 0; c < spec_.n_in && fits && a.rows > 0; c++) {
            if (!spec_.used_channel[c]) continue;
            const pa_column& col = page->columns[c];
            fits = a.nullable[c] == (col.nulls != nullptr);
            if (fits && col.encoding == PA_VARWIDTH) fits = a.dev_var == dev_var;
            // a VARCHAR channel's byte buffer never moves while copies into it are pending
            if (fits && col.encoding == PA_VARWIDTH && col.offsets != nullptr && !dev_var) {
                fits = a.bytes[c] + ((int64_t)col.offsets[n] - col.offsets[0]) <= (int64_t)a.values[c].capacity();
            }
        }
        if (!fits) {
            flush_pending();
            if (next_) return next_->add_input(page);  // the flush started the next generation
            return append_to_arena(page);
        }
        const bool host = page->mem == PA_MEM_HOST;
        const bool readable = !host || (page->flags & PA_PAGE_PINNED) != 0;  // the device can read the page's buffers itself
        const bool defer = readable && (page->flags & PA_PAGE_STABLE) != 0;   // ... and they stay: copy at the arena's launch
        if (a.rows == 0) {
            a.nullable.assign(spec_.n_in, false);
            a.values.resize(spec_.n_in);
            a.nulls.resize(spec_.n_in);
            a.offsets.resize(spec_.n_in);
            a.bytes.assign(spec_.n_in, 0);
            a.dev_var = dev_var;
            a.var_fresh = true;
            for (int c = 0; c < spec_.n_in; c++) {
                if (!spec_.used_channel[c]) continue;
                a.nullable[c] = page->columns[c].nulls != nullptr;
                if (spec_.in_types[c] != PA_VARCHAR) a.values[c].ensure((size_t)kArenaRows * type_width(spec_.in_types[c]));
                if (a.nullable[c]) a.nulls[c].ensure((size_t)kArenaRows);
            }
        }
        CopySeg now[3 * kMaxChannels];
        VarSeg vnow[kInlineVarSegs];
        int m = 0, vm = 0, slot = 0;
        // the device cursors of this append: read from one half of a.cursors, left in the other (deferred appends of one
        // arena launch are planned together: launch_var_append takes the first one's input and the last one's output)
        int64_t* cur_in = nullptr;
        int64_t* cur_out = nullptr;
        if (dev_var) {
            int64_t* cursors = static_cast<int64_t*>(a.cursors.ensure(2 * kVarSlots * sizeof(int64_t)));
            const bool pending = defer && !a.vsegs.empty();  // a deferred append continues the pending plan: same halves
            if (!pending) a.cursor_half ^= 1;
            cur_in = cursors + (a.cursor_half ^ 1) * kVarSlots;
            cur_out = cursors + a.cursor_half * kVarSlots;
        }
        auto seg = [&](const void* src, void* dst, int64_t bytes, int32_t add = 0) {
            CopySeg sg{src, dst, bytes, 0};
            sg.add_i32 = add;
            if (defer) a.segs.push_back(sg);
            else now[m++] = sg;
        };
        for (int c = 0; c < spec_.n_in; c++) {
            if (!spec_.used_channel[c]) continue;
            const pa_column& col = page->columns[c];
            PA_REQUIRE(col.type == spec_.in_types[c], PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared input type");
            PA_REQUIRE(col.values != nullptr, PA_ERR_INVALID_ARGUMENT, "block values is null");
            if (col.nulls) {
                char* dn = a.nulls[c].as<char>() + a.rows;
                if (readable) seg(col.nulls, dn, n);
                else PA_HIP(hipMemcpyAsync(dn, col.nulls, (size_t)n, hipMemcpyHostToDevice, s));
            }
            if (col.encoding == PA_VARWIDTH && dev_var) {
                // device page: where the block's bytes start and how many there are is only known over there
                PA_REQUIRE(col.offsets != nullptr, PA_ERR_INVALID_ARGUMENT, "VARWIDTH block without offsets");
                const int64_t row_bytes = 4 * (int64_t)spec_.in_params[c];
                if (a.rows == 0) a.values[c].ensure((size_t)(row_bytes * kArenaRows));
                VarSeg vs{};
                vs.values = static_cast<const char*>(col.values);
                vs.offsets = col.offsets;
                vs.dst_bytes = a.values[c].as<char>();
                vs.capacity = std::min<int64_t>((int64_t)a.values[c].capacity(), ((int64_t)1 << 31) - 1);
                vs.dst_offsets = static_cast<int32_t*>(a.offsets[c].ensure((size_t)(kArenaRows + 1) * 4)) + a.rows;
                vs.cursor_in = cur_in + slot;
                vs.cursor_out = cur_out + slot;
                vs.rows = (int32_t)n;
                vs.byte_wgs = (int32_t)std::min<int64_t>(std::max<int64_t>(n * std::min<int64_t>(row_bytes, 16) >> 16, 1), 64);
                vs.slot = slot++;
                vs.fresh = a.var_fresh ? 1 : 0;
                if (defer) a.vsegs.push_back(vs);
                else vnow[vm++] = vs;
                continue;
            }
            if (col.encoding == PA_VARWIDTH) {
                // host page: the offsets are readable here.  bytes behind the arena's bytes, offsets rebased by (cursor - first)
                PA_REQUIRE(col.offsets != nullptr, PA_ERR_INVALID_ARGUMENT, "VARWIDTH block without offsets");
                const int64_t first = col.offsets[0], len = (int64_t)col.offsets[n] - first;
                PA_REQUIRE(len >= 0 && a.bytes[c] + len < ((int64_t)1 << 31), PA_ERR_INVALID_ARGUMENT, "bad VARWIDTH offsets");
                int32_t* doff = static_cast<int32_t*>(a.offsets[c].ensure((size_t)(kArenaRows + 1) * 4)) + a.rows;
                if (a.rows == 0) {
                    // sized by the channel's declared bound (VARCHAR(n)), or for this page with room to spare; a later page
                    // that does not fit starts the next arena (see `fits`)
                    const int64_t bound = spec_.in_params[c] > 0 ? std::min<int64_t>(spec_.in_params[c], 64) : 0;
                    a.values[c].ensure((size_t)std::max<int64_t>({bound * kArenaRows, 4 * len, (int64_t)1 << 20}));
                }
                char* dv = a.values[c].as<char>() + a.bytes[c];
                const int32_t delta = (int32_t)(a.bytes[c] - first);
                if (readable) {
                    seg(static_cast<const char*>(col.values) + first, dv, len);
                    // (n + 1 entries: the first one rewrites the previous page's end with the same value)
                    if (delta != 0) seg(col.offsets, doff, (n + 1) * 4, delta);
                    else seg(col.offsets, doff, (n + 1) * 4);
                }
                else {
                    if (len) PA_HIP(hipMemcpyAsync(dv, static_cast<const char*>(col.values) + first, (size_t)len, hipMemcpyHostToDevice, s));
                    PA_HIP(hipMemcpyAsync(doff, col.offsets, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, s));
                    if (delta != 0) {
                        CopySeg sg{doff, doff, (n + 1) * 4, 0};
                        sg.add_i32 = delta;
                        now[m++] = sg;  // in place, behind the copy in stream order
                    }
                }
                a.bytes[c] += len;
                continue;
            }
            const int w = type_width(col.type);
            char* dv = a.values[c].as<char>() + a.rows * w;
            if (readable) seg(col.values, dv, n * w);
            else PA_HIP(hipMemcpyAsync(dv, col.values, (size_t)n * w, hipMemcpyHostToDevice, s));
        }
        if (m > 0) launch_copy_segments_inline(now, m, s);
        if (vm > 0) {
            // (deferred appends recorded before this page come first: the cursor passes through them)
            flush_var_segments(a);
            launch_var_append_inline(vnow, vm, ctl_, s);
        }
        if (slot > 0) a.var_fresh = false;
        a.rows += n;
        if (a.rows >= kArenaRows) flush_pending();
    }

    // launches whatever is pending: the merged range of stable pages and the current arena
    void flush_pending()
    {
        if (next_) next_->flush_pending();
        // a small pending range joins the table of the others; alone, it is launched in place as it is
        if (ranges_ && !ranges_->empty() && run_.rows > 0 && run_.rows < kSmallPageRows && ranges_possible()) retire_run();
        flush_ranges();
        flush_arena();
        flush_run();
    }

    // the deferred VariableWidthBlock appends of an arena: one planning launch and one copy launch for all of them
    template <typename ArenaT> void flush_var_segments(ArenaT& a)
    {
        if (a.vsegs.empty()) return;
        hipStream_t s = stream_.get();
        if (a.vtable_used) PA_HIP(hipEventSynchronize(a.vtable_event));
        else PA_HIP(hipEventCreateWithFlags(&a.vtable_event, hipEventDisableTiming));
        a.vtable_used = true;
        launch_var_append(a.vsegs.data(), a.vsegs.size(), a.h_vtable.ensure(copy_var_table_bytes(a.vsegs.size())),
                          a.d_vtable.ensure(copy_var_table_bytes(a.vsegs.size())), ctl_, s);
        PA_HIP(hipEventRecord(a.vtable_event, s));
        a.vsegs.clear();
    }

    void flush_arena()
    {
        Arena& a = arena_[arena_cur_];
        if (a.rows == 0) return;
        hipStream_t s = stream_.get();
        if (!a.segs.empty()) {
            // the copies of the stable pages gathered in this arena, in one launch.  The staging table is written by the host:
            // the copy of its previous use must have left it
            if (a.table_used) PA_HIP(hipEventSynchronize(a.table_event));
            else PA_HIP(hipEventCreateWithFlags(&a.table_event, hipEventDisableTiming));
            a.table_used = true;
            launch_copy_segments(a.segs.data(), a.segs.size(), a.h_table.ensure(copy_segments_table_bytes(a.segs.size())),
                                 a.d_table.ensure(copy_segments_table_bytes(a.segs.size())), s);
            PA_HIP(hipEventRecord(a.table_event, s));
            a.segs.clear();
        }
        flush_var_segments(a);
        std::vector<pa_column> cols((size_t)spec_.n_in);
        for (int c = 0; c < spec_.n_in; c++) {
            cols[c].type = spec_.in_types[c];
            cols[c].encoding = spec_.in_types[c] == PA_VARCHAR ? PA_VARWIDTH : PA_FLAT;
            if (!spec_.used_channel[c]) continue;
            cols[c].values = a.values[c].ptr();
            cols[c].offsets = spec_.in_types[c] == PA_VARCHAR ? a.offsets[c].as<int32_t>() : nullptr;
            cols[c].nulls = a.nullable[c] ? a.nulls[c].as<uint8_t>() : nullptr;
        }
        pa_page sp{};
        sp.position_count = (int32_t)a.rows;
        sp.channel_count = spec_.n_in;
        sp.columns = cols.data();
        sp.mem = PA_MEM_DEVICE;
        a.rows = 0;
        arena_cur_ ^= 1;
        // the arena is this operator's own: its rows stay put until the launches on it are confirmed -- the other arena
        // takes the next pages, and is only written again after this one's launches were confirmed (kMaxInflight = 2)
        process_page(&sp, true);
    }

    void add_page(const pa_page* page)
    {
        hipStream_t s = stream_.get();
        // interned key channels that arrive as a DictionaryBlock / RLE over strings take the dictionary route: not decoded
        std::vector<int> dict_keys;
        std::vector<bool> needed = spec_.used_channel;
        for (int c = 0; c < spec_.n_in; c++) {
            if (!spec_.interned[c]) continue;
            const pa_column& col = page->columns[c];
            const bool encoded = (col.encoding == PA_DICTIONARY && col.ids != nullptr) || col.encoding == PA_RLE;
            if (!encoded || col.dictionary == nullptr || col.dictionary->encoding != PA_VARWIDTH) continue;
            const int64_t dn = col.encoding == PA_RLE ? 1 : col.dictionary_size;
            if (dn <= 0 || dn > page->position_count) continue;
            needed[c] = false;
            dict_keys.push_back(c);
        }
        if (spec_.join && !join_checked_) {
            const LookupSourceImpl& ls = *spec_.join->ls;
            PA_REQUIRE(ls.built.load(), PA_ERR_ILLEGAL_STATE, "probe page before the lookup source was built");
            if (int32_t e = ls.error.load()) throw Error(e, "hash build failed on device");
            PA_REQUIRE(ls.keyed && !ls.has_duplicates, PA_ERR_ILLEGAL_STATE, "internal: fused probe over a lookup source with duplicate keys");
            join_checked_ = true;
            // the group is the build row whenever the plan allows it: no hashing, no key compares, no spills
            if (grouped_ && !spec_.join->brow_group_proj.empty() && !getenv("PRESTO_AMD_NO_BROW")) mode_ = V_BROW;
        }
        DevPage dp = stager_.stage(page, &needed, s);
        for (int c : dict_keys) intern_dictionary_key(page, c, dp, s);
        intern_keys(dp, s);
        // layout signature of this page
        std::vector<ChannelLayout> layout(spec_.n_in);
        std::string sig;
        bool vec = true;
        for (int c = 0; c < spec_.n_in; c++) {
            layout[c].type = spec_.used_channel[c] ? dp.cols[c].type : spec_.in_types[c];
            // nullability only ever grows: a page without NULLs on a channel that had some runs the nullable kernels with a
            // null valueIsNull pointer, so the state layout changes at most once per channel
            if (spec_.used_channel[c] && dp.cols[c].nulls != nullptr) nullable_seen_[c] = true;
            layout[c].nullable = nullable_seen_[c];
            if (spec_.used_channel[c]) {
                PA_REQUIRE(dp.cols[c].type == spec_.in_types[c], PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared input type");
                vec = vec && ((uintptr_t)dp.cols[c].values % 16 == 0) && ((uintptr_t)dp.cols[c].offsets % 16 == 0) &&
                      ((uintptr_t)dp.cols[c].nulls % 4 == 0);
            }
            sig += layout[c].nullable ? 'n' : '-';
        }
        if (spec_.join) {  // the build columns as channels n_in + v: their nullability is the lookup source's, fixed since the build
            for (size_t v = 0; v < spec_.join->build_cols.size(); v++) {
                ChannelLayout cl;
                cl.type = spec_.join->build_types[v];
                cl.nullable = spec_.join->ls->cols[spec_.join->build_cols[v]].has_nulls;
                layout.push_back(cl);
                sig += cl.nullable ? 'N' : '_';
            }
        }
        run_tiers(sig, layout, dp, vec, 0);
    }

    // rows [start_row, dp.n) of a staged page through the tier mode_ names, moving on to the next tier when it gives up
    void run_tiers(const std::string& sig, const std::vector<ChannelLayout>& layout, const DevPage& dp, bool vec, int64_t start_row)
    {
        for (;;) {
            // launches of the few-groups variant still unconfirmed while another tier takes over: settle them first (their
            // merges write the table the other tiers resize and replicate)
            if (mode_ != V_LDS && !inflight_.empty()) confirm_all();
            if (dp.ranges && mode_ != V_GLOBAL && mode_ != V_LDS) {
                // a table of ranges and a tier without a kernel for tables (the few-groups tier gave up): range by range
                for (const DevPage& r : *dp.ranges) run_tiers(sig, layout, r, range_aligned(r, spec_.used_channel), 0);
                break;
            }
            int partitions = 0;
            if (mode_ == V_GT && partitioned_wanted(sig, layout, &partitions)) {
                run_page_partitioned(sig, layout, dp, vec, partitions, start_row);
                break;
            }
            const Compiled* compiled = nullptr;
            try {
                compiled = &kernel_for(sig, layout, dp.ranges ? (mode_ == V_GLOBAL ? V_GLOBAL_R : V_LDS_R) : mode_);
            }
            catch (const Error& e) {
                // the group state may be too wide for the wave's / the workgroup's LDS budget: move on to the next tier
                // (anything else that is not supported fails again there and surfaces)
                if (e.code != PA_ERR_NOT_SUPPORTED || (mode_ != V_LDS && mode_ != V_LDSH)) throw;
                mode_ = mode_ == V_LDS ? V_LDSH : V_GT;
                continue;
            }
            const Compiled& ck = *compiled;
            resume_from_ = -1;
            cur_sig_ = &sig;
            cur_layout_ = &layout;
            if (run_page(ck, dp, vec, nullptr, start_row)) break;
            if (resume_from_ >= 0) {
                // the rows before resume_from_ are done (or launched and waiting for their confirmation); the rest of the page
                // goes to the tier mode_ now names
                start_row = resume_from_;
                if (start_row >= dp.n) break;
                continue;
            }
            // the page held more groups than the wave's register table: redo it (and every later page) with the
            // workgroup-level LDS table, which itself hands rows it has no room for to the HBM table
            mode_ = V_LDSH;
        }
    }

    void finish() override
    {
        if (finishing_) return;
        PA_REQUIRE(retry_parked(), PA_ERR_INSUFFICIENT_RESOURCES, "finish while a page is still waiting for HBM: the pool's budget does not cover the aggregation");
        flush_pending();
        finishing_ = true;
    }
    bool is_finished() override { return finishing_ && output_done_; }

    bool get_output(pa_page* out) override
    {
        if (!finishing_ || output_done_) return false;
        output_done_ = true;
        confirm_all();
        if (next_) return combine_generations(out);
        build_output();
        if (grouped_ && out_rows_ > 0) decode_interned_keys();
        if (!grouped_ || out_rows_ > 0) {
            publish_output(out_cols_, out_rows_, spec_.output_mem, stream_.get(), out, out_storage_);
            return true;
        }
        return false;  // HashAggregationOperator emits nothing for an empty input (SINGLE step with keys)
    }

    // The accumulator states of this generation as a PARTIAL-format page in HBM (false: no group).
    bool emit_states(pa_page* out)
    {
        flush_pending();
        confirm_all();
        out_partial_ = true;
        spec_.output_mem = PA_MEM_DEVICE;  // host-assembled blocks are uploaded
        build_output();
        if (grouped_ && out_rows_ > 0) decode_interned_keys();
        if (grouped_ && out_rows_ == 0) return false;
        publish_output(out_cols_, out_rows_, PA_MEM_DEVICE, stream_.get(), out, out_storage_);
        return true;
    }

    // Generations exist because a channel's nullability changed the state layout mid-stream.  Their states are combined the
    // way the reference combines partial aggregations (InMemoryHashAggregationBuilder with Step.FINAL / INTERMEDIATE input):
    // every generation emits its states, a FINAL-input operator over [keys, ($hashvalue), states] adds them up and emits
    // what this operator was asked for (final values, or states again for Step.PARTIAL).
    bool combine_generations(pa_page* out)
    {
        Spec cs = combiner_spec();
        combiner_ = std::make_unique<FusedAggregationOperator>(std::move(cs), stream_.get());
        combiner_->out_partial_ = spec_.step == PA_STEP_PARTIAL;
        // the generations' state pages differ in nullability by construction: the combiner starts from the most general
        // layout (every channel nullable), so it never splits into generations itself
        combiner_->nullable_seen_.assign(combiner_->spec_.n_in, true);
        combiner_->is_combiner_ = true;
        for (FusedAggregationOperator* g = this; g != nullptr; g = g->next_.get()) {
            pa_page states{};
            if (g->emit_states(&states)) combiner_->add_input(&states);
        }
        combiner_->finish();
        return combiner_->get_output(out);
    }

    Spec combiner_spec() const
    {
        Spec c;
        c.step = PA_STEP_FINAL;
        c.output_mem = spec_.output_mem;
        c.expected_groups = spec_.expected_groups;
        auto add_channel = [&](int32_t type, int32_t param) {
            c.in_types.push_back(type);
            c.in_params.push_back(param);
            OwnedExpr e;
            pa_expr_node node{};
            node.kind = PA_EXPR_INPUT_REF;
            node.type = type;
            node.channel = c.n_in;
            e.nodes.push_back(node);
            e.strings.emplace_back();
            e.root = 0;
            c.proj.push_back(std::move(e));
            return c.n_in++;
        };
        for (size_t gi = 0; gi < spec_.group_proj.size(); gi++) {
            const OwnedExpr& pe = spec_.proj[spec_.group_proj[gi]];
            int ch = pe.is_input_ref() ? pe.node(pe.root).channel : -1;
            if (ch >= spec_.n_in) ch = -1;  // a build column of the probe stage: not a channel of the page
            const bool interned = ch >= 0 && spec_.interned[ch];
            c.group_proj.push_back(add_channel(interned ? (int32_t)PA_VARCHAR : pe.root_type(), ch >= 0 ? spec_.in_params[ch] : 0));
        }
        c.hash_channel = spec_.hash_channel >= 0 && !spec_.group_proj.empty() ? add_channel(PA_BIGINT, 0) : -1;
        for (const pa_aggregate& ag : spec_.aggs) {
            pa_aggregate f = ag;
            f.mask_channel = -1;
            f.input_channel = add_channel(PA_BIGINT, 0);  // count state
            if (ag.fn != PA_AGG_COUNT && ag.fn != PA_AGG_COUNT_STAR) {
                const int value_proj = spec_.step == PA_STEP_FINAL ? ag.input_channel + 1 : ag.input_channel;
                int32_t t = spec_.proj[value_proj].root_type();
                if (ag.fn == PA_AGG_AVG || (ag.fn == PA_AGG_SUM && t == PA_DOUBLE)) t = PA_DOUBLE;  // sum state: DOUBLE, or BIGINT for integer sums
                else if (ag.fn == PA_AGG_SUM) t = PA_BIGINT;
                add_channel(t, t == PA_VARCHAR ? 7 : 0);  // (min / max over VARCHAR only exist for strings of <= 7 bytes)
            }
            c.aggs.push_back(f);
        }
        finalize_spec(c);
        return c;
    }

    int64_t memory_bytes() override
    {
        return (int64_t)(stager_.bytes() + slab_.capacity() + gt_tag_.capacity() + gt_keys_.capacity() + gt_words_.capacity() + state_.capacity());
    }

    // What isFull() compares with maxPartialMemory: the groups known so far x the bytes of a group's key and state words (and
    // its table slot), over all generations.  Launches still unconfirmed are not counted yet.
    int64_t group_bytes()
    {
        // rows gathered but not launched, or launched but not confirmed, count as one group each (an upper bound: an early
        // flush of a partial aggregation only costs repeated keys)
        int64_t pending = run_.rows + arena_[arena_cur_].rows;
        for (const Inflight& f : inflight_) pending += f.dp.n - f.offset;
        const int64_t slot = 8 * (1 + std::max(w_, 1) + std::max(nw_, 1));
        int64_t b = ((int64_t)std::max(groups_upper_, groups_sum_) + pending) * slot;
        if (next_) b += next_->group_bytes();
        return b;
    }
    const Spec& spec() const { return spec_; }
    void* stream_handle() { return stream_.get(); }

private:
    // One code object + word-kind table per (plan fingerprint, column-layout signature, variant, device), shared by every
    // operator instance of the process: an operator lives for one query (OperatorFactory.createOperator), the generated
    // code for its plan node does not change -- re-generating ~30 KB of source and hashing it per instance cost ~0.1 ms.
    struct Compiled {
        KernelInfo info;
        JitKernel kernel, tail_kernel;
        DevBuf kinds;
    };
    static std::shared_ptr<const Compiled> shared_lookup(const std::string& key)
    {
        std::lock_guard<std::mutex> lock(shared_mutex());
        auto it = shared_cache().find(key);
        return it == shared_cache().end() ? nullptr : it->second;
    }
    static std::mutex& shared_mutex()
    {
        static std::mutex* m = new std::mutex();
        return *m;
    }
    static std::map<std::string, std::shared_ptr<const Compiled>>& shared_cache()
    {
        static auto* c = new std::map<std::string, std::shared_ptr<const Compiled>>();  // leaked: HIP may be gone at exit
        return *c;
    }

    const Compiled& kernel_for(const std::string& sig, const std::vector<ChannelLayout>& layout, int variant)
    {
        std::string key = sig + "|" + std::to_string(variant);
        auto it = compiled_.find(key);
        if (it != compiled_.end()) return *it->second;
        int dev = 0;
        PA_HIP(hipGetDevice(&dev));
        const std::string shared_key = std::to_string(dev) + "|" + key + "|" + plan_fingerprint_;
        if (auto hit = shared_lookup(shared_key)) {
            adopt_layout(*hit);
            compiled_[key] = hit;
            return *hit;
        }
        auto c = std::make_shared<Compiled>();
        c->info = generate(spec_, layout, variant);
        c->kernel = jit_get(c->info.source, c->info.entry);
        if (variant == V_LDS) c->tail_kernel = jit_get(c->info.source, "pa_fused_tail");
        if (variant == V_BROW) c->tail_kernel = jit_get(c->info.source, "pa_brow_keys");
        c->kinds.ensure(sizeof(int32_t) * c->info.word_kind.size());
        PA_HIP(hipMemcpyAsync(c->kinds.ptr(), c->info.word_kind.data(), sizeof(int32_t) * c->info.word_kind.size(), hipMemcpyHostToDevice, stream_.get()));
        PA_HIP(hipStreamSynchronize(stream_.get()));
        adopt_layout(*c);
        {
            std::lock_guard<std::mutex> lock(shared_mutex());
            shared_cache()[shared_key] = c;
        }
        const Compiled& ref = *c;
        compiled_[key] = std::move(c);
        return ref;
    }

    void adopt_layout(const Compiled& c)
    {
        if (!kinds_dev_) kinds_dev_ = c.kinds.as<int32_t>();
        if (!layout_fixed_) {
            nw_ = c.info.nw;
            w_ = c.info.w;
            layout_id_ = c.info.layout_id;
            layout_fixed_ = true;
        }
        // every signature of one state must yield the same layout: a channel that turns nullable adds count words / NULL
        // flags, and states of different layouts cannot be merged word by word -- the page starts the next generation
        if (c.info.layout_id != layout_id_) throw LayoutChange{};
    }

    // replicas wanted for a table of g groups: enough distinct accumulator addresses (>= ~2^17) for the atomics of a
    // launch not to queue on a few of them; none needed once the groups themselves are that many
    static uint32_t desired_replicas(uint64_t g)
    {
        uint64_t r = (1ULL << 17) / std::max<uint64_t>(g, 1);
        uint32_t p = 1;
        while (p * 2 <= r && p < 128) p <<= 1;
        return p;
    }

    // makes room for at least min_groups groups per replica at a load factor of one half; reps = 0 keeps the replica count
    void ensure_table(uint64_t min_groups, uint32_t reps = 0)
    {
        if (reps == 0) reps = gt_rep_;
        if (reps < gt_rep_) min_groups = std::max(min_groups, groups_sum_);  // replicas fold into fewer tables
        uint64_t want = std::max<uint64_t>(1024, 2 * min_groups);
        PA_REQUIRE(want <= (1ULL << 30), PA_ERR_INSUFFICIENT_RESOURCES, "Size of hash table cannot exceed 1 billion entries");
        uint32_t cap = std::max(next_pow2(want), gt_cap_);
        const size_t slot_bytes = 8 * (size_t)(1 + std::max(w_, 1) + nw_);
        while (reps > 1 && (size_t)reps * cap * slot_bytes > (8ULL << 30)) reps >>= 1;
        if (cap == gt_cap_ && reps == gt_rep_) return;
        hipStream_t s = stream_.get();
        drain_merges();  // in-flight merges still write the old table
        DevBuf tag, keys, words, rc;
        const size_t slots = (size_t)reps * cap;
        tag.ensure(slots * 8);
        keys.ensure(slots * 8 * std::max(w_, 1));
        words.ensure(slots * 8 * nw_);
        rc.ensure(128 * 4);
        PA_HIP(hipMemsetAsync(tag.ptr(), 0, slots * 8, s));
        // (PA_GT_KEY_CLEAR: no slot's key words may look like a key before the slot is claimed -- see pa_gt_upsert_n's fast path)
        PA_HIP(hipMemsetAsync(keys.ptr(), 0xA5, slots * 8 * std::max(w_, 1), s));
        PA_HIP(hipMemsetAsync(words.ptr(), 0, slots * 8 * nw_, s));
        PA_HIP(hipMemsetAsync(rc.ptr(), 0, 128 * 4, s));
        if (gt_cap_ > 0) {
            PA_REQUIRE(kinds_dev_ != nullptr, PA_ERR_ILLEGAL_STATE, "group table without a compiled kernel");
            PA_HIP(hipMemsetAsync(ctl_ + 1, 0, 4, s));
            launch_gt_fold(gt_tag_.as<uint64_t>(), gt_keys_.as<uint64_t>(), gt_words_.as<uint64_t>(), gt_cap_, gt_rep_, std::max(w_, 1), nw_,
                           kinds_dev_, tag.as<uint64_t>(), keys.as<uint64_t>(), words.as<uint64_t>(), cap - 1, reps, ctl_ + 1,
                           rc.as<int32_t>(), ctl_, s);
            PA_HIP(hipStreamSynchronize(s));  // the old arrays return to the pool below
        }
        gt_tag_ = std::move(tag);
        gt_keys_ = std::move(keys);
        gt_words_ = std::move(words);
        rep_count_ = std::move(rc);
        gt_cap_ = cap;
        gt_rep_ = reps;
    }

    // group counts of all replicas after a launch: groups_upper_ = the fullest replica (what every replica must have
    // room for), groups_sum_ = upper bound of the distinct groups
    void read_group_counts(hipStream_t s)
    {
        int32_t* h = static_cast<int32_t*>(h_rep_.ensure(128 * 4));
        h[0] = 0;
        if (gt_rep_ > 1) PA_HIP(hipMemcpyAsync(h, rep_count_.ptr(), (size_t)gt_rep_ * 4, hipMemcpyDeviceToHost, s));
        PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 32, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        uint64_t mx = (uint64_t)h_ctl_[1], sum = (uint64_t)h_ctl_[1];
        for (uint32_t r = 1; r < gt_rep_; r++) {
            mx = std::max<uint64_t>(mx, (uint64_t)h[r]);
            sum += (uint64_t)h[r];
        }
        groups_upper_ = mx;
        groups_sum_ = sum;
    }

    void drain_merges()
    {
        if (merge_stream_) PA_HIP(hipStreamSynchronize(merge_stream_));
        merge_pending_[0] = merge_pending_[1] = false;
    }

    // rows of one chunk of a page in a given order (the hash-partitioned path)
    struct RowList {
        const int32_t* rows;   // positions relative to the chunk's first row
        int64_t count;
        int64_t first_row;     // of the chunk in the page
        int64_t chunk_rows;
    };

    // Medium cardinality on the HBM-table tier (G groups, lc / 2 < G <= 8 K): partition the rows by hash(key) mod P so that
    // a partition holds ~lc / 8 groups, then run the LDS-table kernel over the rows in partition order, one contiguous slice
    // per workgroup -- the atomics per row move from HBM (~20 G/s for the whole chip) into LDS.
    bool partitioned_wanted(const std::string& sig, const std::vector<ChannelLayout>& layout, int* partitions)
    {
        if (getenv("PRESTO_AMD_NO_PARTITIONED")) return false;
        if (spec_.join) return false;  // a row's partition would need its probe: the row function runs once per row behind a probe stage
        if (sub_parts_ > 0) {  // partition-owned tables exist: every later page is cut the same way
            *partitions = sub_parts_;
            return true;
        }
        static const uint64_t ldsp_from = [] {
            const char* e = getenv("PRESTO_AMD_LDSP_FROM");
            return (uint64_t)(e ? atoll(e) : 200000);
        }();
        const uint64_t expected = (uint64_t)std::max(spec_.expected_groups, 0);
        // nothing measured yet: only the planner's estimate can name the tier -- when it says "many groups", start with the
        // partition-owned tables at once (a probe launch on the HBM table would leave its groups there, to be folded later)
        if (!gt_probed_ && (expected < ldsp_from || is_combiner_)) return false;
        // measured (64 M rows, 16 B/row, uniform keys; steady state per page): 1 K groups 9 -> 26 G rows/s, 8 K 6 -> 18 G,
        // 100 K 8 -> 11.6 G; beyond ~400 K groups a workgroup's slice holds more groups than its table takes
        const uint64_t g = groups_upper_;
        const uint64_t g_est = std::max(g, expected);
        if (g_est < 256) return false;
        const Compiled* ldsh = nullptr;
        try {
            ldsh = &kernel_for(sig, layout, V_LDSH);
        }
        catch (const Error& e) {
            if (e.code != PA_ERR_NOT_SUPPORTED) throw;
            return false;
        }
        // Many groups: partition-owned tables (V_LDSP).  With the workgroup's table flushed into the HBM table after every
        // launch, a launch costs one HBM upsert per (group, launch) -- at 3 M groups and 2^24-row pages as many atomics as
        // rows / 5, and the tier stays bound by them (9 G rows/s).  A table that belongs to ONE partition for good is loaded
        // from and stored to HBM with plain coalesced accesses instead.  Partitions sized for ~0.4 of a table (they may fill to
        // 3/4 before rows fall through to the HBM table), from what the probe saw or the planner expects, whichever is more.
        if (g_est >= ldsp_from && g_est <= 2048ULL * (uint64_t)ldsh->info.lc * 3 / 4) {
            uint64_t p = next_pow2((uint64_t)((double)g_est / (0.4 * ldsh->info.lc)) + 1);
            *partitions = (int)std::min<uint64_t>(std::max<uint64_t>(p, 64), 2048);
            want_ldsp_ = true;
            return true;
        }
        if (!gt_probed_ || g < 256) return false;
        const uint64_t per = std::max(ldsh->info.lc / 8, 8);
        // at most 512 partitions (+ 1 for filtered rows), each within a quarter of the workgroup's table
        // up to half a table per partition (with one workgroup per partition, see list_grid_hint_): 700 K groups 17 vs 10 G rows/s
        // on the HBM table, 1 M groups even
        // (beyond that the HBM table takes the rows as they come: running ITS kernel over partition-ordered rows, for the
        // locality of the table slice, was measured slower -- 3 M groups 7.2 vs 9.5 G rows/s, 10 M 6.1 vs 7.8: the atomics are
        // bound in the L2 atomic units, not by where the table lines live)
        // (The multisplit takes up to 4096 partitions in one pass, but more than 512 here was measured slower: with the table
        // flushed into HBM after every launch, 2048 partitions of a 2^24-row page are 8 K-row slices whose table set-up and
        // flush outweigh the rows -- 700 K groups 17.6 -> 11.8 G rows/s; the partition-owned tables above take over instead.)
        static const uint64_t max_parts = [] {
            const char* e = getenv("PRESTO_AMD_MAX_PARTITIONS");
            return (uint64_t)(e ? std::max(atoi(e), 2) : 512);
        }();
        if (g > max_parts * (uint64_t)(ldsh->info.lc / 2)) return false;
        uint64_t p = next_pow2((g + per - 1) / per);
        *partitions = (int)std::min<uint64_t>(std::max<uint64_t>(p, 2), max_parts);
        return true;
    }

    void run_page_partitioned(const std::string& sig, const std::vector<ChannelLayout>& layout, const DevPage& dp, bool vec, int partitions,
                              int64_t start_row)
    {
        hipStream_t s = stream_.get();
        // (A variant that also wrote every row's packed key / input words, put them in partition order and let the kernel read
        // them contiguously was measured slower at every cardinality -- 8 K groups 14.8 vs 17.9 G rows/s, 100 K 10.5 vs 11.6 --
        // than letting the LDS-table kernel gather the page rows of its slice, and was removed.)
        const Compiled& hk = kernel_for(sig, layout, V_HASH);
        bool ldsp = want_ldsp_ || sub_parts_ > 0;
        // (needs the reordered columns: fixed-width inputs, few enough for one multisplit)
        int moved = 0;
        for (int c = 0; c < spec_.n_in && ldsp; c++) {
            if (!spec_.used_channel[c]) continue;
            ldsp = !dp.cols[c].varwidth;
            moved += 1 + (dp.cols[c].nulls ? 1 : 0);
        }
        ldsp = ldsp && moved <= kMsplitMaxCols && !getenv("PRESTO_AMD_NO_MSPLIT");
        if (!ldsp && sub_parts_ == 0) want_ldsp_ = false;
        if (!ldsp) partitions = std::min(partitions, 2048);
        const Compiled& lk = kernel_for(sig, layout, ldsp ? V_LDSP : V_LDSH);
        cur_sig_ = &sig;
        cur_layout_ = &layout;
        if (ldsp && sub_parts_ == 0) {
            // the partitions' tables, all empty
            sub_parts_ = partitions;
            const size_t slots = (size_t)partitions * lk.info.lc;
            // (not cleared: the first launch starts every partition's table from zeroes in LDS and stores all of them)
            sub_tag_.ensure(slots * 8);
            sub_keys_.ensure(slots * 8 * std::max(lk.info.w, 1));
            sub_words_.ensure(slots * 8 * lk.info.nw);
            sub_count_.ensure((size_t)partitions * 4);
            sub_lc_ = lk.info.lc;
            sub_fresh_ = true;
        }
        if (ldsp) partitions = sub_parts_;
        const int64_t chunk = (int64_t)1 << 26;
        for (int64_t offset = start_row; offset < dp.n; offset += chunk) {
            const int64_t n = std::min(chunk, dp.n - offset);
            FusedArgs a;
            memset(&a, 0, sizeof a);
            for (int c = 0; c < spec_.n_in; c++) {
                if (!spec_.used_channel[c]) continue;
                const DevColumn& col = dp.cols[c];
                a.v[c] = col.varwidth ? col.values : static_cast<const char*>(col.values) + offset * type_width(col.type);
                a.o[c] = col.offsets ? col.offsets + offset : nullptr;
                a.nl[c] = col.nulls ? col.nulls + offset : nullptr;
            }
            a.n = n;
            a.vec = (vec && offset % 4 == 0) ? 1 : 0;
            a.err = ctl_;
            a.part_ids = static_cast<int32_t*>(part_ids_.ensure((size_t)n * 4));
            a.part_mask = (uint32_t)partitions - 1;
            // the partition pass leaves the multisplit's tile x partition counts behind (tile-major, in the multisplit's scratch)
            void* ms_temp = part_temp_.ensure(std::max(msplit_temp_bytes(n, partitions + 1), partition_temp_bytes(n, partitions + 1)));
            a.sub_count = msplit_counts(ms_temp);
            void* params[] = {&a};
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(msplit_tiles(n), (int64_t)cus_ * 8));
            timer.begin(s);
            PA_HIP(hipModuleLaunchKernel(hk.kernel.fn, grid, 1, 1, hk.info.block, 1, 1, 0, s, params, nullptr));
            int64_t* counts = static_cast<int64_t*>(part_counts_.ensure((size_t)(partitions + 1) * 8));
            // (pays only when a slice of two partitions would overfill the table -- 500 K groups: 15 -> 20 G rows/s; below that
            // the second round of workgroups costs more than the sparser tables save -- 300 K: 24.7 -> 22.9)
            list_grid_hint_ = (2 * groups_upper_ / (uint64_t)partitions > (uint64_t)lk.info.lc * 3 / 8) ? partitions : 0;
            // fixed-width inputs: the used columns themselves are regrouped by partition (LDS-staged multisplit, coalesced both
            // ways) and the LDS-table kernel reads its slice contiguously; with a position list it pays a cache line per row
            // and column.  VARCHAR inputs keep the position list.
            bool reorder = !getenv("PRESTO_AMD_NO_MSPLIT");
            for (int c = 0; c < spec_.n_in && reorder; c++) reorder = !spec_.used_channel[c] || !dp.cols[c].varwidth;
            if (reorder) {
                std::vector<MsplitCol> mc;
                DevPage rp;
                rp.cols.resize(spec_.n_in);
                if (reorder_bufs_.empty()) reorder_bufs_.resize((size_t)spec_.n_in * 2);
                for (int c = 0; c < spec_.n_in; c++) {
                    if (!spec_.used_channel[c]) continue;
                    const DevColumn& col = dp.cols[c];
                    const int w = type_width(col.type);
                    DevColumn& out = rp.cols[c];
                    out.type = col.type;
                    out.values = reorder_bufs_[(size_t)c * 2].ensure((size_t)n * w);
                    mc.push_back(MsplitCol{static_cast<const char*>(col.values) + offset * w, const_cast<void*>(out.values), w, 0});
                    if (col.nulls) {
                        out.nulls = static_cast<const uint8_t*>(reorder_bufs_[(size_t)c * 2 + 1].ensure((size_t)n));
                        mc.push_back(MsplitCol{col.nulls + offset, const_cast<uint8_t*>(out.nulls), 1, 0});
                    }
                }
                reorder = mc.size() <= (size_t)kMsplitMaxCols;
                if (reorder) {
                    launch_msplit(a.part_ids, n, partitions + 1, mc.data(), (int32_t)mc.size(), counts, ms_temp, s, false, true);
                    if (ldsp) {
                        // the kernel finds its rows through the partition boundaries on the device: the host does not need them
                        launch_exclusive_prefix_i64(counts, partitions + 1, static_cast<int64_t*>(part_first_.ensure((size_t)(partitions + 2) * 8)), s);
                        timer.end(s, false);
                        rp.n = (int32_t)n;
                        RowList list{nullptr, n, 0, n};
                        run_page(lk, rp, false, &list);
                        continue;
                    }
                    timer.end(s, false);
                    int64_t dropped = 0;
                    PA_HIP(hipMemcpyAsync(&dropped, counts + partitions, 8, hipMemcpyDeviceToHost, s));
                    PA_HIP(hipStreamSynchronize(s));
                    rp.n = (int32_t)(n - dropped);  // the filtered rows are the last partition
                    RowList list{nullptr, n - dropped, 0, n - dropped};
                    if (list.count > 0) run_page(lk, rp, false, &list);
                    continue;
                }
            }
            int32_t* positions = static_cast<int32_t*>(part_pos_.ensure((size_t)n * 4));
            launch_partition_positions(a.part_ids, n, partitions + 1, positions, counts, ms_temp, s);
            timer.end(s, false);
            int64_t dropped = 0;
            PA_HIP(hipMemcpyAsync(&dropped, counts + partitions, 8, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            RowList list{positions, n - dropped, offset, n};
            if (list.count > 0) run_page(lk, dp, vec, &list);
        }
    }

    // where the group table keeps its tags / accumulator words (static_kernels.hpp, GtStrides)
    const uint64_t* table_tags() const
    {
        return build_rows_table_ && brow_occ_word_ >= 0 ? gt_words_.as<uint64_t>() + (size_t)brow_occ_word_ * gt_cap_ : gt_tag_.as<uint64_t>();
    }
    const uint64_t* table_words() const { return gt_words_.as<uint64_t>(); }
    GtStrides table_strides() const { return GtStrides{1u, gt_cap_, 1u, 0u, build_rows_table_ && brow_occ_word_ >= 0 ? brow_occ_empty_ : 0ULL}; }

    void fill_join_args(FusedArgs& a) const
    {
        const JoinStage& js = *spec_.join;
        const LookupSourceImpl& ls = *js.ls;
        a.jslots = ls.key_slots.ptr();
        a.jmask = ls.probe_mask;
        a.jwrap = ls.probe_wrap;
        a.jbits = ls.bitmap.bits;
        a.jmin = ls.bitmap.min_key;
        a.jrange = ls.bitmap.range;
        a.jrank = ls.rank.words;
        a.jrank_rows = ls.rank.rows;
        a.jrows = ls.n;
        for (size_t v = 0; v < js.build_cols.size(); v++) {
            const BuildColumn& bc = ls.cols[js.build_cols[v]];
            a.bv[v] = bc.values.ptr();
            a.bn[v] = bc.has_nulls ? bc.nulls.as<uint8_t>() : nullptr;
        }
    }

    // BROW: accumulators indexed by build position.  The table has one slot per build row -- it never fills, nothing spills,
    // nothing needs confirming: launches are enqueued and forgotten (the error word is read at finish).
    bool run_page_build_rows(const Compiled& ck, const DevPage& dp, FusedArgs a, int64_t start_row)
    {
        hipStream_t s = stream_.get();
        const KernelInfo& ki = ck.info;
        const uint32_t slots = (uint32_t)std::max(spec_.join->ls->n, 1);
        if (gt_cap_ == 0) {
            brow_occ_word_ = ki.occ_word;
            brow_occ_empty_ = ki.occ_empty;
            gt_keys_.ensure((size_t)slots * 8 * std::max(w_, 1));
            gt_words_.ensure((size_t)slots * 8 * nw_);
            rep_count_.ensure(128 * 4);
            PA_HIP(hipMemsetAsync(gt_words_.ptr(), 0, (size_t)slots * 8 * nw_, s));
            if (brow_occ_word_ < 0) {
                gt_tag_.ensure((size_t)slots * 8);
                PA_HIP(hipMemsetAsync(gt_tag_.ptr(), 0, (size_t)slots * 8, s));
            }
            else if (brow_occ_empty_ != 0) {
                launch_fill_u64(gt_words_.as<uint64_t>() + (size_t)brow_occ_word_ * slots, brow_occ_empty_, (int64_t)slots, s);
            }
            PA_HIP(hipMemsetAsync(rep_count_.ptr(), 0, 128 * 4, s));
            gt_cap_ = slots;
            gt_rep_ = 1;
            build_rows_table_ = true;
        }
        PA_REQUIRE(build_rows_table_ && gt_cap_ == slots && brow_occ_word_ == ki.occ_word, PA_ERR_DEVICE, "internal: build-row table mixed with another table");
        a.gt_tag = gt_tag_.as<uint64_t>();
        a.gt_keys = gt_keys_.as<uint64_t>();
        a.gt_words = gt_words_.as<uint64_t>();
        a.gt_mask = gt_cap_ - 1;  // capacity - 1 (no mask: the slot is the build position)
        a.gt_max_fill = INT32_MAX;
        a.gt_rep_mask = 0;
        a.gt_rep_count = rep_count_.as<int32_t>();
        const int64_t n = dp.n - start_row;
        if (n <= 0) return true;
        if (start_row > 0) {
            for (int c = 0; c < spec_.n_in; c++) {
                if (!spec_.used_channel[c]) continue;
                const DevColumn& col = dp.cols[c];
                if (col.varwidth) a.o[c] = col.offsets + start_row;
                else a.v[c] = static_cast<const char*>(col.values) + start_row * type_width(col.type);
                if (col.nulls) a.nl[c] = col.nulls + start_row;
            }
        }
        a.n = n;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(((n + 3) / 4 + 255) / 256, (int64_t)cus_ * 8));
        void* params[] = {&a};
        timer.begin(s);
        PA_HIP(hipModuleLaunchKernel(ck.kernel.fn, grid, 1, 1, ki.block, 1, 1, 0, s, params, nullptr));
        timer.end(s);
        brow_keys_ = &ck;
        return true;
    }

    // returns false when the LDS variant overflowed and the page must be redone with the HBM table
    // The table of a ranged launch (entry layout: range_entry_words): every range cut into entries of at most kRangeRows rows,
    // each with its own buffer addresses.  Returns the number of entries.
    int64_t fill_range_table(const KernelInfo& ki, const DevPage& dp, FusedArgs& a, hipStream_t s)
    {
        const std::vector<ChannelLayout>& layout = *cur_layout_;
        const int rw = range_entry_words(spec_, layout);
        const int64_t per = ki.variant == V_LDS ? kRangeRowsLds : kRangeRows;
        int64_t entries = 0;
        for (const DevPage& r : *dp.ranges) entries += (r.n + per - 1) / per;
        RangeTable& t = range_table_[range_table_next_];
        range_table_next_ = (range_table_next_ + 1) % 3;
        // the staging table is written by the host: the copy of its previous use must have left it
        if (t.used) PA_HIP(hipEventSynchronize(t.event));
        else PA_HIP(hipEventCreateWithFlags(&t.event, hipEventDisableTiming));
        t.used = true;
        const size_t bytes = (size_t)entries * rw * 8;
        uint64_t* w = static_cast<uint64_t*>(t.host.ensure(bytes));
        for (const DevPage& r : *dp.ranges) {
            const uint64_t vec = range_aligned(r, spec_.used_channel) ? 1 : 0;
            for (int64_t row0 = 0; row0 < r.n; row0 += per) {
                const int64_t n = std::min<int64_t>(per, r.n - row0);
                for (int c = 0; c < spec_.n_in; c++) {
                    if (!spec_.used_channel[c]) continue;
                    const DevColumn& col = r.cols[c];
                    if (layout[c].type == PA_VARCHAR) {
                        *w++ = (uint64_t)(uintptr_t)col.values;
                        *w++ = (uint64_t)(uintptr_t)(col.offsets + row0);
                    }
                    else {
                        *w++ = (uint64_t)(uintptr_t)(static_cast<const char*>(col.values) + row0 * type_width(col.type));
                    }
                    if (layout[c].nullable) *w++ = col.nulls ? (uint64_t)(uintptr_t)(col.nulls + row0) : 0;
                }
                *w++ = (uint64_t)n | (vec << 32);
            }
        }
        void* dev = t.dev.ensure(bytes);
        PA_HIP(hipMemcpyAsync(dev, t.host.ptr(), bytes, hipMemcpyHostToDevice, s));
        PA_HIP(hipEventRecord(t.event, s));
        a.ranges = static_cast<const uint64_t*>(dev);
        a.n_ranges = entries;
        return entries;
    }

    bool run_page(const Compiled& ck, const DevPage& dp, bool vec, const RowList* list = nullptr, int64_t start_row = 0)
    {
        hipStream_t s = stream_.get();
        const KernelInfo& ki = ck.info;
        FusedArgs a;
        memset(&a, 0, sizeof a);
        for (int c = 0; c < spec_.n_in; c++) {
            if (!spec_.used_channel[c]) continue;
            a.v[c] = dp.cols[c].values;
            a.o[c] = dp.cols[c].offsets;
            a.nl[c] = dp.cols[c].nulls;
        }
        a.vec = vec ? 1 : 0;
        a.err = ctl_;
        a.gt_count = ctl_ + 1;
        a.overflow_rows = reinterpret_cast<uint64_t*>(ctl_ + 2);
        if (spec_.join) fill_join_args(a);
        if (ki.variant == V_BROW) return run_page_build_rows(ck, dp, a, start_row);
        // a table of ranges: one launch takes all of them, a workgroup per entry at a time
        int64_t range_entries = 0;
        if (dp.ranges) {
            PA_REQUIRE(ki.ranged && !list && start_row == 0, PA_ERR_DEVICE, "internal: range table handed to a kernel that walks one page");
            range_entries = fill_range_table(ki, dp, a, s);
        }
        int64_t offset = list ? list->first_row : start_row;
        const int64_t total = list ? list->first_row + list->chunk_rows : dp.n;
        // the HBM-table variant bounds the groups one launch can add so that the table can be sized first
        const bool table_tier = ki.variant == V_GT || ki.variant == V_LDSH || ki.variant == V_LDSP;
        const int64_t chunk = table_tier ? (int64_t)1 << 26 : total;
        // LDS variant: head = leading multiple of 256 rows through the vector kernel, tail = the rest through the scalar one
        // (a range table has its tails inside: the one kernel takes everything)
        const int64_t lds_head = dp.ranges ? total : (ki.variant == V_LDS && vec) ? (total & ~(int64_t)255) : 0;
        while (offset < total) {
            int64_t n = std::min(chunk, total - offset);
            // the first launch on the HBM table is a short one: it tells how many groups there are, which decides the
            // number of table replicas for the rest
            if (table_tier && !gt_probed_ && !list) n = std::min<int64_t>(n, (int64_t)1 << 22);
            bool use_tail = false;
            if (ki.variant == V_LDS) {
                if (offset < lds_head) {
                    n = lds_head - offset;
                    // nothing is known about the cardinality yet: a short first launch decides whether the register-table
                    // variant fits, instead of a whole wasted pass over a large page
                    if (!lds_probed_ && n > ((int64_t)1 << 22) && !dp.ranges) n = (int64_t)1 << 20;
                }
                else use_tail = true;
            }
            if (offset > 0) {
                for (int c = 0; c < spec_.n_in; c++) {
                    if (!spec_.used_channel[c]) continue;
                    const DevColumn& col = dp.cols[c];
                    if (col.varwidth) a.o[c] = col.offsets + offset;
                    else a.v[c] = static_cast<const char*>(col.values) + offset * type_width(col.type);
                    if (col.nulls) a.nl[c] = col.nulls + offset;
                }
            }
            a.n = list ? 0 : n;
            int64_t work = use_tail ? n : (n + 3) / 4;
            if (dp.ranges) work = range_entries * ki.block;  // a workgroup per entry
            if (list) {
                a.row_list = list->rows;
                a.n_list = list->count;
                a.list_blocked = list->rows ? 1 : 2;  // 2: rows 0 .. count-1 of (reordered) columns, one contiguous slice per workgroup
                work = list->count;
            }
            int grid;
            if (ki.variant == V_LDS) {
                int per_cu = std::max(1, std::min(16, (int)(160 * 1024 / ((size_t)ki.nw * ki.c * 64 * 8 + 512))));
                grid = (int)std::min<int64_t>((work + 63) / 64, (int64_t)cus_ * per_cu);
            }
            else if (ki.variant == V_LDSP) {
                grid = sub_parts_;  // one workgroup per partition, whatever the page holds
            }
            else if (ki.variant == V_LDSH) {
                grid = (int)std::min<int64_t>((work + ki.block - 1) / ki.block, (int64_t)cus_ * (ki.block == 1024 ? 1 : 2));  // LDS per CU: 160 KB
                // partition-ordered rows: at least one workgroup per partition, so that a workgroup's table meets the groups of
                // one partition (not of the two or three its slice would span with a workgroup per CU)
                if (list && list_grid_hint_ > grid) grid = (int)std::min<int64_t>((work + ki.block - 1) / ki.block, (int64_t)list_grid_hint_);
            }
            else {
                grid = (int)std::min<int64_t>((work + 255) / 256, (int64_t)cus_ * 8);
            }
            grid = std::max(grid, 1);
            // V_LDSH: every workgroup adds up to lc / 2 groups of its LDS table at the end of the launch, and must find room
            uint32_t reps = 1;
            if (ki.variant == V_LDSP) reps = gt_rep_;  // the HBM table only takes the rows that fall through
            else if (ki.variant == V_GT || ki.variant == V_LDSH) {
                const uint32_t want = desired_replicas(gt_probed_ ? groups_upper_ : std::max<uint64_t>(groups_upper_, (uint64_t)std::max(spec_.expected_groups, 1)));
                // change the layout only when it pays: much more replication needed, or far too much held
                reps = (want >= 2 * gt_rep_ || want * 4 <= gt_rep_) ? want : gt_rep_;
                reps = std::min<uint32_t>(reps, next_pow2((uint64_t)grid));
            }
            auto room_for_flush = [&](uint32_t r) { return ki.variant == V_LDSH ? (uint64_t)((grid + r - 1) / r) * (uint64_t)(ki.lc / 2) : (uint64_t)0; };
            uint64_t flush_room = room_for_flush(reps);
            if (ki.variant == V_GLOBAL) {
                a.slab = static_cast<uint64_t*>(slab_.ensure((size_t)grid * ki.nw * 8));
                if (!state_.ptr()) {
                    state_.ensure((size_t)ki.nw * 8);
                    PA_HIP(hipMemsetAsync(state_.ptr(), 0, (size_t)ki.nw * 8, s));
                }
            }
            else if (ki.variant == V_LDS) {
                const int b = lds_page_ & 1;
                if (!merge_stream_) {
                    merge_stream_ = pool_stream_acquire();
                    for (int i = 0; i < 2; i++) {
                        PA_HIP(hipEventCreateWithFlags(&ev_main_[i], hipEventDisableTiming));
                        PA_HIP(hipEventCreateWithFlags(&ev_merge_[i], hipEventDisableTiming));
                    }
                }
                // slab b, its overflow word and ev_main_[b] belong to launch k-2 until that one is confirmed
                while (inflight_.size() >= kMaxInflight) confirm_oldest();
                if (mode_ != V_LDS) {  // a confirmation moved the operator to the next tier: the rows from here on go there
                    resume_from_ = offset;
                    return false;
                }
                // slab b was last read by the merge of page k-2
                if (merge_pending_[b]) PA_HIP(hipStreamWaitEvent(s, ev_merge_[b], 0));
                a.slab = static_cast<uint64_t*>(lds_slab_[b].ensure((size_t)grid * ki.c * (1 + ki.w + ki.nw) * 8));
                a.overflow_rows = reinterpret_cast<uint64_t*>(ctl_ + 2 + 2 * b);
                // the merges of this launch and of the one still in flight add at most 2 * grid * C groups
                ensure_table(groups_upper_ + 2 * (uint64_t)grid * ki.c);
            }
            else {
                drain_merges();
                // sized by the groups seen so far, not by the rows: rows whose new group does not fit are spilled and
                // replayed after a rehash (see below)
                // (ensure_table doubles its argument: the table is kept at most half full)
                // (partition-owned tables: the HBM table only takes what falls through -- no need to size it for the estimate)
                const uint64_t expected = ki.variant == V_LDSP ? 0 : (uint64_t)std::max(spec_.expected_groups, 0);
                ensure_table(std::max<uint64_t>({(uint64_t)16384 / reps, groups_upper_ + groups_upper_ / 4, expected}) + flush_room, reps);
                if (gt_rep_ != reps) {  // the memory bound reduced the replicas
                    flush_room = room_for_flush(gt_rep_);
                    ensure_table(std::max<uint64_t>({(uint64_t)16384 / gt_rep_, groups_upper_ + groups_upper_ / 4, expected}) + flush_room);
                }
                a.spill_rows = static_cast<int32_t*>(spill_[0].ensure((size_t)n * 4));
                a.spill_count = reinterpret_cast<uint32_t*>(ctl_ + 6);
                if (!list) {
                    a.row_list = nullptr;
                    a.n_list = 0;
                }
            }
            a.gt_tag = gt_tag_.as<uint64_t>();
            a.gt_keys = gt_keys_.as<uint64_t>();
            a.gt_words = gt_words_.as<uint64_t>();
            a.gt_mask = gt_cap_ ? gt_cap_ - 1 : 0;
            a.gt_max_fill = ki.variant == V_LDS ? (int32_t)(gt_cap_ - gt_cap_ / 4) : (int32_t)(gt_cap_ / 2 - flush_room);
            a.gt_rep_mask = table_tier ? gt_rep_ - 1 : 0;
            if (ki.variant == V_LDSP) {
                a.sub_tag = sub_tag_.as<uint64_t>();
                a.sub_keys = sub_keys_.as<uint64_t>();
                a.sub_words = sub_words_.as<uint64_t>();
                a.sub_count = sub_count_.as<int32_t>();
                a.part_first = part_first_.as<int64_t>();
                a.row_list = nullptr;
                a.n_list = 0;
                a.list_blocked = 0;
                a.pad3 = sub_fresh_ ? 1 : 0;
                sub_fresh_ = false;
            }
            a.gt_rep_count = rep_count_.as<int32_t>();
            void* params[] = {&a};
            timer.begin(s);
            PA_HIP(hipModuleLaunchKernel(use_tail ? ck.tail_kernel.fn : ck.kernel.fn, grid, 1, 1, ki.block, 1, 1, 0, s, params, nullptr));
            timer.end(s, !use_tail);
            if (ki.variant == V_GLOBAL) {
                launch_merge_global_slab(a.slab, grid, ki.nw, ck.kinds.as<int32_t>(), state_.as<uint64_t>(), ctl_, s);
            }
            else if (ki.variant == V_LDS) {
                // The merge skips itself when the launch overflowed (overflow_rows != 0).  It runs on the merge
                // stream, overlapped with the next page's fused kernel; the host only waits for the fused kernel
                // and the control block (error word, group count, overflow counters).
                const int b = lds_page_ & 1;
                PA_HIP(hipMemcpyAsync(h_ctl_lds(b), ctl_, 32, hipMemcpyDeviceToHost, s));
                PA_HIP(hipEventRecord(ev_main_[b], s));
                PA_HIP(hipStreamWaitEvent(merge_stream_, ev_main_[b], 0));
                launch_merge_lds_slab(a.slab, grid, ki.c, ki.w, ki.nw, ck.kinds.as<int32_t>(), a.gt_tag, a.gt_keys, a.gt_words, a.gt_mask,
                                      a.gt_max_fill, a.gt_count, ctl_, a.overflow_rows,
                                      static_cast<int32_t*>(entry_slot_[b].ensure((size_t)grid * ki.c * 4)), merge_stream_);
                PA_HIP(hipEventRecord(ev_merge_[b], merge_stream_));
                merge_pending_[b] = true;
                lds_page_++;
                // The launch is confirmed later: its overflow word says whether a wave met more groups than its register table
                // holds -- the merge then skipped itself and the rows are redone on the next tier.  The host only ever waits
                // for a launch when the page cannot be read again later (not retained), or for the first launch of all, whose
                // outcome decides the tier of everything that follows.
                Inflight f;
                f.b = b;
                f.dp = dp;
                f.dp.n = (int32_t)(offset + n);
                f.vec = vec;
                f.offset = offset;
                f.sig = *cur_sig_;
                f.layout = *cur_layout_;
                inflight_.push_back(std::move(f));
                if (!retained_ || !lds_probed_) confirm_all();
                else poll_inflight();
                if (mode_ != V_LDS) {
                    resume_from_ = offset + n;
                    return false;
                }
            }
            else {
                // replay loop: grow the table until every row of the launch found room for its group
                int cur = 0;
                for (;;) {
                    read_group_counts(s);
                    raise_if(h_ctl_[0]);
                    const bool first_probe = !gt_probed_;
                    gt_probed_ = true;
                    const uint32_t spilled = (uint32_t)h_ctl_[6];
                    if (ki.variant == V_LDSP) {
                        uint64_t fell;
                        memcpy(&fell, h_ctl_ + 2, 8);
                        if (fell != 0) PA_HIP(hipMemsetAsync(ctl_ + 2, 0, 8, s));
                        sub_fell_ += fell;
                    }
                    if (ki.variant == V_LDSH) {
                        // rows that found no room in the workgroups' LDS tables: when they are a large part of the
                        // page, the cardinality is beyond this variant and later pages go to the HBM table directly
                        uint64_t fell;
                        memcpy(&fell, h_ctl_ + 2, 8);
                        if (fell != 0) PA_HIP(hipMemsetAsync(ctl_ + 2, 0, 8, s));
                        if (fell > (uint64_t)n / 4) mode_ = V_GT;
                    }
                    if (spilled == 0) {
                        // (also after replays of spilled rows: the rest of the page must not crawl through the wrong tier)
                        if (ki.variant == V_LDSH && mode_ == V_GT && !list && offset + n < total) resume_from_ = offset + n;
                        // the probe launch on the HBM table has told the cardinality: when it calls for the hash-partitioned
                        // tiers, the rest of this page already goes there
                        if (ki.variant == V_GT && first_probe && !list && offset + n < total && mode_ == V_GT) {
                            int p = 0;
                            if (partitioned_wanted(*cur_sig_, *cur_layout_, &p)) resume_from_ = offset + n;
                        }
                        break;
                    }
                    PA_HIP(hipMemsetAsync(ctl_ + 6, 0, 4, s));
                    // at least twice the slots (ensure_table doubles its argument)
                    ensure_table(std::max<uint64_t>((uint64_t)gt_cap_ / 2 + 1, groups_upper_ + spilled) + flush_room);
                    FusedArgs r = a;
                    r.n = 0;
                    if (r.list_blocked == 2) r.list_blocked = 1;  // the spilled rows are a real list
                    r.row_list = spill_[cur].as<int32_t>();
                    r.n_list = spilled;
                    r.spill_rows = static_cast<int32_t*>(spill_[cur ^ 1].ensure((size_t)spilled * 4));
                    r.gt_tag = gt_tag_.as<uint64_t>();
                    r.gt_keys = gt_keys_.as<uint64_t>();
                    r.gt_words = gt_words_.as<uint64_t>();
                    r.gt_mask = gt_cap_ - 1;
                    r.gt_max_fill = (int32_t)(gt_cap_ / 2 - flush_room);
                    void* rparams[] = {&r};
                    // (never more workgroups than the launch the table was sized for: V_LDSH flushes per workgroup)
                    // partition-owned tables: the spilled rows are a list over all partitions -- they go to the HBM table, through its kernel
                    const Compiled& rk = ki.variant == V_LDSP ? kernel_for(*cur_sig_, *cur_layout_, V_GT) : ck;
                    int rgrid = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)spilled + rk.info.block - 1) / rk.info.block, (int64_t)grid));
                    timer.begin(s);
                    PA_HIP(hipModuleLaunchKernel(rk.kernel.fn, rgrid, 1, 1, rk.info.block, 1, 1, 0, s, rparams, nullptr));
                    timer.end(s);
                    cur ^= 1;
                }
            }
            offset += n;
            if (resume_from_ >= 0) return false;  // the rest of the page goes to another tier (see run_tiers)
        }
        return true;
    }

    // ---- late confirmation of the few-groups launches ------------------------------------------------------------------
    struct Inflight {
        int b = 0;            // slab / overflow word / event pair of the launch
        DevPage dp;           // the page, cut at the end of the launched rows
        bool vec = false;
        int64_t offset = 0;   // first row of the launch
        std::string sig;
        std::vector<ChannelLayout> layout;
    };
    static constexpr size_t kMaxInflight = 2;
    int32_t* h_ctl_lds(int b) const { return h_ctl_ + 16 + 8 * b; }

    // confirms the launches whose kernel has finished, without waiting
    void poll_inflight()
    {
        while (!inflight_.empty() && hipEventQuery(ev_main_[inflight_.front().b]) == hipSuccess) confirm_oldest();
        (void)hipGetLastError();  // hipErrorNotReady is not an error here
    }
    void confirm_all()
    {
        while (!inflight_.empty()) confirm_oldest();
    }
    void confirm_oldest()
    {
        Inflight f = std::move(inflight_.front());
        inflight_.pop_front();
        PA_HIP(hipEventSynchronize(ev_main_[f.b]));
        const int32_t* hc = h_ctl_lds(f.b);
        uint64_t overflow;
        memcpy(&overflow, hc + 2 + 2 * f.b, 8);
        raise_if(hc[0]);
        if (overflow == 0) {
            groups_upper_ = std::max<uint64_t>(groups_upper_, (uint64_t)hc[1]);  // groups merged so far (in-flight merges are bounded above)
            lds_probed_ = true;
            return;
        }
        // more groups than the wave's register table: the launch's merge skipped itself; its rows -- and every later page --
        // go to the workgroup-level LDS table, which itself hands rows it has no room for to the HBM table
        hipStream_t s = stream_.get();
        drain_merges();
        PA_HIP(hipMemsetAsync(ctl_ + 2 + 2 * f.b, 0, 8, s));
        if (mode_ == V_LDS) mode_ = V_LDSH;
        const int64_t saved = resume_from_;
        const bool saved_retained = retained_;
        retained_ = false;
        run_tiers(f.sig, f.layout, f.dp, f.vec, f.offset);
        retained_ = saved_retained;
        resume_from_ = saved;
    }

    static void raise_if(int32_t code)
    {
        if (code == 0) return;
        switch (code) {
            case PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE: throw Error(code, "numeric value out of range (bigint/integer arithmetic overflow)");
            case PA_ERR_DIVISION_BY_ZERO: throw Error(code, "Division by zero");
            case PA_ERR_INSUFFICIENT_RESOURCES: throw Error(code, "group table capacity exceeded");
            case PA_ERR_NOT_SUPPORTED: throw Error(code, "VARCHAR group key longer than its declared bound / the device key packing supports");
            case PA_ERR_INVALID_ARGUMENT: throw Error(code, "VARCHAR blocks of device pages hold more bytes than their declared VARCHAR(n) allows");
            default: throw Error(code, "device-side error");
        }
    }

    void build_output();
    bool emit_on_device(const KernelInfo& ki, int64_t groups, bool keys_from_build_columns = false);
    void intern_keys(DevPage& dp, hipStream_t s);
    void intern_dictionary_key(const pa_page* page, int c, DevPage& dp, hipStream_t s);
    void decode_interned_keys();
    // channel of group key gi when that channel is interned, else -1
    int interned_channel(int gi) const
    {
        const OwnedExpr& pe = spec_.proj[spec_.group_proj[gi]];
        if (!pe.is_input_ref()) return -1;
        const int c = pe.node(pe.root).channel;
        return c < spec_.n_in && spec_.interned[c] ? c : -1;  // (channels >= n_in: build columns of the probe stage)
    }

    Spec spec_;
    Stream stream_;
    PageStager stager_;
    std::map<std::string, std::shared_ptr<const Compiled>> compiled_;
    std::string plan_fingerprint_;
    bool grouped_ = false, finishing_ = false, output_done_ = false, layout_fixed_ = false;
    int mode_ = V_GLOBAL, cus_ = 256, nw_ = 0, w_ = 0;
    std::string layout_id_;
    std::vector<bool> nullable_seen_;  // per channel: some page so far carried a valueIsNull array
    bool out_partial_ = false;         // the output is the accumulator states (Step.PARTIAL, or a generation to be combined)
    int generation_ = 0;
    bool is_combiner_ = false;
    std::unique_ptr<FusedAggregationOperator> next_;      // later generation: every page from the layout change on
    std::unique_ptr<FusedAggregationOperator> combiner_;  // FINAL-input operator over the generations' states
    DevBuf ctl_buf_;
    PinnedBuf h_ctl_buf_, h_table_;
    int32_t* ctl_ = nullptr;
    int32_t* h_ctl_ = nullptr;
    DevBuf slab_, state_, gt_tag_, gt_keys_, gt_words_;
    // LDS variant: the merge of page k runs on a second stream while the fused kernel of page k+1 streams
    DevBuf lds_slab_[2], entry_slot_[2], spill_[2], dense_keys_, dense_words_, null_flags_, rep_count_;
    DevBuf part_ids_, part_pos_, part_counts_, part_temp_;
    PinnedBuf h_rep_;
    uint32_t gt_rep_ = 1;
    uint64_t groups_sum_ = 0;
    bool gt_probed_ = false, lds_probed_ = false;
    bool join_checked_ = false;  // probe stage: the lookup source was looked at (first page)
    bool build_rows_table_ = false;  // the group table is indexed by build position (BROW)
    const Compiled* brow_keys_ = nullptr;  // its pa_brow_keys kernel
    int brow_occ_word_ = -1;       // KernelInfo::occ_word / occ_empty of the table
    uint64_t brow_occ_empty_ = 0;
    int64_t resume_from_ = -1;
    bool retained_ = false;               // the page being processed stays readable until its launches are confirmed
    // partition-owned tables (V_LDSP)
    int sub_parts_ = 0, sub_lc_ = 0;
    bool want_ldsp_ = false;
    uint64_t sub_fell_ = 0;               // rows that fell through to the HBM table (their partition's table was full)
    DevBuf sub_tag_, sub_keys_, sub_words_, sub_count_, part_first_;
    bool sub_fresh_ = false;       // the partitions' tables exist but no launch has written them yet
    bool parked_ = false;                 // a stable page waits for HBM (see retry_parked)
    pa_page parked_page_{};
    std::vector<pa_column> parked_cols_;
    size_t parked_need_ = 0;
    std::deque<Inflight> inflight_;       // few-groups launches not confirmed yet (at most kMaxInflight)
    const std::string* cur_sig_ = nullptr;                  // signature / layout of the page run_page works on
    const std::vector<ChannelLayout>* cur_layout_ = nullptr;
    // small pages (see gather_small_page)
    struct Run {
        int64_t rows = 0;
        bool flat = true;                 // every used channel is FLAT (a small range can join the arena)
        std::vector<pa_column> cols;      // first page of the range: every later page continues these buffers
    } run_;
    // stable device ranges waiting to be launched as one table (see retire_run)
    static constexpr size_t kMaxRanges = 16384;
    std::shared_ptr<std::vector<DevPage>> ranges_;
    int64_t range_rows_ = 0;
    struct RangeTable {
        PinnedBuf host;
        DevBuf dev;
        hipEvent_t event = nullptr;
        bool used = false;
    } range_table_[3];
    int range_table_next_ = 0;
    static constexpr size_t kArenaMaxSegs = 16384;
    struct Arena {
        int64_t rows = 0;
        std::vector<bool> nullable;
        std::vector<DevBuf> values, nulls, offsets;   // per channel; VARCHAR: values = bytes, offsets = rows + 1 entries
        std::vector<int64_t> bytes;                   // VARCHAR bytes used
        std::vector<CopySeg> segs;                    // copies of stable, device-readable pages, done at the arena's launch
        PinnedBuf h_table;
        DevBuf d_table;
        hipEvent_t table_event = nullptr;
        bool table_used = false;
        // VariableWidthBlocks of device pages (launch_var_append): the byte cursors live in HBM, two halves used in turn
        bool dev_var = false, var_fresh = true;
        int cursor_half = 0;
        DevBuf cursors;
        std::vector<VarSeg> vsegs;                    // deferred appends (stable pages), done at the arena's launch
        PinnedBuf h_vtable;
        DevBuf d_vtable;
        hipEvent_t vtable_event = nullptr;
        bool vtable_used = false;
    } arena_[2];
    int arena_cur_ = 0;
    const int32_t* kinds_dev_ = nullptr;
    hipStream_t merge_stream_ = nullptr;
    hipEvent_t ev_main_[2] = {nullptr, nullptr}, ev_merge_[2] = {nullptr, nullptr};
    bool merge_pending_[2] = {false, false};
    int lds_page_ = 0;
    uint32_t gt_cap_ = 0;
    uint64_t groups_upper_ = 0;
    std::vector<OutColumn> out_cols_;
    std::vector<pa_column> out_storage_;
    int32_t out_rows_ = 0;
    std::vector<std::unique_ptr<StringInterner>> interners_;  // per input channel, for Spec::interned channels
    PageStager dict_stager_;
    int list_grid_hint_ = 0;
    std::vector<DevBuf> reorder_bufs_;   // per channel: values, NULL flags of the partition-ordered copy of a chunk
    std::vector<DevBuf> dict_key_bufs_;  // per interned channel: uploaded ids, key ids, key NULL flags of a dictionary page
};

// VARCHAR group keys without a short bound: the page's strings become ids of the channel's dictionary, and the kernels
// group by the id column (equal strings <=> equal ids, so the grouping is MultiChannelGroupByHash's, exactly).
void FusedAggregationOperator::intern_keys(DevPage& dp, hipStream_t s)
{
    for (int c = 0; c < spec_.n_in; c++) {
        if (!spec_.interned[c]) continue;
        DevColumn& col = dp.cols[c];
        if (col.type == PA_INTEGER && !col.varwidth && col.values != nullptr) continue;  // took the dictionary route
        PA_REQUIRE(col.type == PA_VARCHAR && col.varwidth && col.offsets != nullptr, PA_ERR_INVALID_ARGUMENT,
                   "page block type does not match the declared input type");
        if (interners_.empty()) interners_.resize(spec_.n_in);
        if (!interners_[c]) interners_[c] = std::make_unique<StringInterner>();
        const int32_t* ids = interners_[c]->intern(col.values, col.offsets, col.nulls, dp.n, s);
        col.type = PA_INTEGER;
        col.varwidth = false;
        col.values = ids;
        col.offsets = nullptr;
    }
}

// The dictionary fast path of MultiChannelGroupByHash (MultiChannelGroupByHash.java:465-512: group ids are computed once per
// dictionary entry and looked up through the ids): the dictionary's strings are interned -- a handful of entries instead of
// every row -- and the rows' keys are a gather of those dictionary ids.  Works across pages with different dictionaries,
// since the ids are the operator's own.
void FusedAggregationOperator::intern_dictionary_key(const pa_page* page, int c, DevPage& dp, hipStream_t s)
{
    const pa_column& col = page->columns[c];
    const int64_t n = page->position_count;
    const int64_t dn = col.encoding == PA_RLE ? 1 : col.dictionary_size;
    std::vector<pa_column> cols((size_t)spec_.n_in);
    cols[c] = *col.dictionary;
    pa_page dpage{};
    dpage.position_count = (int32_t)dn;
    dpage.channel_count = spec_.n_in;
    dpage.columns = cols.data();
    dpage.mem = page->mem;
    std::vector<bool> only(spec_.n_in, false);
    only[c] = true;
    DevPage d = dict_stager_.stage(&dpage, &only, s);
    const DevColumn& dcol = d.cols[c];
    PA_REQUIRE(dcol.type == PA_VARCHAR && dcol.varwidth && dcol.offsets != nullptr, PA_ERR_INVALID_ARGUMENT,
               "page block type does not match the declared input type");
    if (interners_.empty()) interners_.resize(spec_.n_in);
    if (!interners_[c]) interners_[c] = std::make_unique<StringInterner>();
    const int32_t* dict_ids = interners_[c]->intern(dcol.values, dcol.offsets, dcol.nulls, dn, s);
    if (dict_key_bufs_.empty()) dict_key_bufs_.resize((size_t)spec_.n_in * 3);
    DevBuf& ids_buf = dict_key_bufs_[(size_t)c * 3], &out_buf = dict_key_bufs_[(size_t)c * 3 + 1], &nulls_buf = dict_key_bufs_[(size_t)c * 3 + 2];
    const int32_t* ids = nullptr;
    if (col.encoding == PA_DICTIONARY && page->mem == PA_MEM_DEVICE) ids = col.ids;
    else {
        int32_t* dev = static_cast<int32_t*>(ids_buf.ensure((size_t)n * 4));
        if (col.encoding == PA_RLE) PA_HIP(hipMemsetAsync(dev, 0, (size_t)n * 4, s));
        else PA_HIP(hipMemcpyAsync(dev, col.ids, (size_t)n * 4, hipMemcpyHostToDevice, s));
        ids = dev;
    }
    int32_t* out = static_cast<int32_t*>(out_buf.ensure((size_t)n * 4));
    launch_gather_flat(dict_ids, 4, ids, n, out, s);
    uint8_t* out_nulls = nullptr;
    if (dcol.nulls) {
        out_nulls = static_cast<uint8_t*>(nulls_buf.ensure((size_t)n));
        launch_gather_nulls(dcol.nulls, ids, n, out_nulls, s);
    }
    DevColumn& k = dp.cols[c];
    k.type = PA_INTEGER;
    k.varwidth = false;
    k.values = out;
    k.offsets = nullptr;
    k.nulls = out_nulls;
}

// The id key columns of the assembled output back to VariableWidthBlocks.
void FusedAggregationOperator::decode_interned_keys()
{
    hipStream_t s = stream_.get();
    for (int gi = 0; gi < (int)spec_.group_proj.size(); gi++) {
        const int c = interned_channel(gi);
        if (c < 0) continue;
        OutColumn& oc = out_cols_[gi];
        const int64_t n = out_rows_;
        if (oc.host_ready && spec_.output_mem != PA_MEM_DEVICE) {  // assembled on the host and not uploaded yet
            oc.values.ensure((size_t)std::max<int64_t>(n, 1) * 4);
            if (n) PA_HIP(hipMemcpyAsync(oc.values.ptr(), oc.h_values.ptr(), (size_t)n * 4, hipMemcpyHostToDevice, s));
            if (oc.has_nulls) {
                oc.nulls.ensure((size_t)std::max<int64_t>(n, 1));
                if (n) PA_HIP(hipMemcpyAsync(oc.nulls.ptr(), oc.h_nulls.ptr(), (size_t)n, hipMemcpyHostToDevice, s));
            }
        }
        DevBuf values, offsets;
        if (n == 0 || interners_.empty() || !interners_[c]) {  // no page ever arrived
            PA_HIP(hipMemsetAsync(offsets.ensure(4), 0, 4, s));
            values.ensure(1);
        }
        else {
            interners_[c]->decode(oc.values.as<int32_t>(), oc.has_nulls ? oc.nulls.as<uint8_t>() : nullptr, n, &values, &offsets, s);
        }
        PA_HIP(hipStreamSynchronize(s));  // the id column goes back to the pool
        oc.values = std::move(values);
        oc.offsets = std::move(offsets);
        oc.type = PA_VARCHAR;
        oc.varwidth = true;
        oc.host_ready = false;
    }
}

// Large grouped results (Q3: millions of groups) never visit the host: k_gt_emit compacts the table and writes the output
// blocks in one pass.  Small results and VARCHAR keys take the host assembly of build_output below.
// keys_from_build_columns (build-row table): the group of slot p is build row p and its keys are build columns, so they are copied
// from there -- no packed key words (pa_brow_keys is not run) and no group count beforehand: `groups` is then the table's capacity,
// an upper bound, and the emit kernel's counter says how many there were.
bool FusedAggregationOperator::emit_on_device(const KernelInfo& ki, int64_t groups, bool keys_from_build_columns)
{
    constexpr int64_t kMinGroups = 4096;
    const int nkeys = (int)spec_.group_proj.size();
    const bool has_hash = nkeys > 0 && spec_.hash_channel >= 0;
    const bool partial = out_partial_;
    if (groups < kMinGroups || groups > INT32_MAX) return false;
    for (int gi = 0; gi < nkeys; gi++) {
        if (ki.keys[gi].type == PA_VARCHAR || ki.keys[gi].type == PA_DECIMAL) return false;  // (DECIMAL keys: their hash is the value itself)
    }
    for (size_t k = 0; k < spec_.aggs.size(); k++) {
        if (ki.agg_limbs[k] > 0) return false;  // DECIMAL sums are put together on the host (decimal_host.hpp)
        const int value_proj = spec_.step == PA_STEP_FINAL ? spec_.aggs[k].input_channel + 1 : spec_.aggs[k].input_channel;
        if (spec_.aggs[k].fn != PA_AGG_COUNT_STAR && spec_.proj[value_proj].root_type() == PA_DECIMAL && (spec_.aggs[k].fn == PA_AGG_MIN || spec_.aggs[k].fn == PA_AGG_MAX)) return false;
    }
    if (keys_from_build_columns && (has_hash || !spec_.join || spec_.join->brow_group_proj.size() != (size_t)nkeys)) return false;
    GtEmitArgs a{};
    int n = 0;
    auto add = [&](int kind, int type) -> GtEmitCol* {
        if (n >= GT_EMIT_MAX_COLS) return nullptr;
        GtEmitCol& c = a.col[n++];
        c.kind = kind;
        c.type = type;
        c.null_word = -1;
        c.width = type_width(type);
        return &c;
    };
    std::vector<bool> nullable;
    for (int gi = 0; gi < nkeys; gi++) {
        const KeyPart& kp = ki.keys[gi];
        if (keys_from_build_columns) {
            const OwnedExpr& pe = spec_.proj[(size_t)spec_.join->brow_group_proj[(size_t)gi]];
            const int v = pe.is_input_ref() ? pe.node(pe.root).channel - spec_.n_in : -1;
            if (v < 0 || v >= (int)spec_.join->build_cols.size()) return false;
            const BuildColumn& bc = spec_.join->ls->cols[(size_t)spec_.join->build_cols[(size_t)v]];
            GtEmitCol* c = add(GT_EMIT_COLUMN, kp.type);
            if (!c || bc.varwidth || type_width(bc.type) != c->width) return false;
            c->src = bc.values.ptr();
            c->src_nulls = bc.has_nulls ? bc.nulls.as<uint8_t>() : nullptr;
            nullable.push_back(bc.has_nulls);
            continue;
        }
        GtEmitCol* c = add(GT_EMIT_KEY, kp.type);
        if (!c) return false;
        c->word = kp.word;
        c->shift = kp.shift;
        c->bits = kp.bits;
        c->null_word = kp.null_word;
        c->null_shift = kp.null_shift;
        const int ic = interned_channel(gi);  // the key's share of $hashvalue is the hash of the string, not of its id
        c->dict_hash = (ic >= 0 && !interners_.empty() && interners_[ic]) ? interners_[ic]->hashes() : nullptr;
        nullable.push_back(kp.null_word >= 0);
    }
    if (has_hash) {
        if (!add(GT_EMIT_HASH, PA_BIGINT)) return false;
        nullable.push_back(false);
    }
    for (size_t k = 0; k < spec_.aggs.size(); k++) {
        const pa_aggregate& ag = spec_.aggs[k];
        const int cw = ki.agg_words[k].first, vw = ki.agg_words[k].second;
        const bool value_is_double = vw >= 0 && ki.word_kind[vw] == W_SUMF;
        const bool min_max = ag.fn == PA_AGG_MIN || ag.fn == PA_AGG_MAX;
        const int value_proj = spec_.step == PA_STEP_FINAL ? ag.input_channel + 1 : ag.input_channel;
        if (partial && min_max) {
            GtEmitCol* c = add(GT_EMIT_STATE, PA_BIGINT);
            if (!c) return false;
            c->word = cw;
            nullable.push_back(false);
        }
        if (min_max && spec_.proj[value_proj].root_type() == PA_VARCHAR) return false;  // strings are assembled on the host
        if (min_max) {  // the value itself (final result, or the value half of the PARTIAL state): NULL while no input was seen
            GtEmitCol* c = add(GT_EMIT_MINMAX, spec_.proj[value_proj].root_type());
            if (!c) return false;
            c->cw = cw;
            c->vw = vw;
            c->shift = ag.fn == PA_AGG_MIN ? 1 : 0;
            nullable.push_back(true);
            continue;
        }
        if (partial) {
            for (int part = 0; part < ((ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_AVG) ? 2 : 1); part++) {
                GtEmitCol* c = add(GT_EMIT_STATE, part == 0 ? PA_BIGINT : (value_is_double ? PA_DOUBLE : PA_BIGINT));
                if (!c) return false;
                c->word = part == 0 ? cw : vw;
                nullable.push_back(false);
            }
            continue;
        }
        const bool as_double = ag.fn == PA_AGG_AVG || (ag.fn == PA_AGG_SUM && value_is_double);
        // sum / avg over REAL narrow their DOUBLE state on output (RealSumAggregation.output, RealAverageAggregation.output)
        const bool real_out = as_double && (ag.input_type == PA_REAL || spec_.proj[value_proj].root_type() == PA_REAL);
        const int type = real_out ? PA_REAL : (as_double ? PA_DOUBLE : ((ag.fn == PA_AGG_SUM) ? spec_.proj[value_proj].root_type() : PA_BIGINT));
        const int kind = ag.fn == PA_AGG_SUM ? GT_EMIT_SUM : (ag.fn == PA_AGG_AVG ? GT_EMIT_AVG : GT_EMIT_COUNT);
        if (ag.fn != PA_AGG_SUM && ag.fn != PA_AGG_AVG && ag.fn != PA_AGG_COUNT && ag.fn != PA_AGG_COUNT_STAR) return false;
        GtEmitCol* c = add(kind, type);
        if (!c) return false;
        c->cw = cw;
        c->vw = vw;
        nullable.push_back(kind != GT_EMIT_COUNT);
    }
    hipStream_t s = stream_.get();
    out_cols_.clear();
    out_cols_.resize(n);
    for (int c = 0; c < n; c++) {
        OutColumn& oc = out_cols_[c];
        oc.type = a.col[c].type;
        a.col[c].values = oc.values.ensure((size_t)groups * a.col[c].width);
        a.col[c].nulls = nullable[c] ? static_cast<uint8_t*>(oc.nulls.ensure((size_t)groups)) : nullptr;
    }
    a.tag = table_tags();
    a.keys = gt_keys_.as<uint64_t>();
    a.words = table_words();
    a.st = table_strides();
    a.cap = gt_cap_;
    a.W = std::max(w_, 1);
    a.NW = nw_;
    a.ncols = n;
    a.counter = reinterpret_cast<uint32_t*>(ctl_ + 7);
    a.null_flags = static_cast<uint32_t*>(null_flags_.ensure(GT_EMIT_MAX_COLS * 4));
    PA_HIP(hipMemsetAsync(ctl_ + 7, 0, 4, s));
    PA_HIP(hipMemsetAsync(a.null_flags, 0, GT_EMIT_MAX_COLS * 4, s));
    launch_gt_emit(a, s);
    uint32_t flags[GT_EMIT_MAX_COLS];
    PA_HIP(hipMemcpyAsync(flags, a.null_flags, sizeof(flags), hipMemcpyDeviceToHost, s));
    PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 32, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    if (keys_from_build_columns) groups = (int64_t)(uint32_t)h_ctl_[7];
    PA_REQUIRE((int64_t)(uint32_t)h_ctl_[7] == groups, PA_ERR_DEVICE, "group table and group count disagree");
    for (int c = 0; c < n; c++) out_cols_[c].has_nulls = flags[c] != 0;
    out_rows_ = (int32_t)groups;
    return true;
}

// Final values: InMemoryHashAggregationBuilder.buildResult (…/InMemoryHashAggregationBuilder.java:244-298) /
// AggregationOperator.getOutput (…/AggregationOperator.java:164-186) with the output functions of SURVEY a15.
// Group counts here are tiny next to the input (Q1: 4 rows), so the states are brought to the host and the
// output blocks assembled there; column order = keys, ($hashvalue), aggregates.
void FusedAggregationOperator::build_output()
{
    hipStream_t s = stream_.get();
    drain_merges();
    // any signature works for the layout (all share nw_/w_): take the first compiled kernel, or build
    // one for the all-non-null layout when no page ever arrived
    if (compiled_.empty()) {
        std::vector<ChannelLayout> layout(spec_.n_in);
        for (int c = 0; c < spec_.n_in; c++) layout[c].type = spec_.in_types[c];
        auto c = std::make_shared<Compiled>();
        c->info = generate(spec_, layout, grouped_ ? V_GT : V_GLOBAL);
        nw_ = c->info.nw;
        w_ = c->info.w;
        compiled_["-"] = std::move(c);
    }
    const KernelInfo& ki = compiled_.begin()->second->info;

    // the error word and, for grouped results, the dense (keys, words) rows of the occupied table slots
    PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 32, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    raise_if(h_ctl_[0]);
    if (grouped_ && sub_parts_ > 0) {
        // Partition-owned tables.  Their HBM form IS a group table of sub_parts_ * lc slots (same arrays, same layouts; only the
        // probe sequence differs, and nothing probes any more).  When the HBM table proper holds no group -- nothing fell
        // through, no other tier ran -- they simply become the table; else their groups are folded into it, one upsert per group.
        std::vector<int32_t> per_part((size_t)sub_parts_);
        PA_HIP(hipMemcpyAsync(per_part.data(), sub_count_.ptr(), (size_t)sub_parts_ * 4, hipMemcpyDeviceToHost, s));
        read_group_counts(s);
        uint64_t total = 0;
        for (int32_t c : per_part) total += (uint64_t)c;
        const uint32_t sub_cap = (uint32_t)sub_parts_ * (uint32_t)sub_lc_;
        if (groups_sum_ == 0) {
            gt_tag_ = std::move(sub_tag_);
            gt_keys_ = std::move(sub_keys_);
            gt_words_ = std::move(sub_words_);
            gt_cap_ = sub_cap;
            gt_rep_ = 1;
            // (the count goes to the device from the pinned control block: no wait)
            h_ctl_[1] = (int32_t)total;
            PA_HIP(hipMemcpyAsync(ctl_ + 1, h_ctl_ + 1, 4, hipMemcpyHostToDevice, s));
            groups_upper_ = groups_sum_ = total;
        }
        else {
            ensure_table(groups_sum_ + total + 1024, 1);
            launch_gt_fold(sub_tag_.as<uint64_t>(), sub_keys_.as<uint64_t>(), sub_words_.as<uint64_t>(), sub_cap, 1, std::max(w_, 1), nw_, kinds_dev_,
                           gt_tag_.as<uint64_t>(), gt_keys_.as<uint64_t>(), gt_words_.as<uint64_t>(), gt_cap_ - 1, 1, ctl_ + 1, rep_count_.as<int32_t>(), ctl_, s);
            PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 32, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            raise_if(h_ctl_[0]);
            sub_tag_.release();
            sub_keys_.release();
            sub_words_.release();
        }
        sub_parts_ = 0;
    }
    std::vector<uint64_t> keys, words;
    int64_t groups = 0;
    if (!grouped_) {
        groups = 1;
        words.assign(nw_, 0);
        if (state_.ptr()) PA_HIP(hipMemcpy(words.data(), state_.ptr(), (size_t)nw_ * 8, hipMemcpyDeviceToHost));
    }
    else if (gt_cap_ > 0) {
        if (gt_rep_ > 1) {
            // fold the replicas into one table: states of one key combine with the aggregates' combine functions
            read_group_counts(s);
            ensure_table(groups_sum_, 1);
            PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 32, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            raise_if(h_ctl_[0]);
        }
        // build-row table of some size: the output blocks straight from the accumulators and the build columns
        if (build_rows_table_ && (int64_t)gt_cap_ >= (1 << 14) && emit_on_device(ki, (int64_t)gt_cap_, true)) return;
        if (build_rows_table_) {
            // build-row table: no kernel counted its groups, and its key words are still to be written -- once per group, from
            // the build columns (pa_brow_keys)
            FusedArgs a;
            memset(&a, 0, sizeof a);
            fill_join_args(a);
            a.err = ctl_;
            a.gt_count = ctl_ + 1;
            PA_HIP(hipMemsetAsync(ctl_ + 1, 0, 4, s));
            a.gt_tag = gt_tag_.as<uint64_t>();
            a.gt_keys = gt_keys_.as<uint64_t>();
            a.gt_words = gt_words_.as<uint64_t>();
            a.gt_mask = gt_cap_ - 1;
            void* params[] = {&a};
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)gt_cap_ + 255) / 256, (int64_t)cus_ * 8));
            PA_HIP(hipModuleLaunchKernel(brow_keys_->tail_kernel.fn, grid, 1, 1, 256, 1, 1, 0, s, params, nullptr));
            PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 32, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            raise_if(h_ctl_[0]);
        }
        groups = h_ctl_[1];
        if (emit_on_device(ki, groups)) return;
        if (groups > 0) {
            const int kw = std::max(w_, 1);
            dense_keys_.ensure((size_t)groups * kw * 8);
            dense_words_.ensure((size_t)groups * nw_ * 8);
            PA_HIP(hipMemsetAsync(ctl_ + 7, 0, 4, s));
            const GtStrides st = table_strides();
            launch_gt_compact(table_tags(), gt_keys_.as<uint64_t>(), table_words(), gt_cap_, kw, nw_, dense_keys_.as<uint64_t>(),
                              dense_words_.as<uint64_t>(), reinterpret_cast<uint32_t*>(ctl_ + 7), s, &st);
            uint8_t* land = static_cast<uint8_t*>(h_table_.ensure((size_t)groups * (kw + nw_) * 8));
            PA_HIP(hipMemcpyAsync(land, dense_keys_.ptr(), (size_t)groups * kw * 8, hipMemcpyDeviceToHost, s));
            PA_HIP(hipMemcpyAsync(land + (size_t)groups * kw * 8, dense_words_.ptr(), (size_t)groups * nw_ * 8, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            const uint64_t* hk = reinterpret_cast<const uint64_t*>(land);
            const uint64_t* hw = hk + (size_t)groups * kw;
            keys.resize((size_t)groups * w_);
            for (int64_t g = 0; g < groups; g++) {
                for (int w = 0; w < w_; w++) keys[(size_t)g * w_ + w] = hk[(size_t)g * kw + w];
            }
            words.assign(hw, hw + (size_t)groups * nw_);
        }
    }
    PA_REQUIRE(groups <= INT32_MAX, PA_ERR_INSUFFICIENT_RESOURCES, "too many groups for one output page");
    out_rows_ = (int32_t)groups;

    const int nkeys = (int)spec_.group_proj.size();
    const bool has_hash = nkeys > 0 && spec_.hash_channel >= 0;
    const bool partial = out_partial_;
    int agg_cols = 0;
    for (const auto& ag : spec_.aggs) agg_cols += (partial && ag.fn != PA_AGG_COUNT && ag.fn != PA_AGG_COUNT_STAR) ? 2 : 1;
    const int ncols = nkeys + (has_hash ? 1 : 0) + agg_cols;
    out_cols_.clear();
    out_cols_.resize(ncols);
    std::vector<std::vector<uint8_t>> host_cols(ncols), host_nulls(ncols);
    std::vector<std::vector<int32_t>> host_offsets(ncols);
    std::vector<int64_t> row_hash(groups, 0);
    int col = 0;
    for (int gi = 0; gi < nkeys; gi++, col++) {
        const KeyPart& kp = ki.keys[gi];
        OutColumn& oc = out_cols_[col];
        // interned key: the id column is assembled here, decode_interned_keys turns it into strings; the string hashes
        // come from the dictionary
        std::vector<uint64_t> dict_hash;
        const int ic = interned_channel(gi);
        if (ic >= 0 && has_hash && groups > 0 && !interners_.empty() && interners_[ic] && interners_[ic]->size()) {
            dict_hash.resize(interners_[ic]->size());
            PA_HIP(hipMemcpyAsync(dict_hash.data(), interners_[ic]->hashes(), dict_hash.size() * 8, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
        }
        oc.type = kp.type;
        oc.varwidth = kp.type == PA_VARCHAR;
        auto& data = host_cols[col];
        auto& nulls = host_nulls[col];
        auto& offs = host_offsets[col];
        nulls.assign(groups ? groups : 1, 0);
        bool any_null = false;
        if (oc.varwidth) offs.push_back(0);
        const uint64_t mask = kp.bits >= 64 ? ~0ULL : ((1ULL << kp.bits) - 1ULL);
        for (int64_t g = 0; g < groups; g++) {
            const uint64_t* kw = &keys[(size_t)g * w_];
            bool is_null = kp.null_word >= 0 && ((kw[kp.null_word] >> kp.null_shift) & 1ULL);
            int64_t h = 0;
            if (is_null) {
                nulls[g] = 1;
                any_null = true;
            }
            uint64_t w0 = (kw[kp.word] >> kp.shift) & mask;
            switch (kp.type) {
                case PA_BIGINT:
                case PA_DECIMAL: {
                    int64_t v = is_null ? 0 : (int64_t)w0;
                    data.insert(data.end(), (uint8_t*)&v, (uint8_t*)&v + 8);
                    if (!is_null) h = kp.type == PA_DECIMAL ? v : host_hash_bigint(v);  // ShortDecimalType.hashCodeOperator: the value
                    break;
                }
                case PA_INTEGER:
                case PA_DATE:
                case PA_REAL: {  // (REAL: the canonical float bits, hashed like the int they are -- RealType.hashCodeOperator)
                    int32_t v = is_null ? 0 : (int32_t)(uint32_t)w0;
                    data.insert(data.end(), (uint8_t*)&v, (uint8_t*)&v + 4);
                    if (!is_null) h = (ic >= 0 && (size_t)(uint32_t)v < dict_hash.size()) ? (int64_t)dict_hash[(uint32_t)v] : host_hash_bigint((int64_t)v);
                    break;
                }
                case PA_BOOLEAN: {
                    uint8_t v = is_null ? 0 : (uint8_t)(w0 != 0);
                    data.push_back(v);
                    if (!is_null) h = (int64_t)host_xxh64_long(v ? 1 : 0);
                    break;
                }
                case PA_DOUBLE: {
                    uint64_t v = is_null ? 0 : w0;
                    data.insert(data.end(), (uint8_t*)&v, (uint8_t*)&v + 8);
                    if (!is_null) h = host_hash_bigint((int64_t)v);  // already canonical (+0, one NaN)
                    break;
                }
                case PA_VARCHAR: {
                    uint8_t bytes[16];
                    int len = 0;
                    if (!is_null) {
                        if (kp.bits == 128) {
                            uint64_t a0 = kw[kp.word], b0 = kw[kp.word + 1];
                            len = (int)(b0 >> 56);
                            for (int b = 0; b < len; b++) bytes[b] = b < 8 ? (uint8_t)(a0 >> (8 * b)) : (uint8_t)(b0 >> (8 * (b - 8)));
                        }
                        else {
                            len = (int)(w0 >> (8 * kp.bound));
                            for (int b = 0; b < len; b++) bytes[b] = (uint8_t)(w0 >> (8 * b));
                        }
                        h = (int64_t)host_xxh64(bytes, len);
                    }
                    data.insert(data.end(), bytes, bytes + len);
                    offs.push_back((int32_t)data.size());
                    break;
                }
                default:
                    break;
            }
            row_hash[g] = (int64_t)(31ULL * (uint64_t)row_hash[g] + (uint64_t)h);  // CombineHashFunction.java:26-29
        }
        oc.has_nulls = any_null;
    }
    if (has_hash) {
        // $hashvalue of the group key == InterpretedHashGenerator over the key columns
        // (HashGenerationOptimizer.java:867-890 defines the precomputed channel as the same function)
        OutColumn& oc = out_cols_[col];
        oc.type = PA_BIGINT;
        auto& data = host_cols[col];
        data.resize((size_t)groups * 8);
        if (groups) memcpy(data.data(), row_hash.data(), (size_t)groups * 8);
        host_nulls[col].assign(groups ? groups : 1, 0);
        col++;
    }
    for (size_t k = 0; k < spec_.aggs.size(); k++) {
        const pa_aggregate& ag = spec_.aggs[k];
        int cw = ki.agg_words[k].first, vw = ki.agg_words[k].second;
        const bool value_is_double = vw >= 0 && ki.word_kind[vw] == W_SUMF;
        if (ag.fn == PA_AGG_MIN || ag.fn == PA_AGG_MAX) {
            // the value back from its order-preserving image (pa_img_*); PARTIAL = [count BIGINT, value], NULL while count == 0
            if (partial) {
                OutColumn& cc = out_cols_[col];
                cc.type = PA_BIGINT;
                host_nulls[col].assign(groups ? groups : 1, 0);
                host_cols[col].resize((size_t)groups * 8);
                const uint64_t one = 1;
                for (int64_t g = 0; g < groups; g++) memcpy(&host_cols[col][(size_t)g * 8], cw >= 0 ? &words[(size_t)g * nw_ + cw] : &one, 8);
                cc.has_nulls = false;
                col++;
            }
            const int value_proj = spec_.step == PA_STEP_FINAL ? ag.input_channel + 1 : ag.input_channel;
            OutColumn& oc = out_cols_[col];
            oc.type = spec_.proj[value_proj].root_type();
            auto& data = host_cols[col];
            auto& nulls = host_nulls[col];
            nulls.assign(groups ? groups : 1, 0);
            bool any_null = false;
            if (oc.type == PA_VARCHAR) {  // image = up to 7 bytes big-endian, then the length (pa_img_str7)
                oc.varwidth = true;
                auto& offs = host_offsets[col];
                offs.assign(1, 0);
                for (int64_t g = 0; g < groups; g++) {
                    const uint64_t* ww = &words[(size_t)g * nw_];
                    if (cw >= 0 && ww[cw] == 0) {
                        nulls[g] = 1;
                        any_null = true;
                    }
                    else {
                        const uint64_t img = ag.fn == PA_AGG_MIN ? ~ww[vw] : ww[vw];
                        const int len = (int)(img & 0xff);
                        for (int b = 0; b < len && b < 7; b++) data.push_back((uint8_t)(img >> (56 - 8 * b)));
                    }
                    offs.push_back((int32_t)data.size());
                }
                oc.has_nulls = any_null;
                col++;
                continue;
            }
            const int width = type_width(oc.type);
            data.assign((size_t)groups * width, 0);
            for (int64_t g = 0; g < groups; g++) {
                const uint64_t* ww = &words[(size_t)g * nw_];
                if (cw >= 0 && ww[cw] == 0) {
                    nulls[g] = 1;
                    any_null = true;
                    continue;
                }
                uint64_t img = ag.fn == PA_AGG_MIN ? ~ww[vw] : ww[vw];
                uint64_t bits;
                if (oc.type == PA_DOUBLE || oc.type == PA_REAL) bits = (img >> 63) ? (img & 0x7fffffffffffffffULL) : ~img;
                else if (oc.type == PA_BOOLEAN) bits = img;
                else bits = img ^ 0x8000000000000000ULL;
                if (oc.type == PA_REAL) {  // the image is the widened value's
                    double d;
                    memcpy(&d, &bits, 8);
                    const float f = (float)d;
                    memcpy(&data[(size_t)g * width], &f, 4);
                    continue;
                }
                memcpy(&data[(size_t)g * width], &bits, (size_t)width);  // little endian: the low bytes are the narrower value
            }
            oc.has_nulls = any_null;
            col++;
            continue;
        }
        if (ki.agg_limbs[k] > 0) {
            // sum / avg over DECIMAL: the limb sums -> the exact total.  SINGLE / FINAL: sum is a DECIMAL(38, s) -- NUMERIC_VALUE_OUT_OF_RANGE
            // at 10^38 (DecimalSumAggregation.outputLongDecimal) --, avg the total / count rounded half up in the input's type
            // (DecimalAverageAggregation.average); PARTIAL: [count BIGINT, sum DECIMAL(38, s)]
            const int limbs = ki.agg_limbs[k];
            if (partial) {
                OutColumn& cc = out_cols_[col];
                cc.type = PA_BIGINT;
                host_nulls[col].assign(groups ? groups : 1, 0);
                host_cols[col].resize((size_t)groups * 8);
                const uint64_t one = 1;
                for (int64_t g = 0; g < groups; g++) memcpy(&host_cols[col][(size_t)g * 8], cw >= 0 ? &words[(size_t)g * nw_ + cw] : &one, 8);
                cc.has_nulls = false;
                col++;
            }
            const int value_proj = spec_.step == PA_STEP_FINAL ? ag.input_channel + 1 : ag.input_channel;
            const int32_t in_type = spec_.step == PA_STEP_FINAL ? ag.input_type : spec_.proj[value_proj].root_type();
            OutColumn& oc = out_cols_[col];
            oc.type = (partial || ag.fn == PA_AGG_SUM) ? PA_LONG_DECIMAL : in_type;
            const int width = type_width(oc.type);
            auto& data = host_cols[col];
            auto& nulls = host_nulls[col];
            nulls.assign(groups ? groups : 1, 0);
            data.assign((size_t)groups * width, 0);
            bool any_null = false;
            for (int64_t g = 0; g < groups; g++) {
                const uint64_t* ww = &words[(size_t)g * nw_];
                const int64_t count = cw >= 0 ? (int64_t)ww[cw] : 1;
                if (count == 0 && !partial) {
                    nulls[g] = 1;
                    any_null = true;
                    continue;
                }
                Wide192 total = decimal_total(ww, vw, limbs);
                if (!partial && ag.fn == PA_AGG_AVG) total = decimal_average(total, count);
                bool neg = false;
                unsigned __int128 mag = 0;
                const unsigned __int128 bound = oc.type == PA_DECIMAL ? ((unsigned __int128)1 << 63) : kTen38;
                PA_REQUIRE(decimal_fits(total, bound, &neg, &mag), PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE, "Decimal overflow");
                if (oc.type == PA_DECIMAL) {
                    const int64_t v = neg ? -(int64_t)(uint64_t)mag : (int64_t)(uint64_t)mag;
                    memcpy(&data[(size_t)g * 8], &v, 8);
                }
                else long_decimal_store(&data[(size_t)g * 16], neg, mag);
            }
            oc.has_nulls = any_null;
            col++;
            continue;
        }
        if (partial) {
            // Step.PARTIAL: the accumulator states themselves -- [count BIGINT] (+ [sum]) per aggregate, the flattened form
            // of the reference's LongState / LongDoubleState / LongLongState intermediate rows
            for (int part = 0; part < ((ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_AVG) ? 2 : 1); part++, col++) {
                OutColumn& oc = out_cols_[col];
                oc.type = part == 0 ? PA_BIGINT : (value_is_double ? PA_DOUBLE : PA_BIGINT);
                auto& data = host_cols[col];
                host_nulls[col].assign(groups ? groups : 1, 0);
                data.resize((size_t)groups * 8);
                const uint64_t one = 1;  // an implicit count (cw == -1) travels as 1: only "zero or not" matters to sum / min / max
                for (int64_t g = 0; g < groups; g++) {
                    const int wi = part == 0 ? cw : vw;
                    memcpy(&data[(size_t)g * 8], wi >= 0 ? &words[(size_t)g * nw_ + wi] : &one, 8);
                }
                oc.has_nulls = false;
            }
            continue;
        }
        OutColumn& oc = out_cols_[col];
        bool as_double = ag.fn == PA_AGG_AVG || (ag.fn == PA_AGG_SUM && value_is_double);
        const int value_proj = spec_.step == PA_STEP_FINAL ? ag.input_channel + 1 : ag.input_channel;
        const bool real_out = as_double && (ag.input_type == PA_REAL || spec_.proj[value_proj].root_type() == PA_REAL);
        oc.type = real_out ? PA_REAL : (as_double ? PA_DOUBLE : ((ag.fn == PA_AGG_SUM) ? spec_.proj[value_proj].root_type() : PA_BIGINT));
        auto& data = host_cols[col];
        auto& nulls = host_nulls[col];
        nulls.assign(groups ? groups : 1, 0);
        bool any_null = false;
        const int width = type_width(oc.type);
        data.resize((size_t)groups * width);
        for (int64_t g = 0; g < groups; g++) {
            const uint64_t* ww = &words[(size_t)g * nw_];
            int64_t count = cw >= 0 ? (int64_t)ww[cw] : 1;
            uint64_t bits = 0;
            switch (ag.fn) {
                case PA_AGG_COUNT_STAR:
                case PA_AGG_COUNT:
                    bits = (uint64_t)count;
                    break;
                case PA_AGG_SUM:
                    if (count == 0) { nulls[g] = 1; any_null = true; }
                    else bits = ww[vw];
                    break;
                case PA_AGG_AVG:
                    if (count == 0) { nulls[g] = 1; any_null = true; }
                    else {
                        double sum;
                        memcpy(&sum, &ww[vw], 8);
                        double avg = sum / (double)count;  // AverageAggregations.java:68-80
                        memcpy(&bits, &avg, 8);
                    }
                    break;
                default:
                    break;
            }
            if (width == 8) memcpy(&data[(size_t)g * 8], &bits, 8);
            else if (real_out) {  // (float) of the DOUBLE sum / average
                double d;
                memcpy(&d, &bits, 8);
                const float f = (float)d;
                memcpy(&data[(size_t)g * 4], &f, 4);
            }
            else {
                int32_t v = (int32_t)(int64_t)bits;
                memcpy(&data[(size_t)g * 4], &v, 4);
            }
        }
        oc.has_nulls = any_null;
        col++;
    }
    // hand the assembled blocks over: pinned host memory for PA_MEM_HOST consumers, HBM otherwise
    const bool to_device = spec_.output_mem == PA_MEM_DEVICE;
    for (int c = 0; c < ncols; c++) {
        OutColumn& oc = out_cols_[c];
        size_t bytes = host_cols[c].size();
        void* hv = oc.h_values.ensure(bytes ? bytes : 1);
        if (bytes) memcpy(hv, host_cols[c].data(), bytes);
        if (oc.varwidth) {
            void* ho = oc.h_offsets.ensure(host_offsets[c].size() * 4);
            memcpy(ho, host_offsets[c].data(), host_offsets[c].size() * 4);
        }
        if (oc.has_nulls) {
            void* hn = oc.h_nulls.ensure(host_nulls[c].size());
            memcpy(hn, host_nulls[c].data(), host_nulls[c].size());
        }
        oc.host_ready = true;
        if (to_device) {
            oc.values.ensure(bytes ? bytes : 1);
            if (bytes) PA_HIP(hipMemcpyAsync(oc.values.ptr(), hv, bytes, hipMemcpyHostToDevice, s));
            if (oc.varwidth) {
                oc.offsets.ensure(host_offsets[c].size() * 4);
                PA_HIP(hipMemcpyAsync(oc.offsets.ptr(), oc.h_offsets.ptr(), host_offsets[c].size() * 4, hipMemcpyHostToDevice, s));
            }
            if (oc.has_nulls) {
                oc.nulls.ensure(host_nulls[c].size());
                PA_HIP(hipMemcpyAsync(oc.nulls.ptr(), oc.h_nulls.ptr(), host_nulls[c].size(), hipMemcpyHostToDevice, s));
            }
        }
    }
    if (to_device) PA_HIP(hipStreamSynchronize(s));
}

}  // namespace

// Step.PARTIAL with maxPartialMemory: HashAggregationOperator's flush state machine (HashAggregationOperator.java:366-378,
// 476-513) around the aggregation.  "Full" -> no input is taken; get_output builds the partial result of what was
// aggregated so far (closeAggregationBuilder, :532-543) and the next page starts an empty aggregation -- here a fresh
// FusedAggregationOperator over the same plan (its kernels are cached per plan, its buffers pooled); the one that emitted stays
// alive until the next call, as its output page does.
class PartialFlushingAggregationOperator : public pa_operator {
public:
    explicit PartialFlushingAggregationOperator(const pa_fused_aggregation_desc* d) : cur_(std::make_unique<FusedAggregationOperator>(d)) {}
    hipStream_t private_stream() override { return cur_->private_stream(); }
    hipStream_t main_stream() override { return cur_->main_stream(); }
    bool needs_input() override { return !finishing_ && !full_ && cur_->needs_input(); }
    bool is_blocked() override { return !full_ && cur_->is_blocked(); }
    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(!full_, PA_ERR_ILLEGAL_STATE, "Aggregation buffer is full");  // HashAggregationOperator.java:425
        emitted_.reset();
        cur_->add_input(page);
        full_ = cur_->group_bytes() > cur_->spec().max_partial_memory;  // InMemoryHashAggregationBuilder.updateIsFull
    }
    bool get_output(pa_page* out) override
    {
        if (finishing_) return cur_->get_output(out);
        if (!full_) return false;  // only flush when finishing or full (:497-499)
        cur_->finish();
        const bool any = cur_->get_output(out);
        if (hipStream_t own = cur_->private_stream()) PA_HIP(hipStreamSynchronize(own));  // the page of an operator about to be parked
        Spec spec = cur_->spec();
        void* stream = cur_->stream_handle();
        flushes_++;
        emitted_ = std::move(cur_);
        cur_ = std::make_unique<FusedAggregationOperator>(std::move(spec), emitted_->private_stream() ? nullptr : stream);
        full_ = false;
        return any;
    }
    void finish() override
    {
        if (finishing_) return;
        finishing_ = true;
        cur_->finish();
    }
    bool is_finished() override { return finishing_ && cur_->is_finished(); }
    int64_t memory_bytes() override { return cur_->memory_bytes() + (emitted_ ? emitted_->memory_bytes() : 0); }

private:
    std::unique_ptr<FusedAggregationOperator> cur_, emitted_;
    bool finishing_ = false, full_ = false;
    int64_t flushes_ = 0;
};

pa_operator* make_fused_aggregation(const pa_fused_aggregation_desc* desc)
{
    PA_REQUIRE(desc != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    if (desc->aggregation.step == PA_STEP_PARTIAL && desc->aggregation.max_partial_memory > 0 && desc->aggregation.group_by_count > 0) {
        return new PartialFlushingAggregationOperator(desc);
    }
    return new FusedAggregationOperator(desc);
}

pa_operator* make_fused_probe_aggregation(const pa_fused_join_aggregation_desc* desc, pa_lookup_source* bridge)
{
    PA_REQUIRE(desc != nullptr && bridge != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    void* stream = desc->aggregation.stream ? desc->aggregation.stream : (desc->join.stream ? desc->join.stream : desc->filter_project.stream);
    return new FusedAggregationOperator(make_spec(desc->filter_project, desc->aggregation, &desc->join, bridge), stream);
}

// a lookup source with the channels of `build` and nothing built: what the code generators need to know of the build side
void lookup_source_shape_for_desc(const pa_hash_builder_desc* build, pa_lookup_source* bridge)
{
    bridge->impl = std::make_shared<LookupSourceImpl>();
    LookupSourceImpl& ls = *bridge->impl;
    PA_REQUIRE(build->input_channel_count > 0 && build->input_channel_count <= 32, PA_ERR_NOT_SUPPORTED, "1..32 build channels");
    ls.cols.resize(build->input_channel_count);
    for (int c = 0; c < build->input_channel_count; c++) ls.cols[c].type = build->input_types[c];
    for (int i = 0; i < build->join_channel_count; i++) {
        PA_REQUIRE(build->join_channels[i] >= 0 && build->join_channels[i] < build->input_channel_count, PA_ERR_INVALID_ARGUMENT, "join channel out of range");
        ls.join_channels.push_back(build->join_channels[i]);
    }
    for (int i = 0; i < build->output_channel_count; i++) {
        PA_REQUIRE(build->output_channels[i] >= 0 && build->output_channels[i] < build->input_channel_count, PA_ERR_INVALID_ARGUMENT, "output channel out of range");
        ls.output_channels.push_back(build->output_channels[i]);
    }
}

// the probe-stage kernels of a descriptor over a lookup source shaped like `build` (no device needed: nothing is allocated)
std::string fused_join_source_for_desc(const pa_fused_join_aggregation_desc* desc, const pa_hash_builder_desc* build, int variant, std::string* entry)
{
    PA_REQUIRE(desc != nullptr && build != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    pa_lookup_source bridge;
    lookup_source_shape_for_desc(build, &bridge);
    Spec s = make_spec(desc->filter_project, desc->aggregation, &desc->join, &bridge);
    std::vector<ChannelLayout> layout(s.n_in);
    for (int c = 0; c < s.n_in; c++) layout[c].type = s.in_types[c];
    for (int32_t t : s.join->build_types) {
        ChannelLayout cl;
        cl.type = t;
        layout.push_back(cl);
    }
    if (variant < 0) variant = s.group_proj.empty() ? V_GLOBAL : (s.join->brow_group_proj.empty() ? V_GT : V_BROW);
    KernelInfo k = generate(s, layout, variant);
    if (entry) *entry = k.entry;
    return k.source;
}

std::string fused_source_for_desc(const pa_fused_aggregation_desc* desc, int variant, std::string* entry)
{
    PA_REQUIRE(desc != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    Spec s = make_spec(desc);
    std::vector<ChannelLayout> layout(s.n_in);
    for (int c = 0; c < s.n_in; c++) layout[c].type = s.in_types[c];
    if (variant < 0) variant = s.group_proj.empty() ? V_GLOBAL : V_LDS;
    PA_REQUIRE(variant == V_GLOBAL ? s.group_proj.empty() : !s.group_proj.empty(), PA_ERR_INVALID_ARGUMENT, "variant does not match the descriptor");
    KernelInfo k = generate(s, layout, variant);
    if (entry) *entry = k.entry;
    return k.source;
}

std::string fused_source_for_layout(const pa_fused_aggregation_desc* desc, int variant, uint64_t nullable_channels)
{
    PA_REQUIRE(desc != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    Spec s = make_spec(desc);
    std::vector<ChannelLayout> layout(s.n_in);
    for (int c = 0; c < s.n_in; c++) {
        layout[c].type = s.in_types[c];
        layout[c].nullable = c < 64 && ((nullable_channels >> c) & 1ULL) != 0;
    }
    PA_REQUIRE((variant >= V_GLOBAL && variant <= V_LDSP) || variant == V_GLOBAL_R || variant == V_LDS_R, PA_ERR_INVALID_ARGUMENT, "variant out of range");
    PA_REQUIRE(variant == V_GLOBAL || variant == V_GLOBAL_R ? s.group_proj.empty() : !s.group_proj.empty(), PA_ERR_INVALID_ARGUMENT,
               "variant does not match the descriptor");
    return generate(s, layout, variant).source;
}

}  // namespace pa
