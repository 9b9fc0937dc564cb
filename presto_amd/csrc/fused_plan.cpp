// fused_plan.cpp -- descriptors of the C ABI -> the fused operator's plan (see fused_plan.hpp).
#include <algorithm>
#include <set>

#include "fused_plan.hpp"

namespace pa {
namespace fused {

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
Spec make_spec(const pa_filter_project_desc& fp, const pa_hash_aggregation_desc& ag, const pa_lookup_join_desc* jd, pa_lookup_source* bridge)
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
        // unbounded strings go through their rank in the channel's dictionary (Spec::ranked, finalize_spec) -- for that the
        // argument has to be a channel of the page itself, not a computed string
        auto varchar_channel = [&](int proj) {
            const OwnedExpr& pe = s.proj[proj];
            if (!pe.is_input_ref()) return false;
            const int ch = pe.node(pe.root).channel;
            return ch >= 0 && ch < s.n_in;
        };
        if ((a.fn == PA_AGG_MIN || a.fn == PA_AGG_MAX) && ag.step != PA_STEP_FINAL && a.input_channel >= 0 && s.proj[a.input_channel].root_type() == PA_VARCHAR) {
            PA_REQUIRE(varchar_channel(a.input_channel), PA_ERR_NOT_SUPPORTED, "min/max over VARCHAR: the argument must be an input channel");
        }
        if (s.step == PA_STEP_FINAL) {
            // intermediate input: [count BIGINT] for count / count(*), [count BIGINT, sum] for sum / avg
            PA_REQUIRE(a.input_channel >= 0 && s.proj[a.input_channel].root_type() == PA_BIGINT, PA_ERR_INVALID_ARGUMENT,
                       "FINAL step: the aggregate's first state channel must be the BIGINT count");
            PA_REQUIRE(a.mask_channel < 0, PA_ERR_INVALID_ARGUMENT, "FINAL step takes no mask");
            if (a.fn == PA_AGG_MIN || a.fn == PA_AGG_MAX) {
                PA_REQUIRE(a.input_channel + 1 < fp.projection_count, PA_ERR_INVALID_ARGUMENT, "FINAL step: missing value state channel");
                PA_REQUIRE(s.proj[a.input_channel + 1].root_type() != PA_VARCHAR || varchar_channel(a.input_channel + 1), PA_ERR_NOT_SUPPORTED,
                           "min/max over VARCHAR: the value state must be an input channel");
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
    // min / max over VARCHAR channels without a bound of at most 7 bytes: by rank (Spec::ranked).  Nothing else may read the
    // channel -- a filter, an expression or a group key would need the strings themselves -- except count(), which only looks at
    // the NULL flag
    s.ranked.assign(s.n_in, false);
    for (const pa_aggregate& a : s.aggs) {
        if (a.fn != PA_AGG_MIN && a.fn != PA_AGG_MAX) continue;
        const int vp = s.step == PA_STEP_FINAL ? a.input_channel + 1 : a.input_channel;
        const OwnedExpr& pe = s.proj[vp];
        if (pe.root_type() != PA_VARCHAR || !pe.is_input_ref()) continue;
        const int c = pe.node(pe.root).channel;
        if (c < 0 || c >= s.n_in || (s.in_params[c] >= 1 && s.in_params[c] <= 7)) continue;
        s.ranked[c] = true;
    }
    for (int c = 0; c < s.n_in; c++) {
        if (!s.ranked[c]) continue;
        PA_REQUIRE(!s.join, PA_ERR_NOT_SUPPORTED, "min/max over long VARCHAR behind the fused probe is not on the device path");
        bool only_minmax = !s.interned[c] && !(s.has_filter && [&] { std::set<int32_t> f; s.filter.collect_channels(&f); return f.count(c) != 0; }());
        for (size_t q = 0; q < s.proj.size() && only_minmax; q++) {
            std::set<int32_t> ch;
            s.proj[q].collect_channels(&ch);
            if (!ch.count(c)) continue;
            only_minmax = s.proj[q].is_input_ref();
            for (int j : s.group_proj) only_minmax = only_minmax && j != (int)q;
            for (const pa_aggregate& a : s.aggs) {
                if (a.mask_channel == (int32_t)q) only_minmax = false;
                const int vp = s.step == PA_STEP_FINAL ? a.input_channel + 1 : a.input_channel;
                if (vp == (int)q && a.fn != PA_AGG_MIN && a.fn != PA_AGG_MAX && a.fn != PA_AGG_COUNT) only_minmax = false;
            }
        }
        PA_REQUIRE(only_minmax, PA_ERR_NOT_SUPPORTED, "min/max over a VARCHAR channel longer than 7 bytes that other expressions read is not on the device path");
        s.in_types[c] = PA_BIGINT;
        for (OwnedExpr& pe : s.proj) {
            if (pe.is_input_ref() && pe.node(pe.root).channel == c) pe.nodes[pe.root].type = PA_BIGINT;
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
        if (s.derived(c) || s.in_types[c] == PA_VARCHAR) s.lazy_channel[c] = false;  // strings are handed to the row function whole
    }
}

}  // namespace fused
}  // namespace pa
