// op_fused_output.cpp -- FusedAggregationOperator (op_fused.hpp): the result page -- groups emitted on the device (k_gt_emit), or a few
// groups assembled on the host.
#include "op_fused.hpp"

namespace pa {
namespace fused_op {

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
    if (spec_.any_ranked()) return false;  // (min / max by rank: the strings come from the host's copy of the dictionary)
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
    // The consumer is a TopN: a bound on its FIRST sort channel is drawn from a sample of the table (the order statistic of 2^14
    // evenly spaced slots that lies above the n-th best group with overwhelming probability, as TopNOperator does for a page,
    // op_topn.cpp) and only the groups not beyond it are emitted -- thousands instead of millions; every group that can be among
    // the n best is among them (ties on the first channel included).  Fewer than n groups under the bound (the sample was
    // unlucky): everything is emitted after all.
    DevBuf topn_keys;
    bool bounded = false;
    if (topn_hint_.n > 0 && !partial && groups >= (int64_t)1 << 16 && groups >= 64 * topn_hint_.n) {
        const int c0 = topn_hint_.channels[0];
        const int64_t cap = (int64_t)gt_cap_;
        constexpr int64_t kSample = (int64_t)1 << 14;
        const double j = (double)topn_hint_.n * (double)kSample / (double)cap;
        const int64_t rank = (int64_t)std::ceil(j + 8.0 + 4.0 * std::sqrt(j));
        if (c0 >= 0 && c0 < n && a.col[c0].kind != GT_EMIT_HASH && cap >= 4 * kSample && rank <= kSample / 8) {
            // (the keys of the kSample sampled slots only -- slot j * (cap / kSample), the sample launch_topn_sample_bound would draw from
            // keys of every slot -- and k_gt_emit computes a slot's key again from the words it reads anyway: until round 4 the keys
            // of all `cap` slots were written out and read back twice, 0.35 GB for Q3's 14.6 M build rows)
            uint64_t* keys = static_cast<uint64_t*>(topn_keys.ensure((size_t)kSample * 8 + 64));
            uint64_t* bound = keys + kSample;
            launch_gt_emit_keys_strided(a, c0, topn_hint_.orders[0], cap / kSample, kSample, keys, s);
            launch_topn_sample_bound(PA_TOPN_KEYS, keys, nullptr, nullptr, kSample, 1, kSample, rank, nullptr, bound, s);
            a.filter_bound = bound;
            a.filter_col = c0;
            a.filter_order = topn_hint_.orders[0];
            bounded = true;
        }
    }
    uint32_t flags[GT_EMIT_MAX_COLS];
    for (;;) {
        PA_HIP(hipMemsetAsync(ctl_ + 7, 0, 4, s));
        PA_HIP(hipMemsetAsync(a.null_flags, 0, GT_EMIT_MAX_COLS * 4, s));
        launch_gt_emit(a, s);
        PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 32, hipMemcpyDeviceToHost, s));
        read_back(flags, a.null_flags, sizeof(flags), s);   // (one wait for both)
        if (!bounded || (int64_t)(uint32_t)h_ctl_[7] >= topn_hint_.n) break;
        bounded = false;
        a.filter_bound = nullptr;
    }
    if (keys_from_build_columns || bounded) groups = (int64_t)(uint32_t)h_ctl_[7];
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
    HostTraceScope trace("  fused.build_output");
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
    const bool sub_tables = grouped_ && sub_parts_ > 0;
    const int32_t* per_part = nullptr;
    if (sub_tables) {
        // (the partitions' group counts, the table's and the error word in ONE round trip)
        per_part = static_cast<const int32_t*>(h_parts_.ensure((size_t)sub_parts_ * 4));
        PA_HIP(hipMemcpyAsync(h_parts_.ptr(), sub_count_.ptr(), (size_t)sub_parts_ * 4, hipMemcpyDeviceToHost, s));
        read_group_counts(s);
    }
    else {
        PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 32, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
    }
    raise_if(h_ctl_[0]);
    if (sub_tables) {
        // Partition-owned tables.  Their HBM form IS a group table of sub_parts_ * lc slots (same arrays, same layouts; only the
        // probe sequence differs, and nothing probes any more).  When the HBM table proper holds no group -- nothing fell
        // through, no other tier ran -- they simply become the table; else their groups are folded into it, one upsert per group.
        uint64_t total = 0;
        for (int p = 0; p < sub_parts_; p++) total += (uint64_t)per_part[p];
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
        // a few groups of an interned key for the host: the strings come from a host copy of the (small) dictionary -- one round trip for
        // the ids it does not hold yet, instead of uploading the id column, decoding it over there and fetching offsets, then bytes
        const std::vector<std::string>* host_dict = nullptr;
        if (ic >= 0 && spec_.output_mem != PA_MEM_DEVICE && groups > 0 && groups <= kHostDecodeGroups && !interners_.empty() && interners_[ic] &&
            interners_[ic]->size() <= kHostDecodeIds) {
            if (host_dict_.empty()) host_dict_.resize(spec_.n_in);
            interners_[ic]->fetch_strings((uint32_t)host_dict_[ic].size(), &host_dict_[ic], s);
            host_dict = &host_dict_[ic];
        }
        oc.type = host_dict ? PA_VARCHAR : kp.type;
        oc.varwidth = oc.type == PA_VARCHAR;
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
                    if (host_dict) {
                        if (!is_null) {
                            PA_REQUIRE((size_t)(uint32_t)v < host_dict->size(), PA_ERR_DEVICE, "internal: key id names no string of the dictionary");
                            const std::string& str = (*host_dict)[(uint32_t)v];
                            data.insert(data.end(), str.begin(), str.end());
                        }
                        offs.push_back((int32_t)data.size());
                    }
                    else data.insert(data.end(), (uint8_t*)&v, (uint8_t*)&v + 4);
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
            if (const int rch = ranked_channel(value_proj); rch >= 0) {  // image = rank << 32 | id of the string (rank_values)
                oc.type = PA_VARCHAR;
                oc.varwidth = true;
                auto& offs = host_offsets[col];
                offs.assign(1, 0);
                const RankedChannel* rc = ranked_.empty() ? nullptr : ranked_[(size_t)rch].get();
                for (int64_t g = 0; g < groups; g++) {
                    const uint64_t* ww = &words[(size_t)g * nw_];
                    const uint64_t img = ag.fn == PA_AGG_MIN ? ~ww[vw] : ww[vw];
                    if ((cw >= 0 && ww[cw] == 0) || ww[vw] == 0 || rc == nullptr) {
                        nulls[g] = 1;
                        any_null = true;
                    }
                    else {
                        const uint32_t id = (uint32_t)(img ^ 0x8000000000000000ULL);
                        PA_REQUIRE(id < rc->strings.size(), PA_ERR_DEVICE, "internal: min/max word names no string of the dictionary");
                        data.insert(data.end(), rc->strings[id].begin(), rc->strings[id].end());
                    }
                    offs.push_back((int32_t)data.size());
                }
                oc.has_nulls = any_null;
                col++;
                continue;
            }
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

}  // namespace fused_op
}  // namespace pa
