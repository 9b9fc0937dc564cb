// op_fused_intern.cpp -- FusedAggregationOperator (op_fused.hpp): VARCHAR group keys as ids of per-channel dictionaries, and the ranks
// that stand in for VARCHAR / long-DECIMAL inputs of min / max.
#include "op_fused.hpp"

namespace pa {
namespace fused_op {

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
        const int64_t bytes_hint = (size_t)c < var_bytes_hint_.size() && spec_.in_types[c] == PA_VARCHAR ? var_bytes_hint_[c] : -1;
        const int32_t* ids = interners_[c]->intern(col.values, col.offsets, col.nulls, dp.n, s, bytes_hint);
        col.type = PA_INTEGER;
        col.varwidth = false;
        col.values = ids;
        col.offsets = nullptr;
    }
}

// min / max over VARCHAR channels without a short bound (Spec::ranked; MaxAggregationFunction / MinAggregationFunction over a
// Slice state compare with VarcharType's operator: unsigned bytes, then the length).  The page's strings are interned; the kernels
// get, in the channel's place, a BIGINT column (rank of the string among all strings of the dictionary) << 32 | id, whose integer
// order is the strings' order -- so every tier's integer min / max serves.  The ranks live on the host: the new strings of a page
// are fetched, sorted into the order of the ones before, and the ranks go back to the device; what the operator has accumulated
// under the old ranks (the HBM table's words, or the ungrouped state) is brought up to date in place -- the id in the low half
// of a word stays, its rank is looked up again.  A page without new strings costs the interning and one pass over the ids.
void FusedAggregationOperator::rank_values(DevPage& dp, hipStream_t s)
{
    for (int c = 0; c < spec_.n_in; c++) {
        if (!spec_.ranked[c] || !spec_.used_channel[c]) continue;
        DevColumn& col = dp.cols[c];
        PA_REQUIRE(col.type == PA_VARCHAR && col.varwidth && col.offsets != nullptr, PA_ERR_INVALID_ARGUMENT,
                   "page block type does not match the declared input type");
        if (interners_.empty()) interners_.resize(spec_.n_in);
        if (!interners_[c]) interners_[c] = std::make_unique<StringInterner>();
        if (ranked_.empty()) ranked_.resize(spec_.n_in);
        if (!ranked_[c]) ranked_[c] = std::make_unique<RankedChannel>();
        RankedChannel& rc = *ranked_[c];
        const int32_t* ids = interners_[c]->intern(col.values, col.offsets, col.nulls, dp.n, s);  // (leaves the stream drained)
        const uint32_t known = (uint32_t)rc.strings.size(), now = interners_[c]->size();
        if (now > known) {
            interners_[c]->fetch_strings(known, &rc.strings, s);
            auto before = [&rc](uint32_t a, uint32_t b) {
                const std::string &x = rc.strings[a], &y = rc.strings[b];
                const int cmp = memcmp(x.data(), y.data(), std::min(x.size(), y.size()));
                return cmp != 0 ? cmp < 0 : x.size() < y.size();
            };
            const size_t mid = rc.order.size();
            for (uint32_t id = known; id < now; id++) rc.order.push_back(id);
            std::sort(rc.order.begin() + (ptrdiff_t)mid, rc.order.end(), before);
            std::inplace_merge(rc.order.begin(), rc.order.begin() + (ptrdiff_t)mid, rc.order.end(), before);
            std::vector<uint32_t> ranks(now);
            for (uint32_t r = 0; r < now; r++) ranks[rc.order[r]] = r;
            uint32_t* dev = static_cast<uint32_t*>(rc.ranks.ensure((size_t)now * 4));
            PA_HIP(hipMemcpyAsync(dev, ranks.data(), (size_t)now * 4, hipMemcpyHostToDevice, s));
            PA_HIP(hipStreamSynchronize(s));
            if (known > 0) rerank_words(c, dev, s);
        }
        int64_t* image = static_cast<int64_t*>(rc.image.ensure((size_t)std::max<int64_t>(dp.n, 1) * 8));
        launch_rank_image(ids, col.nulls, rc.ranks.as<uint32_t>(), dp.n, image, s);
        col.type = PA_BIGINT;
        col.varwidth = false;
        col.values = image;
        col.offsets = nullptr;
    }
}

void FusedAggregationOperator::rerank_words(int c, const uint32_t* ranks, hipStream_t s)
{
    if (last_info_ == nullptr) return;  // nothing was accumulated yet
    std::set<std::pair<int, bool>> done;
    for (size_t k = 0; k < spec_.aggs.size(); k++) {
        const pa_aggregate& ag = spec_.aggs[k];
        if (ag.fn != PA_AGG_MIN && ag.fn != PA_AGG_MAX) continue;
        if (ranked_channel(spec_.step == PA_STEP_FINAL ? ag.input_channel + 1 : ag.input_channel) != c) continue;
        const int vw = last_info_->agg_words[k].second;
        const bool is_min = ag.fn == PA_AGG_MIN;
        if (vw < 0 || !done.insert({vw, is_min}).second) continue;
        if (grouped_) {
            if (gt_cap_ == 0 || gt_words_.ptr() == nullptr) continue;
            for (uint32_t r = 0; r < gt_rep_; r++) {
                launch_rerank_words(gt_words_.as<uint64_t>() + ((uint64_t)r * nw_ + (uint64_t)vw) * gt_cap_, (int64_t)gt_cap_, is_min, ranks, s);
            }
        }
        else if (state_.ptr() != nullptr) {
            launch_rerank_words(state_.as<uint64_t>() + vw, 1, is_min, ranks, s);
        }
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
        if (oc.varwidth) continue;   // build_output wrote the strings themselves (a few groups, for the host)
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

}  // namespace fused_op
}  // namespace pa
