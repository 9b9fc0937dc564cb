// op_default_output.cpp -- HashAggregationOperator's default output rows (GROUPING SETS with a global grouping set).
//
// Reference: core/trino-main/src/main/java/io/trino/operator/HashAggregationOperator.java
//   :120-202  factory arguments globalAggregationGroupIds, produceDefaultOutput, groupIdChannel
//   :386      addInput sets inputProcessed
//   :486-492  getOutput: finishing && !inputProcessed && produceDefaultOutput -> finished, getGlobalAggregationOutput()
//   :545-587  one row per global grouping-set id: NULL in every group-by column except groupIdChannel (the id as BIGINT), the
//             $hashvalue of that row when a hash channel was supplied (calculateDefaultOutputHash, :589-600: combine over the
//             group-by columns of NULL_HASH_CODE = 0 and BigintType.hash(id)), then every aggregate's output over NO input:
//             evaluateIntermediate for a PARTIAL step, evaluateFinal otherwise (count 0, sum / avg / min / max NULL)
//
// The aggregates' empty outputs are what the UNGROUPED aggregation of the same aggregate list emits over no input
// (AggregationOperator always emits its one row): that operator is run once, over nothing, and its row is repeated per id -- types,
// state channels of a PARTIAL step and NULL conventions come from the one place that defines them.
#include <memory>
#include <vector>

#include "host_hash.hpp"
#include "operator.hpp"

namespace pa {

namespace {

int width_of(int32_t type)
{
    switch (type) {
        case PA_BIGINT:
        case PA_DOUBLE:
        case PA_DECIMAL: return 8;
        case PA_INTEGER:
        case PA_DATE:
        case PA_REAL: return 4;
        case PA_BOOLEAN: return 1;
        case PA_LONG_DECIMAL: return 16;
        default: return 0;
    }
}

class DefaultOutputAggregation : public pa_operator {
public:
    DefaultOutputAggregation(pa_operator* inner, const pa_hash_aggregation_desc* d, pa_operator* (*make_flat)(const pa_hash_aggregation_desc*))
        : inner_(inner), make_flat_(make_flat), output_mem_(d->output_mem), stream_(static_cast<hipStream_t>(d->stream)),
          group_id_channel_(d->group_id_channel), hash_(d->hash_channel >= 0), step_(d->step)
    {
        PA_REQUIRE(d->global_aggregation_group_id_count >= 0, PA_ERR_INVALID_ARGUMENT, "negative global grouping-set count");
        PA_REQUIRE(d->global_aggregation_group_id_count == 0 || d->global_aggregation_group_ids != nullptr, PA_ERR_INVALID_ARGUMENT,
                   "global_aggregation_group_ids is null");
        ids_.assign(d->global_aggregation_group_ids, d->global_aggregation_group_ids + d->global_aggregation_group_id_count);
        for (int k = 0; k < d->group_by_count; k++) key_types_.push_back(d->input_types[d->group_by_channels[k]]);
        if (!ids_.empty()) {
            // groupIdChannel.get() (HashAggregationOperator.java:560): present whenever there are global grouping sets
            PA_REQUIRE(group_id_channel_ >= 0 && group_id_channel_ < (int)key_types_.size(), PA_ERR_INVALID_ARGUMENT,
                       "group_id_channel must name one of the group-by columns when there are global grouping sets");
            PA_REQUIRE(key_types_[(size_t)group_id_channel_] == PA_BIGINT, PA_ERR_INVALID_ARGUMENT, "the group id column is BIGINT");
        }
        in_types_.assign(d->input_types, d->input_types + d->input_channel_count);
        if (d->input_type_params) in_params_.assign(d->input_type_params, d->input_type_params + d->input_channel_count);
        aggs_.assign(d->aggregates, d->aggregates + d->aggregate_count);
    }
    hipStream_t private_stream() override { return inner_->private_stream(); }
    hipStream_t main_stream() override { return inner_->main_stream(); }
    bool takes_retained() override { return inner_->takes_retained(); }
    bool needs_input() override { return inner_->needs_input(); }
    bool is_blocked() override { return inner_->is_blocked(); }
    int64_t memory_bytes() override { return inner_->memory_bytes(); }
    pa::KernelTimer& kernel_timer() override { return inner_->kernel_timer(); }
    bool set_output_topn(int64_t n, const int32_t* ch, const int32_t* ord, int32_t count) override { return inner_->set_output_topn(n, ch, ord, count); }
    void add_input(const pa_page* page) override
    {
        input_processed_ = true;  // HashAggregationOperator.java:386
        inner_->add_input(page);
    }
    void finish() override
    {
        finishing_ = true;
        inner_->finish();
    }
    bool is_finished() override { return defaults() ? finished_ : inner_->is_finished(); }
    bool get_output(pa_page* out) override
    {
        if (!defaults()) return inner_->get_output(out);
        if (finished_) return false;
        finished_ = true;  // :489-491 -- "global aggregations always generate an output row with the default aggregation output"
        if (ids_.empty()) return false;  // output.isEmpty() -> null (:583-585)
        build(out);
        return true;
    }
    void close() override { inner_->close(); }

private:
    bool defaults() const { return finishing_ && !input_processed_; }

    struct Col {
        std::vector<uint8_t> values, nulls;
        std::vector<int32_t> offsets;
        bool any_null = false;
        int32_t type = PA_BIGINT, encoding = PA_FLAT;
    };

    void build(pa_page* out)
    {
        const int n = (int)ids_.size();
        std::vector<Col> cols;
        for (size_t k = 0; k < key_types_.size(); k++) {
            Col c;
            c.type = key_types_[k];
            if ((int)k == group_id_channel_) {
                c.values.resize((size_t)n * 8);
                for (int i = 0; i < n; i++) {
                    const int64_t v = ids_[(size_t)i];
                    memcpy(&c.values[(size_t)i * 8], &v, 8);
                }
            }
            else {
                c.any_null = true;
                c.nulls.assign((size_t)n, 1);
                if (c.type == PA_VARCHAR) {
                    c.encoding = PA_VARWIDTH;
                    c.offsets.assign((size_t)n + 1, 0);
                    c.values.assign(1, 0);
                }
                else {
                    PA_REQUIRE(width_of(c.type) > 0, PA_ERR_NOT_SUPPORTED, "default output rows: group-by column type not supported");
                    c.values.assign((size_t)n * width_of(c.type), 0);
                }
            }
            cols.push_back(std::move(c));
        }
        if (hash_) {
            Col c;
            c.values.resize((size_t)n * 8);
            for (int i = 0; i < n; i++) {
                // calculateDefaultOutputHash: result = 31 * result + hash(column), NULL_HASH_CODE = 0 (CombineHashFunction.java:26-29)
                int64_t h = 0;
                for (size_t k = 0; k < key_types_.size(); k++) {
                    const int64_t x = (int)k == group_id_channel_ ? host_hash_bigint((int64_t)ids_[(size_t)i]) : 0;
                    h = (int64_t)((uint64_t)31 * (uint64_t)h + (uint64_t)x);
                }
                memcpy(&c.values[(size_t)i * 8], &h, 8);
            }
            cols.push_back(std::move(c));
        }
        // the aggregates over no input: the ungrouped aggregation of the same list, finished at once
        pa_hash_aggregation_desc g;
        memset(&g, 0, sizeof g);
        g.input_channel_count = (int32_t)in_types_.size();
        g.input_types = in_types_.data();
        g.input_type_params = in_params_.empty() ? nullptr : in_params_.data();
        g.hash_channel = -1;
        g.group_id_channel = -1;
        g.step = step_;
        g.aggregate_count = (int32_t)aggs_.size();
        g.aggregates = aggs_.data();
        g.output_mem = PA_MEM_HOST;
        g.stream = stream_;
        g.state_format = PA_STATES_FLAT;
        std::unique_ptr<pa_operator> empty(make_flat_(&g));
        empty->finish();
        pa_page row;
        memset(&row, 0, sizeof row);
        PA_REQUIRE(empty->get_output(&row) && row.position_count == 1, PA_ERR_DEVICE, "internal: the ungrouped aggregation over no input emitted no row");
        for (int a = 0; a < row.channel_count; a++) {
            const pa_column& rc = row.columns[a];
            Col c;
            c.type = rc.type;
            const bool is_null = rc.nulls != nullptr && rc.nulls[0] != 0;
            c.any_null = is_null;
            if (is_null) c.nulls.assign((size_t)n, 1);
            if (rc.encoding == PA_VARWIDTH) {
                c.encoding = PA_VARWIDTH;
                const int32_t len = is_null ? 0 : rc.offsets[1] - rc.offsets[0];
                c.offsets.resize((size_t)n + 1);
                for (int i = 0; i <= n; i++) c.offsets[(size_t)i] = i * len;
                c.values.assign((size_t)std::max(1, n * len), 0);
                for (int i = 0; i < n && len > 0; i++) memcpy(&c.values[(size_t)i * len], static_cast<const uint8_t*>(rc.values) + rc.offsets[0], (size_t)len);
            }
            else {
                PA_REQUIRE(rc.encoding == PA_FLAT && width_of(rc.type) > 0, PA_ERR_NOT_SUPPORTED, "default output rows: aggregate output block not supported");
                const int w = width_of(rc.type);
                c.values.assign((size_t)n * w, 0);
                if (!is_null) {
                    for (int i = 0; i < n; i++) memcpy(&c.values[(size_t)i * w], rc.values, (size_t)w);
                }
            }
            cols.push_back(std::move(c));
        }
        empty->close();
        empty.reset();

        // hand over: pinned host memory, or HBM for PA_MEM_DEVICE consumers
        const bool dev = output_mem_ == PA_MEM_DEVICE;
        hipStream_t s = stream_ ? stream_ : inner_->main_stream();
        out_cols_.assign(cols.size(), pa_column{});
        host_.clear();
        dev_.clear();
        host_.resize(cols.size() * 3);
        dev_.resize(cols.size() * 3);
        auto place = [&](size_t slot, const void* src, size_t bytes) -> const void* {
            void* h = host_[slot].ensure(bytes ? bytes : 1);
            if (bytes) memcpy(h, src, bytes);
            if (!dev) return h;
            void* d = dev_[slot].ensure(bytes ? bytes : 1);
            if (bytes) PA_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
            return d;
        };
        for (size_t c = 0; c < cols.size(); c++) {
            pa_column& oc = out_cols_[c];
            oc.type = cols[c].type;
            oc.encoding = cols[c].encoding;
            oc.values = place(c * 3, cols[c].values.data(), cols[c].values.size());
            if (cols[c].encoding == PA_VARWIDTH) oc.offsets = static_cast<const int32_t*>(place(c * 3 + 1, cols[c].offsets.data(), cols[c].offsets.size() * 4));
            if (cols[c].any_null) oc.nulls = static_cast<const uint8_t*>(place(c * 3 + 2, cols[c].nulls.data(), cols[c].nulls.size()));
        }
        if (dev) PA_HIP(hipStreamSynchronize(s));
        out->position_count = n;
        out->channel_count = (int32_t)out_cols_.size();
        out->columns = out_cols_.data();
        out->mem = output_mem_;
        out->flags = 0;
    }

    std::unique_ptr<pa_operator> inner_;
    pa_operator* (*make_flat_)(const pa_hash_aggregation_desc*);
    int32_t output_mem_;
    hipStream_t stream_;
    int group_id_channel_;
    bool hash_;
    int32_t step_;
    std::vector<int32_t> ids_, key_types_, in_types_, in_params_;
    std::vector<pa_aggregate> aggs_;
    bool input_processed_ = false, finishing_ = false, finished_ = false;
    std::vector<pa_column> out_cols_;
    std::vector<PinnedBuf> host_;
    std::vector<DevBuf> dev_;
};

}  // namespace

pa_operator* make_default_output_aggregation(pa_operator* inner, const pa_hash_aggregation_desc* desc, pa_operator* (*make_flat)(const pa_hash_aggregation_desc*))
{
    std::unique_ptr<pa_operator> guard(inner);
    auto* op = new DefaultOutputAggregation(guard.get(), desc, make_flat);
    guard.release();
    return op;
}

}  // namespace pa
