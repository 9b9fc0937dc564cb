// op_states.cpp -- intermediate aggregation states in the reference's own format (pa_state_format PA_STATES_REFERENCE).
//
// A Step.PARTIAL HashAggregationOperator of the reference emits, per aggregate, ONE block typed by the aggregate's
// AccumulatorStateSerializer (…/operator/aggregation/state/StateCompiler.java:127-185: a state with one field is that field's
// type, a state with several is an anonymous ROW of the fields sorted by name, :586-625), and a Step.FINAL operator takes the
// same.  The device operators keep states as plain [count] / [count, value] channels; this adapter sits at the boundary of such
// an operator and re-shapes pages -- RowBlocks are (fields[], rowIsNull) and re-wiring them is pointer work; only three things
// touch data: the always-true null flags of LongDoubleState / LongLongState (TwoNullableValueState's initial values), the
// INTEGER / DATE <-> BIGINT width of a min / max state (NullableLongState holds a long), and the count word a min / max state
// does not carry (value IS NULL <=> nothing seen).
#include "operator.hpp"

namespace pa {

// static_kernels.hip
void launch_widen_i32_i64(const int32_t* in, int64_t n, int64_t* out, hipStream_t s);
void launch_narrow_i64_i32(const int64_t* in, int64_t n, int32_t* out, hipStream_t s);
void launch_count_from_nulls(const uint8_t* nulls, int64_t n, int64_t* out, hipStream_t s);

namespace {

struct AggShape {
    int32_t fn;
    int32_t value_type;  // flat format: type of the value channel (sum: DOUBLE / BIGINT; min / max: the input type)
};

// channels of one aggregate in the flat format
int flat_width(int32_t fn) { return (fn == PA_AGG_COUNT || fn == PA_AGG_COUNT_STAR) ? 1 : 2; }

class StateFormatAdapter : public pa_operator {
public:
    // inner: an operator built with PA_STATES_FLAT on `stream`; leading = key (+ $hashvalue) channels in front of the states
    StateFormatAdapter(std::unique_ptr<pa_operator> inner, int step, int leading, std::vector<AggShape> aggs, int32_t /*mem*/, void* stream)
        : inner_(std::move(inner)), step_(step), leading_(leading), aggs_(std::move(aggs)), stream_(static_cast<hipStream_t>(stream))
    {
    }
    hipStream_t private_stream() override { return inner_->private_stream(); }
    hipStream_t main_stream() override { return inner_->main_stream(); }
    bool needs_input() override { return inner_->needs_input(); }
    bool is_blocked() override { return inner_->is_blocked(); }
    void finish() override { inner_->finish(); }
    bool is_finished() override { return inner_->is_finished(); }
    int64_t memory_bytes() override { return inner_->memory_bytes(); }

    void add_input(const pa_page* page) override
    {
        if (step_ != PA_STEP_FINAL) {
            inner_->add_input(page);
            return;
        }
        // reference format -> [count] / [count, value] channels
        PA_REQUIRE(page != nullptr && page->channel_count == leading_ + (int32_t)aggs_.size(), PA_ERR_INVALID_ARGUMENT,
                   "FINAL step (reference state format): one channel per group key, ($hashvalue) and aggregate");
        const int64_t n = page->position_count;
        const bool dev = page->mem == PA_MEM_DEVICE;
        std::vector<pa_column> flat;
        for (int c = 0; c < leading_; c++) flat.push_back(page->columns[c]);
        size_t tmp = 0;
        for (size_t k = 0; k < aggs_.size(); k++) {
            const pa_column& in = page->columns[leading_ + k];
            const AggShape& a = aggs_[k];
            if (a.fn == PA_AGG_COUNT || a.fn == PA_AGG_COUNT_STAR) {
                PA_REQUIRE(in.encoding == PA_FLAT && in.type == PA_BIGINT, PA_ERR_INVALID_ARGUMENT, "count state must be a BIGINT block");
                flat.push_back(in);
                continue;
            }
            if (a.fn == PA_AGG_SUM || a.fn == PA_AGG_AVG) {
                const int want = a.fn == PA_AGG_SUM ? 4 : 2;
                PA_REQUIRE(in.encoding == PA_ROW_FIELDS && in.dictionary != nullptr && in.dictionary_size == want, PA_ERR_INVALID_ARGUMENT,
                           "sum / avg state must be a ROW block of the reference's state fields");
                // sum: ROW(first = count, firstNull, second = sum, secondNull);  avg: ROW(double = sum, long = count)
                const pa_column& cnt = a.fn == PA_AGG_SUM ? in.dictionary[0] : in.dictionary[1];
                const pa_column& val = a.fn == PA_AGG_SUM ? in.dictionary[2] : in.dictionary[0];
                PA_REQUIRE(cnt.type == PA_BIGINT && val.type == a.value_type && cnt.encoding == PA_FLAT && val.encoding == PA_FLAT, PA_ERR_INVALID_ARGUMENT,
                           "sum / avg state: field types do not match the aggregate");
                pa_column c0 = cnt, c1 = val;
                c0.nulls = c1.nulls = nullptr;  // state fields are never NULL (a NULL row cannot occur: rowIsNull is ignored)
                flat.push_back(c0);
                flat.push_back(c1);
                continue;
            }
            // min / max: NullableLong / Double / BooleanState -> [count = value IS NOT NULL, value of the input type]
            const bool narrow = a.value_type == PA_INTEGER || a.value_type == PA_DATE;
            PA_REQUIRE(in.encoding == PA_FLAT && in.type == (narrow ? (int32_t)PA_BIGINT : a.value_type), PA_ERR_INVALID_ARGUMENT,
                       "min / max state: BIGINT for integer types, else the value's type");
            pa_column cnt{}, val = in;
            cnt.type = PA_BIGINT;
            cnt.encoding = PA_FLAT;
            if (dev) {
                int64_t* c = static_cast<int64_t*>(dtmp(tmp++, (size_t)n * 8));
                launch_count_from_nulls(in.nulls, n, c, stream_);
                cnt.values = c;
                if (narrow) {
                    int32_t* v = static_cast<int32_t*>(dtmp(tmp++, (size_t)n * 4));
                    launch_narrow_i64_i32(static_cast<const int64_t*>(in.values), n, v, stream_);
                    val.values = v;
                    val.type = a.value_type;
                }
            }
            else {
                int64_t* c = static_cast<int64_t*>(htmp(tmp++, (size_t)n * 8));
                for (int64_t i = 0; i < n; i++) c[i] = (in.nulls && in.nulls[i]) ? 0 : 1;
                cnt.values = c;
                if (narrow) {
                    int32_t* v = static_cast<int32_t*>(htmp(tmp++, (size_t)n * 4));
                    const int64_t* w = static_cast<const int64_t*>(in.values);
                    for (int64_t i = 0; i < n; i++) v[i] = (int32_t)w[i];
                    val.values = v;
                    val.type = a.value_type;
                }
            }
            flat.push_back(cnt);
            flat.push_back(val);
        }
        pa_page fp = *page;
        fp.channel_count = (int32_t)flat.size();
        fp.columns = flat.data();
        fp.flags = 0;  // the synthesised channels live in this adapter's scratch: not stable
        inner_->add_input(&fp);
    }

    bool get_output(pa_page* out) override
    {
        pa_page flat{};
        if (!inner_->get_output(&flat)) return false;
        if (step_ != PA_STEP_PARTIAL) {
            *out = flat;
            return true;
        }
        // [count] / [count, value] channels -> reference format
        const int64_t n = flat.position_count;
        const bool dev = flat.mem == PA_MEM_DEVICE;
        int expect = leading_;
        for (const AggShape& a : aggs_) expect += flat_width(a.fn);
        PA_REQUIRE(flat.channel_count == expect, PA_ERR_DEVICE, "internal: unexpected PARTIAL output shape");
        cols_.clear();
        fields_.clear();
        fields_.reserve(aggs_.size());
        for (int c = 0; c < leading_; c++) cols_.push_back(flat.columns[c]);
        const uint8_t* all_true = nullptr;
        size_t tmp = 0;
        int at = leading_;
        for (const AggShape& a : aggs_) {
            const pa_column& cnt = flat.columns[at];
            if (a.fn == PA_AGG_COUNT || a.fn == PA_AGG_COUNT_STAR) {
                cols_.push_back(cnt);
                at += 1;
                continue;
            }
            pa_column val = flat.columns[at + 1];
            at += 2;
            if (a.fn == PA_AGG_SUM || a.fn == PA_AGG_AVG) {
                pa_column c0 = cnt;
                c0.nulls = nullptr;
                val.nulls = nullptr;
                std::vector<pa_column> f;
                if (a.fn == PA_AGG_SUM) {
                    if (!all_true) all_true = true_flags(n, dev);
                    pa_column t{};
                    t.type = PA_BOOLEAN;
                    t.encoding = PA_FLAT;
                    t.values = all_true;
                    f = {c0, t, val, t};  // (first, firstNull, second, secondNull)
                }
                else {
                    f = {val, c0};        // (double, long)
                }
                fields_.push_back(std::move(f));
                pa_column row{};
                row.type = PA_ROW;
                row.encoding = PA_ROW_FIELDS;
                row.dictionary = fields_.back().data();
                row.dictionary_size = (int32_t)fields_.back().size();
                cols_.push_back(row);
                continue;
            }
            // min / max: the value alone, NULL while nothing was seen; a long for the integer types
            if (a.value_type == PA_INTEGER || a.value_type == PA_DATE) {
                if (dev) {
                    int64_t* w = static_cast<int64_t*>(dtmp(tmp++, (size_t)std::max<int64_t>(n, 1) * 8));
                    launch_widen_i32_i64(static_cast<const int32_t*>(val.values), n, w, stream_);
                    val.values = w;
                }
                else {
                    int64_t* w = static_cast<int64_t*>(htmp(tmp++, (size_t)std::max<int64_t>(n, 1) * 8));
                    const int32_t* v = static_cast<const int32_t*>(val.values);
                    for (int64_t i = 0; i < n; i++) w[i] = v[i];
                    val.values = w;
                }
                val.type = PA_BIGINT;
            }
            cols_.push_back(val);
        }
        // (device pages: valid in the stream's order; when the library owns the stream the C-ABI wrapper completes them)
        *out = flat;
        out->channel_count = (int32_t)cols_.size();
        out->columns = cols_.data();
        return true;
    }

private:
    void* dtmp(size_t i, size_t bytes)
    {
        if (i >= dtmp_.size()) dtmp_.resize(i + 1);
        return dtmp_[i].ensure(bytes ? bytes : 1);
    }
    void* htmp(size_t i, size_t bytes)
    {
        if (i >= htmp_.size()) htmp_.resize(i + 1);
        return htmp_[i].ensure(bytes ? bytes : 1);
    }
    const uint8_t* true_flags(int64_t n, bool dev)
    {
        const size_t bytes = (size_t)std::max<int64_t>(n, 1);
        if (dev) {
            void* p = d_true_.ensure(bytes);
            PA_HIP(hipMemsetAsync(p, 1, bytes, stream_));
            return static_cast<const uint8_t*>(p);
        }
        void* p = h_true_.ensure(bytes);
        memset(p, 1, bytes);
        return static_cast<const uint8_t*>(p);
    }

    std::unique_ptr<pa_operator> inner_;
    int step_, leading_;
    std::vector<AggShape> aggs_;
    hipStream_t stream_;
    std::vector<pa_column> cols_;
    std::vector<std::vector<pa_column>> fields_;
    std::vector<DevBuf> dtmp_;
    std::vector<PinnedBuf> htmp_;
    DevBuf d_true_;
    PinnedBuf h_true_;
};

}  // namespace

// HashAggregation / Aggregation operator over plain input channels (no fused filter / projection) with the reference's state
// format at its PARTIAL output or FINAL input: the flat-format operator inside an adapter.
pa_operator* make_aggregation_with_reference_states(const pa_hash_aggregation_desc* agg, pa_operator* (*make_flat)(const pa_hash_aggregation_desc*))
{
    PA_REQUIRE(agg->step == PA_STEP_PARTIAL || agg->step == PA_STEP_FINAL, PA_ERR_INVALID_ARGUMENT, "the state format matters for PARTIAL and FINAL steps only");
    const int leading = agg->group_by_count + ((agg->hash_channel >= 0 && agg->group_by_count > 0) ? 1 : 0);
    std::vector<AggShape> shapes;
    pa_hash_aggregation_desc flat = *agg;
    flat.state_format = PA_STATES_FLAT;
    std::vector<int32_t> types;
    std::vector<int32_t> group_by;
    std::vector<pa_aggregate> aggs(agg->aggregates, agg->aggregates + agg->aggregate_count);
    for (const pa_aggregate& a : aggs) {
        // (RealSumAggregation keeps a NullableDoubleState, RealAverageAggregation a LongState + DoubleState pair: other serialized
        // forms than the DOUBLE aggregates' -- not translated here)
        PA_REQUIRE(a.input_type != PA_REAL, PA_ERR_NOT_SUPPORTED, "reference-format states of aggregates over REAL are not on the device path");
    }
    if (agg->step == PA_STEP_PARTIAL) {
        // input is raw rows: nothing to translate; the shapes come from the aggregates' input types
        for (const pa_aggregate& a : aggs) {
            int32_t vt = a.input_type;
            if (a.fn == PA_AGG_AVG) vt = PA_DOUBLE;
            else if (a.fn == PA_AGG_SUM) vt = a.input_type == PA_DOUBLE ? PA_DOUBLE : PA_BIGINT;
            shapes.push_back(AggShape{a.fn, vt});
        }
    }
    else {
        // FINAL: the input page is [keys..., ($hashvalue), one state channel per aggregate]; inside, every sum / avg / min / max
        // state is two channels.  pa_aggregate.input_type = type of the value (sum: DOUBLE / BIGINT; min / max: the input type).
        PA_REQUIRE(agg->input_channel_count == leading + agg->aggregate_count, PA_ERR_INVALID_ARGUMENT,
                   "FINAL step (reference state format): input channels = group keys, ($hashvalue), one channel per aggregate");
        for (int c = 0; c < leading; c++) types.push_back(agg->input_types[c]);
        for (int32_t k = 0; k < agg->aggregate_count; k++) {
            pa_aggregate& a = aggs[k];
            PA_REQUIRE(a.input_channel == leading + k, PA_ERR_INVALID_ARGUMENT, "FINAL step (reference state format): aggregate k reads channel leading + k");
            int32_t vt = a.input_type;
            if (a.fn == PA_AGG_AVG) vt = PA_DOUBLE;
            shapes.push_back(AggShape{a.fn, vt});
            a.input_channel = (int32_t)types.size();
            types.push_back(PA_BIGINT);
            if (flat_width(a.fn) == 2) types.push_back(vt);
        }
        flat.input_channel_count = (int32_t)types.size();
        flat.input_types = types.data();
        flat.input_type_params = nullptr;
        flat.aggregates = aggs.data();
    }
    // the adapter and the operator share one stream: the adapter's scratch is reused in stream order
    void* stream = agg->stream;
    std::unique_ptr<Stream> holder;
    if (!stream) {
        holder = std::make_unique<Stream>(nullptr);
        stream = holder->get();
    }
    flat.stream = stream;
    std::unique_ptr<pa_operator> inner(make_flat(&flat));
    struct Owning : StateFormatAdapter {
        Owning(std::unique_ptr<pa_operator> in, int step, int leading, std::vector<AggShape> a, int32_t mem, void* s, std::unique_ptr<Stream> h)
            : StateFormatAdapter(std::move(in), step, leading, std::move(a), mem, s), holder_(std::move(h))
        {
        }
        // the stream the library made for the pair: device pages are completed before they are handed out
        hipStream_t private_stream() override { return holder_ ? holder_->get() : StateFormatAdapter::private_stream(); }
        ~Owning() override
        {
            if (holder_) (void)hipStreamSynchronize(holder_->get());
        }
        std::unique_ptr<Stream> holder_;  // (pooled streams are never destroyed: the inner operator's destructor may still synchronise it)
    };
    return new Owning(std::move(inner), agg->step, leading, std::move(shapes), agg->output_mem, stream, std::move(holder));
}

}  // namespace pa
