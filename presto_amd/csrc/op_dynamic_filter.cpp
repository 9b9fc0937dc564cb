// op_dynamic_filter.cpp -- DynamicFilterSourceOperator
// (core/trino-main/src/main/java/io/trino/operator/DynamicFilterSourceOperator.java:55-425): sits on the build side of a
// join, passes its pages through untouched and collects, per dynamic-filter channel, the distinct build values -- or,
// once those are too many, the min / max -- which the planner turns into a TupleDomain for the probe-side scan.
//
// The reference walks every position into a TypedSet per channel.  Here a page's channel makes one pass on the device
// (dynfilter_kernels.hip): keys into an open-addressing set in HBM and the page's min / max in the same kernel; VARCHAR
// channels go through the string dictionary of intern_kernels.hip, whose size IS the distinct count.  The host keeps the
// reference's state machine (collecting sets -> min / max only -> gave up) and its limits.
#include <algorithm>
#include <cstring>
#include <memory>

#include "dynfilter_kernels.hpp"
#include "intern_kernels.hpp"
#include "operator.hpp"

namespace pa {
void launch_fill_u64(uint64_t* dst, uint64_t value, int64_t n, hipStream_t s);  // static_kernels.hip

void launch_iota_i32(int32_t* dst, int64_t n, hipStream_t s);  // topn_kernels.hip

namespace {

enum State { COLLECT_SETS, MIN_MAX, GAVE_UP };

struct DomainStore {
    int32_t kind = PA_DOMAIN_ALL;
    int32_t type = PA_BIGINT;
    int32_t count = 0;
    std::vector<uint8_t> values;
    std::vector<int32_t> offsets;
};

struct FilterChannel {
    int32_t index = 0, type = PA_BIGINT;
    bool min_max = false;
    // fixed width
    DevBuf keys, counters, running;
    DfSet set{};
    // VARCHAR
    std::unique_ptr<StringInterner> strings;
    DevBuf null_running;
    // after the last page
    uint32_t distinct = 0;
    bool has_null = false;
};

int width_of(int32_t type) { return type_width(type); }

// deep copy of a page descriptor (the arrays it points to stay the caller's: the output page is the input page)
struct PageRef {
    pa_page page{};
    std::vector<pa_column> columns;
    std::vector<std::unique_ptr<pa_column>> nested;
    void assign(const pa_page* p)
    {
        nested.clear();
        columns.assign(p->columns, p->columns + p->channel_count);
        for (pa_column& c : columns) c.dictionary = copy_nested(c.dictionary);
        page = *p;
        page.columns = columns.data();
    }
    const pa_column* copy_nested(const pa_column* d)
    {
        if (!d) return nullptr;
        nested.push_back(std::make_unique<pa_column>(*d));
        pa_column* mine = nested.back().get();
        mine->dictionary = copy_nested(d->dictionary);
        return mine;
    }
};

class DynamicFilterSourceOperator : public pa_operator {
public:
    explicit DynamicFilterSourceOperator(const pa_dynamic_filter_source_desc* d) : stream_(d->stream)
    {
        PA_REQUIRE(d->input_channel_count > 0 && d->input_types, PA_ERR_INVALID_ARGUMENT, "DynamicFilterSource needs input types");
        PA_REQUIRE(d->filter_channel_count >= 0 && (d->filter_channels || d->filter_channel_count == 0), PA_ERR_INVALID_ARGUMENT,
                   "DynamicFilterSource: bad filter channels");
        PA_REQUIRE(d->max_distinct_values >= 0 && d->max_distinct_values <= (1 << 24), PA_ERR_NOT_SUPPORTED,
                   "max_distinct_values above 2^24 is not on the device path");
        types_.assign(d->input_types, d->input_types + d->input_channel_count);
        max_distinct_ = d->max_distinct_values;
        max_bytes_ = d->max_filter_size_bytes;
        row_limit_ = d->min_max_collection_limit;
        needed_.assign(types_.size(), false);
        hipStream_t s = stream_.get();
        bool any_min_max = false;
        for (int i = 0; i < d->filter_channel_count; i++) {
            FilterChannel ch;
            ch.index = d->filter_channels[i];
            PA_REQUIRE(ch.index >= 0 && ch.index < (int)types_.size(), PA_ERR_INVALID_ARGUMENT, "filter channel out of range");
            ch.type = types_[ch.index];
            needed_[ch.index] = true;
            // :188 -- orderable and not floating point
            ch.min_max = d->min_max_collection_limit > 0 && ch.type != PA_DOUBLE && ch.type != PA_REAL;
            any_min_max = any_min_max || ch.min_max;
            if (ch.type == PA_VARCHAR) {
                ch.strings = std::make_unique<StringInterner>();
                PA_HIP(hipMemsetAsync(ch.null_running.ensure(64), 0, 64, s));
            }
            else {
                // room for the limit, the value that crosses it and the inserts in flight when it is crossed, at load <= 1/2
                uint64_t cap = 4096;
                while (cap < 2 * ((uint64_t)max_distinct_ + 2 + (uint64_t)kDfBlocks * 256)) cap <<= 1;
                ch.set.keys = static_cast<uint64_t*>(ch.keys.ensure(cap * 8));
                ch.set.cap_mask = (uint32_t)(cap - 1);
                ch.set.limit = (uint32_t)max_distinct_;
                ch.set.counters = static_cast<uint32_t*>(ch.counters.ensure(64));
                launch_fill_u64(ch.set.keys, kDfEmpty, (int64_t)cap, s);
                PA_HIP(hipMemsetAsync(ch.set.counters, 0, 64, s));
                PA_HIP(hipMemsetAsync(ch.running.ensure(64), 0, 64, s));
            }
            channels_.push_back(std::move(ch));
        }
        has_min_max_ = any_min_max;  // minValues != null in the reference
        partials_.ensure(df_partials_bytes());
        h_ = static_cast<uint32_t*>(h_buf_.ensure(32 * std::max<size_t>(channels_.size(), 1)));
    }
    ~DynamicFilterSourceOperator() override { (void)hipStreamSynchronize(stream_.get()); }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    bool needs_input() override { return !has_current_ && !finished_; }

    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finished_, PA_ERR_ILLEGAL_STATE, "DynamicFilterSourceOperator: addInput() may not be called after finish()");
        PA_REQUIRE(!has_current_, PA_ERR_ILLEGAL_STATE, "Operator does not need input");
        PA_REQUIRE(page != nullptr, PA_ERR_INVALID_ARGUMENT, "page is null");
        PA_REQUIRE(page->channel_count == (int32_t)types_.size(), PA_ERR_INVALID_ARGUMENT, "page channel count does not match the operator's input types");
        current_.assign(page);
        has_current_ = true;
        if (state_ == GAVE_UP) return;
        const int64_t n = page->position_count;
        if (state_ == MIN_MAX) {
            row_limit_ -= n;
            if (row_limit_ < 0) {  // handleMinMaxCollectionLimitExceeded
                give_up();
                return;
            }
            collect(page, false);
            return;
        }
        row_limit_ -= n;
        collect(page, true);
        // filterSizeInBytes / filterMaxDistinctValues of :249-262.  The reference adds up TypedSet.getRetainedSizeInBytes(),
        // a JVM-layout number; here the size of a set is its values' bytes (VARCHAR: 8-byte padded, + 4 B offsets).
        int64_t bytes = 0;
        uint32_t most = 0;
        for (FilterChannel& ch : channels_) {
            const uint32_t size = ch.distinct + (ch.has_null ? 1u : 0u);
            most = std::max(most, size);
            bytes += ch.type == PA_VARCHAR ? (int64_t)ch.strings->bytes() + 4LL * ch.distinct : (int64_t)ch.distinct * width_of(ch.type);
        }
        if ((int64_t)most > max_distinct_ || bytes > max_bytes_) {  // handleTooLargePredicate
            if (!has_min_max_ || row_limit_ < 0) give_up();
            else state_ = MIN_MAX;  // the min / max of the values collected so far are already in `running`
        }
    }

    bool get_output(pa_page* out) override
    {
        if (!has_current_) return false;
        has_current_ = false;
        *out = current_.page;  // the same Page object in the reference: the caller's buffers
        return true;
    }

    void finish() override
    {
        if (finished_) return;  // Driver may call finish() more than once (:360)
        finished_ = true;
        if (state_ == GAVE_UP) return;
        domains_.assign(channels_.size(), DomainStore{});
        for (size_t i = 0; i < channels_.size(); i++) {
            domains_[i].type = channels_[i].type;
            if (state_ == COLLECT_SETS) values_domain(channels_[i], &domains_[i]);
            else if (channels_[i].min_max) range_domain(channels_[i], &domains_[i]);
        }
        all_ = channels_.empty();  // TupleDomain.withColumnDomains of nothing
        notified_ = true;
    }
    bool is_finished() override { return !has_current_ && finished_; }

    int64_t memory_bytes() override
    {
        int64_t b = (int64_t)stager_.bytes();
        for (const FilterChannel& ch : channels_) b += (int64_t)ch.keys.capacity();
        return b;
    }

    // dynamicPredicateConsumer.accept(...): 0 while it has not been called
    int32_t poll(int32_t* is_all, pa_domain* out, int32_t capacity)
    {
        if (!notified_) return 0;
        *is_all = all_ ? 1 : 0;
        if (all_) return 1;
        PA_REQUIRE(capacity >= (int32_t)channels_.size() && out != nullptr, PA_ERR_INVALID_ARGUMENT, "domain array too small");
        for (size_t i = 0; i < channels_.size(); i++) {
            const DomainStore& d = domains_[i];
            pa_domain& o = out[i];
            memset(&o, 0, sizeof(o));
            o.kind = d.kind;
            o.value_count = d.count;
            o.values.type = d.type;
            o.values.encoding = d.type == PA_VARCHAR ? PA_VARWIDTH : PA_FLAT;
            o.values.values = d.values.data();
            o.values.offsets = d.type == PA_VARCHAR ? d.offsets.data() : nullptr;
        }
        return 1;
    }

private:
    void give_up()
    {
        state_ = GAVE_UP;
        all_ = true;  // TupleDomain.all(): every probe-side value may be read
        notified_ = true;
    }

    void collect(const pa_page* page, bool with_sets)
    {
        const int64_t n = page->position_count;
        if (n == 0 || channels_.empty()) return;
        hipStream_t s = stream_.get();
        DevPage dp = stager_.stage(page, &needed_, s);
        int64_t* partials = partials_.as<int64_t>();
        timer.begin(s);
        for (FilterChannel& ch : channels_) {
            const DevColumn& col = dp.cols[ch.index];
            PA_REQUIRE(col.type == ch.type, PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared input type");
            if (ch.type == PA_VARCHAR) {
                // the dictionary is the set; NULLs: "any NULL" = max over the valueIsNull bytes read as a BOOLEAN column
                ch.strings->intern(col.values, col.offsets, col.nulls, n, s);
                if (col.nulls && with_sets) launch_df_collect(PA_BOOLEAN, col.nulls, nullptr, n, nullptr, partials, ch.null_running.as<int64_t>(), s);
                continue;
            }
            if (!with_sets && !ch.min_max) continue;
            launch_df_collect(ch.type, col.values, col.nulls, n, with_sets ? &ch.set : nullptr, ch.min_max ? partials : nullptr,
                              ch.running.as<int64_t>(), s);
        }
        timer.end(s, true);
        if (!with_sets) return;
        // the sizes decide the state: one small read-back per channel
        for (size_t i = 0; i < channels_.size(); i++) {
            FilterChannel& ch = channels_[i];
            if (ch.type == PA_VARCHAR) PA_HIP(hipMemcpyAsync(h_ + 8 * i, ch.null_running.ptr(), 24, hipMemcpyDeviceToHost, s));
            else PA_HIP(hipMemcpyAsync(h_ + 8 * i, ch.set.counters, 12, hipMemcpyDeviceToHost, s));
        }
        PA_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < channels_.size(); i++) read_sizes(channels_[i], h_ + 8 * i);
    }
    void read_sizes(FilterChannel& ch, const uint32_t* h)
    {
        if (ch.type == PA_VARCHAR) {
            int64_t running[3];
            memcpy(running, h, 24);
            ch.distinct = ch.strings->size();
            ch.has_null = running[2] != 0 && running[1] != 0;  // max over the NULL flags
        }
        else {
            ch.distinct = h[0] + (h[2] ? 1u : 0u);
            ch.has_null = h[1] != 0;
        }
    }

    // convertToDomain (:403-418): the non-null, non-NaN values, as a sorted discrete set
    void values_domain(FilterChannel& ch, DomainStore* d)
    {
        hipStream_t s = stream_.get();
        d->kind = PA_DOMAIN_VALUES;
        if (ch.type == PA_VARCHAR) {
            std::vector<std::string> strs = download_strings(ch);
            std::sort(strs.begin(), strs.end());  // VarcharType compares unsigned bytes, as std::string does
            store_strings(strs, d);
            if (strs.empty()) d->kind = PA_DOMAIN_NONE;
            return;
        }
        const uint64_t cap = (uint64_t)ch.set.cap_mask + 1;
        DevBuf out, count;
        uint64_t* dev = static_cast<uint64_t*>(out.ensure(cap * 8));
        uint32_t* cnt = static_cast<uint32_t*>(count.ensure(64));
        launch_df_values(ch.set, dev, cnt, s);
        uint32_t m = 0, flags[3];
        PA_HIP(hipMemcpyAsync(&m, cnt, 4, hipMemcpyDeviceToHost, s));
        PA_HIP(hipMemcpyAsync(flags, ch.set.counters, 12, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        std::vector<uint64_t> keys(m);
        if (m) PA_HIP(hipMemcpyAsync(keys.data(), dev, (size_t)m * 8, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        if (flags[2]) keys.push_back(kDfEmpty);
        if (ch.type == PA_DOUBLE || ch.type == PA_REAL) {
            // "join doesn't match rows with NaN values"; the rest in Double.compare order (REAL keys are the widened floats)
            keys.erase(std::remove(keys.begin(), keys.end(), 0x7ff8000000000000ULL), keys.end());
            std::sort(keys.begin(), keys.end(), [](uint64_t a, uint64_t b) {
                double x, y;
                memcpy(&x, &a, 8);
                memcpy(&y, &b, 8);
                return x < y;
            });
            if (ch.type == PA_REAL) {  // back to the IntArrayBlock's raw float bits
                for (uint64_t& k : keys) {
                    double x;
                    memcpy(&x, &k, 8);
                    const float f = (float)x;
                    uint32_t bits;
                    memcpy(&bits, &f, 4);
                    k = bits;
                }
            }
        }
        else std::sort(keys.begin(), keys.end(), [](uint64_t a, uint64_t b) { return (int64_t)a < (int64_t)b; });
        store_fixed(keys, ch.type, d);
        if (keys.empty()) d->kind = PA_DOMAIN_NONE;  // ValueSet.copyOf of nothing, nullAllowed = false
    }

    // :374-389: [min, max] of an orderable channel, Domain.none when every value was NULL
    void range_domain(FilterChannel& ch, DomainStore* d)
    {
        hipStream_t s = stream_.get();
        if (ch.type == PA_VARCHAR) {
            std::vector<std::string> strs = download_strings(ch);
            if (strs.empty()) {
                d->kind = PA_DOMAIN_NONE;
                return;
            }
            auto mm = std::minmax_element(strs.begin(), strs.end());
            std::vector<std::string> two{*mm.first, *mm.second};
            store_strings(two, d);
            d->kind = PA_DOMAIN_RANGE;
            return;
        }
        int64_t running[3];
        PA_HIP(hipMemcpyAsync(running, ch.running.ptr(), 24, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        if (!running[2]) {
            d->kind = PA_DOMAIN_NONE;
            return;
        }
        std::vector<uint64_t> two{(uint64_t)running[0], (uint64_t)running[1]};
        store_fixed(two, ch.type, d);
        d->kind = PA_DOMAIN_RANGE;
    }

    std::vector<std::string> download_strings(FilterChannel& ch)
    {
        hipStream_t s = stream_.get();
        const int64_t m = ch.strings->size();
        std::vector<std::string> strs;
        if (m == 0) return strs;
        DevBuf ids, values, offsets;
        launch_iota_i32(static_cast<int32_t*>(ids.ensure((size_t)m * 4)), m, s);
        ch.strings->decode(ids.as<int32_t>(), nullptr, m, &values, &offsets, s);
        std::vector<int32_t> off((size_t)m + 1);
        PA_HIP(hipMemcpyAsync(off.data(), offsets.ptr(), off.size() * 4, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        std::vector<char> bytes((size_t)std::max(off[m], 1));
        if (off[m]) PA_HIP(hipMemcpyAsync(bytes.data(), values.ptr(), (size_t)off[m], hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        strs.reserve((size_t)m);
        for (int64_t i = 0; i < m; i++) strs.emplace_back(bytes.data() + off[i], bytes.data() + off[i + 1]);
        return strs;
    }
    static void store_strings(const std::vector<std::string>& strs, DomainStore* d)
    {
        d->count = (int32_t)strs.size();
        d->offsets.assign(1, 0);
        d->values.clear();
        for (const std::string& v : strs) {
            d->values.insert(d->values.end(), v.begin(), v.end());
            d->offsets.push_back((int32_t)d->values.size());
        }
        if (d->values.empty()) d->values.push_back(0);
    }
    static void store_fixed(const std::vector<uint64_t>& keys, int32_t type, DomainStore* d)
    {
        const int w = width_of(type);
        d->count = (int32_t)keys.size();
        d->values.assign(std::max<size_t>(keys.size() * w, 1), 0);
        for (size_t i = 0; i < keys.size(); i++) memcpy(&d->values[i * w], &keys[i], (size_t)w);  // little endian: low bytes = narrower value
    }

    Stream stream_;
    std::vector<int32_t> types_;
    std::vector<bool> needed_;
    std::vector<FilterChannel> channels_;
    int64_t max_distinct_ = 0, max_bytes_ = 0, row_limit_ = 0;
    bool has_min_max_ = false;
    State state_ = COLLECT_SETS;
    PageStager stager_;
    DevBuf partials_;
    PinnedBuf h_buf_;
    uint32_t* h_ = nullptr;
    PageRef current_;
    bool has_current_ = false, finished_ = false, notified_ = false, all_ = false;
    std::vector<DomainStore> domains_;
};

}  // namespace

pa_operator* make_dynamic_filter_source(const pa_dynamic_filter_source_desc* desc)
{
    PA_REQUIRE(desc != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    return new DynamicFilterSourceOperator(desc);
}

int32_t dynamic_filter_poll(pa_operator* op, int32_t* is_all, pa_domain* domains, int32_t capacity)
{
    auto* df = dynamic_cast<DynamicFilterSourceOperator*>(op);
    PA_REQUIRE(df != nullptr, PA_ERR_INVALID_ARGUMENT, "not a DynamicFilterSourceOperator");
    return df->poll(is_all, domains, capacity);
}

}  // namespace pa
