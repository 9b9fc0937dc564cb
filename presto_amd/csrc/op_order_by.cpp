// op_order_by.cpp -- OrderByOperator (core/trino-main/src/main/java/io/trino/operator/OrderByOperator.java:45-330): collect every
// input page (PagesIndex.addPage), sort by (sortChannels, sortOrders) when the input ends (PagesIndex.sort ->
// PagesIndexOrdering with SimplePagesIndexComparator), emit the output channels in order.
//
// The reference quick-sorts row addresses with a multi-channel comparator.  Here: a stable least-significant-digit radix
// sort of a row permutation on device.  Every sort channel is turned into order-preserving 64-bit images
// (topn_kernels.hip: integers, DATE, BOOLEAN, DOUBLE in Double.compare order; a VARCHAR as its 8-byte chunks, zero padded,
// with the length as the least significant key -- Slice.compareTo: unsigned bytes, a proper prefix sorts first) plus a
// one-bit NULL digit above them (SortOrder: NULLS FIRST / LAST, independent of ASC / DESC); channels are processed from the
// last sort channel to the first, each image by one stable sort of (image, row id) pairs over the bits in which the images
// differ (sort_kernels.hip), the NULL digit by the stable partition the exchange already uses.  Fully tied rows keep arrival
// order.  A first sort channel of integers without NULL rows that is also an output channel is written from the sorted images.
#include <algorithm>
#include <cstring>

#include "operator.hpp"
#include "scan_kernels.hpp"
#include "sort_kernels.hpp"
#include "static_kernels.hpp"
#include "topn_kernels.hpp"

namespace pa {

void launch_sort_null_digits(const uint8_t* nulls, const int32_t* perm, int64_t n, int nulls_first, int32_t* digits, hipStream_t s);
void launch_varchar_chunk_keys(const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, int chunk, int descending,
                               uint64_t* keys, hipStream_t s);
void launch_iota_i32(int32_t* dst, int64_t n, hipStream_t s);
int32_t varchar_max_length(const int32_t* offsets, int64_t n, void* temp_dev_8, hipStream_t s);

namespace {

// all rows of the input so far, column by column (PagesIndex.addPage; flat device arrays: address == position)
struct Accumulated {
    int32_t type = PA_BIGINT;
    bool varwidth = false, has_nulls = false;
    DevBuf values, offsets, nulls;
    int64_t bytes = 0;  // VARCHAR bytes used
};

class OrderByOperator : public pa_operator {
public:
    explicit OrderByOperator(const pa_order_by_desc* d) : stream_(d->stream)
    {
        require_device();
        PA_REQUIRE(d->input_channel_count > 0 && d->input_types, PA_ERR_INVALID_ARGUMENT, "OrderBy needs input types");
        PA_REQUIRE(d->sort_channel_count > 0 && d->sort_channels && d->sort_orders, PA_ERR_INVALID_ARGUMENT, "OrderBy needs sort channels");
        types_.assign(d->input_types, d->input_types + d->input_channel_count);
        if (d->output_channels) output_channels_.assign(d->output_channels, d->output_channels + d->output_channel_count);
        sort_channels_.assign(d->sort_channels, d->sort_channels + d->sort_channel_count);
        sort_orders_.assign(d->sort_orders, d->sort_orders + d->sort_channel_count);
        for (int c : output_channels_) PA_REQUIRE(c >= 0 && c < (int)types_.size(), PA_ERR_INVALID_ARGUMENT, "output channel out of range");
        for (size_t i = 0; i < sort_channels_.size(); i++) {
            PA_REQUIRE(sort_channels_[i] >= 0 && sort_channels_[i] < (int)types_.size(), PA_ERR_INVALID_ARGUMENT, "sort channel out of range");
            PA_REQUIRE(sort_orders_[i] >= 0 && sort_orders_[i] <= 3, PA_ERR_INVALID_ARGUMENT, "unknown sort order");
        }
        output_mem_ = d->output_mem;
        cols_.resize(types_.size());
        needed_.assign(types_.size(), false);
        for (int c : output_channels_) needed_[(size_t)c] = true;
        for (int c : sort_channels_) needed_[(size_t)c] = true;
        for (size_t c = 0; c < types_.size(); c++) {
            cols_[c].type = types_[c];
            cols_[c].varwidth = types_[c] == PA_VARCHAR;
        }
    }
    ~OrderByOperator() override
    {
        (void)hipStreamSynchronize(stream_.get());
        for (Held& h : held_) {
            if (h.release) h.release(h.release_ctx);
        }
    }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    bool needs_input() override { return !finishing_; }

    bool takes_retained() override { return true; }

    // PagesIndex.addPage keeps the Page; so does this for device pages that stay where they are (PA_PAGE_STABLE, PA_PAGE_RETAINED) and
    // hold only flat fixed-width blocks in the needed channels: they are listed, not copied.  One such page alone is sorted in place; several
    // are laid behind each other when the input ends (the same copies, later).  Every other page is copied when it arrives -- its
    // buffers are the caller's again when add_input returns -- after the pages listed before it.
    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(page != nullptr && page->channel_count == (int32_t)types_.size(), PA_ERR_INVALID_ARGUMENT, "page does not match the operator's input types");
        const bool retained = (page->flags & PA_PAGE_RETAINED) != 0 && page->release != nullptr;
        if (page->position_count == 0) {
            if (retained) page->release(page->release_ctx);
            return;
        }
        const int64_t m = page->position_count;
        if (rows_ + m > INT32_MAX) {
            if (retained) page->release(page->release_ctx);   // (nothing of it was read)
            throw Error(PA_ERR_INSUFFICIENT_RESOURCES, "too many rows for one sort");
        }
        if (page->mem == PA_MEM_DEVICE && (retained || (page->flags & PA_PAGE_STABLE) != 0) && holdable(page)) {
            Held h;
            h.rows = m;
            h.cols.assign(page->columns, page->columns + page->channel_count);
            if (retained) {
                h.release = page->release;
                h.release_ctx = page->release_ctx;
            }
            held_.push_back(std::move(h));
            rows_ += m;
            return;
        }
        // (a retained page that is copied goes back to its owner when the copies have run -- also when a check below throws)
        struct ReleaseAtExit {
            const pa_page* p;
            bool on;
            hipStream_t s;
            ~ReleaseAtExit()
            {
                if (!on) return;
                (void)hipStreamSynchronize(s);
                p->release(p->release_ctx);
            }
        } release_at_exit{page, retained, stream_.get()};
        flush_held();
        hipStream_t s = stream_.get();
        DevPage dp = stager_.stage(page, &needed_, s);
        append(dp);
        PA_HIP(hipStreamSynchronize(s));  // the stager's buffers are reused by the next page
    }

    bool holdable(const pa_page* page) const
    {
        for (size_t c = 0; c < types_.size(); c++) {
            if (!needed_[c]) continue;
            const pa_column& in = page->columns[c];
            if (in.encoding != PA_FLAT || in.type != types_[c] || in.values == nullptr || type_width(in.type) <= 0) return false;
        }
        return true;
    }

    // the listed pages behind the rows copied so far, in arrival order; their owners get them back
    void flush_held()
    {
        if (held_.empty()) return;
        hipStream_t s = stream_.get();
        const int64_t total = rows_;
        int64_t listed = 0;
        for (const Held& h : held_) listed += h.rows;
        rows_ = total - listed;   // (append counts them again)
        std::vector<Held> held;
        held.swap(held_);
        struct ReleaseAll {
            std::vector<Held>& held;
            hipStream_t s;
            ~ReleaseAll()
            {
                (void)hipStreamSynchronize(s);
                for (Held& h : held) {
                    if (h.release) h.release(h.release_ctx);
                }
            }
        } release_all{held, s};
        for (const Held& h : held) {
            DevPage dp;
            dp.n = (int32_t)h.rows;
            dp.cols.resize(types_.size());
            for (size_t c = 0; c < types_.size(); c++) {
                dp.cols[c].type = types_[c];
                if (!needed_[c]) continue;
                dp.cols[c].values = h.cols[c].values;
                dp.cols[c].nulls = h.cols[c].nulls;
            }
            append(dp);
        }
    }

    void append(const DevPage& dp)
    {
        hipStream_t s = stream_.get();
        const int64_t m = dp.n;
        for (size_t c = 0; c < types_.size(); c++) {
            if (!needed_[c]) continue;
            const DevColumn& in = dp.cols[c];
            PA_REQUIRE(in.type == types_[c], PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared input type");
            Accumulated& a = cols_[c];
            if (a.varwidth) {
                int32_t ends[2];
                PA_HIP(hipMemcpyAsync(&ends[0], in.offsets, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipMemcpyAsync(&ends[1], in.offsets + m, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
                const int64_t add = ends[1] - ends[0];
                PA_REQUIRE(a.bytes + add <= INT32_MAX, PA_ERR_INSUFFICIENT_RESOURCES, "VARCHAR column exceeds 2 GB");
                int32_t* off = static_cast<int32_t*>(a.offsets.reserve_keep((size_t)(rows_ + m + 1) * 4, (size_t)(rows_ ? rows_ + 1 : 0) * 4, s));
                launch_offsets_append(in.offsets, m, (int32_t)a.bytes, off + rows_, rows_ == 0, s);
                char* v = static_cast<char*>(a.values.reserve_keep((size_t)(a.bytes + add + 1), (size_t)a.bytes, s));
                if (add) PA_HIP(hipMemcpyAsync(v + a.bytes, static_cast<const char*>(in.values) + ends[0], (size_t)add, hipMemcpyDeviceToDevice, s));
                a.bytes += add;
            }
            else {
                const size_t w = (size_t)type_width(a.type);
                char* v = static_cast<char*>(a.values.reserve_keep((size_t)(rows_ + m) * w, (size_t)rows_ * w, s));
                PA_HIP(hipMemcpyAsync(v + (size_t)rows_ * w, in.values, (size_t)m * w, hipMemcpyDeviceToDevice, s));
            }
            if (in.nulls || a.has_nulls) {
                uint8_t* nl = static_cast<uint8_t*>(a.nulls.reserve_keep((size_t)(rows_ + m), a.has_nulls ? (size_t)rows_ : 0, s));
                if (!a.has_nulls && rows_ > 0) PA_HIP(hipMemsetAsync(nl, 0, (size_t)rows_, s));
                if (in.nulls) PA_HIP(hipMemcpyAsync(nl + rows_, in.nulls, (size_t)m, hipMemcpyDeviceToDevice, s));
                else PA_HIP(hipMemsetAsync(nl + rows_, 0, (size_t)m, s));
                a.has_nulls = true;
            }
        }
        rows_ += m;
    }

    void finish() override { finishing_ = true; }
    bool is_finished() override { return finishing_ && output_done_; }

    bool get_output(pa_page* out) override
    {
        if (!finishing_ || output_done_) return false;
        output_done_ = true;
        if (rows_ == 0) return false;
        hipStream_t s = stream_.get();
        if (held_.size() == 1 && held_[0].rows == rows_) {
            // the one page of the input, still where its owner put it: its block arrays ARE the columns (released when the operator goes)
            for (size_t c = 0; c < types_.size(); c++) {
                if (!needed_[c]) continue;
                const pa_column& in = held_[0].cols[c];
                cols_[c].values.borrow(in.values, (size_t)rows_ * type_width(types_[c]));
                if (in.nulls) {
                    cols_[c].nulls.borrow(in.nulls, (size_t)rows_);
                    cols_[c].has_nulls = true;
                }
            }
        }
        else flush_held();
        const int64_t n = rows_;
        int32_t* perm = static_cast<int32_t*>(perm_[0].ensure((size_t)n * 4));
        int32_t* next = static_cast<int32_t*>(perm_[1].ensure((size_t)n * 4));
        int32_t* digits = static_cast<int32_t*>(digits_.ensure((size_t)n * 4));
        int32_t* pos = static_cast<int32_t*>(pos_.ensure((size_t)n * 4));
        int64_t* counts = static_cast<int64_t*>(counts_.ensure(256 * 8));
        uint64_t* keys = static_cast<uint64_t*>(keys_.ensure((size_t)n * 8));
        void* temp = part_temp_.ensure(partition_temp_bytes(n, 256));
        timer.begin(s);
        bool identity = true;  // the permutation is still 0, 1, 2, ...: the images are in the current order as they are, and perm is not
                               // written yet (the first sort takes the row ids as implied; whoever else needs them writes them first)
        bool written = false;
        auto materialize = [&] {
            if (identity && !written) launch_iota_i32(perm, n, s);
            written = true;
        };
        auto pass = [&](int partitions) {
            // one stable radix pass: rows grouped by digit, arrival order kept inside a digit; perm' = perm o pos
            launch_partition_positions(digits, n, partitions, pos, counts, temp, s);
            materialize();
            launch_gather_flat(perm, 4, pos, n, next, s);
            std::swap(perm, next);
            identity = false;
        };
        // Stable sort of (image, row id) PAIRS: the images are brought into the current order once (one gather), then the pairs are sorted
        // by the bits in which the images differ at all (OR / AND of the images: keys below 2^40 are sorted by 40 bits, not 64) --
        // launch_sort_pairs, sort_kernels.hip.  The first version sorted the permutation alone and fetched every pass's digits through
        // it (a random 8-byte read per row and pass: 2^24 rows by a BIGINT key 6.5 ms), the second regrouped the pairs in eight LDS-staged
        // 8-bit passes of its own (count, scan, scatter: 2.3 ms).
        uint64_t* kp[2] = {static_cast<uint64_t*>(pair_keys_[0].ensure((size_t)n * 8)), static_cast<uint64_t*>(pair_keys_[1].ensure((size_t)n * 8))};
        // Output channels that can ride along with the pairs of the LAST sort (the first sort channel's) instead of being gathered by the
        // sorted row ids afterwards: flat 4- / 8-byte channels without NULL rows, when that sort starts from the identity permutation (one
        // sort channel, or the channels behind it all constant) and no NULL digit follows it.  payload_of[j] = its slot, or -1.
        const std::vector<int>& outs = output_channels_;
        std::vector<int> payload_of(outs.size(), -1);
        SortPayload payload;
        memset(&payload, 0, sizeof payload);
        {
            const Accumulated& f = cols_[(size_t)sort_channels_[0]];
            const bool images_serve = !f.varwidth && !f.has_nulls && (f.type == PA_BIGINT || f.type == PA_INTEGER || f.type == PA_DATE);
            if (!f.varwidth && !f.has_nulls && !getenv("PRESTO_AMD_SORT_NO_PAYLOAD")) {
                for (size_t j = 0; j < outs.size() && payload.count < PA_SORT_MAX_PAYLOAD; j++) {
                    const Accumulated& a = cols_[(size_t)outs[j]];
                    const int w = a.varwidth ? 0 : type_width(a.type);
                    if (a.has_nulls || (w != 4 && w != 8)) continue;
                    if (outs[j] == sort_channels_[0] && images_serve) continue;   // (written from the sorted images)
                    payload_of[j] = payload.count;
                    payload.in[payload.count] = a.values.ptr();
                    payload.out[payload.count] = payload_out_[payload.count].ensure((size_t)n * w);
                    payload.width[payload.count] = w;
                    payload.count++;
                }
            }
        }
        bool payload_moved = false;
        const size_t sort_temp_bytes = sort_pairs_temp_bytes(n, payload.count);
        void* sort_temp = radix_temp_.ensure(sort_temp_bytes);
        uint64_t* or_and = static_cast<uint64_t*>(or_and_.ensure(key_or_and_bytes()));
        std::vector<uint64_t> h_or_and(key_or_and_bytes() / 8);
        const uint64_t* sorted_images = nullptr;  // the images of the channel sorted by last, in their sorted order (valid until perm changes again)
        // pairs: OR / AND of the images already in or_and (the image kernel left them there), 0 = still to be computed
        // crowded: images of doubles (exponent bits) or text crowd under few bit prefixes -- the sort takes its bucket bounds from a sample
        auto sort_by_image = [&](int pairs, bool last, bool crowded) {
            const uint64_t* in = keys;
            if (!identity) {
                launch_gather_flat(keys, 8, perm, n, kp[0], s);
                in = kp[0];
            }
            const int blocks = pairs > 0 ? pairs : launch_key_or_and(in, n, or_and, s);
            read_back(h_or_and.data(), or_and, (size_t)blocks * 16, s);
            uint64_t h[2] = {0ULL, ~0ULL};
            for (int b = 0; b < blocks; b++) {
                h[0] |= h_or_and[2 * (size_t)b];
                h[1] &= h_or_and[2 * (size_t)b + 1];
            }
            const uint64_t varying = h[0] ^ h[1];
            sorted_images = nullptr;
            if (varying == 0ULL) return;  // every image the same: the order stays
            int begin_bit = __builtin_ctzll(varying);
            const int end_bit = 64 - __builtin_clzll(varying);
            // (rocPRIM 4.0's merge-sort path for small inputs builds its mask of the bit range with 1 << end_bit: a range that ends at
            // bit 64 without starting at bit 0 compares nothing -- such ranges are widened to the whole key)
            if (end_bit == 64) begin_bit = 0;
            const bool carry = last && identity && payload.count > 0;
            const int path = launch_sort_pairs(in, identity ? nullptr : perm, pos, kp[1], next, n, begin_bit, end_bit, sort_temp, sort_temp_bytes, s,
                                               carry ? &payload : nullptr, crowded ? PA_SORT_HINT_CROWDED : PA_SORT_HINT_SPREAD);
            payload_moved = carry && path == PA_SORT_BUCKETS;
            timer.set_name(path == PA_SORT_LIBRARY ? "rocprim_radix_sort_pairs" : "pa_sort_buckets");   // (pa_op_kernel_name: which sort the last image took)
            std::swap(perm, next);
            identity = false;
            sorted_images = kp[1];
        };
        for (int i = (int)sort_channels_.size() - 1; i >= 0; i--) {
            const Accumulated& a = cols_[(size_t)sort_channels_[i]];
            const int order = sort_orders_[i];
            const bool descending = order >= 2, nulls_first = (order & 1) == 0;
            const uint8_t* nulls = a.has_nulls ? a.nulls.as<uint8_t>() : nullptr;
            if (a.varwidth) {
                // least significant first: the length, then the 8-byte chunks from the last to the first
                const int32_t max_len = varchar_max_length(a.offsets.as<int32_t>(), n, counts, s);
                const int chunks = (max_len + 7) / 8;
                for (int chunk = chunks; chunk >= 0; chunk--) {
                    launch_varchar_chunk_keys(a.values.ptr(), a.offsets.as<int32_t>(), nulls, n, chunk == chunks ? -1 : chunk, descending ? 1 : 0, keys, s);
                    sort_by_image(0, false, chunk != chunks);   // (the length key spreads; the byte chunks do not)
                }
            }
            else {
                // value image; NULL rows get one constant image (they keep arrival order among themselves) and their place
                // relative to the values is decided by the separate NULL digit below
                const int pairs = launch_topn_keys_or_and(a.type, a.values.ptr(), nullptr, nulls, n, descending ? PA_DESC_NULLS_LAST : PA_ASC_NULLS_LAST, keys, or_and, s);
                sort_by_image(pairs, i == 0 && nulls == nullptr, a.type == PA_DOUBLE || a.type == PA_REAL);
            }
            if (nulls) {
                materialize();
                launch_sort_null_digits(nulls, perm, n, nulls_first ? 1 : 0, digits, s);
                pass(2);
                sorted_images = nullptr;
            }
        }
        materialize();   // (no sort ran: every image the same)
        timer.end(s);
        // the first sort channel as an output channel: its sorted images ARE the column (integers without NULL rows), no gather
        const int first_channel = sort_channels_[0];
        const bool first_descending = sort_orders_[0] >= 2;
        {
            const Accumulated& f = cols_[(size_t)first_channel];
            if (f.varwidth || f.has_nulls || !(f.type == PA_BIGINT || f.type == PA_INTEGER || f.type == PA_DATE)) sorted_images = nullptr;
        }
        // output channels in sorted order (PagesIndex.appendTo)
        out_cols_.clear();
        out_cols_.resize(outs.size());
        // the fixed-width channels (and every NULL flag array) are gathered through the permutation by ONE launch: the permutation is
        // read once, the random reads of all channels are in flight together
        GatherMultiArgs gm;
        memset(&gm, 0, sizeof gm);
        gm.positions[0] = perm;
        gm.count = n;
        auto flush_gather = [&] {
            if (gm.ncols > 0) launch_gather_multi(gm, s);
            gm.ncols = 0;
        };
        for (size_t j = 0; j < outs.size(); j++) {
            const Accumulated& a = cols_[(size_t)outs[j]];
            OutColumn& oc = out_cols_[j];
            oc.type = a.type;
            oc.varwidth = a.varwidth;
            const uint8_t* nulls = a.has_nulls ? a.nulls.as<uint8_t>() : nullptr;
            oc.has_nulls = nulls != nullptr;
            if (gm.ncols == GATHER_MULTI_MAX_COLS) flush_gather();
            GatherMultiCol& gc = gm.col[gm.ncols];
            memset(&gc, 0, sizeof gc);
            gc.width = 1;
            if (a.varwidth) {
                int32_t* lens = static_cast<int32_t*>(oc.offsets.ensure((size_t)(n + 1) * 4));
                int32_t* total = reinterpret_cast<int32_t*>(counts);
                launch_varwidth_lengths(perm, n, a.offsets.as<int32_t>(), nulls, lens, s);
                launch_exclusive_scan_i32(lens, lens, n, total, scan_temp_.ensure(scan_temp_bytes(n)), s);
                int32_t h_total = 0;
                PA_HIP(hipMemcpyAsync(&h_total, total, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
                launch_varwidth_copy(perm, n, a.offsets.as<int32_t>(), a.values.as<uint8_t>(), nulls, lens,
                                     static_cast<uint8_t*>(oc.values.ensure((size_t)(h_total > 0 ? h_total : 1))), total, s);
            }
            else if (payload_moved && payload_of[j] >= 0) {
                oc.values = std::move(payload_out_[payload_of[j]]);   // the sort brought the channel along
            }
            else if (sorted_images != nullptr && outs[j] == first_channel) {
                launch_topn_values_of_keys(a.type, sorted_images, n, first_descending, oc.values.ensure((size_t)n * type_width(a.type)), s);
            }
            else {
                const int w = type_width(a.type);
                if (w == 1 || w == 4 || w == 8) {
                    gc.src = a.values.ptr();
                    gc.dst = oc.values.ensure((size_t)n * w);
                    gc.width = w;
                }
                else launch_gather_flat(a.values.ptr(), w, perm, n, oc.values.ensure((size_t)n * w), s);   // (LONG_DECIMAL: 16 bytes)
            }
            if (nulls) {
                gc.src_nulls = nulls;
                gc.dst_nulls = static_cast<uint8_t*>(oc.nulls.ensure((size_t)n));
            }
            if (gc.dst || gc.dst_nulls) gm.ncols++;
        }
        flush_gather();
        publish_output(out_cols_, (int32_t)n, output_mem_, s, out, out_storage_);
        return true;
    }

    int64_t memory_bytes() override
    {
        int64_t b = 0;
        for (const auto& a : cols_) b += (int64_t)(a.values.capacity() + a.offsets.capacity() + a.nulls.capacity());
        return b;
    }

private:
    Stream stream_;
    PageStager stager_;
    std::vector<int32_t> types_, sort_orders_;
    std::vector<int> output_channels_, sort_channels_;
    std::vector<bool> needed_;
    std::vector<Accumulated> cols_;
    // device pages listed instead of copied (add_input)
    struct Held {
        int64_t rows = 0;
        std::vector<pa_column> cols;
        void (*release)(void*) = nullptr;
        void* release_ctx = nullptr;
    };
    std::vector<Held> held_;
    int64_t rows_ = 0;
    int32_t output_mem_ = PA_MEM_HOST;
    bool finishing_ = false, output_done_ = false;
    DevBuf payload_out_[PA_SORT_MAX_PAYLOAD];
    DevBuf perm_[2], digits_, pos_, counts_, keys_, part_temp_, scan_temp_, pair_keys_[2], radix_temp_, or_and_;
    std::vector<OutColumn> out_cols_;
    std::vector<pa_column> out_storage_;
};

}  // namespace

pa_operator* make_order_by(const pa_order_by_desc* desc)
{
    PA_REQUIRE(desc != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    return new OrderByOperator(desc);
}

}  // namespace pa
