// op_topn.cpp -- TopNOperator (core/trino-main/src/main/java/io/trino/operator/TopNOperator.java:30-160; TopNProcessor.java:
// 35-110; GroupedTopNBuilder.java with a single group): keep the N best rows under (sortChannels, sortOrders), emit them
// in order once the input is finished.
//
// The reference compares every input row with the root of an N-row heap (SimplePageWithPositionComparator.java:45-70).
// Device side here: every page is cut down to the rows whose FIRST sort key is not beyond the N-th best first key seen
// so far (topn_kernels.hip: order-preserving 64-bit key, radix selection, stable compaction + gathers); only those rows
// -- at least N per page at first, fewer and fewer later -- cross to the host, which holds the survivors and does the
// exact multi-channel comparison (all sort channels, NULL placement, ties in arrival order) at the end.  Rows tied with
// the N-th on the first key are kept, so the result is exact; the reference leaves the order of fully tied rows open.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "operator.hpp"
#include "scan_kernels.hpp"
#include "topn_kernels.hpp"

namespace pa {

namespace {

struct HostColumn {
    int32_t type = PA_BIGINT;
    std::vector<uint8_t> values;    // fixed width elements, or VARCHAR bytes
    std::vector<int32_t> offsets;   // VARCHAR: rows + 1 entries
    std::vector<uint8_t> nulls;     // one per row
};

class TopNOperator : public pa_operator {
public:
    explicit TopNOperator(const pa_topn_desc* d) : stream_(d->stream)
    {
        require_device();
        PA_REQUIRE(d->input_channel_count > 0 && d->input_types, PA_ERR_INVALID_ARGUMENT, "TopN needs input types");
        PA_REQUIRE(d->n >= 0, PA_ERR_INVALID_ARGUMENT, "n must be positive");  // TopNOperator.java:62
        PA_REQUIRE(d->sort_channel_count > 0 && d->sort_channels && d->sort_orders, PA_ERR_INVALID_ARGUMENT, "TopN needs sort channels");
        types_.assign(d->input_types, d->input_types + d->input_channel_count);
        sort_channels_.assign(d->sort_channels, d->sort_channels + d->sort_channel_count);
        sort_orders_.assign(d->sort_orders, d->sort_orders + d->sort_channel_count);
        for (size_t i = 0; i < sort_channels_.size(); i++) {
            PA_REQUIRE(sort_channels_[i] >= 0 && sort_channels_[i] < (int)types_.size(), PA_ERR_INVALID_ARGUMENT, "sort channel out of range");
            PA_REQUIRE(sort_orders_[i] >= 0 && sort_orders_[i] <= 3, PA_ERR_INVALID_ARGUMENT, "unknown sort order");
        }
        n_ = d->n;
        output_mem_ = d->output_mem;
        store_.resize(types_.size());
        for (size_t c = 0; c < types_.size(); c++) {
            store_[c].type = types_[c];
            if (types_[c] == PA_VARCHAR) store_[c].offsets.push_back(0);
        }
        h_hist_ = static_cast<uint32_t*>(h_hist_buf_.ensure(256 * 4 + 64));
        finishing_ = n_ == 0;  // TopNOperator.java:93-95: LIMIT 0 is finished from the start
        output_done_ = n_ == 0;
    }
    ~TopNOperator() override { (void)hipStreamSynchronize(stream_.get()); }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    bool needs_input() override { return !finishing_; }

    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(page != nullptr, PA_ERR_INVALID_ARGUMENT, "page is null");
        PA_REQUIRE(page->channel_count == (int32_t)types_.size(), PA_ERR_INVALID_ARGUMENT, "page channel count does not match the operator's input types");
        if (page->position_count == 0) return;
        hipStream_t s = stream_.get();
        DevPage dp = stager_.stage(page, nullptr, s);
        const int64_t n = dp.n;
        for (size_t c = 0; c < types_.size(); c++) PA_REQUIRE(dp.cols[c].type == types_[c], PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared input type");
        const DevColumn& first = dp.cols[sort_channels_[0]];
        timer.begin(s);
        if (add_filtered(dp, first, n)) {
            timer.end(s);
            return;
        }
        // the exact way: a key per row, radix selection of the page's N-th best, ties refined channel by channel
        uint64_t* keys = static_cast<uint64_t*>(keys_.ensure((size_t)n * 8));
        launch_topn_keys(first.type, first.values, first.offsets, first.nulls, n, sort_orders_[0], keys, s);
        // the N-th best first key of this page alone bounds the N-th best overall
        if (n >= n_) threshold_ = std::min(threshold_, topn_select_kth(keys, n, n_, select_temp_.ensure(topn_select_temp_bytes()), h_hist_, s));
        const int32_t* positions = nullptr;
        int64_t count = n;
        if (threshold_ != ~0ULL) {
            int32_t* part = static_cast<int32_t*>(part_.ensure((size_t)n * 4));
            int32_t* pos = static_cast<int32_t*>(pos_.ensure((size_t)n * 4));
            int64_t* counts = static_cast<int64_t*>(counts_.ensure(64));
            launch_topn_flag(keys, n, threshold_, part, s);
            if (n >= n_) refine_ties(dp, keys, n, part, s);
            launch_partition_positions(part, n, 2, pos, counts, part_temp_.ensure(partition_temp_bytes(n, 2)), s);
            int64_t h_counts[2];
            PA_HIP(hipMemcpyAsync(h_counts, counts, 16, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            count = h_counts[0];
            positions = pos;
        }
        timer.end(s);
        if (count == 0) return;
        append_rows(dp, keys, positions, count);
        prune();
    }

    void finish() override { finishing_ = true; }
    bool is_finished() override { return finishing_ && output_done_; }

    bool get_output(pa_page* out) override
    {
        if (!finishing_ || output_done_) return false;
        output_done_ = true;
        const int64_t rows = (int64_t)store_keys_.size();
        if (rows == 0) return false;
        // (rows beyond the bound cannot be among the N best: they need not be compared at all -- the bound is the exact N-th best
        // first key whenever a page went the sampled way, and a few thousand candidates then shrink to N and its ties)
        std::vector<int64_t> order;
        order.reserve((size_t)rows);
        for (int64_t i = 0; i < rows; i++) {
            if (store_keys_[(size_t)i] <= threshold_) order.push_back(i);
        }
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return compare_rows(a, b) < 0; });
        const int64_t m = std::min<int64_t>((int64_t)order.size(), n_);
        build_output(order, m);
        publish_output(out_cols_, (int32_t)m, output_mem_, stream_.get(), out, out_storage_);
        return true;
    }

    int64_t memory_bytes() override { return (int64_t)(stager_.bytes() + keys_.capacity() + keys2_.capacity() + state_.capacity() + part_.capacity() + pos_.capacity()); }

private:
    // The common case in ONE pass over the first sort channel: a bound for the page is drawn from a sample of its rows -- the
    // order statistic of 2^14 evenly spaced rows that lies above the page's N-th best key with overwhelming probability -- and
    // the rows not beyond min(that, the bound carried over from earlier pages) are collected as they are met, a few thousand
    // of 2^26 (the exact way walks the page a dozen times: keys, two histogram passes, compaction, flags, states, partition --
    // 34 G rows/s for the 100 best of 2^26 rows).  The exact N-th best key of everything kept so far then becomes the next
    // page's bound.  Whatever the sample cannot vouch for goes the exact way: fewer than N rows under a bound that came from
    // the sample (the sample was unlucky), or more rows than the sample promised (ties on the first key: refine_ties).
    // true: the page was taken.
    bool add_filtered(const DevPage& dp, const DevColumn& first, int64_t n)
    {
        if (getenv("PRESTO_AMD_TOPN_EXACT")) return false;
        hipStream_t s = stream_.get();
        // 2^14 sample rows (a workgroup's registers hold that many keys: sample and selection are one launch, topn_kernels.hip);
        // 2^16 for the largest pages, where a four times tighter bound saves more host work on the candidates than the second
        // launch and its walks over the sample array cost (2^26 rows: 0.45 vs 0.70 ms per page)
        const int64_t kSample = n >= ((int64_t)1 << 25) ? (int64_t)1 << 16 : (int64_t)1 << 14;
        bool sampled = false;
        int64_t expect = 0;
        uint32_t* counter = static_cast<uint32_t*>(counts_.ensure(256)) + 32;  // [count, pad, bound (8 bytes)]
        uint64_t* device_bound = reinterpret_cast<uint64_t*>(counter + 2);
        if (n >= n_ && n >= 4 * kSample) {
            const double j = (double)n_ * (double)kSample / (double)n;
            const int64_t rank = (int64_t)std::ceil(j + 8.0 + 4.0 * std::sqrt(j));
            if (rank <= kSample / 8) {
                launch_topn_sample_bound(first.type, first.values, first.offsets, first.nulls, n, sort_orders_[0], kSample, rank,
                                         static_cast<uint64_t*>(keys2_.ensure((size_t)kSample * 8)), device_bound, s);
                sampled = true;
                expect = rank * (n / kSample + 1);
            }
        }
        auto tighten = [&] {
            // the exact N-th best first key of everything kept so far bounds every later page (and what prune keeps)
            if ((int64_t)store_keys_.size() >= n_) {
                std::vector<uint64_t> k(store_keys_);
                std::nth_element(k.begin(), k.begin() + (n_ - 1), k.end());
                threshold_ = std::min(threshold_, k[(size_t)n_ - 1]);
            }
            prune();
        };
        if (!sampled && threshold_ == ~0ULL) {
            if (n > ((int64_t)1 << 14)) return false;  // nothing bounds the page
            // a small first page (an upstream operator that already cut its output down, Q3's hinted aggregation): every row is a
            // candidate -- one kernel for the keys, one copy to the host -- instead of a selection over a few thousand rows
            uint64_t* all = static_cast<uint64_t*>(keys_.ensure((size_t)n * 8));
            launch_topn_keys(first.type, first.values, first.offsets, first.nulls, n, sort_orders_[0], all, s);
            append_rows(dp, all, nullptr, n);
            tighten();
            return true;
        }
        const int64_t capacity = std::min<int64_t>(n, sampled ? 2 * expect + kSample : (int64_t)1 << 20);
        int32_t* positions = static_cast<int32_t*>(pos_.ensure((size_t)capacity * 4));
        uint64_t* keys = static_cast<uint64_t*>(keys_.ensure((size_t)capacity * 8));
        launch_topn_filter(first.type, first.values, first.offsets, first.nulls, n, sort_orders_[0], threshold_, sampled ? device_bound : nullptr,
                           (uint32_t)capacity, positions, keys, counter, s);
        struct Landed {
            uint32_t count, pad;
            uint64_t bound;
        };
        Landed* h = reinterpret_cast<Landed*>(h_hist_);
        PA_HIP(hipMemcpyAsync(h, counter, 16, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        const int64_t count = (int64_t)h->count;
        if (count > capacity) return false;
        // fewer than N rows are only the whole truth under the carried-over bound (which holds by construction)
        if (sampled && count < n_ && h->bound < threshold_) return false;
        if (count == 0) return true;
        // (the matches were collected as the waves met them: every row takes its arrival number along -- store_seq_ -- and fully
        // tied rows are put in arrival order by the final comparison, not by the order they are stored in)
        append_rows(dp, keys, positions, count, true);
        tighten();
        return true;
    }

    // Many rows tied with the bound on the first sort key (ORDER BY a low-cardinality column ... LIMIT n; all-equal keys): they
    // would all cross to the host and be sorted there -- 12 M tied rows took 5.8 s.  A page contributes at most its OWN n best
    // rows to the result, so the page's ties are cut down on the device: among the rows tied so far, the next sort channel's
    // keys are selected from the same way, channel after channel, and rows tied on every channel are kept in arrival order (what
    // the host's stable sort would do).  A VARCHAR channel ends the refinement (its key is the first 8 bytes: equal keys need not
    // be equal strings), keeping the remaining ties whole.  Rewrites `part` (0 = kept).
    void refine_ties(const DevPage& dp, const uint64_t* keys0, int64_t n, int32_t* part, hipStream_t s)
    {
        if (types_[(size_t)sort_channels_[0]] == PA_VARCHAR) return;
        uint8_t* state = static_cast<uint8_t*>(state_.ensure((size_t)n));
        int64_t* dcounts = static_cast<int64_t*>(counts_.ensure(64)) + 4;
        launch_topn_state(keys0, n, threshold_, true, state, s);
        int64_t h[2];
        auto count = [&] {
            launch_topn_count_states(state, n, dcounts, s);
            PA_HIP(hipMemcpyAsync(h, dcounts, 16, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
        };
        count();
        int64_t need = n_ - h[0];   // ties still wanted
        if (need <= 0 || h[1] <= std::max<int64_t>(4 * n_, 1 << 16) || h[1] <= need) return;  // few ties: the host sorts them out
        uint64_t* keys = nullptr;
        for (size_t level = 1; level < sort_channels_.size() && h[1] > need; level++) {
            const DevColumn& col = dp.cols[(size_t)sort_channels_[level]];
            if (col.type == PA_VARCHAR) {
                need = h[1];  // keep every remaining tie
                break;
            }
            keys = static_cast<uint64_t*>(keys2_.ensure((size_t)n * 8));
            launch_topn_keys(col.type, col.values, col.offsets, col.nulls, n, sort_orders_[level], keys, s);
            launch_topn_mask_keys(state, n, keys, s);
            const uint64_t thr = topn_select_kth(keys, n, need, select_temp_.ensure(topn_select_temp_bytes()), h_hist_, s);
            launch_topn_state(keys, n, thr, false, state, s);
            count();
            need = n_ - h[0];
        }
        int32_t* flags = static_cast<int32_t*>(tie_rank_.ensure((size_t)(n + 1) * 4));
        launch_topn_tie_flags(state, n, flags, s);
        launch_exclusive_scan_i32(flags, flags, n, nullptr, scan_temp_.ensure(scan_temp_bytes(n)), s);
        launch_topn_state_partition(state, flags, n, std::max<int64_t>(need, 0), part, s);
    }

    // selected rows of the page -> host store (Block.copyPositions on device, then one D2H per column)
    // (keys_compacted: keys[i] belongs to positions[i]; else keys is the page's key array)
    void append_rows(const DevPage& dp, const uint64_t* keys, const int32_t* positions, int64_t count, bool keys_compacted = false)
    {
        hipStream_t s = stream_.get();
        const int64_t n = dp.n;
        auto fetch = [&](const void* dev, size_t bytes, std::vector<uint8_t>& dst) {
            void* land = land_.ensure(bytes ? bytes : 1);
            if (bytes) PA_HIP(hipMemcpyAsync(land, dev, bytes, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            dst.insert(dst.end(), static_cast<uint8_t*>(land), static_cast<uint8_t*>(land) + bytes);
        };
        // keys, NULL flags and fixed-width values of the selected rows: gathered side by side into one buffer, ONE copy to the host
        // (a copy and a wait per column was 0.1 ms of Q3's TopN)
        struct Piece {
            size_t at, bytes;
        };
        std::vector<Piece> pieces;
        size_t total_bytes = 0;
        auto reserve = [&](size_t bytes) {
            pieces.push_back(Piece{total_bytes, bytes});
            total_bytes += (bytes + 15) & ~(size_t)15;
            return pieces.size() - 1;
        };
        const size_t key_piece = reserve((size_t)count * 8);
        const size_t pos_piece = positions ? reserve((size_t)count * 4) : (size_t)-1;
        std::vector<size_t> null_piece(types_.size(), (size_t)-1), value_piece(types_.size(), (size_t)-1);
        for (size_t c = 0; c < types_.size(); c++) {
            const DevColumn& col = dp.cols[c];
            if (col.nulls) null_piece[c] = reserve((size_t)count);
            if (!col.varwidth) value_piece[c] = reserve((size_t)count * type_width(col.type));
        }
        uint8_t* side = static_cast<uint8_t*>(gather_.ensure(total_bytes ? total_bytes : 1));
        auto place = [&](size_t piece, const void* src, int width) {
            void* dst = side + pieces[piece].at;
            if (positions) {
                if (width == 0) launch_gather_nulls(static_cast<const uint8_t*>(src), positions, count, static_cast<uint8_t*>(dst), s);
                else launch_gather_flat(src, width, positions, count, dst, s);
            }
            else if (pieces[piece].bytes) {
                PA_HIP(hipMemcpyAsync(dst, src, pieces[piece].bytes, hipMemcpyDeviceToDevice, s));
            }
        };
        if (keys_compacted) PA_HIP(hipMemcpyAsync(side + pieces[key_piece].at, keys, (size_t)count * 8, hipMemcpyDeviceToDevice, s));
        else place(key_piece, keys, 8);
        if (positions) PA_HIP(hipMemcpyAsync(side + pieces[pos_piece].at, positions, (size_t)count * 4, hipMemcpyDeviceToDevice, s));
        for (size_t c = 0; c < types_.size(); c++) {
            const DevColumn& col = dp.cols[c];
            if (col.nulls) place(null_piece[c], col.nulls, 0);
            if (!col.varwidth) place(value_piece[c], col.values, type_width(col.type));
        }
        uint8_t* landed = static_cast<uint8_t*>(side_land_.ensure(total_bytes ? total_bytes : 1));  // (land_ is the VARCHAR columns')
        if (total_bytes) PA_HIP(hipMemcpyAsync(landed, side, total_bytes, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        {
            const size_t old = store_keys_.size();
            store_keys_.resize(old + (size_t)count);
            memcpy(store_keys_.data() + old, landed + pieces[key_piece].at, (size_t)count * 8);
            // arrival number of every row: (page, position in the page)
            store_seq_.resize(old + (size_t)count);
            const int32_t* hp = positions ? reinterpret_cast<const int32_t*>(landed + pieces[pos_piece].at) : nullptr;
            for (int64_t i = 0; i < count; i++) store_seq_[old + (size_t)i] = (pages_seen_ << 32) | (uint64_t)(uint32_t)(hp ? hp[i] : (int32_t)i);
            pages_seen_++;
        }
        for (size_t c = 0; c < types_.size(); c++) {
            const DevColumn& col = dp.cols[c];
            HostColumn& hc = store_[c];
            if (col.nulls) hc.nulls.insert(hc.nulls.end(), landed + pieces[null_piece[c]].at, landed + pieces[null_piece[c]].at + (size_t)count);
            else hc.nulls.insert(hc.nulls.end(), (size_t)count, 0);
            if (!col.varwidth) {
                const Piece& p = pieces[value_piece[c]];
                hc.values.insert(hc.values.end(), landed + p.at, landed + p.at + p.bytes);
                continue;
            }
            // VARCHAR: lengths -> exclusive scan -> byte copy (as FilterAndProject's copyPositions), or the whole block
            std::vector<uint8_t> raw_off, raw_bytes;
            if (positions) {
                int32_t* lens = static_cast<int32_t*>(var_off_.ensure((size_t)(count + 1) * 4));
                int32_t* total = static_cast<int32_t*>(counts_.ensure(64)) + 8;
                launch_varwidth_lengths(positions, count, col.offsets, col.nulls, lens, s);
                launch_exclusive_scan_i32(lens, lens, count, total, scan_temp_.ensure(scan_temp_bytes(count)), s);
                int32_t h_total = 0;
                PA_HIP(hipMemcpyAsync(&h_total, total, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
                uint8_t* bytes = static_cast<uint8_t*>(var_bytes_.ensure((size_t)(h_total > 0 ? h_total : 1)));
                launch_varwidth_copy(positions, count, col.offsets, static_cast<const uint8_t*>(col.values), col.nulls, lens, bytes, total, s);
                fetch(lens, (size_t)(count + 1) * 4, raw_off);
                fetch(bytes, (size_t)h_total, raw_bytes);
            }
            else {
                fetch(col.offsets, (size_t)(n + 1) * 4, raw_off);
                const int32_t* o = reinterpret_cast<const int32_t*>(raw_off.data());
                const int32_t lo = o[0], hi = o[n];
                fetch(static_cast<const uint8_t*>(col.values) + lo, (size_t)(hi - lo), raw_bytes);
            }
            const int32_t* o = reinterpret_cast<const int32_t*>(raw_off.data());
            const int32_t base = (int32_t)hc.values.size() - o[0];
            for (int64_t i = 1; i <= count; i++) hc.offsets.push_back(o[i] + base);
            hc.values.insert(hc.values.end(), raw_bytes.begin(), raw_bytes.end());
        }
    }

    // keeps the host store bounded: the N-th smallest first key of the survivors tightens the threshold
    void prune()
    {
        const size_t rows = store_keys_.size();
        if ((int64_t)rows < std::max<int64_t>(8 * n_, 1 << 20)) return;
        std::vector<uint64_t> k(store_keys_);
        std::nth_element(k.begin(), k.begin() + (n_ - 1), k.end());
        threshold_ = std::min(threshold_, k[(size_t)n_ - 1]);
        std::vector<int64_t> keep;
        for (size_t i = 0; i < rows; i++) {
            if (store_keys_[i] <= threshold_) keep.push_back((int64_t)i);
        }
        if (keep.size() == rows) return;
        std::vector<uint64_t> nk(keep.size());
        std::vector<uint64_t> ns(keep.size());
        for (size_t i = 0; i < keep.size(); i++) {
            nk[i] = store_keys_[(size_t)keep[i]];
            ns[i] = store_seq_[(size_t)keep[i]];
        }
        store_keys_.swap(nk);
        store_seq_.swap(ns);
        for (auto& hc : store_) {
            HostColumn out;
            out.type = hc.type;
            copy_rows(hc, keep, (int64_t)keep.size(), out);
            hc = std::move(out);
        }
    }

    static void copy_rows(const HostColumn& src, const std::vector<int64_t>& rows, int64_t m, HostColumn& dst)
    {
        dst.nulls.resize((size_t)m);
        if (src.type == PA_VARCHAR) {
            dst.offsets.assign(1, 0);
            for (int64_t i = 0; i < m; i++) {
                const size_t r = (size_t)rows[(size_t)i];
                dst.nulls[(size_t)i] = src.nulls[r];
                dst.values.insert(dst.values.end(), src.values.begin() + src.offsets[r], src.values.begin() + src.offsets[r + 1]);
                dst.offsets.push_back((int32_t)dst.values.size());
            }
            return;
        }
        const size_t w = (size_t)type_width(src.type);
        dst.values.resize((size_t)m * w);
        for (int64_t i = 0; i < m; i++) {
            const size_t r = (size_t)rows[(size_t)i];
            dst.nulls[(size_t)i] = src.nulls[r];
            memcpy(&dst.values[(size_t)i * w], &src.values[r * w], w);
        }
    }

    // SimplePageWithPositionComparator.compareTo (…/operator/SimplePageWithPositionComparator.java:45-70) with
    // SortOrder.compareBlockValue (core/trino-spi/src/main/java/io/trino/spi/connector/SortOrder.java:58-84)
    int compare_rows(int64_t a, int64_t b) const
    {
        for (size_t i = 0; i < sort_channels_.size(); i++) {
            const HostColumn& hc = store_[(size_t)sort_channels_[i]];
            const int order = sort_orders_[i];
            const bool ascending = order < 2, nulls_first = (order & 1) == 0;
            const bool an = hc.nulls[(size_t)a] != 0, bn = hc.nulls[(size_t)b] != 0;
            int c = 0;
            if (an && bn) c = 0;
            else if (an) c = nulls_first ? -1 : 1;
            else if (bn) c = nulls_first ? 1 : -1;
            else {
                c = compare_values(hc, a, b);
                if (!ascending) c = -c;
            }
            if (c != 0) return c;
        }
        // fully tied rows: arrival order (what a stable sort of the rows as they came would leave)
        return store_seq_[(size_t)a] < store_seq_[(size_t)b] ? -1 : (store_seq_[(size_t)a] > store_seq_[(size_t)b] ? 1 : 0);
    }

    static int compare_values(const HostColumn& hc, int64_t a, int64_t b)
    {
        switch (hc.type) {
            case PA_BIGINT: {
                int64_t x, y;
                memcpy(&x, &hc.values[(size_t)a * 8], 8);
                memcpy(&y, &hc.values[(size_t)b * 8], 8);
                return x < y ? -1 : (x > y ? 1 : 0);
            }
            case PA_INTEGER:
            case PA_DATE: {
                int32_t x, y;
                memcpy(&x, &hc.values[(size_t)a * 4], 4);
                memcpy(&y, &hc.values[(size_t)b * 4], 4);
                return x < y ? -1 : (x > y ? 1 : 0);
            }
            case PA_BOOLEAN: return (int)(hc.values[(size_t)a] != 0) - (int)(hc.values[(size_t)b] != 0);
            case PA_DOUBLE: {
                // Double.compare: numeric order, -0.0 < 0.0, NaN (one value) above everything
                auto image = [&](int64_t r) {
                    double d;
                    memcpy(&d, &hc.values[(size_t)r * 8], 8);
                    uint64_t bits;
                    if (d != d) bits = 0x7ff8000000000000ULL;
                    else memcpy(&bits, &d, 8);
                    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ULL);
                };
                const uint64_t x = image(a), y = image(b);
                return x < y ? -1 : (x > y ? 1 : 0);
            }
            case PA_REAL: {  // Float.compare (RealType.comparisonOperator): the order of the widened values' images
                auto image = [&](int64_t r) {
                    float f;
                    memcpy(&f, &hc.values[(size_t)r * 4], 4);
                    const double d = (double)f;
                    uint64_t bits;
                    if (d != d) bits = 0x7ff8000000000000ULL;
                    else memcpy(&bits, &d, 8);
                    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ULL);
                };
                const uint64_t x = image(a), y = image(b);
                return x < y ? -1 : (x > y ? 1 : 0);
            }
            case PA_VARCHAR: {
                const int32_t ao = hc.offsets[(size_t)a], al = hc.offsets[(size_t)a + 1] - ao;
                const int32_t bo = hc.offsets[(size_t)b], bl = hc.offsets[(size_t)b + 1] - bo;
                const int c = memcmp(hc.values.data() + ao, hc.values.data() + bo, (size_t)std::min(al, bl));  // Slice.compareTo
                if (c != 0) return c < 0 ? -1 : 1;
                return al < bl ? -1 : (al > bl ? 1 : 0);
            }
            default: return 0;
        }
    }

    void build_output(const std::vector<int64_t>& order, int64_t m)
    {
        hipStream_t s = stream_.get();
        out_cols_.clear();
        out_cols_.resize(types_.size());
        const bool to_device = output_mem_ == PA_MEM_DEVICE;
        for (size_t c = 0; c < types_.size(); c++) {
            HostColumn hc;
            hc.type = types_[c];
            copy_rows(store_[c], order, m, hc);
            OutColumn& oc = out_cols_[c];
            oc.type = types_[c];
            oc.varwidth = types_[c] == PA_VARCHAR;
            oc.has_nulls = std::any_of(hc.nulls.begin(), hc.nulls.end(), [](uint8_t v) { return v != 0; });
            const size_t bytes = hc.values.size();
            memcpy(oc.h_values.ensure(bytes ? bytes : 1), hc.values.data(), bytes);
            if (oc.varwidth) memcpy(oc.h_offsets.ensure(hc.offsets.size() * 4), hc.offsets.data(), hc.offsets.size() * 4);
            if (oc.has_nulls) memcpy(oc.h_nulls.ensure(hc.nulls.size()), hc.nulls.data(), hc.nulls.size());
            oc.host_ready = true;
            if (to_device) {
                oc.values.ensure(bytes ? bytes : 1);
                if (bytes) PA_HIP(hipMemcpyAsync(oc.values.ptr(), oc.h_values.ptr(), bytes, hipMemcpyHostToDevice, s));
                if (oc.varwidth) {
                    oc.offsets.ensure(hc.offsets.size() * 4);
                    PA_HIP(hipMemcpyAsync(oc.offsets.ptr(), oc.h_offsets.ptr(), hc.offsets.size() * 4, hipMemcpyHostToDevice, s));
                }
                if (oc.has_nulls) {
                    oc.nulls.ensure(hc.nulls.size());
                    PA_HIP(hipMemcpyAsync(oc.nulls.ptr(), oc.h_nulls.ptr(), hc.nulls.size(), hipMemcpyHostToDevice, s));
                }
            }
        }
        if (to_device) PA_HIP(hipStreamSynchronize(s));
    }

    Stream stream_;
    PageStager stager_;
    std::vector<int32_t> types_, sort_channels_, sort_orders_;
    int64_t n_ = 0;
    int32_t output_mem_ = PA_MEM_HOST;
    bool finishing_ = false, output_done_ = false;
    uint64_t threshold_ = ~0ULL;  // rows whose first-channel key is above it cannot be among the N best
    DevBuf keys_, part_, pos_, counts_, part_temp_, select_temp_, gather_, var_off_, var_bytes_, scan_temp_;
    DevBuf keys2_, state_, tie_rank_;  // refine_ties
    PinnedBuf land_, side_land_, h_hist_buf_, found_land_;
    uint32_t* h_hist_ = nullptr;
    std::vector<HostColumn> store_;
    std::vector<uint64_t> store_keys_;
    std::vector<uint64_t> store_seq_;   // arrival number of every stored row: (page << 32) | position
    uint64_t pages_seen_ = 0;
    std::vector<OutColumn> out_cols_;
    std::vector<pa_column> out_storage_;
};

}  // namespace

pa_operator* make_topn(const pa_topn_desc* desc)
{
    PA_REQUIRE(desc != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    return new TopNOperator(desc);
}

}  // namespace pa
