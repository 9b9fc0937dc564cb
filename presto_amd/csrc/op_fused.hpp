// op_fused.hpp -- the fused scan-filter-project-aggregate operator's class (host side): pages in, launches, the tiers' hand-overs, the
// Operator protocol with device work in flight.  See op_fused.cpp for what the operator replaces and the kernel tiers; the members that
// intern VARCHAR keys and rank min / max inputs are defined in op_fused_intern.cpp, the output (emit on the device, host assembly of
// small results) in op_fused_output.cpp.  Included by those three files only.
#pragma once

#include <atomic>
#include <algorithm>
#include <cmath>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>

#include "decimal_host.hpp"
#include "exchange_kernels.hpp"
#include "exprgen.hpp"
#include "host_hash.hpp"
#include "jit.hpp"
#include "join_source.hpp"
#include "operator.hpp"
#include "fused_plan.hpp"
#include "rowgen.hpp"
#include "scan_kernels.hpp"
#include "static_kernels.hpp"
#include "topn_kernels.hpp"
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

using namespace fused;

namespace fused_op {


inline uint32_t next_pow2(uint64_t v)
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
        // min / max by rank (Spec::ranked): what the operator has accumulated is re-ranked when a page brings new strings, and the
        // HBM table (or the ungrouped state) is where that is done -- the tiers with tables of their own stay out
        if (spec_.any_ranked() && grouped_) mode_ = V_GT;
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
        release_everything();  // retained pages the operator still holds go back to their owner: nothing reads them any more
        for (hipEvent_t e : release_events_) (void)hipEventDestroy(e);
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
    // The consumer of this operator's output is a TopN(n; sort channels / orders over the OUTPUT channels): groups that cannot be
    // among its n best rows may be left out (set_output_topn, pa_aggregation_set_output_topn_hint).
    struct TopNHint {
        int64_t n = 0;
        std::vector<int32_t> channels, orders;
    } topn_hint_;
    bool set_output_topn(int64_t n, const int32_t* channels, const int32_t* orders, int32_t count) override
    {
        if (n <= 0 || count <= 0 || !grouped_ || out_partial_ || spec_.step == PA_STEP_PARTIAL) return false;
        topn_hint_.n = n;
        topn_hint_.channels.assign(channels, channels + count);
        topn_hint_.orders.assign(orders, orders + count);
        return true;
    }

    bool lookup_source_ready() const { return !spec_.join || spec_.join->ls->built.load(); }

    bool takes_retained() override { return true; }

    bool needs_input() override
    {
        poll_releases();
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

    // A page whose buffers stay valid after add_input returns: PA_PAGE_STABLE (until the operator is closed) or PA_PAGE_RETAINED
    // (until the operator calls the page's release -- see "retained pages" below)
    static bool page_stays(const pa_page* page) { return (page->flags & (PA_PAGE_STABLE | PA_PAGE_RETAINED)) != 0; }

    void take_page(const pa_page* page)
    {
        const uint64_t launched = timer.begun();
        try {
            if (next_) {
                next_->add_input(page);  // (the generation in charge registers the page's release itself)
                return;
            }
            cur_rel_set_ = (page->flags & PA_PAGE_RETAINED) != 0 && page->release != nullptr;
            cur_rel_ = Release{page->release, page->release_ctx};
            if (gather_small_page(page)) return;
            flush_pending();
            if (next_) {  // the flush met a layout change and started the next generation: the page belongs there
                cur_rel_set_ = false;
                next_->add_input(page);
                return;
            }
            process_page(page, page_stays(page) && page->mem == PA_MEM_DEVICE);
            if (cur_rel_set_) {  // launched as a page of its own: released once its launches are done and confirmed
                std::vector<Release> rel{cur_rel_};
                cur_rel_set_ = false;
                release_checkpoint(std::move(rel), true);
            }
        }
        catch (const PoolExhausted& e) {
            // only a page that can be read again later, and of which nothing has been launched, can wait
            if (!page_stays(page) || timer.begun() != launched || is_combiner_) throw;
            parked_page_ = *page;
            // (a release that already travels with a pending structure is not registered again when the page is taken up again)
            if (!cur_rel_set_) parked_page_.flags &= ~PA_PAGE_RETAINED;
            cur_rel_set_ = false;
            parked_cols_.assign(page->columns, page->columns + page->channel_count);
            parked_need_ = e.bytes;
            parked_ = true;
        }
    }

    // ---- retained pages ---------------------------------------------------------------------------------------------------
    // PA_PAGE_RETAINED: the page's owner keeps its buffers valid and unchanged until this operator calls page->release(ctx) -- what a
    // reference does for a Java Page.  Such a page is taken like a stable one (merged into a range, listed in a range table, copied
    // into the arena at the arena's launch -- no launch and no wait per page); its release travels with the structure the page went
    // to and is called, from inside a later call on this handle, once the launches that read the page have finished AND been
    // confirmed (a few-groups launch that met too many groups is redone from the same buffers, confirm_oldest).
    struct Release {
        void (*fn)(void*) = nullptr;
        void* ctx = nullptr;
    };
    struct ReleaseBatch {
        hipEvent_t event = nullptr;
        uint64_t seq = 0;            // 0: the event alone decides; else every few-groups launch up to this one must be confirmed
        std::vector<Release> rel;
    };
    Release take_cur_release()
    {
        cur_rel_set_ = false;
        return cur_rel_;
    }
    void release_checkpoint(std::vector<Release> rel, bool needs_confirm)
    {
        if (rel.empty()) return;
        ReleaseBatch b;
        if (release_events_.empty()) PA_HIP(hipEventCreateWithFlags(&b.event, hipEventDisableTiming));
        else {
            b.event = release_events_.back();
            release_events_.pop_back();
        }
        PA_HIP(hipEventRecord(b.event, stream_.get()));
        b.seq = needs_confirm ? launch_seq_ : 0;
        b.rel = std::move(rel);
        release_batches_.push_back(std::move(b));
    }
    void poll_releases()
    {
        while (!release_batches_.empty()) {
            ReleaseBatch& b = release_batches_.front();
            if (b.seq != 0 && !inflight_.empty() && inflight_.front().seq <= b.seq) break;
            if (hipEventQuery(b.event) != hipSuccess) {
                (void)hipGetLastError();  // hipErrorNotReady is not an error here
                break;
            }
            for (const Release& r : b.rel) r.fn(r.ctx);
            release_events_.push_back(b.event);
            release_batches_.pop_front();
        }
    }
    // every release the operator still owes, now: the caller has made sure nothing reads the pages any more (stream drained)
    void release_everything()
    {
        for (ReleaseBatch& b : release_batches_) {
            for (const Release& r : b.rel) r.fn(r.ctx);
            release_events_.push_back(b.event);
        }
        release_batches_.clear();
        auto fire = [](std::vector<Release>& v) {
            for (const Release& r : v) r.fn(r.ctx);
            v.clear();
        };
        fire(run_rel_);
        fire(ranges_rel_);
        fire(carry_rel_);
        fire(arena_[0].rel);
        fire(arena_[1].rel);
        if (cur_rel_set_) {
            cur_rel_set_ = false;
            cur_rel_.fn(cur_rel_.ctx);
        }
    }

    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(page != nullptr, PA_ERR_INVALID_ARGUMENT, "page is null");
        PA_REQUIRE(page->channel_count == spec_.n_in, PA_ERR_INVALID_ARGUMENT, "page channel count does not match the operator's input types");
        poll_releases();
        if (page->position_count == 0) {
            if ((page->flags & PA_PAGE_RETAINED) != 0 && page->release) page->release(page->release_ctx);
            return;
        }
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
    // stable ranges below this many rows wait for each other in a range table (one launch for all of them); a range of 2^24 rows
    // -- 0.5 GB of Q6 columns -- pays for a launch and its merges of its own.  (Measured, profiles/r04_page_sweep.md: with the limit at
    // 2^21 rows, pages of 2^22 rows that do not continue each other got a launch each and ran at half the rate of 2^20-row pages.)
    static constexpr int64_t kRangeTableRows = (int64_t)1 << 24;
    static int64_t gather_rows()
    {
        const char* e = getenv("PRESTO_AMD_GATHER_ROWS");  // tests and sweeps move the launch threshold
        return e ? std::max<int64_t>(strtoll(e, nullptr, 10), 1) : (int64_t)1 << 26;
    }
    static constexpr int64_t kArenaRows = (int64_t)1 << 22;

    bool channel_plain(const pa_column& col, int c) const
    {
        if (col.encoding == PA_FLAT) return col.type == spec_.in_types[c] || spec_.derived(c);
        return col.encoding == PA_VARWIDTH;
    }

    // true: the page was taken (merged into the pending range / copied into the arena)
    bool gather_small_page(const pa_page* page)
    {
        const int64_t n = page->position_count;
        const bool stable_dev = page_stays(page) && page->mem == PA_MEM_DEVICE;
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
                if (cur_rel_set_) run_rel_.push_back(take_cur_release());
                if (run_.rows >= gather_rows()) flush_pending();
                return true;
            }
            retire_run();
            if (next_) return false;  // (add_input hands the page to the generation a flush started)
        }
        bool plain = true, flat = true;
        for (int c = 0; c < spec_.n_in && plain; c++) {
            if (!spec_.used_channel[c]) continue;
            plain = channel_plain(page->columns[c], c) && !spec_.derived(c);
            flat = flat && page->columns[c].encoding == PA_FLAT;
        }
        if (!plain) return false;
        if (stable_dev && n < gather_rows()) {
            // a range starts here: whatever its size, the next page may continue it
            run_.rows = n;
            run_.flat = flat;
            run_.cols.assign(page->columns, page->columns + page->channel_count);
            if (cur_rel_set_) run_rel_.push_back(take_cur_release());
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
        if (run_.rows < kRangeTableRows && ranges_possible()) {
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
            ranges_rel_.insert(ranges_rel_.end(), run_rel_.begin(), run_rel_.end());
            run_rel_.clear();
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
        carry_rel_.insert(carry_rel_.end(), run_rel_.begin(), run_rel_.end());  // the releases of the run's pages go where its rows go
        run_rel_.clear();
        append_to_arena(&sp);
    }

    // stable device ranges can be handed over as a table when the tier in charge has a kernel for it
    bool ranges_possible() const
    {
        if (spec_.join || getenv("PRESTO_AMD_NO_RANGES")) return false;
        if (mode_ != V_GLOBAL && mode_ != V_LDS) return false;
        for (int c = 0; c < spec_.n_in; c++) {
            if (spec_.used_channel[c] && spec_.derived(c)) return false;
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
        std::vector<Release> rel;
        rel.swap(ranges_rel_);
        const uint64_t launched = timer.begun();
        try {
            process_ranges(set, rows);
        }
        catch (const PoolExhausted&) {
            // nothing of the table was launched: it waits with the page that is being parked (take_page)
            if (timer.begun() == launched && !next_) {
                ranges_ = std::make_shared<std::vector<DevPage>>(*set);
                range_rows_ = rows;
                ranges_rel_.swap(rel);
            }
            else release_checkpoint(std::move(rel), true);
            throw;
        }
        release_checkpoint(std::move(rel), true);
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
        std::vector<Release> rel;
        rel.swap(run_rel_);
        try {
            process_page(&sp, true);
        }
        catch (...) {
            release_checkpoint(std::move(rel), true);  // (whatever of the range was launched is in the stream in front of the event)
            throw;
        }
        release_checkpoint(std::move(rel), true);
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
            if (next_) {  // the flush started the next generation
                std::vector<Release> carried;
                carried.swap(carry_rel_);
                // the page itself: the generation registers its release (it sees the flag); a gathered run's releases cannot travel
                // through add_input -- they are called once the generation has launched and confirmed the run's rows
                if (cur_rel_set_ && (page->flags & PA_PAGE_RETAINED) != 0 && page->release == cur_rel_.fn && page->release_ctx == cur_rel_.ctx) cur_rel_set_ = false;
                next_->add_input(page);
                if (!carried.empty()) {
                    next_->flush_pending();
                    next_->confirm_all();
                    release_checkpoint(std::move(carried), false);
                }
                return;
            }
            return append_to_arena(page);
        }
        // a pageable host page: its arrays go behind each other into pinned memory, and ONE launch reads them from there (a copy per
        // array is ~4 us of enqueueing each)
        bool pinned_copy = false;
        if (page->mem == PA_MEM_HOST && (page->flags & PA_PAGE_PINNED) == 0) {
            if (const pa_page* pinned = pinned_copy_.copy(page, &spec_.used_channel)) {
                page = pinned;
                pinned_copy = true;
            }
        }
        const bool host = page->mem == PA_MEM_HOST;
        const bool readable = !host || (page->flags & PA_PAGE_PINNED) != 0;  // the device can read the page's buffers itself
        const bool defer = readable && page_stays(page);                       // ... and they stay: copy at the arena's launch
        if (cur_rel_set_) a.rel.push_back(take_cur_release());
        a.rel.insert(a.rel.end(), carry_rel_.begin(), carry_rel_.end());
        carry_rel_.clear();
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
        if (pinned_copy) pinned_copy_.used(s);
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
        if (ranges_ && !ranges_->empty() && run_.rows > 0 && run_.rows < kRangeTableRows && ranges_possible()) retire_run();
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
        {
            // the arena holds copies: the pages it was filled from are free once the copies in the stream have run
            std::vector<Release> rel;
            rel.swap(a.rel);
            release_checkpoint(std::move(rel), false);
        }
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
        // VARCHAR channels gathered from host pages: the host placed the bytes, so it knows how many there are (StringInterner::intern)
        struct HintScope {
            std::vector<int64_t>& hint;
            ~HintScope() { hint.clear(); }
        } hint_scope{var_bytes_hint_};
        if (!a.dev_var) var_bytes_hint_.assign(a.bytes.begin(), a.bytes.end());
        a.rows = 0;
        arena_cur_ ^= 1;
        // the arena is this operator's own: its rows stay put until the launches on it are confirmed -- the other arena
        // takes the next pages, and is only written again after this one's launches were confirmed (kMaxInflight = 2)
        process_page(&sp, true);
    }

    void add_page(const pa_page* page)
    {
        HostTraceScope trace("  fused.add_page");
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
        rank_values(dp, s);
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
        poll_releases();
    }
    bool is_finished() override { return finishing_ && output_done_; }

    bool get_output(pa_page* out) override
    {
        poll_releases();
        if (!finishing_ || output_done_) return false;
        output_done_ = true;
        confirm_all();
        if (next_) return combine_generations(out);
        build_output();
        if (grouped_ && out_rows_ > 0) decode_interned_keys();
        if (!grouped_ || out_rows_ > 0) {
            publish_output(out_cols_, out_rows_, spec_.output_mem, stream_.get(), out, out_storage_);
            poll_releases();
            return true;
        }
        poll_releases();
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
                // (min / max over VARCHAR: strings of <= 7 bytes as their image, any other string by rank -- its state is the string)
                if (ranked_channel(value_proj) >= 0) add_channel(PA_VARCHAR, 0);
                else add_channel(t, t == PA_VARCHAR ? 7 : 0);
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
        // V_LDS: what the first launch of earlier operators of this plan found -- 1: every wave's register table held its groups,
        // 2: one overflowed.  An operator that finds 1 here does not cut a short probe launch off its first page and wait for it:
        // the whole page goes out and is confirmed late, like every later page (a wrong guess is the redo path of confirm_oldest).
        mutable std::atomic<int> lds_verdict{0};
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
        HostTraceScope trace("    fused.kernel_for");
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
        if (variant == V_LDS) c->tail_kernel = jit_get(c->info.source, c->info.entry + "_tail");
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

    // Groups of the whole input, from the d distinct keys among the first n rows, as if the keys were drawn uniformly from G
    // values: d = G (1 - exp(-n / G)).  (Skewed keys make it an overestimate; it only ever chooses between tiers.)
    static constexpr int64_t kProbeRows = (int64_t)1 << 18;
    static constexpr uint32_t kLdshReplicas = 4;
    static uint64_t estimate_groups(uint64_t d, uint64_t n)
    {
        if (d == 0 || n == 0 || d * 8 < n) return d;  // most rows repeat a key already seen: d is about all there is
        if (d * 100 >= n * 98) return 32 * n;         // nearly every row a new key: "many" is all that can be said
        double lo = (double)d, hi = 64.0 * (double)n;
        for (int i = 0; i < 60; i++) {
            const double g = 0.5 * (lo + hi);
            if (g * (1.0 - std::exp(-(double)n / g)) < (double)d) lo = g;
            else hi = g;
        }
        return (uint64_t)hi;
    }

    // the HBM table and everything in it is given up (see lone_probe)
    void drop_table()
    {
        hipStream_t s = stream_.get();
        drain_merges();
        gt_tag_.release();
        gt_keys_.release();
        gt_words_.release();
        rep_count_.release();
        gt_cap_ = 0;
        gt_rep_ = 1;
        PA_HIP(hipMemsetAsync(ctl_ + 1, 0, 4, s));
        groups_upper_ = groups_sum_ = 0;
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
        if (spec_.any_ranked()) return false;  // (re-ranking walks the HBM table)
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
        const uint64_t g = std::max(groups_upper_, probed_groups_);
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
        HostTraceScope trace("    fused.run_page_build_rows");
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
        // every wave walks one contiguous range of the page and all ranges are equally long: two rounds of as many workgroups as
        // the device holds at once (a grid that is not a multiple of that leaves CUs idle in the last round)
        int resident = 4;
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&resident, ck.kernel.fn, ki.block, 0) != hipSuccess || resident <= 0) resident = 4;
        int per_cu = resident * 2;
        if (const char* e = getenv("PRESTO_AMD_BROW_GRID")) per_cu = std::max(1, atoi(e));  // (measurement switch: workgroups per CU)
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(((n + 3) / 4 + 255) / 256, (int64_t)cus_ * per_cu));
        void* params[] = {&a};
        timer.set_name(ck.kernel.name);
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
        last_info_ = &ki;
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
            // (the distinct keys of 2^18 rows tell hundreds of groups from hundreds of thousands, and -- by how many of the rows
            // were new keys, estimate_groups -- those from millions; the rows of a probe that sends the page to the
            // partition-owned tables are redone there -- see lone_probe)
            if (table_tier && !gt_probed_ && !list) n = std::min<int64_t>(n, kProbeRows);
            bool use_tail = false;
            if (ki.variant == V_LDS) {
                if (offset < lds_head) {
                    n = lds_head - offset;
                    // nothing is known about the cardinality yet: a short first launch decides whether the register-table
                    // variant fits, instead of a whole wasted pass over a large page
                    if (!lds_probed_ && ck.lds_verdict.load(std::memory_order_relaxed) == 1) lds_probed_ = true;
                    lds_compiled_ = &ck;
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
                uint32_t want = desired_replicas(gt_probed_ ? groups_upper_ : std::max<uint64_t>(groups_upper_, (uint64_t)std::max(spec_.expected_groups, 1)));
                // (the LDS-table tier reaches the HBM table once per group and workgroup, its workgroups each starting elsewhere in
                // their tables: a few replicas are plenty, and every replica is memory to clear and a table to fold at the end)
                if (ki.variant == V_LDSH) want = std::min<uint32_t>(want, kLdshReplicas);
                // change the layout only when it pays: much more replication needed, or far too much held
                reps = (want >= 2 * gt_rep_ || want * 4 <= gt_rep_) ? want : gt_rep_;
                reps = std::min<uint32_t>(reps, next_pow2((uint64_t)grid));
                // the launch that tells the cardinality counts into ONE table: the groups of replicas cannot be told apart from
                // the outside (their sum counts a group once per replica that met it), and its few workgroups' flushes are
                // no load on anybody's addresses
                if (ki.variant == V_LDSH && !gt_probed_ && !list && groups_upper_ == 0) reps = 1;
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
            if (!use_tail) timer.set_name(ck.kernel.name);
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
                f.seq = ++launch_seq_;
                inflight_.push_back(std::move(f));
                if (!lds_probed_) {
                    // the first launch of all: when it gives up, its rows are not redone on their own -- the page goes to the next
                    // tier from this launch's first row on, so that the tier's own probe sees a whole page in front of it
                    confirm_all_but_last();
                    if (!confirm_oldest(false)) {
                        resume_from_ = offset;
                        return false;
                    }
                }
                else if (!retained_) confirm_all();
                else poll_inflight();
                if (mode_ != V_LDS) {
                    resume_from_ = offset + n;
                    return false;
                }
            }
            else {
                // replay loop: grow the table until every row of the launch found room for its group
                int cur = 0;
                // the launch that tells the cardinality, on a table that held nothing before it
                const bool lone_probe = !gt_probed_ && !list && groups_upper_ == 0 && groups_sum_ == 0 && sub_parts_ == 0 && !is_combiner_;
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
                        // the probe launch is short -- its workgroups' tables have room for all they see -- so the groups it
                        // found speak instead: more than a workgroup's table takes, and the tiers behind the HBM table's probe
                        // (hash-partitioned LDS tables, partition-owned tables) do better from the next row on
                        if (lone_probe && first_probe && estimate_groups(groups_sum_, (uint64_t)n) > (uint64_t)ki.lc / 2) mode_ = V_GT;
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
                        // ... and when it calls for the partition-owned tables, the probe's own groups are given up: left in the
                        // HBM table they would make every group of the partitions' tables pay an upsert there at the end (the
                        // fold: 0.24 ms for 1 M groups), where redoing the probe's rows with the rest of the page costs a few
                        // per cent of the page
                        if (lone_probe && resume_from_ >= 0 && mode_ == V_GT && !getenv("PRESTO_AMD_KEEP_PROBE")) {
                            int p = 0;
                            probed_groups_ = std::max(probed_groups_, estimate_groups(groups_sum_, (uint64_t)n));
                            if (partitioned_wanted(*cur_sig_, *cur_layout_, &p) && want_ldsp_) {
                                drop_table();
                                resume_from_ = offset;
                            }
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
        uint64_t seq = 0;     // position among the operator's few-groups launches (release_checkpoint)
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
    void confirm_all_but_last()
    {
        while (inflight_.size() > 1) confirm_oldest();
    }
    // false: the launch met more groups than its tier takes (redo: its rows are redone on the next tier here and now)
    bool confirm_oldest(bool redo = true)
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
            if (lds_compiled_) lds_compiled_->lds_verdict.store(1, std::memory_order_relaxed);
            return true;
        }
        if (lds_compiled_) lds_compiled_->lds_verdict.store(2, std::memory_order_relaxed);
        // more groups than the wave's register table: the launch's merge skipped itself; its rows -- and every later page --
        // go to the workgroup-level LDS table, which itself hands rows it has no room for to the HBM table
        hipStream_t s = stream_.get();
        drain_merges();
        PA_HIP(hipMemsetAsync(ctl_ + 2 + 2 * f.b, 0, 8, s));
        if (mode_ == V_LDS) mode_ = V_LDSH;
        if (!redo) return false;
        const int64_t saved = resume_from_;
        const bool saved_retained = retained_;
        retained_ = false;
        run_tiers(f.sig, f.layout, f.dp, f.vec, f.offset);
        retained_ = saved_retained;
        resume_from_ = saved;
        return false;
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
    void rank_values(DevPage& dp, hipStream_t s);
    void rerank_words(int c, const uint32_t* ranks, hipStream_t s);
    // channel of the page projection `proj` names when min / max go through its rank (Spec::ranked), else -1
    int ranked_channel(int proj) const
    {
        const OwnedExpr& pe = spec_.proj[(size_t)proj];
        if (!pe.is_input_ref()) return -1;
        const int c = pe.node(pe.root).channel;
        return c >= 0 && c < spec_.n_in && spec_.ranked[c] ? c : -1;
    }
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
    PinnedPageCopy pinned_copy_;
    std::vector<int64_t> var_bytes_hint_;   // per channel, while an arena of host pages is processed: bytes of its VARCHAR block
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
    PinnedBuf h_ctl_buf_, h_table_, h_parts_;
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
    const Compiled* lds_compiled_ = nullptr;  // the few-groups kernel in use (its lds_verdict is this plan's memory between operators)
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
    // retained pages (see take_page)
    Release cur_rel_;
    bool cur_rel_set_ = false;
    std::vector<Release> run_rel_, ranges_rel_, carry_rel_;
    std::deque<ReleaseBatch> release_batches_;
    std::vector<hipEvent_t> release_events_;
    uint64_t launch_seq_ = 0;
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
        std::vector<Release> rel;                     // releases of the retained pages copied (or to be copied) into this arena
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
    uint64_t probed_groups_ = 0;          // groups a probe launch saw before its table was given up (drop_table)
    std::vector<OutColumn> out_cols_;
    std::vector<pa_column> out_storage_;
    int32_t out_rows_ = 0;
    std::vector<std::unique_ptr<StringInterner>> interners_;  // per input channel, for Spec::interned channels
    PageStager dict_stager_;
    int list_grid_hint_ = 0;
    std::vector<DevBuf> reorder_bufs_;   // per channel: values, NULL flags of the partition-ordered copy of a chunk
    std::vector<DevBuf> dict_key_bufs_;  // per interned channel: uploaded ids, key ids, key NULL flags of a dictionary page
    // per ranked channel (Spec::ranked): the dictionary's strings on the host, their ids in string order, the ranks on the device, the
    // page's image column
    struct RankedChannel {
        std::vector<std::string> strings;
        std::vector<uint32_t> order;
        DevBuf ranks, image;
    };
    std::vector<std::unique_ptr<RankedChannel>> ranked_;
    // host copies of small key dictionaries (build_output's host assembly of a few groups)
    static constexpr int64_t kHostDecodeGroups = 4096;
    static constexpr uint32_t kHostDecodeIds = 4096;
    std::vector<std::vector<std::string>> host_dict_;
    const KernelInfo* last_info_ = nullptr;  // layout of the accumulator words in the table (the last launch's)
};

}  // namespace fused_op
}  // namespace pa
