// op_fused_join.cpp -- FilterAndProjectOperator -> LookupJoinOperator [-> (Hash)AggregationOperator] behind one operator handle
// (pa_fused_join_aggregation_desc, pa_fused_join_desc).
//
// The reference runs the three operators of such a pipeline in one Driver (LocalExecutionPlanner.visitAggregation over
// visitJoin's probe side; Driver.processInternal moves pages between neighbours, Driver.java:355-457).  Two executions of the
// same composition live behind this handle, chosen when the build side has published its lookup source:
//   * one generated kernel (op_fused.cpp, "probe stage"; for the join without an aggregation behind it: the FilterAndProject kernels
//     with the probe inside, op_filter_project.cpp) when the lookup source has one integer key and no duplicate keys;
//   * the device operators behind each other, pages moved between them the way Driver.processInternal does, in every
//     other case (duplicate keys -- a probe row then has several matches --, several join keys, other key types).
// Both are created up front (creation does no device work to speak of), so that everything a descriptor can get wrong is
// reported by the factory call, and the one not taken is dropped with the first page.
#include <memory>
#include <vector>

#include "join_source.hpp"
#include "operator.hpp"

namespace pa {

bool lookup_source_built(pa_lookup_source* ls) { return ls != nullptr && ls->impl != nullptr && ls->impl->built.load(); }
bool lookup_source_unique_keyed(pa_lookup_source* ls)
{
    return lookup_source_built(ls) && ls->impl->keyed && !ls->impl->has_duplicates;
}

namespace {

class FusedJoinAggregationOperator : public pa_operator {
public:
    // FilterAndProject -> LookupJoin (no aggregation behind it)
    FusedJoinAggregationOperator(const pa_fused_join_desc* d, pa_lookup_source* bridge) : bridge_(*bridge)
    {
        PA_REQUIRE(bridge->impl != nullptr, PA_ERR_ILLEGAL_STATE, "lookup source has no build operator yet");
        PA_REQUIRE(d->join.join_type == PA_JOIN_INNER && d->join.filter == nullptr, PA_ERR_NOT_SUPPORTED, "the fused join is an inner join without a filter function");
        void* stream = shared_stream(d->join.stream ? d->join.stream : d->filter_project.stream);
        pa_filter_project_desc fp = d->filter_project;
        fp.output_mem = PA_MEM_DEVICE;
        fp.stream = stream;
        fp.min_output_page_bytes = fp.min_output_page_rows = fp.max_output_page_bytes = 0;  // (MergePages sits behind the join)
        fp.output_handover = 0;  // (inside the chain the page goes to the join in stream order; the hand-over is the one-pass form's)
        pa_lookup_join_desc join = d->join;
        join.stream = stream;
        try {
            pa_filter_project_desc one = d->filter_project;
            one.output_mem = d->join.output_mem;
            one.stream = stream;
            fused_.reset(make_filter_project_probe(&one, &d->join, bridge));
        }
        catch (const Error& e) {
            if (e.code != PA_ERR_NOT_SUPPORTED) throw;
        }
        // (a lookup source that is already built has told which execution runs: the other one is not even made)
        if (!(fused_ && lookup_source_unique_keyed(bridge))) {
            chain_.emplace_back(make_filter_project(&fp));
            chain_.emplace_back(make_lookup_join(&join, bridge));
        }
        stream_ = static_cast<hipStream_t>(stream);
        finish_sent_.assign(chain_.size(), false);
    }
    FusedJoinAggregationOperator(const pa_fused_join_aggregation_desc* d, pa_lookup_source* bridge) : bridge_(*bridge)
    {
        PA_REQUIRE(bridge->impl != nullptr, PA_ERR_ILLEGAL_STATE, "lookup source has no build operator yet");
        PA_REQUIRE(d->join.join_type == PA_JOIN_INNER, PA_ERR_NOT_SUPPORTED, "the fused join-aggregation is an inner join");
        void* stream = shared_stream(d->aggregation.stream ? d->aggregation.stream : (d->join.stream ? d->join.stream : d->filter_project.stream));
        // the chain: intermediate pages stay in HBM, everything on one stream
        pa_filter_project_desc fp = d->filter_project;
        fp.output_mem = PA_MEM_DEVICE;
        fp.stream = stream;
        pa_lookup_join_desc join = d->join;
        join.output_mem = PA_MEM_DEVICE;
        join.stream = stream;
        pa_hash_aggregation_desc agg = d->aggregation;
        agg.stream = stream;
        try {
            pa_fused_join_aggregation_desc f = *d;
            f.aggregation.stream = stream;
            fused_.reset(make_fused_probe_aggregation(&f, bridge));
        }
        catch (const Error& e) {
            if (e.code != PA_ERR_NOT_SUPPORTED) throw;  // shapes the one-kernel form does not cover run as the chain
        }
        if (!(fused_ && lookup_source_unique_keyed(bridge))) {
            chain_.emplace_back(make_filter_project(&fp));
            chain_.emplace_back(make_lookup_join(&join, bridge));
            chain_.emplace_back(make_hash_aggregation(&agg));
        }
        stream_ = static_cast<hipStream_t>(stream);
        finish_sent_.assign(chain_.size(), false);
    }
    ~FusedJoinAggregationOperator() override
    {
        // the members run on own_stream_: they go first, the stream (drained) after them
        if (own_stream_) (void)hipStreamSynchronize(own_stream_->get());
        fused_.reset();
        chain_.clear();
    }
    hipStream_t main_stream() override { return stream_; }
    // every member runs on ONE stream: the caller's, or -- when no descriptor names one -- a stream this handle owns.  (Members
    // with pooled streams of their own would hand device pages to each other inside pump() with nothing ordering the streams:
    // only the C-ABI wrappers drain a private stream when a device page changes hands.)
    hipStream_t private_stream() override { return own_stream_ ? own_stream_->get() : nullptr; }

    bool needs_input() override
    {
        if (finishing_ || !choose()) return false;
        return fused_ ? fused_->needs_input() : chain_[0]->needs_input();
    }
    bool is_blocked() override
    {
        if (!choose()) return !finishing_;  // the lookup source future (LookupJoinOperator.java:63, 100)
        return fused_ ? fused_->is_blocked() : false;
    }
    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(choose(), PA_ERR_ILLEGAL_STATE, "probe page before the lookup source was built");
        if (fused_) {
            fused_->add_input(page);
            return;
        }
        chain_[0]->add_input(page);
        pump();
    }
    void finish() override
    {
        if (finishing_) return;
        finishing_ = true;
        if (!choose()) {
            // no page ever came and the build side is not done: an inner join over nothing is nothing, whatever it builds
            fused_.reset();
        }
        if (fused_) {
            fused_->finish();
            return;
        }
        chain_[0]->finish();
        pump();
    }
    bool get_output(pa_page* out) override
    {
        if (fused_) return fused_->get_output(out);
        pump();
        return chain_.back()->get_output(out);
    }
    bool is_finished() override { return fused_ ? fused_->is_finished() : chain_.back()->is_finished(); }
    // (the one-kernel form and the aggregation at the end of the chain both take it; the hint may arrive before the lookup source
    // decides which of the two runs, so it goes to both)
    bool set_output_topn(int64_t n, const int32_t* channels, const int32_t* orders, int32_t count) override
    {
        bool taken = false;
        if (fused_) taken = fused_->set_output_topn(n, channels, orders, count) || taken;
        if (!chain_.empty() && chain_.back()) taken = chain_.back()->set_output_topn(n, channels, orders, count) || taken;
        return taken;
    }
    int64_t memory_bytes() override
    {
        if (fused_) return fused_->memory_bytes();
        int64_t b = 0;
        for (auto& op : chain_) b += op ? op->memory_bytes() : 0;
        return b;
    }
    KernelTimer& kernel_timer() override { return fused_ ? fused_->kernel_timer() : (!chain_.empty() && chain_.back() ? chain_.back()->kernel_timer() : timer); }
    // which execution runs (tests, DESIGN's measurements): 1 = one kernel, 2 = operator chain, 0 = not decided yet
    int execution() const { return !chosen_ ? 0 : (fused_ ? 1 : 2); }

private:
    void* shared_stream(void* given)
    {
        if (given) return given;
        own_stream_ = std::make_unique<Stream>(nullptr);
        return own_stream_->get();
    }

    // true once the lookup source is there and the execution is fixed
    bool choose()
    {
        if (chosen_) return true;
        if (!lookup_source_built(&bridge_)) return false;
        if (fused_ && !lookup_source_unique_keyed(&bridge_)) fused_.reset();
        if (fused_) chain_.clear();
        chosen_ = true;
        return true;
    }

    // Driver.processInternal over the three operators: a page moves to the next operator whenever that one takes input
    void pump()
    {
        for (bool moved = true; moved;) {
            moved = false;
            for (size_t i = 0; i + 1 < chain_.size(); i++) {
                pa_operator* cur = chain_[i].get();
                pa_operator* next = chain_[i + 1].get();
                if (!cur->is_finished() && next->needs_input()) {
                    pa_page page;
                    memset(&page, 0, sizeof page);
                    if (cur->get_output(&page)) {
                        next->add_input(&page);
                        moved = true;
                    }
                }
                if (cur->is_finished() && !finish_sent_[i + 1]) {
                    next->finish();
                    finish_sent_[i + 1] = true;
                    moved = true;
                }
            }
        }
    }

    pa_lookup_source bridge_;
    std::unique_ptr<Stream> own_stream_;
    std::vector<std::unique_ptr<pa_operator>> chain_;
    std::unique_ptr<pa_operator> fused_;
    hipStream_t stream_ = nullptr;
    bool chosen_ = false, finishing_ = false;
    std::vector<bool> finish_sent_;
};

}  // namespace

pa_operator* make_fused_join(const pa_fused_join_desc* desc, pa_lookup_source* bridge)
{
    PA_REQUIRE(desc != nullptr && bridge != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    return new FusedJoinAggregationOperator(desc, bridge);
}

pa_operator* make_fused_join_aggregation(const pa_fused_join_aggregation_desc* desc, pa_lookup_source* bridge)
{
    PA_REQUIRE(desc != nullptr && bridge != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    return new FusedJoinAggregationOperator(desc, bridge);
}

}  // namespace pa
