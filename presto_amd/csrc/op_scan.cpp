// op_scan.cpp -- ScanFilterAndProjectOperator: a source operator that pulls pages from a page source and pipes them through
// PageProcessor + MergePages, loading lazy blocks only when they are needed.
//
// Reference path replaced:
//   ScanFilterAndProjectOperator.SplitToPages / ConnectorPageSourceToPages
//       (…/operator/ScanFilterAndProjectOperator.java:185-292, 357-400): getNextPage -> recordMaterializedBytes(page, …)
//       (:391, sizes of the LazyBlocks as they are loaded) -> PageProcessor -> MergePages
//   the lazy-load rule of PageProcessor (…/operator/project/PageProcessor.java:307-347): the filter sees only its own input
//       channels (InputChannels.getInputChannels, …/InputChannels.java:49-52); a projection's channels are loaded
//       (Block.getLoadedBlock, :341-343) only when it runs, i.e. when the filter selected at least one position (:127-129)
// On the device this is where a lazy block pays most: a block that is never loaded never crosses PCIe.  A page is processed
// in two phases -- (1) the filter's channels are loaded and staged, a filter-only PageProcessor counts the selected
// positions; (2) only when some survive are the remaining channels of the projections loaded, staged next to the first ones
// and the page handed to the FilterAndProject operator as a device page.  Channels no expression reads are never loaded.
#include <set>

#include "exprgen.hpp"
#include "operator.hpp"

namespace pa {
namespace {

class ScanFilterProjectOperator : public pa_operator {
public:
    ScanFilterProjectOperator(const pa_filter_project_desc* d, const pa_page_source* source) : stream_(checked(d, source)->stream)
    {
        source_ = *source;
        n_in_ = d->input_channel_count;
        types_.assign(d->input_types, d->input_types + n_in_);
        // channels the filter / the projections read
        filter_ch_.assign(n_in_, false);
        proj_ch_.assign(n_in_, false);
        std::set<int32_t> fc, pc;
        if (d->filter) OwnedExpr::copy(*d->filter).collect_channels(&fc);
        for (int32_t j = 0; j < d->projection_count; j++) OwnedExpr::copy(d->projections[j]).collect_channels(&pc);
        for (int32_t c : fc) {
            PA_REQUIRE(c >= 0 && c < n_in_, PA_ERR_INVALID_ARGUMENT, "expression references a channel outside the page");
            filter_ch_[c] = true;
        }
        for (int32_t c : pc) {
            PA_REQUIRE(c >= 0 && c < n_in_, PA_ERR_INVALID_ARGUMENT, "expression references a channel outside the page");
            proj_ch_[c] = true;
        }
        bool proj_only = false;
        for (int c = 0; c < n_in_; c++) proj_only = proj_only || (proj_ch_[c] && !filter_ch_[c]);
        two_phase_ = d->filter != nullptr && proj_only;
        pa_filter_project_desc inner = *d;
        inner.stream = stream_.get();
        inner_.reset(make_filter_project(&inner));
        if (two_phase_) {
            pa_filter_project_desc count = *d;
            count.projection_count = 0;  // PageProcessor without projections: a channel-less page with the selected count
            count.projections = nullptr;
            count.min_output_page_bytes = 0;
            count.min_output_page_rows = 0;
            count.output_mem = PA_MEM_HOST;
            count.stream = stream_.get();
            count_.reset(make_filter_project(&count));
        }
    }
    static const pa_filter_project_desc* checked(const pa_filter_project_desc* d, const pa_page_source* source)
    {
        PA_REQUIRE(d != nullptr && source != nullptr && source->next_page != nullptr, PA_ERR_INVALID_ARGUMENT, "page source without next_page");
        return d;
    }
    ~ScanFilterProjectOperator() override
    {
        (void)hipStreamSynchronize(stream_.get());
        close_source();
    }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    bool needs_input() override { return false; }  // a source operator (WorkProcessorSourceOperatorAdapter)
    void add_input(const pa_page*) override { throw Error(PA_ERR_ILLEGAL_STATE, "a scan operator takes no input pages"); }
    void finish() override
    {
        // SourceOperator.finish / close of the page source: stop pulling, flush what MergePages holds
        if (!source_done_) {
            source_done_ = true;
            close_source();
            inner_->finish();
        }
    }
    bool is_finished() override { return source_done_ && inner_->is_finished(); }
    int64_t memory_bytes() override { return (int64_t)(stage_filter_.bytes() + stage_proj_.bytes()) + inner_->memory_bytes(); }

    bool get_output(pa_page* out) override
    {
        // at most one page is pulled per call: a Driver calls again (and may yield in between, Driver.java:355-457)
        if (inner_->get_output(out)) return true;
        if (source_done_) return false;
        pull_one_page();
        return inner_->get_output(out);
    }

    void stats(int64_t* rows, int64_t* bytes, int64_t* loaded, int64_t* skipped) const
    {
        if (rows) *rows = rows_in_;
        if (bytes) *bytes = bytes_loaded_;
        if (loaded) *loaded = blocks_loaded_;
        if (skipped) *skipped = blocks_skipped_;
    }

private:
    void close_source()
    {
        if (source_.close && !source_closed_) source_.close(source_.ctx);
        source_closed_ = true;
    }

    // LazyBlock.getLoadedBlock for the channels of `which` that are not loaded yet; counts what was materialised
    void load(std::vector<pa_column>& cols, const std::vector<bool>& which, int32_t n)
    {
        for (int c = 0; c < n_in_; c++) {
            if (!which[c] || loaded_[c]) continue;
            if (cols[c].values == nullptr && cols[c].dictionary == nullptr) {
                PA_REQUIRE(source_.load_block != nullptr, PA_ERR_INVALID_ARGUMENT, "the page source returned an unloaded block but has no load_block");
                pa_column col{};
                const int64_t rc = source_.load_block(source_.ctx, c, &col);
                if (rc < 0) throw Error((int32_t)rc, "the page source failed to load a block");
                cols[c] = col;
            }
            loaded_[c] = true;
            blocks_loaded_++;
            // recordMaterializedBytes: Block.getSizeInBytes of the loaded block ((width + 1) per position for fixed width,
            // bytes + 5 per position for VARCHAR; LongArrayBlock.java:72, VariableWidthBlock.java)
            const pa_column& col = cols[c];
            if (col.encoding == PA_FLAT) bytes_loaded_ += (int64_t)n * (type_width(col.type) + 1);
            else if (col.encoding == PA_VARWIDTH && col.offsets && page_mem_ == PA_MEM_HOST) bytes_loaded_ += (int64_t)(col.offsets[n] - col.offsets[0]) + 5LL * n;
            else bytes_loaded_ += 5LL * n;
        }
    }

    void pull_one_page()
    {
        pa_page page{};
        const int32_t rc = source_.next_page(source_.ctx, &page);
        if (rc < 0) throw Error(rc, "the page source failed");
        if (rc == 0) {  // ConnectorPageSource.isFinished
            finish();
            return;
        }
        PA_REQUIRE(page.channel_count == n_in_ && (page.columns != nullptr || n_in_ == 0), PA_ERR_INVALID_ARGUMENT,
                   "page source: channel count does not match the operator's input types");
        const int32_t n = page.position_count;
        rows_in_ += n;
        if (n == 0) return;
        page_mem_ = page.mem;
        std::vector<pa_column> cols(page.columns, page.columns + n_in_);
        loaded_.assign(n_in_, false);
        hipStream_t s = stream_.get();
        if (two_phase_) {
            // phase 1: the filter's channels, and how many positions it selects
            load(cols, filter_ch_, n);
            pa_page fpage = page;
            fpage.columns = cols.data();
            DevPage staged = stage_filter_.stage(&fpage, &filter_ch_, s);
            std::vector<pa_column> dcols(n_in_);
            for (int c = 0; c < n_in_; c++) {
                dcols[c].type = types_[c];
                dcols[c].encoding = types_[c] == PA_VARCHAR ? PA_VARWIDTH : PA_FLAT;
                if (!filter_ch_[c]) continue;
                dcols[c].type = staged.cols[c].type;
                dcols[c].encoding = staged.cols[c].varwidth ? PA_VARWIDTH : PA_FLAT;
                dcols[c].values = staged.cols[c].values;
                dcols[c].offsets = staged.cols[c].offsets;
                dcols[c].nulls = staged.cols[c].nulls;
            }
            pa_page dpage{};
            dpage.position_count = n;
            dpage.channel_count = n_in_;
            dpage.columns = dcols.data();
            dpage.mem = PA_MEM_DEVICE;
            count_->add_input(&dpage);
            pa_page counted{};
            const bool any = count_->get_output(&counted) && counted.position_count > 0;
            if (!any) {
                // PageProcessor.java:127-129: no position selected -> no projection runs -> their blocks stay unloaded
                for (int c = 0; c < n_in_; c++) {
                    if (proj_ch_[c] && !filter_ch_[c]) blocks_skipped_++;
                }
                return;
            }
            // phase 2: the projections' other channels next to the staged ones; the operator sees one device page
            std::vector<bool> rest(n_in_, false);
            for (int c = 0; c < n_in_; c++) rest[c] = proj_ch_[c] && !filter_ch_[c];
            load(cols, rest, n);
            fpage.columns = cols.data();
            DevPage staged2 = stage_proj_.stage(&fpage, &rest, s);
            for (int c = 0; c < n_in_; c++) {
                if (!rest[c]) continue;
                dcols[c].type = staged2.cols[c].type;
                dcols[c].encoding = staged2.cols[c].varwidth ? PA_VARWIDTH : PA_FLAT;
                dcols[c].values = staged2.cols[c].values;
                dcols[c].offsets = staged2.cols[c].offsets;
                dcols[c].nulls = staged2.cols[c].nulls;
            }
            inner_->add_input(&dpage);
            return;
        }
        std::vector<bool> used(n_in_, false);
        for (int c = 0; c < n_in_; c++) used[c] = filter_ch_[c] || proj_ch_[c];
        load(cols, used, n);
        pa_page full = page;
        full.columns = cols.data();
        inner_->add_input(&full);
    }

    Stream stream_;
    pa_page_source source_{};
    int n_in_ = 0;
    std::vector<int32_t> types_;
    std::vector<bool> filter_ch_, proj_ch_, loaded_;
    bool two_phase_ = false, source_done_ = false, source_closed_ = false;
    int32_t page_mem_ = PA_MEM_HOST;
    std::unique_ptr<pa_operator> inner_, count_;
    PageStager stage_filter_, stage_proj_;
    int64_t rows_in_ = 0, bytes_loaded_ = 0, blocks_loaded_ = 0, blocks_skipped_ = 0;
};

}  // namespace

pa_operator* make_scan_filter_project(const pa_filter_project_desc* desc, const pa_page_source* source)
{
    return new ScanFilterProjectOperator(desc, source);
}

void scan_stats(pa_operator* op, int64_t* rows, int64_t* bytes, int64_t* loaded, int64_t* skipped)
{
    auto* scan = dynamic_cast<ScanFilterProjectOperator*>(op);
    PA_REQUIRE(scan != nullptr, PA_ERR_INVALID_ARGUMENT, "not a ScanFilterAndProject operator");
    scan->stats(rows, bytes, loaded, skipped);
}

}  // namespace pa
