// device_page.cpp -- see device_page.hpp.
#include "device_page.hpp"
#include "scan_kernels.hpp"

namespace pa {

void* PageStager::arena(size_t index, size_t bytes)
{
    if (index >= bufs_.size()) bufs_.resize(index + 1);
    return bufs_[index].ensure(bytes);
}

size_t PageStager::bytes() const
{
    size_t total = packed_dev_[0].capacity() + packed_dev_[1].capacity();
    for (const auto& b : bufs_) total += b.capacity();
    return total;
}

PageStager::~PageStager()
{
    for (Packed& p : packed_) {
        if (p.done) {
            if (p.pending) (void)hipEventSynchronize(p.done);   // the pinned block goes back to the pool: not under a running copy
            (void)hipEventDestroy(p.done);
        }
    }
}

// The packed path of small host pages (kPackedLimit).  false: the page does not qualify (an encoded block, too large), nothing done.
bool PageStager::stage_packed(const pa_page* page, const std::vector<bool>* needed, hipStream_t stream, DevPage& out)
{
    const int64_t n = page->position_count;
    struct Piece { const void* src; size_t bytes; size_t at; };
    Piece pieces[3 * 64];
    if (page->channel_count > 64) return false;
    size_t count = 0, total = 0;
    auto add = [&](const void* src, size_t bytes) -> size_t {
        size_t at = total;
        pieces[count++] = Piece{src, bytes, at};
        total += (bytes + 255) & ~(size_t)255;
        return at;
    };
    // first pass: every block qualifies, and the layout
    for (int32_t c = 0; c < page->channel_count; c++) {
        const pa_column& col = page->columns[c];
        if (needed && (c >= (int32_t)needed->size() || !(*needed)[c])) continue;
        if (col.encoding == PA_FLAT) {
            int w = type_width(col.type);
            if (w <= 0 || (col.values == nullptr && n > 0)) return false;   // the general path reports it
        }
        else if (col.encoding == PA_VARWIDTH) {
            if (col.type != PA_VARCHAR || col.offsets == nullptr) return false;
        }
        else return false;
    }
    size_t bytes = 0;
    for (int32_t c = 0; c < page->channel_count; c++) {
        const pa_column& col = page->columns[c];
        if (needed && (c >= (int32_t)needed->size() || !(*needed)[c])) continue;
        bytes += col.encoding == PA_FLAT ? (size_t)n * type_width(col.type) : (size_t)(n + 1) * 4 + (size_t)col.offsets[n];
        bytes += col.nulls ? (size_t)n : 0;
    }
    if (bytes == 0 || bytes > kPackedLimit) return false;
    const size_t turn = packed_at_++ & 1;
    Packed& slot = packed_[turn];
    if (slot.done == nullptr) PA_HIP(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    if (slot.pending) PA_HIP(hipEventSynchronize(slot.done));   // two pages back: done long ago
    slot.pending = false;
    out.n = page->position_count;
    out.cols.assign(page->channel_count, DevColumn{});
    struct Where { size_t values = 0, offsets = 0, nulls = 0; bool has_nulls = false; };
    std::vector<Where> where(page->channel_count);
    for (int32_t c = 0; c < page->channel_count; c++) {
        const pa_column& col = page->columns[c];
        out.cols[c].type = col.type;
        if (needed && (c >= (int32_t)needed->size() || !(*needed)[c])) continue;
        Where& w = where[c];
        if (col.encoding == PA_FLAT) w.values = add(col.values, (size_t)n * type_width(col.type));
        else {
            w.offsets = add(col.offsets, (size_t)(n + 1) * 4);
            w.values = add(col.values, (size_t)col.offsets[n]);
        }
        if (col.nulls) {
            w.nulls = add(col.nulls, (size_t)n);
            w.has_nulls = true;
        }
    }
    uint8_t* host = static_cast<uint8_t*>(slot.host.ensure(total));
    for (size_t i = 0; i < count; i++)
        if (pieces[i].bytes) memcpy(host + pieces[i].at, pieces[i].src, pieces[i].bytes);
    uint8_t* dev = static_cast<uint8_t*>(packed_dev_[turn].ensure(total));
    PA_HIP(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, stream));
    PA_HIP(hipEventRecord(slot.done, stream));
    slot.pending = true;
    for (int32_t c = 0; c < page->channel_count; c++) {
        const pa_column& col = page->columns[c];
        if (needed && (c >= (int32_t)needed->size() || !(*needed)[c])) continue;
        DevColumn& d = out.cols[c];
        const Where& w = where[c];
        d.values = dev + w.values;
        if (col.encoding == PA_VARWIDTH) {
            d.varwidth = true;
            d.offsets = reinterpret_cast<const int32_t*>(dev + w.offsets);
        }
        d.nulls = w.has_nulls ? dev + w.nulls : nullptr;
    }
    return true;
}

PinnedPageCopy::~PinnedPageCopy()
{
    for (Slot& p : slots_) {
        if (p.done) {
            if (p.pending) (void)hipEventSynchronize(p.done);   // the pinned block goes back to the pool: not under a running read
            (void)hipEventDestroy(p.done);
        }
    }
}

const pa_page* PinnedPageCopy::copy(const pa_page* page, const std::vector<bool>* needed)
{
    const int64_t n = page->position_count;
    if (n <= 0 || page->mem != PA_MEM_HOST) return nullptr;
    auto skip = [&](int32_t c) { return needed && (c >= (int32_t)needed->size() || !(*needed)[c]); };
    auto pad = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    size_t total = 0;
    for (int32_t c = 0; c < page->channel_count; c++) {
        const pa_column& col = page->columns[c];
        if (skip(c)) continue;
        if (col.encoding == PA_FLAT) {
            const int w = type_width(col.type);
            if (w <= 0 || col.values == nullptr) return nullptr;   // (the general path reports it)
            total += pad((size_t)n * w);
        }
        else if (col.encoding == PA_VARWIDTH) {
            if (col.type != PA_VARCHAR || col.offsets == nullptr || col.offsets[n] < 0 || (col.offsets[n] > 0 && col.values == nullptr)) return nullptr;
            total += pad((size_t)(n + 1) * 4) + pad((size_t)col.offsets[n]);
        }
        else return nullptr;
        if (col.nulls) total += pad((size_t)n);
        if (total > PageStager::kPackedLimit) return nullptr;
    }
    if (total == 0) return nullptr;
    Slot& slot = slots_[at_++ & 1];
    if (slot.done == nullptr) PA_HIP(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    if (slot.pending) PA_HIP(hipEventSynchronize(slot.done));   // two pages back
    slot.pending = false;
    char* base = static_cast<char*>(slot.host.ensure(total));
    size_t at = 0;
    auto put = [&](const void* src, size_t bytes) -> const void* {
        char* dst = base + at;
        if (bytes) memcpy(dst, src, bytes);
        at += pad(bytes);
        return dst;
    };
    slot.cols.assign(page->columns, page->columns + page->channel_count);
    for (int32_t c = 0; c < page->channel_count; c++) {
        if (skip(c)) continue;
        const pa_column& col = page->columns[c];
        pa_column& out = slot.cols[c];
        if (col.encoding == PA_FLAT) out.values = put(col.values, (size_t)n * type_width(col.type));
        else {
            out.offsets = static_cast<const int32_t*>(put(col.offsets, (size_t)(n + 1) * 4));
            out.values = put(col.values, (size_t)col.offsets[n]);   // from byte 0: the offsets stay what they are
        }
        if (col.nulls) out.nulls = static_cast<const uint8_t*>(put(col.nulls, (size_t)n));
    }
    slot.page = *page;
    slot.page.columns = slot.cols.data();
    slot.page.flags = PA_PAGE_PINNED;
    slot.page.release = nullptr;
    slot.page.release_ctx = nullptr;
    cur_ = &slot;
    return &slot.page;
}

void PinnedPageCopy::used(hipStream_t stream)
{
    if (cur_ == nullptr) return;
    PA_HIP(hipEventRecord(cur_->done, stream));
    cur_->pending = true;
    cur_ = nullptr;
}

static const void* to_device(const void* src, size_t bytes, bool is_device, void* dst, hipStream_t stream)
{
    if (src == nullptr) return nullptr;
    if (is_device) return src;
    if (bytes > 0) PA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
    return dst;
}

DevPage PageStager::stage(const pa_page* page, const std::vector<bool>* needed, hipStream_t stream)
{
    PA_REQUIRE(page != nullptr, PA_ERR_INVALID_ARGUMENT, "page is null");
    PA_REQUIRE(page->position_count >= 0 && page->channel_count >= 0, PA_ERR_INVALID_ARGUMENT, "negative page dimensions");
    PA_REQUIRE(page->channel_count == 0 || page->columns != nullptr, PA_ERR_INVALID_ARGUMENT, "page columns is null");
    const bool dev = page->mem == PA_MEM_DEVICE;
    const int64_t n = page->position_count;
    DevPage out;
    if (!dev && n > 0 && stage_packed(page, needed, stream, out)) return out;
    out.n = page->position_count;
    out.cols.resize(page->channel_count);
    next_ = 0;
    for (int32_t c = 0; c < page->channel_count; c++) {
        const pa_column& col = page->columns[c];
        DevColumn& d = out.cols[c];
        d.type = col.type;
        if (needed && (c >= (int32_t)needed->size() || !(*needed)[c])) {
            next_ += 9;
            continue;
        }
        size_t slot = next_;
        next_ += 9;
        if (col.encoding == PA_FLAT) {
            int w = type_width(col.type);
            PA_REQUIRE(w > 0, PA_ERR_NOT_SUPPORTED, "FLAT block of a variable-width type");
            PA_REQUIRE(col.values != nullptr || n == 0, PA_ERR_INVALID_ARGUMENT, "block values is null");
            d.values = to_device(col.values, (size_t)n * w, dev, dev ? nullptr : arena(slot, (size_t)n * w), stream);
            d.nulls = static_cast<const uint8_t*>(
                to_device(col.nulls, (size_t)n, dev, (dev || !col.nulls) ? nullptr : arena(slot + 1, (size_t)n), stream));
        }
        else if (col.encoding == PA_VARWIDTH) {
            PA_REQUIRE(col.type == PA_VARCHAR, PA_ERR_NOT_SUPPORTED, "VARWIDTH block of a fixed-width type");
            PA_REQUIRE(col.offsets != nullptr, PA_ERR_INVALID_ARGUMENT, "VARWIDTH block without offsets");
            d.varwidth = true;
            if (dev) {
                d.values = col.values;
                d.offsets = col.offsets;
                d.nulls = col.nulls;
            }
            else {
                size_t total = (size_t)col.offsets[n];
                d.offsets = static_cast<const int32_t*>(
                    to_device(col.offsets, (size_t)(n + 1) * 4, false, arena(slot, (size_t)(n + 1) * 4), stream));
                d.values = to_device(col.values, total, false, arena(slot + 1, total ? total : 1), stream);
                d.nulls = static_cast<const uint8_t*>(
                    to_device(col.nulls, (size_t)n, false, col.nulls ? arena(slot + 2, (size_t)n) : nullptr, stream));
            }
        }
        else if (col.encoding == PA_DICTIONARY || col.encoding == PA_RLE) {
            // decode (DictionaryBlock / RunLengthEncodedBlock -> flat) with a gather kernel
            PA_REQUIRE(col.dictionary != nullptr, PA_ERR_INVALID_ARGUMENT, "dictionary block without dictionary");
            const pa_column& dict = *col.dictionary;
            if (dict.encoding == PA_VARWIDTH) {
                // DictionaryBlock / RLE over a VariableWidthBlock (what ORC / Parquet readers hand over for strings):
                // Block.copyPositions by the ids -- lengths, exclusive scan, byte copy
                PA_REQUIRE(dict.type == PA_VARCHAR && dict.offsets != nullptr, PA_ERR_INVALID_ARGUMENT, "bad variable-width dictionary");
                const int64_t dn = col.encoding == PA_RLE ? 1 : col.dictionary_size;
                PA_REQUIRE(dn > 0 || n == 0, PA_ERR_INVALID_ARGUMENT, "empty dictionary");
                d.type = PA_VARCHAR;
                d.varwidth = true;
                const int32_t* doff;
                const void* dvals;
                const uint8_t* dnulls;
                if (dev) {
                    doff = dict.offsets;
                    dvals = dict.values;
                    dnulls = dict.nulls;
                }
                else {
                    const size_t bytes = (size_t)dict.offsets[dn];
                    doff = static_cast<const int32_t*>(to_device(dict.offsets, (size_t)(dn + 1) * 4, false, arena(slot, (size_t)(dn + 1) * 4), stream));
                    dvals = to_device(dict.values, bytes, false, arena(slot + 1, bytes ? bytes : 1), stream);
                    dnulls = static_cast<const uint8_t*>(to_device(dict.nulls, (size_t)dn, false, dict.nulls ? arena(slot + 2, (size_t)dn) : nullptr, stream));
                }
                const size_t rows = (size_t)(n > 0 ? n : 1);
                int32_t* ids = static_cast<int32_t*>(arena(slot + 4, rows * 4));
                if (col.encoding == PA_RLE) {
                    PA_HIP(hipMemsetAsync(ids, 0, rows * 4, stream));
                }
                else {
                    PA_REQUIRE(col.ids != nullptr, PA_ERR_INVALID_ARGUMENT, "dictionary block without ids");
                    if (dev) ids = const_cast<int32_t*>(col.ids);
                    else PA_HIP(hipMemcpyAsync(ids, col.ids, (size_t)n * 4, hipMemcpyHostToDevice, stream));
                }
                int32_t* out_off = static_cast<int32_t*>(arena(slot + 3, (rows + 1) * 4));
                int32_t* total = static_cast<int32_t*>(arena(slot + 6, 64));
                PA_HIP(hipMemsetAsync(out_off, 0, (rows + 1) * 4, stream));
                PA_HIP(hipMemsetAsync(total, 0, 4, stream));
                int32_t h_total = 0;
                if (n > 0) {
                    launch_varwidth_lengths(ids, n, doff, dnulls, out_off, stream);
                    launch_exclusive_scan_i32(out_off, out_off, n, total, arena(slot + 7, scan_temp_bytes(n)), stream);
                    PA_HIP(hipMemcpyAsync(&h_total, total, 4, hipMemcpyDeviceToHost, stream));
                    PA_HIP(hipStreamSynchronize(stream));
                }
                uint8_t* out_bytes = static_cast<uint8_t*>(arena(slot + 5, (size_t)(h_total > 0 ? h_total : 1)));
                if (n > 0) launch_varwidth_copy(ids, n, doff, static_cast<const uint8_t*>(dvals), dnulls, out_off, out_bytes, total, stream);
                uint8_t* fnulls = nullptr;
                if (dnulls) {
                    fnulls = static_cast<uint8_t*>(arena(slot + 8, rows));
                    if (n > 0) launch_gather_nulls(dnulls, ids, n, fnulls, stream);
                }
                d.values = out_bytes;
                d.offsets = out_off;
                d.nulls = fnulls;
                continue;
            }
            PA_REQUIRE(dict.encoding == PA_FLAT, PA_ERR_NOT_SUPPORTED, "dictionary / RLE over nested blocks is decoded on the Java side");
            int w = type_width(dict.type);
            PA_REQUIRE(w > 0, PA_ERR_NOT_SUPPORTED, "dictionary of a variable-width type");
            d.type = dict.type;
            int64_t dn = col.encoding == PA_RLE ? 1 : col.dictionary_size;
            PA_REQUIRE(dn > 0 || n == 0, PA_ERR_INVALID_ARGUMENT, "empty dictionary");
            const void* dvals = to_device(dict.values, (size_t)dn * w, dev, dev ? nullptr : arena(slot, (size_t)dn * w), stream);
            const uint8_t* dnulls = static_cast<const uint8_t*>(
                to_device(dict.nulls, (size_t)dn, dev, (dev || !dict.nulls) ? nullptr : arena(slot + 1, (size_t)dn), stream));
            void* flat = arena(slot + 2, (size_t)(n > 0 ? n : 1) * w);
            uint8_t* fnulls = dnulls ? static_cast<uint8_t*>(arena(slot + 3, (size_t)(n > 0 ? n : 1))) : nullptr;
            if (col.encoding == PA_RLE) {
                launch_fill_flat(flat, w, dvals, n, stream);
                if (dnulls) launch_fill_flat(fnulls, 1, dnulls, n, stream);
            }
            else {
                PA_REQUIRE(col.ids != nullptr, PA_ERR_INVALID_ARGUMENT, "dictionary block without ids");
                const int32_t* ids = static_cast<const int32_t*>(
                    to_device(col.ids, (size_t)n * 4, dev, dev ? nullptr : arena(slot + 4, (size_t)(n > 0 ? n : 1) * 4), stream));
                launch_gather_flat(dvals, w, ids, n, flat, stream);
                if (dnulls) launch_gather_nulls(dnulls, ids, n, fnulls, stream);
            }
            d.values = flat;
            d.nulls = fnulls;
        }
        else {
            throw Error(PA_ERR_NOT_SUPPORTED, "unknown block encoding");
        }
    }
    return out;
}

void publish_output(std::vector<OutColumn>& cols, int32_t n, int32_t mem, hipStream_t stream, pa_page* out,
                    std::vector<pa_column>& storage)
{
    storage.assign(cols.size() ? cols.size() : 1, pa_column{});
    bool need_sync = false;
    for (size_t c = 0; c < cols.size(); c++) {
        OutColumn& o = cols[c];
        pa_column& p = storage[c];
        p.type = o.type;
        p.encoding = o.varwidth ? PA_VARWIDTH : PA_FLAT;
        const void* dv = o.is_view ? o.view_values : o.values.ptr();
        const int32_t* doff = o.is_view ? o.view_offsets : o.offsets.as<int32_t>();
        const uint8_t* dn = o.is_view ? o.view_nulls : (o.has_nulls ? o.nulls.as<uint8_t>() : nullptr);
        if (mem == PA_MEM_DEVICE) {
            p.values = dv;
            p.offsets = o.varwidth ? doff : nullptr;
            p.nulls = dn;
            continue;
        }
        if (o.host_ready) {
            p.values = o.h_values.ptr();
            p.offsets = o.varwidth ? o.h_offsets.as<int32_t>() : nullptr;
            p.nulls = o.has_nulls ? o.h_nulls.as<uint8_t>() : nullptr;
            continue;
        }
        if (o.varwidth) {
            int32_t* ho = static_cast<int32_t*>(o.h_offsets.ensure((size_t)(n + 1) * 4));
            PA_HIP(hipMemcpyAsync(ho, doff, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, stream));
            PA_HIP(hipStreamSynchronize(stream));
            size_t total = (size_t)ho[n];
            void* hv = o.h_values.ensure(total ? total : 1);
            if (total) PA_HIP(hipMemcpyAsync(hv, dv, total, hipMemcpyDeviceToHost, stream));
            p.values = hv;
            p.offsets = ho;
        }
        else {
            size_t bytes = (size_t)n * type_width(o.type);
            void* hv = o.h_values.ensure(bytes ? bytes : 1);
            if (bytes) PA_HIP(hipMemcpyAsync(hv, dv, bytes, hipMemcpyDeviceToHost, stream));
            p.values = hv;
        }
        if (dn) {
            void* hn = o.h_nulls.ensure((size_t)(n > 0 ? n : 1));
            if (n) PA_HIP(hipMemcpyAsync(hn, dn, (size_t)n, hipMemcpyDeviceToHost, stream));
            p.nulls = static_cast<const uint8_t*>(hn);
        }
        need_sync = true;
    }
    if (need_sync) PA_HIP(hipStreamSynchronize(stream));
    out->position_count = n;
    out->channel_count = (int32_t)cols.size();
    out->columns = storage.data();
    out->mem = mem;
    out->flags = 0;
}

}  // namespace pa
