// device_page.hpp -- Pages as columnar HBM buffers.
//
// An input pa_page is normalised into a DevPage whose columns are FLAT or VARWIDTH device arrays:
//   * PA_MEM_DEVICE pages are used in place (zero copy);
//   * PA_MEM_HOST pages (what the JNI shim hands over after copying out of the JVM heap, SURVEY 8b
//     "Ownership") are copied column by column into the operator's staging arena on its stream;
//   * DICTIONARY / RLE blocks are decoded by a gather kernel (DictionaryBlock.getLoadedBlock semantics).
// Output pages are produced in HBM and, for PA_MEM_HOST consumers, landed in pinned host memory.
#pragma once

#include <memory>
#include <vector>

#include "common.hpp"

namespace pa {

struct DevColumn {
    int32_t type = PA_BIGINT;
    bool varwidth = false;
    const void* values = nullptr;
    const int32_t* offsets = nullptr;
    const uint8_t* nulls = nullptr;  // nullptr = no nulls
};

struct DevPage {
    int32_t n = 0;
    std::vector<DevColumn> cols;
    // set: the page is the rows of these ranges, one behind the other (stable device pages that do not continue each other in
    // memory, handed to ONE launch as a table: op_fused.hpp); n is their total, cols those of the first
    std::shared_ptr<const std::vector<DevPage>> ranges;
};

class PageStager {
public:
    PageStager() = default;
    PageStager(const PageStager&) = delete;
    PageStager& operator=(const PageStager&) = delete;
    ~PageStager();
    // `needed` (may be null) limits staging to the channels an operator actually reads.
    DevPage stage(const pa_page* page, const std::vector<bool>* needed, hipStream_t stream);
    size_t bytes() const;

    // A host page of FLAT / VARWIDTH blocks up to this many bytes travels as ONE copy: its arrays are laid behind each other in a
    // pinned block and land in one HBM allocation (an enqueued copy costs ~4 us whatever its size: the 19 arrays of a 4-row page of
    // Q1's PARTIAL states were 85 us of staging, one copy is 6).
    static constexpr size_t kPackedLimit = 256 << 10;

private:
    void* arena(size_t index, size_t bytes);
    bool stage_packed(const pa_page* page, const std::vector<bool>* needed, hipStream_t stream, DevPage& out);
    std::vector<DevBuf> bufs_;
    size_t next_ = 0;
    // the pinned blocks of the packed path: two, taken in turn; `done` = the copy out of the block has been executed
    struct Packed {
        PinnedBuf host;
        hipEvent_t done = nullptr;
        bool pending = false;
    };
    Packed packed_[2];
    DevBuf packed_dev_[2];
    size_t packed_at_ = 0;
};

// A small pageable host page laid into pinned memory, so that ONE launch (or copy) can read all its arrays: an enqueued copy costs
// ~4 us whatever its size, and a page of Q1's PARTIAL states has 19 arrays of 4 rows.  Two pinned blocks taken in turn: the page of
// a call stays valid until the call after the next.
class PinnedPageCopy {
public:
    PinnedPageCopy() = default;
    PinnedPageCopy(const PinnedPageCopy&) = delete;
    PinnedPageCopy& operator=(const PinnedPageCopy&) = delete;
    ~PinnedPageCopy();
    // The same page with every needed FLAT / VARWIDTH array in pinned memory (flags = PA_PAGE_PINNED), or nullptr when the page does
    // not qualify (encoded blocks, more than PageStager::kPackedLimit bytes, no rows).  After enqueueing what reads it: used(stream).
    const pa_page* copy(const pa_page* page, const std::vector<bool>* needed);
    void used(hipStream_t stream);

private:
    struct Slot {
        PinnedBuf host;
        hipEvent_t done = nullptr;
        bool pending = false;
        std::vector<pa_column> cols;
        pa_page page{};
    };
    Slot slots_[2];
    size_t at_ = 0;
    Slot* cur_ = nullptr;
};

// One output Block under construction in HBM.
struct OutColumn {
    int32_t type = PA_BIGINT;
    bool varwidth = false;
    bool has_nulls = false;
    DevBuf values, offsets, nulls;
    // host landing zone (PA_MEM_HOST outputs)
    PinnedBuf h_values, h_offsets, h_nulls;
    // zero-copy view of an input column (identity projection of a fully selected device page)
    const void* view_values = nullptr;
    const int32_t* view_offsets = nullptr;
    const uint8_t* view_nulls = nullptr;
    bool is_view = false;
    // the h_* buffers already hold the block (operators that assemble small results on the host)
    bool host_ready = false;
};

// Fills `out` (whose `columns` array has room for cols.size() entries) from device columns; for
// PA_MEM_HOST copies the first `n` positions to pinned memory and synchronises the stream.
void publish_output(std::vector<OutColumn>& cols, int32_t n, int32_t mem, hipStream_t stream, pa_page* out,
                    std::vector<pa_column>& storage);

// static kernels used by staging (static_kernels.hip)
void launch_gather_flat(const void* src, int elem_bytes, const int32_t* positions, int64_t count, void* dst, hipStream_t s);
void launch_gather_nulls(const uint8_t* src, const int32_t* positions, int64_t count, uint8_t* dst, hipStream_t s);
// positions may hold -1 = NULL row; elem_bytes 0 = only the NULL flags (VARCHAR columns)
void launch_gather_or_null(const void* src, int elem_bytes, const uint8_t* src_nulls, const int32_t* positions, int64_t count, void* dst,
                           uint8_t* dst_nulls, hipStream_t s);
void launch_fill_flat(void* dst, int elem_bytes, const void* src_one, int64_t count, hipStream_t s);

}  // namespace pa
