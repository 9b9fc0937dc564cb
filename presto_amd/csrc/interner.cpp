// interner.cpp -- host side of the string dictionaries (intern_kernels.hpp): sizing, growth and the id -> string decode.
#include <algorithm>

#include "intern_kernels.hpp"
#include "scan_kernels.hpp"

namespace pa {

namespace {
// a page is interned in slices of this many rows, so that "every row may be a new string" bounds the growth of the
// table and the id arrays by the slice and not by the page
constexpr int64_t kSliceRows = 4 << 20;
constexpr uint32_t kMinCap = 1u << 12;
}  // namespace

InternTable StringInterner::view() const
{
    InternTable t{};
    t.tag = tag_.as<uint64_t>();
    t.meta = meta_.as<uint64_t>();
    t.off = off_.as<uint32_t>();
    t.cap_mask = cap_ - 1;
    t.id_off = id_off_.as<uint32_t>();
    t.id_len = id_len_.as<uint32_t>();
    t.id_hash = id_hash_.as<uint64_t>();
    t.arena = arena_.as<uint64_t>();
    t.counters = counters_.as<uint32_t>();
    return t;
}

// Room for `rows` new strings of `bytes` bytes in total: table load factor <= 1/2, id arrays, arena (8-byte padded).
void StringInterner::reserve(int64_t rows, int64_t bytes, hipStream_t s)
{
    if (!counters_.ptr()) {
        counters_.ensure(64);
        PA_HIP(hipMemsetAsync(counters_.ptr(), 0, 64, s));
    }
    const uint64_t need_ids = (uint64_t)ids_ + (uint64_t)rows;
    const uint64_t need_words = (uint64_t)words_ + (uint64_t)bytes / 8 + (uint64_t)rows;
    PA_REQUIRE(need_ids < (1ULL << 30) && need_words < (1ULL << 32), PA_ERR_INSUFFICIENT_RESOURCES, "VARCHAR key dictionary is full");
    uint32_t cap = std::max(cap_, kMinCap);
    while ((uint64_t)cap < 2 * need_ids) cap <<= 1;
    if (cap != cap_) {
        DevBuf ntag, nmeta, noff;
        ntag.ensure((size_t)cap * 8);
        nmeta.ensure((size_t)cap * 8);
        noff.ensure((size_t)cap * 4);
        PA_HIP(hipMemsetAsync(ntag.ptr(), 0, (size_t)cap * 8, s));
        const uint32_t old_cap = cap_;
        DevBuf otag = std::move(tag_), ometa = std::move(meta_), ooff = std::move(off_);
        tag_ = std::move(ntag);
        meta_ = std::move(nmeta);
        off_ = std::move(noff);
        cap_ = cap;
        if (old_cap && ids_) launch_intern_rehash(otag.as<uint64_t>(), ometa.as<uint64_t>(), ooff.as<uint32_t>(), old_cap, view(), s);
        PA_HIP(hipStreamSynchronize(s));  // the old arrays go back to the pool here
    }
    id_off_.reserve_keep((size_t)need_ids * 4, (size_t)ids_ * 4, s);
    id_len_.reserve_keep((size_t)need_ids * 4, (size_t)ids_ * 4, s);
    id_hash_.reserve_keep((size_t)need_ids * 8, (size_t)ids_ * 8, s);
    arena_.reserve_keep((size_t)need_words * 8, (size_t)words_ * 8, s);
}

void StringInterner::settle()
{
    if (!pending_) return;
    PA_HIP(hipStreamSynchronize(pending_stream_));
    const uint32_t* h = h_.as<uint32_t>();
    ids_ = h[0];
    words_ = h[1];
    pending_ = false;
}

const int32_t* StringInterner::intern(const void* values, const int32_t* offsets, const uint8_t* nulls, int64_t n, hipStream_t s, int64_t bytes_hint)
{
    settle();
    int32_t* ids = static_cast<int32_t*>(ids_out_.ensure((size_t)std::max<int64_t>(n, 1) * 4));
    uint32_t* h = static_cast<uint32_t*>(h_.ensure(64));
    for (int64_t at = 0; at < n; at += kSliceRows) {
        const int64_t rows = std::min(kSliceRows, n - at);
        int64_t bytes = bytes_hint;
        if (bytes < 0) {
            int32_t* hb = reinterpret_cast<int32_t*>(h + 4);
            PA_HIP(hipMemcpyAsync(hb, offsets + at, 4, hipMemcpyDeviceToHost, s));
            PA_HIP(hipMemcpyAsync(hb + 1, offsets + at + rows, 4, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            PA_REQUIRE(hb[1] >= hb[0], PA_ERR_INVALID_ARGUMENT, "VARCHAR offsets are not ascending");
            bytes = (int64_t)hb[1] - hb[0];
        }
        reserve(rows, bytes, s);
        launch_intern(view(), values, offsets + at, nulls ? nulls + at : nullptr, rows, ids + at, s);
        PA_HIP(hipMemcpyAsync(h, counters_.ptr(), 8, hipMemcpyDeviceToHost, s));
        pending_ = true;
        pending_stream_ = s;
        if (at + rows < n) settle();   // the next slice's reservation starts from these
    }
    return ids;
}

void StringInterner::decode(const int32_t* ids, const uint8_t* nulls, int64_t n, DevBuf* values, DevBuf* offsets, hipStream_t s)
{
    settle();
    int32_t* offs = static_cast<int32_t*>(offsets->ensure((size_t)(n + 1) * 4));
    if (n == 0) {
        PA_HIP(hipMemsetAsync(offs, 0, 4, s));
        values->ensure(1);
        return;
    }
    PA_REQUIRE(cap_ != 0, PA_ERR_ILLEGAL_STATE, "internal: ids without a dictionary");
    launch_intern_lengths(view(), ids, nulls, n, offs, s);
    DevBuf temp, total;
    int32_t* t = static_cast<int32_t*>(total.ensure(64));
    launch_exclusive_scan_i32(offs, offs, n, t, temp.ensure(scan_temp_bytes(n)), s);
    PA_HIP(hipMemcpyAsync(offs + n, t, 4, hipMemcpyDeviceToDevice, s));
    int32_t* h = static_cast<int32_t*>(h_.ensure(64));
    PA_HIP(hipMemcpyAsync(h + 8, t, 4, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    const int32_t bytes = h[8];
    PA_REQUIRE(bytes >= 0, PA_ERR_INSUFFICIENT_RESOURCES, "VARCHAR key column exceeds 2 GiB");
    uint8_t* out = static_cast<uint8_t*>(values->ensure((size_t)std::max(bytes, 1)));
    launch_intern_bytes(view(), ids, nulls, n, offs, out, s);
}

void StringInterner::fetch_strings(uint32_t from, std::vector<std::string>* out, hipStream_t s)
{
    settle();
    if (from >= ids_) return;
    const uint32_t n = ids_ - from;
    // a small dictionary (the keys of a FINAL step, a handful of flags): offsets, lengths and the whole arena in one round trip
    const bool whole = (uint64_t)words_ * 8 <= (64u << 10);
    uint32_t* land = static_cast<uint32_t*>(h_fetch_.ensure((size_t)n * 8 + (whole ? (size_t)words_ * 8 : 0) + 8));
    uint32_t* off = land;
    uint32_t* len = land + n;
    PA_HIP(hipMemcpyAsync(off, id_off_.as<uint32_t>() + from, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    PA_HIP(hipMemcpyAsync(len, id_len_.as<uint32_t>() + from, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (whole && words_) PA_HIP(hipMemcpyAsync(land + 2 * (size_t)n, arena_.ptr(), (size_t)words_ * 8, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    // (the arena is handed out by an atomic cursor: the new ids' strings lie somewhere between the lowest of their offsets and its end)
    uint32_t lo = 0;
    std::vector<uint64_t> fetched;
    const uint64_t* words = reinterpret_cast<const uint64_t*>(land + 2 * (size_t)n);
    if (!whole) {
        lo = words_;
        for (uint32_t i = 0; i < n; i++) lo = std::min(lo, off[i]);
        fetched.resize((size_t)(words_ - lo) + 1);
        if (words_ > lo) {
            PA_HIP(hipMemcpyAsync(fetched.data(), arena_.as<uint64_t>() + lo, (size_t)(words_ - lo) * 8, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
        }
        words = fetched.data();
    }
    out->reserve(out->size() + n);
    for (uint32_t i = 0; i < n; i++) {
        PA_REQUIRE(off[i] >= lo && (uint64_t)(off[i] - lo) * 8 + len[i] <= (uint64_t)(words_ - lo) * 8, PA_ERR_DEVICE, "internal: dictionary entry outside the arena");
        out->emplace_back(reinterpret_cast<const char*>(words + (off[i] - lo)), (size_t)len[i]);
    }
}

}  // namespace pa
