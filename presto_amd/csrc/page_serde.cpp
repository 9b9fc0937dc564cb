// page_serde.cpp -- the wire form of a Page, so that exchange pages can be read from / written for Java
// workers without a CPU pass over the rows:
//   SerializedPage frame   PagesSerdeUtil.writeSerializedPage (core/trino-main/src/main/java/io/trino/execution/buffer/
//                          PagesSerdeUtil.java:66-74): positionCount int, pageCodecMarkers byte, uncompressedSize int, size int
//   payload                PagesSerdeUtil.writeRawPage (:45-52): channelCount int, then per block the length-prefixed encoding
//                          name (InternalBlockEncodingSerde.java:56-80) and the encoding's body:
//     LONG_ARRAY / INT_ARRAY / BYTE_ARRAY   core/trino-spi/.../block/LongArrayBlockEncoding.java:38-61 (Int / Byte alike):
//                          positionCount int, nulls as bits (EncoderUtil.java:35-72), then all values when the block has no
//                          NULL array, else nonNullCount int + the non-null values only
//     VARIABLE_WIDTH       VariableWidthBlockEncoding.java:37-58: positionCount int, the END offset of every position (ints),
//                          nulls as bits, totalLength int, the bytes
// Markers (PageCodecMarker.java:27-30): COMPRESSED = the payload is one LZ4 block (PagesSerde.java:74-95: kept only when it
// shrinks the payload to <= 0.8 of its size); ENCRYPTED (the spill cipher) stays with the Java PagesSerde.  The LZ4 block codec
// below is written from the public block-format description (token = literal length << 4 | match length - 4, 255-extension
// bytes, 2-byte little-endian offsets, last 5 bytes literal, last match starting >= 12 bytes before the end): any decoder of the
// format -- airlift's Lz4Decompressor on the Java side -- reads what it writes, and it reads any encoder's output.  It runs on
// the host: the frame is host memory on its way to / from the network.
// Device side: null bytes <-> bits, compaction / re-expansion of the non-null values (stable partition + gather / scatter),
// offset re-basing; the host only writes the handful of header ints and moves each segment with one copy.
#include <cstring>

#include "operator.hpp"
#include "scan_kernels.hpp"

namespace pa {

namespace {

const char* encoding_name(int32_t type)
{
    switch (type) {
        case PA_BIGINT:
        case PA_DOUBLE: return "LONG_ARRAY";   // DoubleType stores doubleToLongBits in a LongArrayBlock (DoubleType.java:98-108)
        case PA_INTEGER:
        case PA_DATE:
        case PA_REAL: return "INT_ARRAY";      // RealType stores floatToRawIntBits in an IntArrayBlock (RealType.java)
        case PA_BOOLEAN: return "BYTE_ARRAY";
        case PA_VARCHAR: return "VARIABLE_WIDTH";
        default: throw Error(PA_ERR_NOT_SUPPORTED, "block type has no wire encoding here");
    }
}

struct Writer {
    uint8_t* base;
    int64_t cap;
    int64_t pos = 0;
    void need(int64_t bytes) { PA_REQUIRE(pos + bytes <= cap, PA_ERR_INSUFFICIENT_RESOURCES, "serialization buffer too small"); }
    void i32(int32_t v) { need(4); memcpy(base + pos, &v, 4); pos += 4; }
    void u8(uint8_t v) { need(1); base[pos++] = v; }
    void raw(const void* p, int64_t n) { need(n); memcpy(base + pos, p, (size_t)n); pos += n; }
    uint8_t* reserve(int64_t n) { need(n); uint8_t* p = base + pos; pos += n; return p; }
};

// ---- LZ4 block format ----
int64_t lz4_max_compressed(int64_t n) { return n + n / 255 + 16; }

int64_t lz4_compress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap)
{
    const uint8_t* const end = src + n;
    uint8_t* op = dst;
    uint8_t* const oend = dst + cap;
    const uint8_t* anchor = src;
    auto emit = [&](const uint8_t* lit, int64_t lit_len, int64_t match_len, int offset) -> bool {
        if (op + 1 + lit_len + lit_len / 255 + 8 + match_len / 255 > oend) return false;
        uint8_t* token = op++;
        int64_t l = lit_len;
        if (l >= 15) {
            *token = 15 << 4;
            for (l -= 15; l >= 255; l -= 255) *op++ = 255;
            *op++ = (uint8_t)l;
        }
        else *token = (uint8_t)(l << 4);
        memcpy(op, lit, (size_t)lit_len);
        op += lit_len;
        if (match_len == 0) return true;  // final literals
        *op++ = (uint8_t)(offset & 255);
        *op++ = (uint8_t)(offset >> 8);
        int64_t m = match_len - 4;
        if (m >= 15) {
            *token |= 15;
            for (m -= 15; m >= 255; m -= 255) *op++ = 255;
            *op++ = (uint8_t)m;
        }
        else *token |= (uint8_t)m;
        return true;
    };
    if (n >= 13) {
        std::vector<int32_t> table(1 << 14, -1);
        const uint8_t* ip = src;
        const uint8_t* const mflimit = end - 12;   // a match may not start later
        const uint8_t* const matchlimit = end - 5; // ... nor reach into the last 5 bytes
        while (ip < mflimit) {
            uint32_t v;
            memcpy(&v, ip, 4);
            const uint32_t h = (v * 2654435761u) >> 18;
            const int32_t cand = table[h];
            table[h] = (int32_t)(ip - src);
            uint32_t cv = 0;
            if (cand >= 0) memcpy(&cv, src + cand, 4);
            if (cand < 0 || cv != v || (ip - src) - cand > 65535) {
                ip++;
                continue;
            }
            const uint8_t* m = src + cand;
            const uint8_t* p = ip + 4;
            const uint8_t* q = m + 4;
            while (p < matchlimit && *p == *q) { p++; q++; }
            if (!emit(anchor, ip - anchor, p - ip, (int)(ip - m))) return -1;
            ip = p;
            anchor = ip;
        }
    }
    if (!emit(anchor, end - anchor, 0, 0)) return -1;
    return op - dst;
}

// returns the decompressed size, or -1 for a malformed block / one that does not fit dst
int64_t lz4_decompress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap)
{
    const uint8_t* ip = src;
    const uint8_t* const iend = src + n;
    uint8_t* op = dst;
    uint8_t* const oend = dst + cap;
    while (ip < iend) {
        const uint8_t token = *ip++;
        int64_t lit = token >> 4;
        if (lit == 15) {
            uint8_t b;
            do {
                if (ip >= iend) return -1;
                b = *ip++;
                lit += b;
            } while (b == 255);
        }
        if (lit > iend - ip || lit > oend - op) return -1;
        memcpy(op, ip, (size_t)lit);
        ip += lit;
        op += lit;
        if (ip >= iend) break;  // the last sequence has literals only
        if (iend - ip < 2) return -1;
        const int offset = ip[0] | (ip[1] << 8);
        ip += 2;
        if (offset == 0 || offset > op - dst) return -1;
        int64_t ml = token & 15;
        if (ml == 15) {
            uint8_t b;
            do {
                if (ip >= iend) return -1;
                b = *ip++;
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (ml > oend - op) return -1;
        const uint8_t* m = op - offset;
        for (int64_t i = 0; i < ml; i++) op[i] = m[i];  // overlapping copies replicate (offset < length)
        op += ml;
    }
    return op - dst;
}

struct Reader {
    const uint8_t* base;
    int64_t size;
    int64_t pos = 0;
    void need(int64_t bytes) const { PA_REQUIRE(bytes >= 0 && pos + bytes <= size, PA_ERR_INVALID_ARGUMENT, "serialized page is truncated"); }
    int32_t i32() { need(4); int32_t v; memcpy(&v, base + pos, 4); pos += 4; return v; }
    uint8_t u8() { need(1); return base[pos++]; }
    const uint8_t* take(int64_t n) { need(n); const uint8_t* p = base + pos; pos += n; return p; }
};

}  // namespace

int64_t serialize_page(const pa_page* page, void* out_host, int64_t capacity, hipStream_t s, bool compress)
{
    PA_REQUIRE(page != nullptr && out_host != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    require_device();
    PageStager stager;
    DevPage dp = stager.stage(page, nullptr, s);  // host pages are uploaded, dictionary / RLE blocks decoded
    const int64_t n = dp.n;
    Writer w{static_cast<uint8_t*>(out_host), capacity};
    w.i32((int32_t)n);
    w.u8(0);                          // PageCodecMarker set: neither COMPRESSED nor ENCRYPTED nor CHECKSUMMED
    uint8_t* sizes = w.reserve(8);    // uncompressedSize, size: patched below
    const int64_t payload_start = w.pos;
    w.i32((int32_t)dp.cols.size());
    DevBuf bits, part, pos, counts, temp, packed_vals, ends;
    for (const DevColumn& col : dp.cols) {
        const char* name = encoding_name(col.type);
        w.i32((int32_t)strlen(name));
        w.raw(name, (int64_t)strlen(name));
        w.i32((int32_t)n);
        int32_t total_bytes = 0, first = 0;
        if (col.varwidth) {
            // END offset of every position
            int32_t h[2] = {0, 0};
            if (n > 0) {
                launch_varwidth_ends(col.offsets, n, static_cast<int32_t*>(ends.ensure((size_t)n * 4)), s);
                PA_HIP(hipMemcpyAsync(w.reserve(n * 4), ends.ptr(), (size_t)n * 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipMemcpyAsync(&h[0], col.offsets, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipMemcpyAsync(&h[1], col.offsets + n, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
            }
            first = h[0];
            total_bytes = h[1] - h[0];
        }
        // nulls as bits
        w.u8(col.nulls ? 1 : 0);
        if (col.nulls && n > 0) {
            launch_pack_null_bits(col.nulls, n, static_cast<uint8_t*>(bits.ensure((size_t)(n + 7) / 8)), s);
            PA_HIP(hipMemcpyAsync(w.reserve((n + 7) / 8), bits.ptr(), (size_t)(n + 7) / 8, hipMemcpyDeviceToHost, s));
        }
        if (col.varwidth) {
            w.i32(total_bytes);
            if (total_bytes > 0) PA_HIP(hipMemcpyAsync(w.reserve(total_bytes), static_cast<const uint8_t*>(col.values) + first, (size_t)total_bytes, hipMemcpyDeviceToHost, s));
            continue;
        }
        const int width = type_width(col.type);
        if (!col.nulls) {
            if (n > 0) PA_HIP(hipMemcpyAsync(w.reserve(n * width), col.values, (size_t)n * width, hipMemcpyDeviceToHost, s));
            continue;
        }
        // the non-null values only: stable partition by the NULL flag, gather
        int64_t non_null = 0;
        if (n > 0) {
            int32_t* flags = static_cast<int32_t*>(part.ensure((size_t)n * 4));
            int32_t* positions = static_cast<int32_t*>(pos.ensure((size_t)n * 4));
            int64_t* cnt = static_cast<int64_t*>(counts.ensure(64));
            launch_null_flag(col.nulls, n, flags, s);
            launch_partition_positions(flags, n, 2, positions, cnt, temp.ensure(partition_temp_bytes(n, 2)), s);
            int64_t h[2];
            PA_HIP(hipMemcpyAsync(h, cnt, 16, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            non_null = h[0];
            if (non_null > 0) launch_gather_flat(col.values, width, positions, non_null, packed_vals.ensure((size_t)non_null * width), s);
        }
        w.i32((int32_t)non_null);
        if (non_null > 0) PA_HIP(hipMemcpyAsync(w.reserve(non_null * width), packed_vals.ptr(), (size_t)non_null * width, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));  // packed_vals / positions are reused by the next column
    }
    PA_HIP(hipStreamSynchronize(s));
    const int32_t payload = (int32_t)(w.pos - payload_start);
    memcpy(sizes, &payload, 4);      // uncompressedSizeInBytes
    memcpy(sizes + 4, &payload, 4);  // sizeInBytes of the (uncompressed) slice that follows
    if (compress && payload > 0) {
        // PagesSerde.serialize (PagesSerde.java:74-95): compress the payload, keep the result only at <= 0.8 of its size
        std::vector<uint8_t> packed((size_t)lz4_max_compressed(payload));
        const int64_t c = lz4_compress(w.base + payload_start, payload, packed.data(), (int64_t)packed.size());
        if (c > 0 && (double)c / payload <= 0.8) {
            memcpy(w.base + payload_start, packed.data(), (size_t)c);
            w.base[4] = 1;  // PageCodecMarker.COMPRESSED
            const int32_t size = (int32_t)c;
            memcpy(sizes + 4, &size, 4);
            return payload_start + c;
        }
    }
    return w.pos;
}

// a deserialized page and the device buffers behind it
struct PageBuffer {
    std::vector<OutColumn> cols;
    std::vector<pa_column> storage;
    pa_page page{};
};

}  // namespace pa

struct pa_page_buffer {
    pa::PageBuffer impl;
};

namespace pa {

pa_page_buffer* deserialize_page(const void* bytes, int64_t size, hipStream_t s, const int32_t* expected_types, int32_t expected_count)
{
    PA_REQUIRE(bytes != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    require_device();
    Reader frame{static_cast<const uint8_t*>(bytes), size};
    const int32_t n = frame.i32();
    const uint8_t markers = frame.u8();
    PA_REQUIRE((markers & ~1) == 0, PA_ERR_NOT_SUPPORTED, "encrypted pages are decoded by the Java PagesSerde (the spill cipher lives there)");
    const int32_t uncompressed = frame.i32(), length = frame.i32();
    PA_REQUIRE(n >= 0 && uncompressed >= 0 && length >= 0 && frame.pos + length <= size, PA_ERR_INVALID_ARGUMENT, "bad SerializedPage header");
    // the payload, and nothing behind it: every read below is confined to [payload, payload + length) of THIS frame -- the
    // bytes come from another worker
    std::vector<uint8_t> inflated;
    Reader r{frame.base + frame.pos, length};
    if (markers & 1) {  // PageCodecMarker.COMPRESSED (PagesSerde.java:139-156)
        inflated.resize((size_t)std::max(uncompressed, 1));
        PA_REQUIRE(lz4_decompress(frame.base + frame.pos, length, inflated.data(), uncompressed) == uncompressed, PA_ERR_INVALID_ARGUMENT,
                   "compressed page does not inflate to its uncompressed size");
        r = Reader{inflated.data(), uncompressed};
    }
    else PA_REQUIRE(uncompressed == length, PA_ERR_INVALID_ARGUMENT, "bad SerializedPage header");
    const int32_t channels = r.i32();
    PA_REQUIRE(channels >= 0 && channels <= 4096, PA_ERR_INVALID_ARGUMENT, "bad channel count");
    PA_REQUIRE(expected_types == nullptr || expected_count == channels, PA_ERR_INVALID_ARGUMENT, "page has a different number of channels than the declared types");
    auto out = std::make_unique<pa_page_buffer>();
    PageBuffer& pb = out->impl;
    pb.cols.resize((size_t)channels);
    DevBuf stage_bits, stage_vals, part, pos, counts, temp, stage_ends;
    for (int32_t c = 0; c < channels; c++) {
        const int32_t name_len = r.i32();
        PA_REQUIRE(name_len > 0 && name_len < 64, PA_ERR_INVALID_ARGUMENT, "bad block encoding name");
        const std::string name(reinterpret_cast<const char*>(r.take(name_len)), (size_t)name_len);
        OutColumn& oc = pb.cols[(size_t)c];
        const int32_t positions = r.i32();
        PA_REQUIRE(positions == n, PA_ERR_INVALID_ARGUMENT, "block position count differs from the page's");
        const uint8_t* ends_bytes = nullptr;
        int width = 0;
        // the wire carries encodings, not types: the consumer's declared types tell DOUBLE from BIGINT and DATE from INTEGER
        const int32_t want = expected_types ? expected_types[c] : -1;
        if (name == "LONG_ARRAY") {
            PA_REQUIRE(want < 0 || want == PA_BIGINT || want == PA_DOUBLE, PA_ERR_INVALID_ARGUMENT, "LONG_ARRAY block for a channel declared otherwise");
            oc.type = want == PA_DOUBLE ? PA_DOUBLE : PA_BIGINT;
            width = 8;
        }
        else if (name == "INT_ARRAY") {
            PA_REQUIRE(want < 0 || want == PA_INTEGER || want == PA_DATE || want == PA_REAL, PA_ERR_INVALID_ARGUMENT, "INT_ARRAY block for a channel declared otherwise");
            oc.type = want == PA_DATE ? PA_DATE : (want == PA_REAL ? PA_REAL : PA_INTEGER);
            width = 4;
        }
        else if (name == "BYTE_ARRAY") {
            PA_REQUIRE(want < 0 || want == PA_BOOLEAN, PA_ERR_INVALID_ARGUMENT, "BYTE_ARRAY block for a channel declared otherwise");
            oc.type = PA_BOOLEAN;
            width = 1;
        }
        else if (name == "VARIABLE_WIDTH") {
            PA_REQUIRE(want < 0 || want == PA_VARCHAR, PA_ERR_INVALID_ARGUMENT, "VARIABLE_WIDTH block for a channel declared otherwise");
            oc.type = PA_VARCHAR;
            oc.varwidth = true;
            ends_bytes = r.take((int64_t)n * 4);
        }
        else throw Error(PA_ERR_NOT_SUPPORTED, "block encoding " + name + " is decoded on the Java side");
        const bool may_have_null = r.u8() != 0;
        oc.has_nulls = may_have_null;
        if (may_have_null && n > 0) {
            const int64_t nb = ((int64_t)n + 7) / 8;
            PA_HIP(hipMemcpyAsync(stage_bits.ensure((size_t)nb), r.take(nb), (size_t)nb, hipMemcpyHostToDevice, s));
            launch_unpack_null_bits(stage_bits.as<uint8_t>(), n, static_cast<uint8_t*>(oc.nulls.ensure((size_t)n)), s);
        }
        else if (may_have_null) oc.nulls.ensure(1);
        if (oc.varwidth) {
            int32_t* offs = static_cast<int32_t*>(oc.offsets.ensure((size_t)(n + 1) * 4));
            if (n > 0) PA_HIP(hipMemcpyAsync(stage_ends.ensure((size_t)n * 4), ends_bytes, (size_t)n * 4, hipMemcpyHostToDevice, s));
            launch_varwidth_from_ends(stage_ends.as<int32_t>(), n, offs, s);
            const int32_t total = r.i32();
            PA_REQUIRE(total >= 0, PA_ERR_INVALID_ARGUMENT, "negative VARCHAR length");
            {
                // the end offsets index the bytes that follow: non-negative, ascending, the last one the total -- every
                // downstream varwidth kernel trusts them
                int32_t prev = 0;
                for (int32_t i = 0; i < n; i++) {
                    int32_t e;
                    memcpy(&e, ends_bytes + (size_t)i * 4, 4);
                    PA_REQUIRE(e >= prev && e <= total, PA_ERR_INVALID_ARGUMENT, "VARIABLE_WIDTH end offsets are not ascending within the block's bytes");
                    prev = e;
                }
                PA_REQUIRE(n == 0 || prev == total, PA_ERR_INVALID_ARGUMENT, "VARIABLE_WIDTH end offsets do not end at the block's byte count");
            }
            oc.values.ensure((size_t)(total > 0 ? total : 1));
            if (total > 0) PA_HIP(hipMemcpyAsync(oc.values.ptr(), r.take(total), (size_t)total, hipMemcpyHostToDevice, s));
            PA_HIP(hipStreamSynchronize(s));  // stage_ends is reused
            continue;
        }
        oc.values.ensure((size_t)std::max(n, 1) * width);
        if (!may_have_null) {
            if (n > 0) PA_HIP(hipMemcpyAsync(oc.values.ptr(), r.take((int64_t)n * width), (size_t)n * width, hipMemcpyHostToDevice, s));
            continue;
        }
        const int32_t non_null = r.i32();
        PA_REQUIRE(non_null >= 0 && non_null <= n, PA_ERR_INVALID_ARGUMENT, "bad non-null count");
        if (n > 0) {
            // values of the non-null positions back in place (NULL positions hold 0, as LongArrayBlockEncoding.readBlock leaves them)
            PA_HIP(hipMemsetAsync(oc.values.ptr(), 0, (size_t)n * width, s));
            int32_t* flags = static_cast<int32_t*>(part.ensure((size_t)n * 4));
            int32_t* where = static_cast<int32_t*>(pos.ensure((size_t)n * 4));
            int64_t* cnt = static_cast<int64_t*>(counts.ensure(64));
            launch_null_flag(oc.nulls.as<uint8_t>(), n, flags, s);
            launch_partition_positions(flags, n, 2, where, cnt, temp.ensure(partition_temp_bytes(n, 2)), s);
            int64_t h[2];
            PA_HIP(hipMemcpyAsync(h, cnt, 16, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            PA_REQUIRE(h[0] == non_null, PA_ERR_INVALID_ARGUMENT, "non-null count does not match the null bits");
            if (non_null > 0) {
                PA_HIP(hipMemcpyAsync(stage_vals.ensure((size_t)non_null * width), r.take((int64_t)non_null * width), (size_t)non_null * width, hipMemcpyHostToDevice, s));
                launch_scatter_flat(stage_vals.ptr(), width, where, non_null, oc.values.ptr(), s);
            }
            PA_HIP(hipStreamSynchronize(s));  // staging buffers are reused by the next column
        }
    }
    PA_HIP(hipStreamSynchronize(s));
    publish_output(pb.cols, n, PA_MEM_DEVICE, s, &pb.page, pb.storage);
    return out.release();
}

void page_buffer_page(pa_page_buffer* buffer, pa_page* out) { *out = buffer->impl.page; }
void page_buffer_free(pa_page_buffer* buffer) { delete buffer; }

}  // namespace pa
