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
// Uncompressed, unencrypted, no checksum (markers = 0): LZ4 / AES / XXH64 framing stay with the Java PagesSerde.
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
        case PA_DATE: return "INT_ARRAY";
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

int64_t serialize_page(const pa_page* page, void* out_host, int64_t capacity, hipStream_t s)
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

pa_page_buffer* deserialize_page(const void* bytes, int64_t size, hipStream_t s)
{
    PA_REQUIRE(bytes != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    require_device();
    Reader r{static_cast<const uint8_t*>(bytes), size};
    const int32_t n = r.i32();
    const uint8_t markers = r.u8();
    PA_REQUIRE(markers == 0, PA_ERR_NOT_SUPPORTED, "compressed / encrypted / checksummed pages are decoded by the Java PagesSerde");
    const int32_t uncompressed = r.i32(), length = r.i32();
    PA_REQUIRE(n >= 0 && uncompressed == length && r.pos + length <= size, PA_ERR_INVALID_ARGUMENT, "bad SerializedPage header");
    const int32_t channels = r.i32();
    PA_REQUIRE(channels >= 0 && channels <= 4096, PA_ERR_INVALID_ARGUMENT, "bad channel count");
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
        if (name == "LONG_ARRAY") { oc.type = PA_BIGINT; width = 8; }        // the declared column type tells BIGINT from DOUBLE
        else if (name == "INT_ARRAY") { oc.type = PA_INTEGER; width = 4; }
        else if (name == "BYTE_ARRAY") { oc.type = PA_BOOLEAN; width = 1; }
        else if (name == "VARIABLE_WIDTH") {
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
