// op_exchange.cpp -- the hash-partitioned exchange between the ranks (GPUs) of one node.
//
// Reference path replaced:
//   producer  PartitionedOutputOperator.PagePartitioner.partitionPage   (…/operator/PartitionedOutputOperator.java:411-431)
//             PartitioningExchanger.accept                              (…/operator/exchange/PartitioningExchanger.java:59-82)
//   routing   HashGenerator.getPartition                                (…/operator/HashGenerator.java:24-35)
//             LocalPartitionGenerator.getPartition                      (…/operator/exchange/LocalPartitionGenerator.java:45-65)
//   consumer  ExchangeOperator.getOutput                                (…/operator/ExchangeOperator.java)
// The reference appends every row to the PageBuilder of its partition, serialises full builders and lets the consumers pull
// them over HTTP.  Here a PartitionedOutput operator regroups each page by destination rank on the device (row hash ->
// partition id -> stable LDS-staged multisplit of every column, so a destination's rows keep their page order) and appends
// the runs to per-destination column buffers in HBM; when every sink of the rank has finished, the exchange source runs the
// two collectives of the whole exchange -- an all-gather of the (rows, VARCHAR bytes, nullability) count rows and ONE
// variable all-to-all of packed column segments (grouped ncclSend / ncclRecv, RCCL over xGMI) -- and hands the received rows
// out as one page, ordered by (source rank, source position).  One exchange = two collectives on every rank, whatever the
// page counts: the collective order cannot diverge between ranks.
#include <algorithm>
#include <atomic>
#include <mutex>

#include "comm.hpp"
#include "exchange_kernels.hpp"
#include "operator.hpp"
#include "scan_kernels.hpp"
#include "static_kernels.hpp"

namespace pa {
namespace {

inline int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }

// rows of one destination rank: one growing buffer per column (+ NULL flags, + lengths / bytes for VARCHAR)
struct DestColumn {
    DevBuf values;   // flat: elements; VARCHAR: the strings' bytes
    DevBuf lengths;  // VARCHAR: int32 per row
    DevBuf nulls;    // 1 B per row once the column has shown NULLs
    int64_t bytes = 0;  // VARCHAR bytes used
};
struct Dest {
    std::vector<DestColumn> cols;
    int64_t rows = 0;
};

}  // namespace
}  // namespace pa

struct pa_exchange {
    pa_comm* comm = nullptr;
    std::vector<int32_t> types;
    std::vector<int32_t> partition_channels;
    int hash_channel = -1;
    int local_rule = 1;
    int sink_count = 1;
    std::mutex mu;                  // sinks on different Driver threads append under it
    std::vector<pa::Dest> dest;     // per destination rank
    std::vector<bool> nullable;     // per channel: some page carried a valueIsNull array
    int sinks_created = 0, sinks_finished = 0;
    bool transferred = false;
    // stream of the sink that appended last (under mu): another sink drains it before it touches the destinations' buffers --
    // reserve_keep copies and recycles them on the caller's stream, and the previous appender's copy kernel may still write there
    hipStream_t last_appender = nullptr;
    // statistics
    int64_t rows_sent = 0, rows_received = 0, bytes_remote = 0;
    double transfer_ms = 0;
    int world() const { return comm->world; }
};

namespace pa {
namespace {

class PartitionedOutputOperator : public pa_operator {
public:
    PartitionedOutputOperator(pa_exchange* ex, void* stream) : ex_(ex), stream_(stream)
    {
        std::lock_guard<std::mutex> lock(ex_->mu);
        PA_REQUIRE(ex_->sinks_created < ex_->sink_count, PA_ERR_ILLEGAL_STATE, "more PartitionedOutput operators than pa_exchange_desc.sink_count");
        ex_->sinks_created++;
    }
    ~PartitionedOutputOperator() override { (void)hipStreamSynchronize(stream_.get()); }

    hipStream_t main_stream() override { return stream_.get(); }
    bool needs_input() override { return !finishing_; }
    bool get_output(pa_page*) override { return false; }
    bool is_finished() override { return finishing_; }
    void finish() override
    {
        if (finishing_) return;
        finishing_ = true;
        PA_HIP(hipStreamSynchronize(stream_.get()));  // the appended runs are complete before the source packs them
        std::lock_guard<std::mutex> lock(ex_->mu);
        if (ex_->last_appender == stream_.get()) ex_->last_appender = nullptr;  // (drained above; the stream may go away with the sink)
        ex_->sinks_finished++;
    }
    int64_t memory_bytes() override { return (int64_t)stager_.bytes(); }

    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(page != nullptr && page->channel_count == (int32_t)ex_->types.size(), PA_ERR_INVALID_ARGUMENT,
                   "page channel count does not match the exchange's types");
        const int64_t n = page->position_count;
        if (n == 0) return;
        hipStream_t s = stream_.get();
        const int W = ex_->world();
        const int C = (int)ex_->types.size();
        DevPage dp = stager_.stage(page, nullptr, s);
        for (int c = 0; c < C; c++) {
            PA_REQUIRE(dp.cols[c].type == ex_->types[c], PA_ERR_INVALID_ARGUMENT, "page block type does not match the exchange's declared type");
        }
        // ---- row -> destination rank ----
        int64_t* raw = static_cast<int64_t*>(raw_hash_.ensure((size_t)n * 8));
        const int64_t* raw_hash = raw;
        if (ex_->hash_channel >= 0) {
            raw_hash = static_cast<const int64_t*>(dp.cols[ex_->hash_channel].values);  // precomputed $hashvalue
        }
        else {
            HashPageArgs ha;
            memset(&ha, 0, sizeof ha);
            ha.ncols = (int32_t)ex_->partition_channels.size();
            for (int i = 0; i < ha.ncols; i++) {
                const DevColumn& col = dp.cols[ex_->partition_channels[i]];
                ha.col[i].values = col.values;
                ha.col[i].offsets = col.offsets;
                ha.col[i].nulls = col.nulls;
                ha.col[i].type = col.type;
            }
            ha.n = n;
            ha.out = raw;
            launch_hash_page(ha, s);
        }
        int32_t* part = static_cast<int32_t*>(part_.ensure((size_t)n * 4));
        launch_partition_ids(raw_hash, n, W, ex_->local_rule, part, s);
        // ---- every fixed-width column (and NULL flag array) regrouped by destination in one stable multisplit pass; VARCHAR
        //      columns through the regrouped row positions ----
        std::vector<MsplitCol> mc;
        struct Moved { int channel; int kind; DevBuf* buf; int width; };  // kind 0 values, 1 nulls, 2 positions
        std::vector<Moved> moved;
        size_t slot = 0;
        auto temp = [&](size_t bytes) -> DevBuf* {
            if (slot >= temp_.size()) temp_.push_back(std::make_unique<DevBuf>());
            temp_[slot]->ensure(bytes ? bytes : 1);
            return temp_[slot++].get();
        };
        bool any_varchar = false;
        for (int c = 0; c < C; c++) {
            const DevColumn& col = dp.cols[c];
            if (col.varwidth) any_varchar = true;
            else {
                const int w = type_width(col.type);
                DevBuf* b = temp((size_t)n * w);
                mc.push_back(MsplitCol{col.values, b->ptr(), w, 0});
                moved.push_back(Moved{c, 0, b, w});
            }
            if (col.nulls) {
                DevBuf* b = temp((size_t)n);
                mc.push_back(MsplitCol{col.nulls, b->ptr(), 1, 0});
                moved.push_back(Moved{c, 1, b, 1});
            }
        }
        DevBuf* positions = nullptr;
        if (any_varchar) {
            int32_t* iota = static_cast<int32_t*>(iota_.ensure((size_t)n * 4));
            launch_iota_i32(iota, n, s);
            positions = temp((size_t)n * 4);
            mc.push_back(MsplitCol{iota, positions->ptr(), 4, 0});
            moved.push_back(Moved{-1, 2, positions, 4});
        }
        PA_REQUIRE(mc.size() <= (size_t)kMsplitMaxCols, PA_ERR_NOT_SUPPORTED, "too many columns for one exchange page");
        int64_t* counts_dev = static_cast<int64_t*>(counts_.ensure((size_t)(W + 1) * 8));
        launch_msplit(part, n, W, mc.data(), (int32_t)mc.size(), counts_dev, msplit_temp_.ensure(msplit_temp_bytes(n, W)), s, true);
        int64_t* h = static_cast<int64_t*>(h_counts_.ensure((size_t)(2 * W + 2) * 8));
        PA_HIP(hipMemcpyAsync(h, counts_dev, (size_t)W * 8, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        std::vector<int64_t> first((size_t)W + 1, 0);
        for (int d = 0; d < W; d++) first[d + 1] = first[d] + h[d];
        PA_REQUIRE(first[W] == n, PA_ERR_DEVICE, "internal: the multisplit lost rows");
        // VARCHAR: lengths + bytes in destination order, and the byte total of every destination
        struct Var { int channel; DevBuf* lengths; DevBuf* offsets; DevBuf* bytes; std::vector<int64_t> first_byte; };
        std::vector<Var> vars;
        for (int c = 0; c < C && any_varchar; c++) {
            const DevColumn& col = dp.cols[c];
            if (!col.varwidth) continue;
            Var v;
            v.channel = c;
            v.lengths = temp((size_t)n * 4);
            v.offsets = temp((size_t)(n + 1) * 4);
            const int32_t* pos = positions->as<int32_t>();
            launch_varwidth_lengths(pos, n, col.offsets, col.nulls, v.lengths->as<int32_t>(), s);
            int32_t* total = static_cast<int32_t*>(temp(64)->ptr());
            PA_HIP(hipMemsetAsync(total, 0, 4, s));
            launch_exclusive_scan_i32(v.lengths->as<int32_t>(), v.offsets->as<int32_t>(), n, total, temp(scan_temp_bytes(n))->ptr(), s);
            PA_HIP(hipMemcpyAsync(v.offsets->as<int32_t>() + n, total, 4, hipMemcpyDeviceToDevice, s));
            // byte offset at every destination boundary
            int32_t* hb = reinterpret_cast<int32_t*>(h + W);
            for (int d = 0; d <= W; d++) {
                PA_HIP(hipMemcpyAsync(hb + d, v.offsets->as<int32_t>() + first[d], 4, hipMemcpyDeviceToHost, s));
            }
            PA_HIP(hipStreamSynchronize(s));
            PA_REQUIRE(hb[W] >= 0, PA_ERR_INSUFFICIENT_RESOURCES, "more than 2 GiB of VARCHAR bytes in one exchange page");
            v.first_byte.assign(hb, hb + W + 1);
            v.bytes = temp((size_t)std::max<int64_t>(hb[W], 1));
            launch_varwidth_copy(pos, n, col.offsets, static_cast<const uint8_t*>(col.values), col.nulls, v.offsets->as<int32_t>(),
                                 v.bytes->as<uint8_t>(), total, s);
            vars.push_back(std::move(v));
        }
        // ---- append the runs to the destinations' buffers ----
        std::lock_guard<std::mutex> lock(ex_->mu);
        if (ex_->last_appender != nullptr && ex_->last_appender != s) PA_HIP(hipStreamSynchronize(ex_->last_appender));
        ex_->last_appender = s;
        std::vector<CopySeg> segs;
        for (int c = 0; c < C; c++) {
            if (dp.cols[c].nulls && !ex_->nullable[c]) {
                // first page with NULLs on this channel: the rows appended so far get zero flags
                ex_->nullable[c] = true;
                for (int d = 0; d < W; d++) {
                    Dest& dst = ex_->dest[d];
                    if (dst.rows == 0) continue;
                    // (room for this page's rows too, so that the append below does not move the buffer again)
                    dst.cols[c].nulls.reserve_keep((size_t)(dst.rows + h[d]), 0, s);
                    PA_HIP(hipMemsetAsync(dst.cols[c].nulls.ptr(), 0, (size_t)dst.rows, s));
                }
            }
        }
        for (int d = 0; d < W; d++) {
            const int64_t cnt = h[d];
            if (cnt == 0) continue;
            Dest& dst = ex_->dest[d];
            for (const Moved& m : moved) {
                if (m.kind == 2) continue;
                DestColumn& dc = dst.cols[m.channel];
                DevBuf& buf = m.kind == 0 ? dc.values : dc.nulls;
                char* base = static_cast<char*>(buf.reserve_keep((size_t)(dst.rows + cnt) * m.width, (size_t)dst.rows * m.width, s));
                segs.push_back(CopySeg{m.buf->as<char>() + first[d] * m.width, base + dst.rows * m.width, cnt * m.width, 0});
            }
            for (int c = 0; c < C; c++) {
                // a nullable channel whose page has no NULLs: zero flags for these rows
                if (ex_->nullable[c] && !dp.cols[c].nulls) {
                    DestColumn& dc = dst.cols[c];
                    char* base = static_cast<char*>(dc.nulls.reserve_keep((size_t)(dst.rows + cnt), (size_t)dst.rows, s));
                    segs.push_back(CopySeg{nullptr, base + dst.rows, cnt, 0});
                }
            }
            for (const Var& v : vars) {
                DestColumn& dc = dst.cols[v.channel];
                char* lb = static_cast<char*>(dc.lengths.reserve_keep((size_t)(dst.rows + cnt) * 4, (size_t)dst.rows * 4, s));
                segs.push_back(CopySeg{v.lengths->as<char>() + first[d] * 4, lb + dst.rows * 4, cnt * 4, 0});
                const int64_t nb = v.first_byte[d + 1] - v.first_byte[d];
                char* bb = static_cast<char*>(dc.values.reserve_keep((size_t)std::max<int64_t>(dc.bytes + nb, 1), (size_t)dc.bytes, s));
                segs.push_back(CopySeg{v.bytes->as<char>() + v.first_byte[d], bb + dc.bytes, nb, 0});
                dc.bytes += nb;
            }
            dst.rows += cnt;
        }
        if (!segs.empty()) {
            launch_copy_segments(segs.data(), segs.size(), h_table_.ensure(copy_segments_table_bytes(segs.size())),
                                 d_table_.ensure(copy_segments_table_bytes(segs.size())), s);
        }
        ex_->rows_sent += n;
        // the staging / temporary buffers are reused by the next page: stream order protects them (same stream)
    }

private:
    pa_exchange* ex_;
    Stream stream_;
    PageStager stager_;
    bool finishing_ = false;
    DevBuf raw_hash_, part_, iota_, counts_, msplit_temp_, d_table_;
    PinnedBuf h_counts_, h_table_;
    std::vector<std::unique_ptr<DevBuf>> temp_;  // per-page scratch, reused from page to page (stream order protects it)
};

class ExchangeSourceOperator : public pa_operator {
public:
    ExchangeSourceOperator(pa_exchange* ex, int32_t output_mem, void* stream) : ex_(ex), output_mem_(output_mem), stream_(stream) {}
    ~ExchangeSourceOperator() override { (void)hipStreamSynchronize(stream_.get()); }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    bool needs_input() override { return false; }
    void add_input(const pa_page*) override { throw Error(PA_ERR_ILLEGAL_STATE, "an exchange source takes no input"); }
    void finish() override { closed_ = true; }  // SourceOperator.finish: stop producing
    bool is_finished() override { return closed_ || done_; }
    bool is_blocked() override { return !closed_ && !done_ && !sinks_done(); }
    int64_t memory_bytes() override
    {
        int64_t b = (int64_t)(send_blob_.capacity() + recv_blob_.capacity());
        for (const OutColumn& o : out_cols_) b += (int64_t)(o.values.capacity() + o.offsets.capacity() + o.nulls.capacity());
        return b;
    }

    bool get_output(pa_page* out) override
    {
        if (closed_ || done_ || !sinks_done()) return false;
        done_ = true;
        const int64_t rows = transfer();
        if (rows == 0) return false;
        publish_output(out_cols_, (int32_t)rows, output_mem_, stream_.get(), out, out_storage_);
        return true;
    }

private:
    bool sinks_done()
    {
        std::lock_guard<std::mutex> lock(ex_->mu);
        return ex_->sinks_finished >= ex_->sink_count;
    }

    // byte size of rank src's segment of column c for `rows` rows inside a blob (16-byte aligned pieces)
    struct ColLayout { int64_t values, lengths, nulls; };

    int64_t transfer()
    {
        std::lock_guard<std::mutex> lock(ex_->mu);
        PA_REQUIRE(!ex_->transferred, PA_ERR_ILLEGAL_STATE, "the exchange was already consumed");
        ex_->transferred = true;
        hipStream_t s = stream_.get();
        pa_comm* comm = ex_->comm;
        const int W = comm->world, me = comm->rank;
        const int C = (int)ex_->types.size();
        std::vector<int> var;  // VARCHAR channels
        for (int c = 0; c < C; c++) {
            if (ex_->types[c] == PA_VARCHAR) var.push_back(c);
        }
        const int V = (int)var.size();
        // ---- count rows: [rows -> d] [bytes of VARCHAR channel v -> d]... [nullable c] ----
        const int L = W * (1 + V) + C;
        std::vector<int64_t> row((size_t)L, 0), all((size_t)L * W, 0);
        for (int d = 0; d < W; d++) {
            row[d] = ex_->dest[d].rows;
            for (int v = 0; v < V; v++) row[(size_t)W * (1 + v) + d] = ex_->dest[d].cols[var[v]].bytes;
        }
        for (int c = 0; c < C; c++) row[(size_t)W * (1 + V) + c] = ex_->nullable[c] ? 1 : 0;
        comm_all_gather_i64(comm, row.data(), all.data(), L, s);
        auto rows_from_to = [&](int src, int dst) { return all[(size_t)src * L + dst]; };
        auto bytes_from_to = [&](int src, int v, int dst) { return all[(size_t)src * L + (size_t)W * (1 + v) + dst]; };
        auto nullable_at = [&](int src, int c) { return all[(size_t)src * L + (size_t)W * (1 + V) + c] != 0; };
        // the piece of a blob that holds `rows` rows of source `src` for one destination: per channel values / lengths+bytes / nulls
        auto blob_bytes = [&](int src, int dst) {
            const int64_t rows = rows_from_to(src, dst);
            int64_t b = 0;
            if (rows == 0) return b;
            int vi = 0;
            for (int c = 0; c < C; c++) {
                if (ex_->types[c] == PA_VARCHAR) {
                    b += align16(rows * 4) + align16(bytes_from_to(src, vi, dst));
                    vi++;
                }
                else b += align16(rows * type_width(ex_->types[c]));
                if (nullable_at(src, c)) b += align16(rows);
            }
            return b;
        };
        std::vector<int64_t> soff((size_t)W), sbytes((size_t)W), roff((size_t)W), rbytes((size_t)W);
        int64_t stotal = 0, rtotal = 0, total_rows = 0;
        for (int p = 0; p < W; p++) {
            soff[p] = stotal;
            sbytes[p] = blob_bytes(me, p);
            stotal += sbytes[p];
            roff[p] = rtotal;
            rbytes[p] = blob_bytes(p, me);
            rtotal += rbytes[p];
            total_rows += rows_from_to(p, me);
        }
        PA_REQUIRE(total_rows < ((int64_t)1 << 31), PA_ERR_INSUFFICIENT_RESOURCES, "an exchange delivers at most 2^31 - 1 rows to one rank");
        // ---- pack: destination p's blob = its column segments back to back ----
        char* sb = static_cast<char*>(send_blob_.ensure((size_t)std::max<int64_t>(stotal, 16)));
        char* rb = static_cast<char*>(recv_blob_.ensure((size_t)std::max<int64_t>(rtotal, 16)));
        std::vector<CopySeg> segs;
        for (int p = 0; p < W; p++) {
            const Dest& dst = ex_->dest[p];
            const int64_t rows = dst.rows;
            if (rows == 0) continue;
            char* at = sb + soff[p];
            for (int c = 0; c < C; c++) {
                const DestColumn& dc = dst.cols[c];
                if (ex_->types[c] == PA_VARCHAR) {
                    segs.push_back(CopySeg{dc.lengths.ptr(), at, rows * 4, 0});
                    at += align16(rows * 4);
                    segs.push_back(CopySeg{dc.values.ptr(), at, dc.bytes, 0});
                    at += align16(dc.bytes);
                }
                else {
                    const int w = type_width(ex_->types[c]);
                    segs.push_back(CopySeg{dc.values.ptr(), at, rows * w, 0});
                    at += align16(rows * w);
                }
                if (ex_->nullable[c]) {
                    segs.push_back(CopySeg{dc.nulls.ptr(), at, rows, 0});
                    at += align16(rows);
                }
            }
        }
        if (!segs.empty()) {
            launch_copy_segments(segs.data(), segs.size(), h_table_.ensure(copy_segments_table_bytes(segs.size())),
                                 d_table_.ensure(copy_segments_table_bytes(segs.size())), s);
        }
        // ---- the all-to-all ----
        hipEvent_t e0, e1;
        PA_HIP(hipEventCreate(&e0));
        PA_HIP(hipEventCreate(&e1));
        PA_HIP(hipEventRecord(e0, s));
        comm_all_to_all_v(comm, sb, soff.data(), sbytes.data(), rb, roff.data(), rbytes.data(), s);
        PA_HIP(hipEventRecord(e1, s));
        // the senders' buffers are not needed any more once the blob is packed (the stream has passed the pack by e0)
        // ---- unpack into one page: channel c = the segments of the sources in rank order ----
        out_cols_.clear();
        out_cols_.resize((size_t)C);
        segs.clear();
        std::vector<int64_t> var_total((size_t)V, 0);
        for (int v = 0; v < V; v++) {
            for (int p = 0; p < W; p++) var_total[v] += bytes_from_to(p, v, me);
            PA_REQUIRE(var_total[v] < ((int64_t)1 << 31), PA_ERR_INSUFFICIENT_RESOURCES, "more than 2 GiB of VARCHAR bytes delivered to one rank by one exchange");
        }
        std::vector<DevBuf> lengths((size_t)V);
        {
            int vi = 0;
            for (int c = 0; c < C; c++) {
                OutColumn& o = out_cols_[c];
                o.type = ex_->types[c];
                bool any_null = false;
                for (int p = 0; p < W; p++) any_null = any_null || (nullable_at(p, c) && rows_from_to(p, me) > 0);
                o.has_nulls = any_null;
                if (any_null) o.nulls.ensure((size_t)std::max<int64_t>(total_rows, 1));
                if (o.type == PA_VARCHAR) {
                    o.varwidth = true;
                    o.offsets.ensure((size_t)(total_rows + 1) * 4);
                    o.values.ensure((size_t)std::max<int64_t>(var_total[vi], 1));
                    lengths[vi].ensure((size_t)std::max<int64_t>(total_rows, 1) * 4);
                    vi++;
                }
                else o.values.ensure((size_t)std::max<int64_t>(total_rows, 1) * type_width(o.type));
            }
        }
        {
            int64_t row0 = 0;
            std::vector<int64_t> byte0((size_t)V, 0);
            for (int p = 0; p < W; p++) {
                const int64_t rows = rows_from_to(p, me);
                if (rows == 0) continue;
                const char* at = rb + roff[p];
                int vi = 0;
                for (int c = 0; c < C; c++) {
                    OutColumn& o = out_cols_[c];
                    if (o.type == PA_VARCHAR) {
                        segs.push_back(CopySeg{at, lengths[vi].as<char>() + row0 * 4, rows * 4, 0});
                        at += align16(rows * 4);
                        const int64_t nb = bytes_from_to(p, vi, me);
                        segs.push_back(CopySeg{at, o.values.as<char>() + byte0[vi], nb, 0});
                        at += align16(nb);
                        byte0[vi] += nb;
                        vi++;
                    }
                    else {
                        const int w = type_width(o.type);
                        segs.push_back(CopySeg{at, o.values.as<char>() + row0 * w, rows * w, 0});
                        at += align16(rows * w);
                    }
                    if (nullable_at(p, c)) {
                        segs.push_back(CopySeg{at, o.nulls.as<char>() + row0, rows, 0});
                        at += align16(rows);
                    }
                    else if (o.has_nulls) {
                        segs.push_back(CopySeg{nullptr, o.nulls.as<char>() + row0, rows, 0});
                    }
                }
                row0 += rows;
            }
        }
        if (!segs.empty()) {
            launch_copy_segments(segs.data(), segs.size(), h_table2_.ensure(copy_segments_table_bytes(segs.size())),
                                 d_table2_.ensure(copy_segments_table_bytes(segs.size())), s);
        }
        // VARCHAR offsets = exclusive scan of the concatenated lengths
        DevBuf scan_temp, total_dev;
        {
            int vi = 0;
            for (int c = 0; c < C; c++) {
                OutColumn& o = out_cols_[c];
                if (o.type != PA_VARCHAR) continue;
                int32_t* t = static_cast<int32_t*>(total_dev.ensure(64));
                PA_HIP(hipMemsetAsync(t, 0, 4, s));
                if (total_rows > 0) {
                    launch_exclusive_scan_i32(lengths[vi].as<int32_t>(), o.offsets.as<int32_t>(), total_rows, t, scan_temp.ensure(scan_temp_bytes(total_rows)), s);
                }
                PA_HIP(hipMemcpyAsync(o.offsets.as<int32_t>() + total_rows, t, 4, hipMemcpyDeviceToDevice, s));
                PA_HIP(hipStreamSynchronize(s));  // scan_temp / total_dev are reused by the next VARCHAR channel
                vi++;
            }
        }
        PA_HIP(hipStreamSynchronize(s));
        float ms = 0;
        PA_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        ex_->transfer_ms += ms;
        ex_->rows_received += total_rows;
        for (int p = 0; p < W; p++) {
            if (p != me) ex_->bytes_remote += sbytes[p];
        }
        // the destinations' buffers and the blobs go back to the pool (everything on the stream is complete)
        ex_->dest.clear();
        ex_->dest.resize((size_t)W);
        send_blob_.release();
        recv_blob_.release();
        return total_rows;
    }

    pa_exchange* ex_;
    int32_t output_mem_;
    Stream stream_;
    bool closed_ = false, done_ = false;
    DevBuf send_blob_, recv_blob_, d_table_, d_table2_;
    PinnedBuf h_table_, h_table2_;
    std::vector<OutColumn> out_cols_;
    std::vector<pa_column> out_storage_;
};

}  // namespace

pa_exchange* exchange_new(const pa_exchange_desc* d, pa_comm* comm)
{
    require_device();
    PA_REQUIRE(d != nullptr && comm != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    PA_REQUIRE(d->channel_count > 0 && d->channel_count <= 32 && d->types != nullptr, PA_ERR_NOT_SUPPORTED, "1..32 exchange channels");
    PA_REQUIRE(comm->world <= 256, PA_ERR_NOT_SUPPORTED, "at most 256 ranks");
    auto ex = std::make_unique<pa_exchange>();
    ex->comm = comm;
    ex->types.assign(d->types, d->types + d->channel_count);
    for (int32_t t : ex->types) PA_REQUIRE(t >= PA_BIGINT && t <= PA_VARCHAR, PA_ERR_NOT_SUPPORTED, "unknown exchange channel type");
    ex->hash_channel = d->hash_channel;
    if (ex->hash_channel >= 0) {
        PA_REQUIRE(ex->hash_channel < d->channel_count && ex->types[ex->hash_channel] == PA_BIGINT, PA_ERR_INVALID_ARGUMENT, "hash channel must be a BIGINT channel");
    }
    else {
        PA_REQUIRE(d->partition_channel_count > 0 && d->partition_channel_count <= 16 && d->partition_channels != nullptr, PA_ERR_INVALID_ARGUMENT,
                   "1..16 partition channels (or a hash channel)");
        for (int i = 0; i < d->partition_channel_count; i++) {
            PA_REQUIRE(d->partition_channels[i] >= 0 && d->partition_channels[i] < d->channel_count, PA_ERR_INVALID_ARGUMENT, "partition channel out of range");
            ex->partition_channels.push_back(d->partition_channels[i]);
        }
    }
    const bool pow2 = (comm->world & (comm->world - 1)) == 0;
    ex->local_rule = d->partition_rule < 0 ? (pow2 ? 1 : 0) : d->partition_rule;
    PA_REQUIRE(!ex->local_rule || pow2, PA_ERR_INVALID_ARGUMENT, "partitionCount must be a power of 2");  // LocalPartitionGenerator.java:38
    ex->sink_count = d->sink_count > 0 ? d->sink_count : 1;
    ex->dest.resize((size_t)comm->world);
    for (Dest& dst : ex->dest) dst.cols.resize(ex->types.size());
    ex->nullable.assign(ex->types.size(), false);
    return ex.release();
}

void exchange_delete(pa_exchange* ex) { delete ex; }

void exchange_stats(pa_exchange* ex, int64_t* rows_sent, int64_t* rows_received, int64_t* bytes_remote, double* transfer_ms)
{
    std::lock_guard<std::mutex> lock(ex->mu);
    if (rows_sent) *rows_sent = ex->rows_sent;
    if (rows_received) *rows_received = ex->rows_received;
    if (bytes_remote) *bytes_remote = ex->bytes_remote;
    if (transfer_ms) *transfer_ms = ex->transfer_ms;
}

pa_operator* make_partitioned_output(pa_exchange* ex, void* stream)
{
    PA_REQUIRE(ex != nullptr, PA_ERR_INVALID_ARGUMENT, "exchange is null");
    return new PartitionedOutputOperator(ex, stream);
}

pa_operator* make_exchange_source(pa_exchange* ex, int32_t output_mem, void* stream)
{
    PA_REQUIRE(ex != nullptr, PA_ERR_INVALID_ARGUMENT, "exchange is null");
    return new ExchangeSourceOperator(ex, output_mem, stream);
}

}  // namespace pa
