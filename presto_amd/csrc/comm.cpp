// comm.cpp -- see comm.hpp.  RCCL (ncclSend / ncclRecv groups over xGMI) or the host's callbacks.
#include "comm.hpp"

#include <rccl/rccl.h>

#include <algorithm>

namespace pa {

#define PA_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            throw ::pa::Error(PA_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(r_));      \
        }                                                                                              \
    } while (0)

static_assert(sizeof(ncclUniqueId) == PA_COMM_ID_BYTES, "PA_COMM_ID_BYTES must match ncclUniqueId");

void comm_unique_id(void* out128)
{
    require_device();
    ncclUniqueId id;
    PA_NCCL(ncclGetUniqueId(&id));
    memcpy(out128, &id, sizeof id);
}

pa_comm* comm_create_rccl(const void* unique_id, int32_t rank, int32_t world)
{
    require_device();
    PA_REQUIRE(unique_id != nullptr && world >= 1 && rank >= 0 && rank < world, PA_ERR_INVALID_ARGUMENT, "bad communicator arguments");
    auto c = std::make_unique<pa_comm>();
    c->rank = rank;
    c->world = world;
    PA_HIP(hipGetDevice(&c->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    PA_NCCL(ncclCommInitRank(&comm, world, id, rank));
    c->nccl = comm;
    return c.release();
}

pa_comm* comm_create_host(const pa_host_transport* t, int32_t rank, int32_t world)
{
    require_device();
    PA_REQUIRE(t != nullptr && t->all_gather_i64 != nullptr && t->all_to_all_v != nullptr, PA_ERR_INVALID_ARGUMENT, "host transport without callbacks");
    PA_REQUIRE(world >= 1 && rank >= 0 && rank < world, PA_ERR_INVALID_ARGUMENT, "bad communicator arguments");
    auto c = std::make_unique<pa_comm>();
    c->rank = rank;
    c->world = world;
    PA_HIP(hipGetDevice(&c->device));
    c->host = true;
    c->transport = *t;
    return c.release();
}

static void host_status(int32_t rc, const char* what)
{
    if (rc < 0) throw Error(rc, std::string("host transport: ") + what + " failed");
}

void comm_all_gather_i64(pa_comm* c, const int64_t* send, int64_t* recv, int32_t count, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(c->mu);
    c->collectives++;
    if (c->host) {
        host_status(c->transport.all_gather_i64(c->transport.ctx, send, recv, count), "all_gather_i64");
        return;
    }
    const size_t bytes = (size_t)count * 8;
    char* d = static_cast<char*>(c->dev_scratch.ensure(bytes * (size_t)(c->world + 1)));
    char* h = static_cast<char*>(c->host_scratch.ensure(bytes * (size_t)(c->world + 1)));
    memcpy(h, send, bytes);
    PA_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
    PA_NCCL(ncclAllGather(d, d + bytes, (size_t)count, ncclInt64, static_cast<ncclComm_t>(c->nccl), s));
    PA_HIP(hipMemcpyAsync(h + bytes, d + bytes, bytes * (size_t)c->world, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    memcpy(recv, h + bytes, bytes * (size_t)c->world);
}

void comm_all_reduce_i64(pa_comm* c, int64_t* values, int32_t count, int op, hipStream_t s)
{
    if (c->world == 1) return;
    // through the all-gather: counts and key ranges are a handful of words, and the host transport needs nothing else
    std::vector<int64_t> all((size_t)count * c->world);
    comm_all_gather_i64(c, values, all.data(), count, s);
    for (int32_t i = 0; i < count; i++) {
        int64_t v = all[i];
        for (int32_t r = 1; r < c->world; r++) {
            const int64_t x = all[(size_t)r * count + i];
            v = op == COMM_SUM ? v + x : (op == COMM_MIN ? std::min(v, x) : std::max(v, x));
        }
        values[i] = v;
    }
}

void comm_all_reduce_sum_u64(pa_comm* c, uint64_t* dev_words, int64_t words, hipStream_t s)
{
    if (c->world == 1 || words <= 0) return;
    std::lock_guard<std::mutex> lock(c->mu);
    c->collectives++;
    if (c->host) {
        // host transport: gather every rank's words through the byte all-to-all (each rank sends its whole array to every peer)
        const size_t bytes = (size_t)words * 8;
        PinnedBuf mine, all;
        char* hm = static_cast<char*>(mine.ensure(bytes));
        char* ha = static_cast<char*>(all.ensure(bytes * (size_t)c->world));
        PA_HIP(hipMemcpyAsync(hm, dev_words, bytes, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        std::vector<int64_t> soff((size_t)c->world, 0), sb((size_t)c->world, (int64_t)bytes), roff((size_t)c->world), rb((size_t)c->world, (int64_t)bytes);
        for (int32_t r = 0; r < c->world; r++) roff[r] = (int64_t)bytes * r;
        host_status(c->transport.all_to_all_v(c->transport.ctx, hm, soff.data(), sb.data(), ha, roff.data(), rb.data()), "all_to_all_v");
        uint64_t* acc = reinterpret_cast<uint64_t*>(hm);
        memset(acc, 0, bytes);
        for (int32_t r = 0; r < c->world; r++) {
            const uint64_t* w = reinterpret_cast<const uint64_t*>(ha + bytes * (size_t)r);
            for (int64_t i = 0; i < words; i++) acc[i] += w[i];
        }
        PA_HIP(hipMemcpyAsync(dev_words, hm, bytes, hipMemcpyHostToDevice, s));
        PA_HIP(hipStreamSynchronize(s));
        return;
    }
    PA_NCCL(ncclAllReduce(dev_words, dev_words, (size_t)words, ncclUint64, ncclSum, static_cast<ncclComm_t>(c->nccl), s));
    c->payload_bytes_remote += (int64_t)words * 8 * 2 * (c->world - 1) / c->world;
}

void comm_all_to_all_v(pa_comm* c, const void* send, const int64_t* send_off, const int64_t* send_bytes, void* recv, const int64_t* recv_off,
                       const int64_t* recv_bytes, hipStream_t s)
{
    std::lock_guard<std::mutex> lock(c->mu);
    c->collectives++;
    const int W = c->world;
    for (int p = 0; p < W; p++) {
        if (p != c->rank) c->payload_bytes_remote += send_bytes[p];
    }
    if (c->host) {
        int64_t st = 0, rt = 0;
        for (int p = 0; p < W; p++) {
            st = std::max(st, send_off[p] + send_bytes[p]);
            rt = std::max(rt, recv_off[p] + recv_bytes[p]);
        }
        PinnedBuf hs, hr;
        char* a = static_cast<char*>(hs.ensure((size_t)std::max<int64_t>(st, 1)));
        char* b = static_cast<char*>(hr.ensure((size_t)std::max<int64_t>(rt, 1)));
        if (st > 0) PA_HIP(hipMemcpyAsync(a, send, (size_t)st, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        host_status(c->transport.all_to_all_v(c->transport.ctx, a, send_off, send_bytes, b, recv_off, recv_bytes), "all_to_all_v");
        if (rt > 0) PA_HIP(hipMemcpyAsync(recv, b, (size_t)rt, hipMemcpyHostToDevice, s));
        PA_HIP(hipStreamSynchronize(s));
        return;
    }
    ncclComm_t comm = static_cast<ncclComm_t>(c->nccl);
    // one send and one receive per peer in one group: the pattern of an all-to-all over point-to-point xGMI links
    // (1/W of the rows stays local -- RCCL turns the self pair into a device copy -- each of the W - 1 links carries 1/W)
    PA_NCCL(ncclGroupStart());
    for (int i = 0; i < W; i++) {
        // start with the next rank so that the ranks do not all address peer 0 first
        const int p = (c->rank + 1 + i) % W;
        if (send_bytes[p] > 0) {
            PA_NCCL(ncclSend(static_cast<const char*>(send) + send_off[p], (size_t)send_bytes[p], ncclInt8, p, comm, s));
        }
        if (recv_bytes[p] > 0) {
            PA_NCCL(ncclRecv(static_cast<char*>(recv) + recv_off[p], (size_t)recv_bytes[p], ncclInt8, p, comm, s));
        }
    }
    PA_NCCL(ncclGroupEnd());
}

// Pre-flight of a communicator: a `bytes_per_peer`-byte all-to-all whose every byte names (source rank, destination rank, position),
// an all-reduce of a device word array and a host-side all-gather, each verified on the receiving side.  A mismatch is
// PA_ERR_DEVICE with the first offending peer in the message; a transport that never answers is the caller's watchdog's to catch.
void comm_preflight(pa_comm* c, int64_t bytes_per_peer, hipStream_t s)
{
    PA_REQUIRE(bytes_per_peer > 0 && bytes_per_peer % 8 == 0, PA_ERR_INVALID_ARGUMENT, "pre-flight payload must be a positive multiple of 8 bytes");
    const int W = c->world;
    const size_t per = (size_t)bytes_per_peer, total = per * (size_t)W;
    auto pattern = [](int src, int dst, size_t i) { return (uint8_t)(src * 131 + dst * 31 + (int)(i * 2654435761u >> 24) + 7); };
    PinnedBuf hs, hr;
    DevBuf ds, dr;
    uint8_t* a = static_cast<uint8_t*>(hs.ensure(total));
    uint8_t* b = static_cast<uint8_t*>(hr.ensure(total));
    for (int p = 0; p < W; p++) {
        for (size_t i = 0; i < per; i++) a[(size_t)p * per + i] = pattern(c->rank, p, i);
    }
    memset(b, 0, total);
    char* dsp = static_cast<char*>(ds.ensure(total));
    char* drp = static_cast<char*>(dr.ensure(total));
    PA_HIP(hipMemcpyAsync(dsp, a, total, hipMemcpyHostToDevice, s));
    PA_HIP(hipMemsetAsync(drp, 0, total, s));
    std::vector<int64_t> off((size_t)W), len((size_t)W, (int64_t)per);
    for (int p = 0; p < W; p++) off[p] = (int64_t)per * p;
    const int64_t before = c->payload_bytes_remote;
    comm_all_to_all_v(c, dsp, off.data(), len.data(), drp, off.data(), len.data(), s);
    PA_HIP(hipMemcpyAsync(b, drp, total, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    c->payload_bytes_remote = before;  // (the statistics are the exchanges')
    for (int p = 0; p < W; p++) {
        for (size_t i = 0; i < per; i++) {
            if (b[(size_t)p * per + i] != pattern(p, c->rank, i)) {
                throw Error(PA_ERR_DEVICE, "communicator pre-flight: all-to-all byte " + std::to_string(i) + " from rank " + std::to_string(p) + " to rank " +
                                               std::to_string(c->rank) + " arrived damaged");
            }
        }
    }
    // all-reduce: word i of rank r = (r + 1) * (i + 1); the sum over the ranks = W (W + 1) / 2 * (i + 1)
    const int64_t words = std::min<int64_t>(bytes_per_peer / 8, 1 << 17);
    uint64_t* w = reinterpret_cast<uint64_t*>(a);
    for (int64_t i = 0; i < words; i++) w[i] = (uint64_t)(c->rank + 1) * (uint64_t)(i + 1);
    PA_HIP(hipMemcpyAsync(dsp, w, (size_t)words * 8, hipMemcpyHostToDevice, s));
    comm_all_reduce_sum_u64(c, reinterpret_cast<uint64_t*>(dsp), words, s);
    PA_HIP(hipMemcpyAsync(b, dsp, (size_t)words * 8, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    c->payload_bytes_remote = before;
    const uint64_t* got = reinterpret_cast<const uint64_t*>(b);
    for (int64_t i = 0; i < words; i++) {
        if (got[i] != (uint64_t)W * (uint64_t)(W + 1) / 2 * (uint64_t)(i + 1)) {
            throw Error(PA_ERR_DEVICE, "communicator pre-flight: all-reduce word " + std::to_string(i) + " is wrong on rank " + std::to_string(c->rank));
        }
    }
    int64_t mine[2] = {c->rank, 0x5EED0000LL + c->rank};
    std::vector<int64_t> all((size_t)2 * W);
    comm_all_gather_i64(c, mine, all.data(), 2, s);
    for (int p = 0; p < W; p++) {
        if (all[(size_t)2 * p] != p || all[(size_t)2 * p + 1] != 0x5EED0000LL + p) {
            throw Error(PA_ERR_DEVICE, "communicator pre-flight: all-gather slot " + std::to_string(p) + " is wrong on rank " + std::to_string(c->rank));
        }
    }
}

}  // namespace pa

pa_comm::~pa_comm()
{
    if (nccl) (void)ncclCommDestroy(static_cast<ncclComm_t>(nccl));
}
