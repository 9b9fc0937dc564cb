// comm.hpp -- the communicator of the partitioned exchange: one rank per GPU, collectives over RCCL (xGMI inside a node),
// or handed to the host through callbacks (pa_host_transport: ranks that share one GPU in the tests, or a host that
// routes bytes itself).
#pragma once

#include <mutex>
#include <vector>

#include "common.hpp"

struct pa_comm {
    int32_t rank = 0, world = 1;
    int device = 0;
    void* nccl = nullptr;            // ncclComm_t when the transport is RCCL
    bool host = false;
    pa_host_transport transport{};   // host == true
    std::mutex mu;                   // one collective at a time per communicator (RCCL calls on one comm must not interleave)
    pa::DevBuf dev_scratch;          // count rows, small reductions
    pa::PinnedBuf host_scratch;
    // statistics over the communicator's life
    int64_t payload_bytes_remote = 0, collectives = 0;
    ~pa_comm();
};

namespace pa {

enum CommOp { COMM_SUM = 0, COMM_MIN = 1, COMM_MAX = 2 };

void comm_unique_id(void* out128);
pa_comm* comm_create_rccl(const void* unique_id, int32_t rank, int32_t world);
pa_comm* comm_create_host(const pa_host_transport* t, int32_t rank, int32_t world);

// every rank contributes `count` int64 (host memory) and receives world * count in rank order; blocking
void comm_all_gather_i64(pa_comm* c, const int64_t* send, int64_t* recv, int32_t count, hipStream_t s);
// in-place reduction of `count` int64 in host memory; blocking
void comm_all_reduce_i64(pa_comm* c, int64_t* values, int32_t count, int op, hipStream_t s);
// in-place SUM of `words` uint64 in device memory, enqueued on s (bitmaps whose set bits are disjoint between the ranks:
// SUM == OR; RCCL has no bitwise reduction)
void comm_all_reduce_sum_u64(pa_comm* c, uint64_t* dev_words, int64_t words, hipStream_t s);
// variable all-to-all of bytes between device buffers: rank p receives send[send_off[p] .. + send_bytes[p]); one grouped
// ncclSend / ncclRecv per peer.  Enqueued on s (RCCL) or staged through pinned host memory and done on return (host transport).
void comm_all_to_all_v(pa_comm* c, const void* send, const int64_t* send_off, const int64_t* send_bytes, void* recv, const int64_t* recv_off,
                       const int64_t* recv_bytes, hipStream_t s);

// checked all-to-all + all-reduce + all-gather over the communicator (collective; see comm.cpp)
void comm_preflight(pa_comm* c, int64_t bytes_per_peer, hipStream_t s);

}  // namespace pa
