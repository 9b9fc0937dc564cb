// pool.cpp -- size-class caches for HBM, pinned host memory and streams.  hipMalloc / hipHostMalloc /
// hipStreamCreate cost 0.1-several ms each; an operator instance lives for one query, so without reuse
// the allocator would dominate short queries (a fresh operator per query is the reference's model too:
// OperatorFactory.createOperator, …/operator/OperatorFactory.java:18-50).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace pa {
namespace {

// A cached HBM block and the stream that may still be working on it: blocks are released in the middle of a pipeline (a
// staging arena that grows, a table that is rehashed) while kernels enqueued earlier still read them.  A request made on
// the same stream may take the block at once (stream order), any other stream only once that stream has drained.
struct DeviceBlock {
    void* ptr;
    hipStream_t last;  // nullptr: released by code that had synchronised (or outside any operator call)
};

struct Pools {
    std::mutex mu;
    // key: (device, size class)
    std::multimap<std::pair<int, size_t>, DeviceBlock> device_free;
    std::multimap<size_t, void*> pinned_free;
    std::multimap<int, hipStream_t> stream_free;
    size_t device_cached = 0, pinned_cached = 0;
    size_t device_in_use = 0;   // HBM handed out and not returned yet
    size_t device_limit = 0;    // 0 = none: budget for device_in_use (pa_memory_set_limit)
};
Pools& pools()
{
    static Pools* p = new Pools();  // leaked on purpose: HIP may already be torn down at static destruction
    return *p;
}

// powers of two below 1 MiB; above, four steps per octave (<= 25 % slack on the multi-GB page buffers)
size_t size_class(size_t bytes)
{
    size_t c = 256;
    while (c < bytes) c <<= 1;
    if (c <= (1ULL << 20)) return c;
    const size_t step = c >> 3;  // c/2 < bytes <= c: classes c/2 + k * c/8
    return (bytes + step - 1) / step * step;
}

// Freed blocks kept for the next operator.  A Q3 pipeline over 2^28-row pages holds tens of GB of transient page
// buffers, and re-creating them with hipMalloc costs hundreds of ms; the card has 288 GB.
size_t max_cached_device()
{
    static const size_t v = [] {
        const char* e = getenv("PRESTO_AMD_POOL_BYTES");
        return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)(96ULL << 30);
    }();
    return v;
}
constexpr size_t kMaxCachedPinned = 1ULL << 30;

thread_local hipStream_t t_scope_stream = nullptr;

}  // namespace

void pool_set_limit(int64_t bytes)
{
    Pools& p = pools();
    std::lock_guard<std::mutex> lock(p.mu);
    p.device_limit = bytes > 0 ? (size_t)bytes : 0;
}

bool pool_has_room(size_t bytes)
{
    Pools& p = pools();
    std::lock_guard<std::mutex> lock(p.mu);
    return p.device_limit == 0 || p.device_in_use + size_class(bytes) <= p.device_limit;
}

void pool_stats(int64_t* in_use, int64_t* cached, int64_t* limit)
{
    Pools& p = pools();
    std::lock_guard<std::mutex> lock(p.mu);
    if (in_use) *in_use = (int64_t)p.device_in_use;
    if (cached) *cached = (int64_t)p.device_cached;
    if (limit) *limit = (int64_t)p.device_limit;
}

hipStream_t pool_scope_stream(hipStream_t s)
{
    hipStream_t prev = t_scope_stream;
    t_scope_stream = s;
    return prev;
}

void pool_forget_stream(hipStream_t s)
{
    if (!s) return;
    Pools& p = pools();
    std::lock_guard<std::mutex> lock(p.mu);
    for (auto& kv : p.device_free) {
        if (kv.second.last == s) kv.second.last = nullptr;
    }
}

void* pool_device_alloc(size_t bytes, size_t* granted)
{
    size_t c = size_class(bytes);
    int dev = 0;
    PA_HIP(hipGetDevice(&dev));
    {
        Pools& p = pools();
        std::lock_guard<std::mutex> lock(p.mu);
        if (p.device_limit != 0 && p.device_in_use + c > p.device_limit) throw PoolExhausted(c);
        p.device_in_use += c;  // (given back below if the driver itself has nothing left)
        auto range = p.device_free.equal_range({dev, c});
        for (auto it = range.first; it != range.second; ++it) {
            hipStream_t last = it->second.last;
            // hipStreamQuery: hipErrorNotReady while work is pending; anything else (idle, or a stream its owner destroyed,
            // which waits for its work) means the block is quiescent
            if (last != nullptr && last != t_scope_stream && hipStreamQuery(last) == hipErrorNotReady) continue;
            void* ptr = it->second.ptr;
            p.device_free.erase(it);
            p.device_cached -= c;
            *granted = c;
            // PRESTO_AMD_POOL_SCRUB=<byte>: a recycled block is overwritten before it is handed out -- shows code that relies on
            // what a previous owner left behind (debugging aid; the scrub waits for the device)
            static const int scrub = [] {
                const char* e = getenv("PRESTO_AMD_POOL_SCRUB");
                return e ? (int)strtol(e, nullptr, 0) & 0xff : -1;
            }();
            if (scrub >= 0) {
                (void)hipDeviceSynchronize();
                (void)hipMemset(ptr, scrub, c);
                (void)hipDeviceSynchronize();
            }
            return ptr;
        }
        (void)hipGetLastError();
    }
    void* ptr = nullptr;
    hipError_t e = hipMalloc(&ptr, c);
    if (e == hipErrorOutOfMemory) {
        // give the cached blocks back to the driver and try once more
        (void)hipGetLastError();
        std::vector<void*> cached;
        {
            Pools& p = pools();
            std::lock_guard<std::mutex> lock(p.mu);
            for (auto it = p.device_free.begin(); it != p.device_free.end();) {
                if (it->first.first == dev) {
                    cached.push_back(it->second.ptr);
                    p.device_cached -= it->first.second;
                    it = p.device_free.erase(it);
                }
                else ++it;
            }
        }
        PA_HIP(hipDeviceSynchronize());
        for (void* q : cached) (void)hipFree(q);
        e = hipMalloc(&ptr, c);
    }
    if (e != hipSuccess) {
        Pools& p = pools();
        std::lock_guard<std::mutex> lock(p.mu);
        p.device_in_use -= c;
    }
    PA_HIP(e);
    *granted = c;
    return ptr;
}

void pool_device_free(void* ptr, size_t granted)
{
    if (!ptr) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    Pools& p = pools();
    {
        std::lock_guard<std::mutex> lock(p.mu);
        p.device_in_use -= std::min(p.device_in_use, granted);
        if (p.device_cached + granted <= max_cached_device()) {
            p.device_free.emplace(std::make_pair(dev, granted), DeviceBlock{ptr, t_scope_stream});
            p.device_cached += granted;
            return;
        }
    }
    (void)hipFree(ptr);
}

void* pool_pinned_alloc(size_t bytes, size_t* granted)
{
    size_t c = size_class(bytes);
    {
        Pools& p = pools();
        std::lock_guard<std::mutex> lock(p.mu);
        auto it = p.pinned_free.find(c);
        if (it != p.pinned_free.end()) {
            void* ptr = it->second;
            p.pinned_free.erase(it);
            p.pinned_cached -= c;
            *granted = c;
            return ptr;
        }
    }
    void* ptr = nullptr;
    PA_HIP(hipHostMalloc(&ptr, c, hipHostMallocDefault));
    *granted = c;
    return ptr;
}

void pool_pinned_free(void* ptr, size_t granted)
{
    if (!ptr) return;
    Pools& p = pools();
    {
        std::lock_guard<std::mutex> lock(p.mu);
        if (p.pinned_cached + granted <= kMaxCachedPinned) {
            p.pinned_free.emplace(granted, ptr);
            p.pinned_cached += granted;
            return;
        }
    }
    (void)hipHostFree(ptr);
}

hipStream_t pool_stream_acquire()
{
    int dev = 0;
    PA_HIP(hipGetDevice(&dev));
    {
        Pools& p = pools();
        std::lock_guard<std::mutex> lock(p.mu);
        auto it = p.stream_free.find(dev);
        if (it != p.stream_free.end()) {
            hipStream_t s = it->second;
            p.stream_free.erase(it);
            return s;
        }
    }
    hipStream_t s = nullptr;
    PA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}

void pool_stream_release(hipStream_t s)
{
    if (!s) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    // everything enqueued by the previous owner must be done before its buffers are recycled
    (void)hipStreamSynchronize(s);
    pool_forget_stream(s);
    Pools& p = pools();
    std::lock_guard<std::mutex> lock(p.mu);
    p.stream_free.emplace(dev, s);
}

}  // namespace pa
