// common.hpp -- host-side plumbing of libpresto_amd.so: error model, HIP RAII helpers, streams.
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/presto_amd.h"

namespace pa {

// Internal error; converted into (status, thread-local message) at the C ABI, never crosses it.
struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& msg);

#define PA_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            throw ::pa::Error(PA_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));       \
        }                                                                                              \
    } while (0)

#define PA_REQUIRE(cond, code, msg)                   \
    do {                                              \
        if (!(cond)) throw ::pa::Error((code), (msg)); \
    } while (0)

inline int type_width(int32_t t)
{
    switch (t) {
        case PA_LONG_DECIMAL:
            return 16;
        case PA_BIGINT:
        case PA_DOUBLE:
        case PA_DECIMAL:
            return 8;
        case PA_INTEGER:
        case PA_DATE:
        case PA_REAL:
            return 4;
        case PA_BOOLEAN:
            return 1;
        default:
            return 0;
    }
}

// Thrown by pool_device_alloc when the HBM budget (pa_memory_set_limit) does not cover the request: the counterpart of a memory
// reservation the reference's memory pool cannot grant right now (OperatorContext.isWaitingForMemory, Operator.java:69-80) --
// an operator that can put its page aside reports is_blocked until other operators have released memory.
struct PoolExhausted : Error {
    size_t bytes;
    explicit PoolExhausted(size_t b) : Error(PA_ERR_INSUFFICIENT_RESOURCES, "HBM budget of the process exhausted (pa_memory_set_limit)"), bytes(b) {}
};
void pool_set_limit(int64_t bytes);              // 0 = unlimited
bool pool_has_room(size_t bytes);                 // would a request of this size be granted now?
void pool_stats(int64_t* in_use, int64_t* cached, int64_t* limit);

// size-class caches (pool.cpp)
void* pool_device_alloc(size_t bytes, size_t* granted);
void pool_device_free(void* ptr, size_t granted);
void* pool_pinned_alloc(size_t bytes, size_t* granted);
void pool_pinned_free(void* ptr, size_t granted);
// Stream of the operator call the thread is in (set by the C-ABI wrappers): blocks released meanwhile are tagged with it and
// handed to another stream only once it has drained; returns the previous value.  pool_forget_stream: the stream is about to
// be destroyed (the caller has synchronised it).
hipStream_t pool_scope_stream(hipStream_t s);
void pool_forget_stream(hipStream_t s);
hipStream_t pool_stream_acquire();
void pool_stream_release(hipStream_t s);

// Growable HBM allocation (never shrinks; reused across pages).
class DevBuf {
public:
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p_(o.p_), cap_(o.cap_), borrowed_(o.borrowed_) { o.p_ = nullptr; o.cap_ = 0; o.borrowed_ = false; }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; cap_ = o.cap_; borrowed_ = o.borrowed_; o.p_ = nullptr; o.cap_ = 0; o.borrowed_ = false; }
        return *this;
    }
    ~DevBuf() { release(); }
    // `bytes` of somebody else's HBM, read in place (a retained page's block array): never freed here; growing past it moves the
    // contents into an allocation of the buffer's own (reserve_keep)
    void borrow(const void* p, size_t bytes)
    {
        release();
        p_ = const_cast<void*>(p);
        cap_ = bytes;
        borrowed_ = true;
    }
    bool borrowed() const { return borrowed_; }
    void* ensure(size_t bytes)
    {
        if (bytes > cap_) {
            release();
            p_ = pool_device_alloc(bytes, &cap_);
        }
        return p_;
    }
    // grows keeping the first `used` bytes; the old allocation is recycled only after the copy completed
    void* reserve_keep(size_t bytes, size_t used, hipStream_t s)
    {
        if (bytes <= cap_) return p_;
        size_t ncap = 0;
        size_t want = bytes > 2 * cap_ ? bytes : 2 * cap_;
        void* np = pool_device_alloc(want, &ncap);
        if (p_ && used) {
            PA_HIP(hipMemcpyAsync(np, p_, used, hipMemcpyDeviceToDevice, s));
            PA_HIP(hipStreamSynchronize(s));
        }
        release();
        p_ = np;
        cap_ = ncap;
        return p_;
    }
    void release()
    {
        if (p_ && !borrowed_) pool_device_free(p_, cap_);
        p_ = nullptr;
        cap_ = 0;
        borrowed_ = false;
    }
    void* ptr() const { return p_; }
    template <typename T> T* as() const { return static_cast<T*>(p_); }
    size_t capacity() const { return cap_; }

private:
    void* p_ = nullptr;
    size_t cap_ = 0;
    bool borrowed_ = false;
};

// Growable pinned host allocation (D2H landing zone / H2D staging).
class PinnedBuf {
public:
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    PinnedBuf(PinnedBuf&& o) noexcept : p_(o.p_), cap_(o.cap_) { o.p_ = nullptr; o.cap_ = 0; }
    PinnedBuf& operator=(PinnedBuf&& o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; cap_ = o.cap_; o.p_ = nullptr; o.cap_ = 0; }
        return *this;
    }
    ~PinnedBuf() { release(); }
    void* ensure(size_t bytes)
    {
        if (bytes > cap_) {
            release();
            p_ = pool_pinned_alloc(bytes, &cap_);
        }
        return p_;
    }
    void release()
    {
        if (p_) pool_pinned_free(p_, cap_);
        p_ = nullptr;
        cap_ = 0;
    }
    void* ptr() const { return p_; }
    template <typename T> T* as() const { return static_cast<T*>(p_); }
    size_t capacity() const { return cap_; }

private:
    void* p_ = nullptr;
    size_t cap_ = 0;
};

// A few words from the device for a host decision: through a pinned landing zone of the calling thread, then the wait.  (Into pageable
// memory -- a variable on the stack -- the runtime stages the copy itself and blocks: 22 us per round trip against 16,
// scripts/micro/readback.cpp.)
inline void read_back(void* dst, const void* src_dev, size_t bytes, hipStream_t s)
{
    static thread_local PinnedBuf* land = new PinnedBuf();   // (leaked: the pool may be gone when a thread ends)
    void* p = land->ensure(bytes < 256 ? 256 : bytes);
    PA_HIP(hipMemcpyAsync(p, src_dev, bytes, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    memcpy(dst, p, bytes);
}

int device_cu_count();
void require_device();

// A stream either borrowed from the host (desc.stream) or owned by the operator.
class Stream {
public:
    explicit Stream(void* borrowed)
    {
        require_device();  // operators construct their stream first: no device => PA_ERR_NO_DEVICE, loudly
        if (borrowed) {
            s_ = static_cast<hipStream_t>(borrowed);
            owned_ = false;
        }
        else {
            s_ = pool_stream_acquire();
            owned_ = true;
        }
    }
    Stream(const Stream&) = delete;
    Stream& operator=(const Stream&) = delete;
    ~Stream()
    {
        if (owned_ && s_) pool_stream_release(s_);
    }
    hipStream_t get() const { return s_; }
    bool owned() const { return owned_; }
    void sync() const { PA_HIP(hipStreamSynchronize(s_)); }

private:
    hipStream_t s_ = nullptr;
    bool owned_ = false;
};

// HIP-event stopwatch around the launches of an operator's dominant kernel (pa_op_kernel_time).
class KernelTimer {
public:
    ~KernelTimer()
    {
        for (auto& p : pairs_) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    }
    void begin(hipStream_t s)
    {
        if (next_ == pairs_.size()) {
            if (pairs_.size() >= 4096) { drain(); }
            else {
                hipEvent_t a, b;
                PA_HIP(hipEventCreate(&a));
                PA_HIP(hipEventCreate(&b));
                pairs_.emplace_back(a, b);
            }
        }
        PA_HIP(hipEventRecord(pairs_[next_].first, s));
        begun_++;
    }
    uint64_t begun() const { return begun_; }  // timed launches started so far
    // dominant = false: the time still counts, the launch does not (e.g. the few-row tail launch of a page)
    void end(hipStream_t s, bool dominant = true)
    {
        PA_HIP(hipEventRecord(pairs_[next_].second, s));
        if (minor_.size() <= next_) minor_.resize(next_ + 1);
        minor_[next_] = !dominant;
        next_++;
    }
    void drain()
    {
        for (size_t i = 0; i < next_; i++) {
            PA_HIP(hipEventSynchronize(pairs_[i].second));
            float ms = 0;
            PA_HIP(hipEventElapsedTime(&ms, pairs_[i].first, pairs_[i].second));
            total_ms_ += ms;
            if (!minor_[i]) launches_++;
        }
        next_ = 0;
    }
    double total_ms() const { return total_ms_; }
    int64_t launches() const { return launches_; }
    // the dominant kernel's name as a kernel trace shows it (pa_op_kernel_name); set by the operator at its launch
    void set_name(const std::string& n) { if (name_ != n) name_ = n; }
    const std::string& name() const { return name_; }

private:
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs_;
    std::vector<char> minor_;
    size_t next_ = 0;
    uint64_t begun_ = 0;
    double total_ms_ = 0;
    int64_t launches_ = 0;
    std::string name_;
};

int device_cu_count();
void require_device();

// a phase of host work in the PRESTO_AMD_HOST_TRACE listing (abi.cpp); costs one branch when the trace is off
struct HostTraceScope {
    const char* name;
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit HostTraceScope(const char* n);
    ~HostTraceScope();
};

}  // namespace pa
