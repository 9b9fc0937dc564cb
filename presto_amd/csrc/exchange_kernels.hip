// exchange_kernels.hip -- see exchange_kernels.hpp.  Pure HBM copies: a partitioned exchange keeps the rows of every
// destination as raw column segments, and packing them into the send blob / unpacking the received blob into the consumer's
// page is a segment-table copy, one launch whatever the number of (rank, column) pairs.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "exchange_kernels.hpp"

namespace pa {

constexpr int kCopyThreads = 256;
constexpr int64_t kCopyChunk = 64 * 1024;  // bytes per workgroup: 16 x (256 lanes x 16 B)

size_t copy_segments_table_bytes(size_t n) { return (n > 0 ? n : 1) * sizeof(CopySeg); }

__device__ __forceinline__ void copy_chunk(const CopySeg* segs, int nseg);

struct InlineSegs {
    CopySeg seg[kInlineSegs];
};
__global__ __launch_bounds__(kCopyThreads) void k_copy_segments_inline(InlineSegs t, int nseg) { copy_chunk(t.seg, nseg); }
__global__ __launch_bounds__(kCopyThreads) void k_copy_segments(const CopySeg* __restrict__ segs, int nseg) { copy_chunk(segs, nseg); }

__device__ __forceinline__ void copy_chunk(const CopySeg* segs, int nseg)
{
    // the segment this workgroup's chunk belongs to: last one whose first_chunk <= blockIdx.x
    const int64_t b = blockIdx.x;
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].first_chunk <= b) lo = mid;
        else hi = mid - 1;
    }
    const CopySeg sg = segs[lo];
    const int64_t off = (b - sg.first_chunk) * kCopyChunk;
    const int64_t len = sg.bytes - off < kCopyChunk ? sg.bytes - off : kCopyChunk;
    char* dst = static_cast<char*>(sg.dst) + off;
    if (sg.src == nullptr) {
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
            const uint4 z = make_uint4(0, 0, 0, 0);
            const int64_t v = len >> 4;
            for (int64_t i = threadIdx.x; i < v; i += kCopyThreads) reinterpret_cast<uint4*>(dst)[i] = z;
            for (int64_t i = (v << 4) + threadIdx.x; i < len; i += kCopyThreads) dst[i] = 0;
        }
        else {
            for (int64_t i = threadIdx.x; i < len; i += kCopyThreads) dst[i] = 0;
        }
        return;
    }
    const char* src = static_cast<const char*>(sg.src) + off;
    if (sg.add_i32 != 0) {
        const int64_t v = len >> 2;
        for (int64_t i = threadIdx.x; i < v; i += kCopyThreads) reinterpret_cast<int32_t*>(dst)[i] = reinterpret_cast<const int32_t*>(src)[i] + sg.add_i32;
        return;
    }
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
        const int64_t v = len >> 4;
        for (int64_t i = threadIdx.x; i < v; i += kCopyThreads) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (int64_t i = (v << 4) + threadIdx.x; i < len; i += kCopyThreads) dst[i] = src[i];
    }
    else if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 3) == 0) {
        const int64_t v = len >> 2;
        for (int64_t i = threadIdx.x; i < v; i += kCopyThreads) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
        for (int64_t i = (v << 2) + threadIdx.x; i < len; i += kCopyThreads) dst[i] = src[i];
    }
    else {
        for (int64_t i = threadIdx.x; i < len; i += kCopyThreads) dst[i] = src[i];
    }
}

void launch_copy_segments_inline(CopySeg* segs, int n, hipStream_t s)
{
    PA_REQUIRE(n <= kInlineSegs, PA_ERR_NOT_SUPPORTED, "too many segments for an inline copy");
    InlineSegs t;
    int m = 0;
    int64_t chunks = 0;
    for (int i = 0; i < n; i++) {
        if (segs[i].bytes <= 0) continue;
        t.seg[m] = segs[i];
        t.seg[m].first_chunk = chunks;
        chunks += (segs[i].bytes + kCopyChunk - 1) / kCopyChunk;
        m++;
    }
    if (m == 0) return;
    hipLaunchKernelGGL(k_copy_segments_inline, (int)chunks, kCopyThreads, 0, s, t, m);
    PA_HIP(hipGetLastError());
}

void launch_copy_segments(CopySeg* segs, size_t n, void* host_table, void* dev_table, hipStream_t s)
{
    // drop empty segments, number the chunks
    size_t m = 0;
    int64_t chunks = 0;
    for (size_t i = 0; i < n; i++) {
        if (segs[i].bytes <= 0) continue;
        segs[m] = segs[i];
        segs[m].first_chunk = chunks;
        chunks += (segs[m].bytes + kCopyChunk - 1) / kCopyChunk;
        m++;
    }
    if (m == 0) return;
    PA_REQUIRE(chunks < (int64_t)1 << 31, PA_ERR_INSUFFICIENT_RESOURCES, "exchange copy larger than 128 TB");
    memcpy(host_table, segs, m * sizeof(CopySeg));
    PA_HIP(hipMemcpyAsync(dev_table, host_table, m * sizeof(CopySeg), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_copy_segments, (int)chunks, kCopyThreads, 0, s, static_cast<const CopySeg*>(dev_table), (int)m);
    PA_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_or_words(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, int64_t words, int reps)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) {
        uint64_t v = dst[i];
        for (int r = 0; r < reps; r++) v |= src[(int64_t)r * words + i];
        dst[i] = v;
    }
}

void launch_or_words(uint64_t* dst, const uint64_t* src, int64_t words, int32_t reps, hipStream_t s)
{
    if (words <= 0 || reps <= 0) return;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((words + 255) / 256, 4096));
    hipLaunchKernelGGL(k_or_words, grid, 256, 0, s, dst, src, words, (int)reps);
    PA_HIP(hipGetLastError());
}

}  // namespace pa
