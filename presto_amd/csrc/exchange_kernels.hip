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

// len bytes by one workgroup, with the widest accesses the two sides can be aligned for together
__device__ __forceinline__ void copy_bytes(char* dst, const char* src, int64_t len)
{
    const uintptr_t d = reinterpret_cast<uintptr_t>(dst), x = d ^ reinterpret_cast<uintptr_t>(src);
    if ((x & 3) != 0) {
        for (int64_t i = threadIdx.x; i < len; i += kCopyThreads) dst[i] = src[i];
        return;
    }
    const int w = (x & 15) == 0 ? 16 : 4;
    int64_t head = (int64_t)((w - (d & (w - 1))) & (w - 1));
    if (head > len) head = len;
    for (int64_t i = threadIdx.x; i < head; i += kCopyThreads) dst[i] = src[i];
    dst += head;
    src += head;
    len -= head;
    if (w == 16) {
        const int64_t v = len >> 4;
        for (int64_t i = threadIdx.x; i < v; i += kCopyThreads) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (int64_t i = (v << 4) + threadIdx.x; i < len; i += kCopyThreads) dst[i] = src[i];
    }
    else {
        const int64_t v = len >> 2;
        for (int64_t i = threadIdx.x; i < v; i += kCopyThreads) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
        for (int64_t i = (v << 2) + threadIdx.x; i < len; i += kCopyThreads) dst[i] = src[i];
    }
}

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
    copy_bytes(dst, src, len);
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

// ---- VariableWidthBlocks of device pages behind an arena's bytes (see VarSeg) ----
constexpr int64_t kVarOffsetsPerWg = kCopyChunk / 4;

size_t copy_var_table_bytes(size_t n) { return (n > 0 ? n : 1) * sizeof(VarSeg); }

// workgroup `local` of a segment whose start / first / len are known: the first ones rebase the offsets, the others copy bytes
__device__ __forceinline__ void var_copy_part(const VarSeg& sg, int64_t local, int64_t start, int32_t first, int32_t len)
{
    const int64_t off_wgs = ((int64_t)sg.rows + 1 + kVarOffsetsPerWg - 1) / kVarOffsetsPerWg;
    if (local < off_wgs) {
        const int64_t k0 = local * kVarOffsetsPerWg;
        const int64_t k1 = k0 + kVarOffsetsPerWg < (int64_t)sg.rows + 1 ? k0 + kVarOffsetsPerWg : (int64_t)sg.rows + 1;
        const int32_t delta = (int32_t)start - first;
        for (int64_t k = k0 + threadIdx.x; k < k1; k += kCopyThreads) sg.dst_offsets[k] = sg.offsets[k] + delta;
        return;
    }
    for (int64_t chunk = local - off_wgs; chunk * kCopyChunk < len; chunk += sg.byte_wgs) {
        const int64_t off = chunk * kCopyChunk;
        const int64_t part = len - off < kCopyChunk ? len - off : kCopyChunk;
        copy_bytes(sg.dst_bytes + start + off, sg.values + first + off, part);
    }
}

template <typename Table> __device__ __forceinline__ int var_find(const Table& segs, int nseg)
{
    const int64_t b = blockIdx.x;
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].first_wg <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

struct InlineVarSegs {
    VarSeg seg[kInlineVarSegs];
};
// one page: every workgroup reads the cursor of its segment itself; the segment's first workgroup moves it on (other slot)
__global__ __launch_bounds__(kCopyThreads) void k_var_append_inline(InlineVarSegs t, int nseg, int32_t* err)
{
    const int i = var_find(t.seg, nseg);
    const VarSeg& sg = t.seg[i];
    const int64_t local = (int64_t)blockIdx.x - sg.first_wg;
    const int64_t start = sg.fresh ? 0 : *sg.cursor_in;
    const int32_t first = sg.offsets[0];
    const int64_t len = (int64_t)sg.offsets[sg.rows] - first;
    const bool bad = start < 0 || len < 0 || start + len > sg.capacity;
    if (local == 0 && threadIdx.x == 0) {
        *sg.cursor_out = bad ? -1 : start + len;
        if (bad && *err == 0) atomicCAS(err, 0, (int32_t)PA_ERR_INVALID_ARGUMENT);
    }
    if (bad) return;
    var_copy_part(sg, local, start, first, (int32_t)len);
}

struct VarSlots {
    const int64_t* in[kVarSlots];
    int64_t* out[kVarSlots];
    int32_t fresh[kVarSlots];
};
// many pages: start / first / len of every segment, by an exclusive prefix sum over the segments of each slot in table order
__global__ __launch_bounds__(kCopyThreads) void k_var_plan(VarSeg* __restrict__ segs, int nseg, VarSlots slots, int nslots, int32_t* err)
{
    __shared__ int64_t scan[kCopyThreads];
    __shared__ int64_t running[kVarSlots];
    __shared__ int failed;
    const int t = threadIdx.x;
    if (t < kVarSlots) running[t] = t < nslots && !slots.fresh[t] ? *slots.in[t] : 0;
    if (t == 0) failed = 0;
    __syncthreads();
    for (int base = 0; base < nseg; base += kCopyThreads) {
        const int i = base + t;
        int32_t first = 0, slot = -1;
        int64_t len = 0;
        if (i < nseg) {
            first = segs[i].offsets[0];
            len = (int64_t)segs[i].offsets[segs[i].rows] - first;
            slot = segs[i].slot;
            if (len < 0) { failed = 1; len = 0; }
        }
        for (int sl = 0; sl < nslots; sl++) {
            scan[t] = slot == sl ? len : 0;
            __syncthreads();
            for (int d = 1; d < kCopyThreads; d <<= 1) {
                const int64_t add = t >= d ? scan[t - d] : 0;
                __syncthreads();
                scan[t] += add;
                __syncthreads();
            }
            const int64_t incl = scan[t], total = scan[kCopyThreads - 1];
            const int64_t before = running[sl];
            __syncthreads();
            if (slot == sl) {
                const int64_t start = before + incl - len;
                const bool bad = before < 0 || start + len > segs[i].capacity;
                if (bad) failed = 1;
                segs[i].start = bad ? -1 : start;
                segs[i].first = first;
                segs[i].len = (int32_t)len;
            }
            if (t == 0) running[sl] = before + total;
            __syncthreads();
        }
    }
    if (t < nslots) *slots.out[t] = failed ? -1 : running[t];
    if (t == 0 && failed && *err == 0) atomicCAS(err, 0, (int32_t)PA_ERR_INVALID_ARGUMENT);
}

__global__ __launch_bounds__(kCopyThreads) void k_var_copy(const VarSeg* __restrict__ segs, int nseg)
{
    const int i = var_find(segs, nseg);
    const VarSeg sg = segs[i];
    if (sg.start < 0) return;
    var_copy_part(sg, (int64_t)blockIdx.x - sg.first_wg, sg.start, sg.first, sg.len);
}

static int64_t var_number_wgs(VarSeg* segs, size_t n)
{
    int64_t wgs = 0;
    for (size_t i = 0; i < n; i++) {
        PA_REQUIRE(segs[i].rows >= 0 && segs[i].byte_wgs >= 1 && segs[i].slot >= 0 && segs[i].slot < kVarSlots, PA_ERR_INVALID_ARGUMENT,
                   "bad variable-width segment");
        segs[i].first_wg = wgs;
        wgs += ((int64_t)segs[i].rows + 1 + kVarOffsetsPerWg - 1) / kVarOffsetsPerWg + segs[i].byte_wgs;
    }
    PA_REQUIRE(wgs < (int64_t)1 << 31, PA_ERR_INSUFFICIENT_RESOURCES, "variable-width append too large for one launch");
    return wgs;
}

void launch_var_append_inline(VarSeg* segs, int n, int32_t* err, hipStream_t s)
{
    if (n <= 0) return;
    PA_REQUIRE(n <= kInlineVarSegs, PA_ERR_NOT_SUPPORTED, "too many variable-width blocks for an inline append");
    const int64_t wgs = var_number_wgs(segs, (size_t)n);
    InlineVarSegs t;
    for (int i = 0; i < n; i++) t.seg[i] = segs[i];
    hipLaunchKernelGGL(k_var_append_inline, (int)wgs, kCopyThreads, 0, s, t, n, err);
    PA_HIP(hipGetLastError());
}

void launch_var_append(VarSeg* segs, size_t n, void* host_table, void* dev_table, int32_t* err, hipStream_t s)
{
    if (n == 0) return;
    const int64_t wgs = var_number_wgs(segs, n);
    // per slot: the cursor its first segment reads and the one its last segment leaves
    VarSlots slots{};
    int nslots = 0;
    bool seen[kVarSlots] = {};
    for (size_t i = 0; i < n; i++) {
        const int sl = segs[i].slot;
        if (!seen[sl]) {
            seen[sl] = true;
            slots.in[sl] = segs[i].cursor_in;
            slots.fresh[sl] = segs[i].fresh;
        }
        slots.out[sl] = segs[i].cursor_out;
        if (sl + 1 > nslots) nslots = sl + 1;
    }
    for (int sl = 0; sl < nslots; sl++) PA_REQUIRE(seen[sl], PA_ERR_INVALID_ARGUMENT, "variable-width slots must be numbered densely");
    memcpy(host_table, segs, n * sizeof(VarSeg));
    PA_HIP(hipMemcpyAsync(dev_table, host_table, n * sizeof(VarSeg), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_var_plan, 1, kCopyThreads, 0, s, static_cast<VarSeg*>(dev_table), (int)n, slots, nslots, err);
    PA_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_var_copy, (int)wgs, kCopyThreads, 0, s, static_cast<const VarSeg*>(dev_table), (int)n);
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
