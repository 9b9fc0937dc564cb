// join_kernels.hip -- hash join build and probe on device.
//
// Layout follows the reference's PagesHash: key[hashSize] (int32, -1 = empty) holds a *build position*,
// positionLinks[n] chains equal keys, the head of a chain is the LAST inserted (= highest) position and
// the chain descends, so a probe row emits its matches in descending build position
// (…/operator/join/PagesHash.java:77-120, …/ArrayPositionLinks.java:45-50).
// Build pages are concatenated into flat columns, so address == position.
// The probe is random access into key[] and the build key columns: transaction-bound, not streaming.
#include <hip/hip_runtime.h>

#include "join_kernels.hpp"
#include "scan_kernels.hpp"
#include "kernels/pa_device.h"

namespace pa {

static inline int grid_for(int64_t work)
{
    int64_t g = (work + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

__device__ __forceinline__ bool jcol_is_null(const JoinCol& c, i32 r) { return c.nulls && c.nulls[r]; }

// positionEqualsRowIgnoreNulls: plain EQUAL on non-null values (DOUBLE: ==, so NaN never matches)
__device__ __forceinline__ bool jcol_equal(const JoinCol& a, i32 ra, const JoinCol& b, i32 rb)
{
    switch (a.type) {
        case PA_BIGINT: return ((const i64*)a.values)[ra] == ((const i64*)b.values)[rb];
        case PA_INTEGER:
        case PA_DATE: return ((const i32*)a.values)[ra] == ((const i32*)b.values)[rb];
        case PA_DOUBLE: return ((const double*)a.values)[ra] == ((const double*)b.values)[rb];
        case PA_REAL: return ((const float*)a.values)[ra] == ((const float*)b.values)[rb];  // RealType.equalOperator: NaN matches nothing
        case PA_BOOLEAN: return (((const u8*)a.values)[ra] != 0) == (((const u8*)b.values)[rb] != 0);
        case PA_VARCHAR: {
            i32 oa = a.offsets[ra], ob = b.offsets[rb];
            return pa_str_eq((const u8*)a.values + oa, a.offsets[ra + 1] - oa, (const u8*)b.values + ob, b.offsets[rb + 1] - ob);
        }
        default: return false;
    }
}
__device__ __forceinline__ bool keys_equal(const JoinKeys& a, i32 ra, const JoinKeys& b, i32 rb)
{
    for (int c = 0; c < a.ncols; c++) {
        if (!jcol_equal(a.col[c], ra, b.col[c], rb)) return false;
    }
    return true;
}
__device__ __forceinline__ bool keys_have_null(const JoinKeys& k, i32 r)
{
    for (int c = 0; c < k.ncols; c++) {
        if (jcol_is_null(k.col[c], r)) return true;
    }
    return false;
}

// Phase A: every non-null build position finds / claims the slot of its key; the slot ends up holding the
// highest position of the key (atomicMax), exactly the head the reference's sequential insertion leaves.
__global__ __launch_bounds__(256) void k_join_build_slots(JoinKeys build, const i64* __restrict__ raw_hash, i32 n, i32* key, u32 mask,
                                                          i32* __restrict__ slot_of, i32* err)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const i32 p = (i32)i;
        if (keys_have_null(build, p)) {  // PagesHash.java:95-97: rows with a NULL key are not inserted
            slot_of[p] = -1;
            continue;
        }
        u32 pos = (u32)pa_murmur3_fmix((u64)raw_hash[p]) & mask;
        u32 probes = 0;
        for (;;) {
            i32 cur = __hip_atomic_load(&key[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == -1) {
                i32 old = atomicCAS(&key[pos], -1, p);
                if (old == -1) break;
                cur = old;
            }
            // build columns are immutable: any position ever stored in this slot carries the slot's key
            if (raw_hash[cur] == raw_hash[p] && keys_equal(build, cur, build, p)) {
                atomicMax(&key[pos], p);
                break;
            }
            pos = (pos + 1) & mask;
            if (++probes > mask) {
                pa_raise(err, PA_DEV_ERR_RESOURCES);
                break;
            }
        }
        slot_of[p] = (i32)pos;
    }
}

// Phase B: positionLinks.  Each key's chain is a list sorted by descending position starting at the head
// (= key[slot]).  A non-head position inserts itself with a lock-free sorted-list insertion (no deletions):
// publication order = links[p] store (sc1) -> s_waitcnt vmcnt(0) -> CAS on the predecessor.
__global__ __launch_bounds__(256) void k_join_build_links(i32 n, const i32* __restrict__ key, const i32* __restrict__ slot_of, i32* links)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const i32 p = (i32)i;
        const i32 slot = slot_of[p];
        if (slot < 0) continue;
        i32 a = key[slot];
        if (a == p) continue;  // the head; its link is set by whoever inserts right after it
        for (;;) {
            i32 nxt = __hip_atomic_load(&links[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (nxt > p) {
                a = nxt;
                continue;
            }
            __hip_atomic_store(&links[p], nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicCAS(&links[a], nxt, p) == nxt) break;
        }
    }
}

__global__ __launch_bounds__(256) void k_fill_i32(i32* __restrict__ dst, i32 v, i64 n)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) dst[i] = v;
}
void launch_fill_i32(int32_t* dst, int32_t value, int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fill_i32, grid_for(n), 256, 0, s, dst, value, (i64)n);
    PA_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_rebase_offsets(const i32* __restrict__ in, i32 in_base, i32 out_base, i64 n, i32* __restrict__ out)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) out[i] = in[i] - in_base + out_base;
}
void launch_rebase_offsets(const int32_t* in, int32_t in_base, int32_t out_base, int64_t n_plus_1, int32_t* out, hipStream_t s)
{
    if (n_plus_1 <= 0) return;
    hipLaunchKernelGGL(k_rebase_offsets, grid_for(n_plus_1), 256, 0, s, in, in_base, out_base, (i64)n_plus_1, out);
    PA_HIP(hipGetLastError());
}

void launch_join_build(const JoinKeys& build, const int64_t* raw_hash, int32_t n, int32_t* key, uint32_t mask, int32_t* slot_of,
                       int32_t* links, int32_t* err, hipStream_t s)
{
    launch_fill_i32(key, -1, (int64_t)mask + 1, s);
    if (n <= 0) return;
    launch_fill_i32(links, -1, n, s);
    hipLaunchKernelGGL(k_join_build_slots, grid_for(n), 256, 0, s, build, (const i64*)raw_hash, n, key, mask, slot_of, err);
    hipLaunchKernelGGL(k_join_build_links, grid_for(n), 256, 0, s, n, (const i32*)key, (const i32*)slot_of, links);
    PA_HIP(hipGetLastError());
}

// Probe-side table for the general case (several key channels, VARCHAR, DOUBLE ...): build position + the upper half of the
// mixed hash per 8-byte slot, so that a probe that meets another key's slot moves on without touching the build rows (the
// reference's positionToHashes byte filter, PagesHash.java:85-90,182-196, kept inside the slot).  Like the keyed table it is
// separate from PagesHash.key[] and has its own size (load <= 1/2), filled from the occupied slots of key[].
__global__ __launch_bounds__(256) void k_join_tag_slots_clear(u64* __restrict__ tagged, i64 size)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < size; i += (i64)gridDim.x * 256) tagged[i] = ~0ULL;
}
__global__ __launch_bounds__(256) void k_join_tag_slots(const i32* __restrict__ key, i64 hash_size, const i64* __restrict__ raw_hash, u64* tagged,
                                                        u32 mask)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < hash_size; i += (i64)gridDim.x * 256) {
        const i32 p = key[i];
        if (p == -1) continue;
        const u64 mixed = (u64)pa_murmur3_fmix((u64)raw_hash[p]);
        const u64 entry = (mixed & 0xffffffff00000000ULL) | (u64)(u32)p;
        u32 pos = (u32)mixed & mask;
        while (atomicCAS((unsigned long long*)&tagged[pos], ~0ULL, (unsigned long long)entry) != ~0ULL) pos = (pos + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void k_join_probe_count(JoinKeys build, JoinKeys probe, const i64* __restrict__ probe_hash, i32 n_probe,
                                                          const u64* __restrict__ tagged, u32 mask, const i32* __restrict__ links,
                                                          i32* __restrict__ head, i32* __restrict__ counts, int probe_outer)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n_probe; i += (i64)gridDim.x * 256) {
        const i32 r = (i32)i;
        i32 h = -1;
        if (!keys_have_null(probe, r)) {  // JoinProbe.java:89-91
            // linear probing, a 64-byte line (8 slots) per memory round trip
            const u64 mixed = (u64)pa_murmur3_fmix((u64)probe_hash[r]);
            u32 pos = (u32)mixed & mask;
            const ulonglong2* lines = (const ulonglong2*)tagged;
            bool done = false;
            for (u32 seen = 0; !done && seen <= mask;) {
                const u32 base = pos & ~7u, first = pos & 7u;
                ulonglong2 q[4];
#pragma unroll
                for (int k = 0; k < 4; k++) q[k] = lines[(base >> 1) + k];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (done || (u32)k < first) continue;
                    const u64 t = (k & 1) ? q[k >> 1].y : q[k >> 1].x;
                    const i32 cur = (i32)(u32)t;
                    if (cur == -1) done = true;
                    else if (((t ^ mixed) >> 32) == 0ULL && keys_equal(build, cur, probe, r)) {
                        h = cur;
                        done = true;
                    }
                }
                seen += 8u - first;
                pos = (base + 8u) & mask;
            }
        }
        head[r] = h;
        i32 c = 0;
        if (probe_outer & 2) c = h != -1 ? 1 : 0;  // outputSingleMatch: the head of the chain only
        else for (i32 j = h; j != -1; j = links[j]) c++;
        counts[r] = (c == 0 && (probe_outer & 1)) ? 1 : c;  // DefaultPageJoiner.outerJoinCurrentPosition: one NULL-extended row
    }
}

__global__ __launch_bounds__(256) void k_join_probe_emit(const i32* __restrict__ head, const i32* __restrict__ offsets, i32 n_probe, i32 total,
                                                         const i32* __restrict__ links, i32* __restrict__ probe_idx, i32* __restrict__ build_pos,
                                                         int probe_outer, u8* __restrict__ visited)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n_probe; i += (i64)gridDim.x * 256) {
        const i32 r = (i32)i;
        i32 o = offsets[r];
        const i32 h = head[r];
        if (h == -1) {
            if (probe_outer & 1) {  // LookupJoinPageBuilder.appendNullForBuild
                probe_idx[o] = r;
                build_pos[o] = -1;
            }
            continue;
        }
        // the count pass knows how many rows this probe row emits (1 under outputSingleMatch, DefaultPageJoiner.java:276-278):
        // the chain is followed only while more are due, so a unique key never touches positionLinks
        const i32 cnt = ((r + 1 < n_probe) ? offsets[r + 1] : total) - o;
        i32 j = h;
        for (i32 k = 0; k < cnt; k++) {
            probe_idx[o + k] = r;
            build_pos[o + k] = j;
            if (visited) visited[j] = 1;  // OuterLookupSource.appendTo -> positionVisited (same value from every writer)
            if (k + 1 < cnt) j = links[j];
        }
    }
}

__device__ __forceinline__ u64 join_key_bits(const JoinCol& c, i32 r)
{
    return c.type == PA_BIGINT ? (u64)((const i64*)c.values)[r] : (u64)(i64)((const i32*)c.values)[r];  // INTEGER / DATE sign-extended
}

// The probe-side table is the lookup structure only -- PagesHash.key[] (the reference's layout, fill 0.75) stays what the
// build made.  It has its own size (load <= 1/2: an unsuccessful linear probe visits ~2.5 slots instead of ~8.5 at 0.75,
// and most probe rows of a selective join are unsuccessful) and is filled from the occupied slots of key[]: one entry per
// distinct key, so inserting needs no key comparison.
__global__ __launch_bounds__(256) void k_join_key_slots_clear(JoinKeySlot* __restrict__ slots, i64 size)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < size; i += (i64)gridDim.x * 256) {
        JoinKeySlot s;
        s.key = 0ULL;
        s.head = -1;
        s.count = 0;
        slots[i] = s;
    }
}
__global__ __launch_bounds__(256) void k_join_key_slots(const i32* __restrict__ key, i64 hash_size, JoinCol build_key, const i64* __restrict__ raw_hash,
                                                        const i32* __restrict__ links, JoinKeySlot* slots, u32 mask)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < hash_size; i += (i64)gridDim.x * 256) {
        const i32 p = key[i];
        if (p == -1) continue;
        u32 pos = (u32)pa_murmur3_fmix((u64)raw_hash[p]) & mask;
        while (atomicCAS(&slots[pos].head, -1, p) != -1) pos = (pos + 1) & mask;  // load <= 1/2: a free slot exists
        slots[pos].key = join_key_bits(build_key, p);
        i32 c = 1;
        for (i32 j = links[p]; j != -1; j = links[j]) c++;
        slots[pos].count = c;
    }
}

// ---- keyed joins: the probe-side table is built directly from the build rows ---------------------------------------------------
// (PagesHash.key[] -- the reference's layout, only ever read by the parity tests -- is then built on demand.)  Slot protocol:
// head -1 empty -> -2 claimed (key being written) -> build position; equal keys raise the head with atomicMax, so it ends at
// the highest position of the key, the head the reference's sequential insertion leaves.  The loop is wave-uniform: a lane
// that finds a claimed slot looks again in the next round, and the claiming lane (possibly of the same wave) publishes
// inside its own round.
// (Measured, 14.6 M unique keys into 2^25 slots: 1.5 ms -- ~30 G table accesses per second, the rate at which HBM opens rows for
// scattered memory-side operations, whatever their kind: claiming through a 64-bit CAS on the key word plus atomicMax on the head,
// two operations per row and no waiting, took 1.77 ms; dropping the load in front of the CAS 1.62 ms.  Only a build that writes
// every table line once -- rows partitioned by hash first, tables assembled in LDS -- gets below that.)
constexpr i32 kSlotBusy = -2;
__global__ __launch_bounds__(256) void k_join_keyed_build(JoinCol build_key, const i64* __restrict__ raw_hash, i32 n, JoinKeySlot* slots, u32 mask,
                                                          i32* __restrict__ slot_of, i32* err)
{
    const i64 padded = ((i64)n + 255) & ~(i64)255;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < padded; i += (i64)gridDim.x * 256) {
        const i32 p = (i32)i;
        bool pending = i < n && !jcol_is_null(build_key, p);  // PagesHash.java:95-97: rows with a NULL key are not inserted
        if (i < n && !pending) slot_of[p] = -1;
        const u64 v = pending ? join_key_bits(build_key, p) : 0ULL;
        const i64 raw = pending ? (raw_hash ? raw_hash[p] : pa_hash_bigint((i64)v)) : 0;
        u32 pos = (u32)pa_murmur3_fmix((u64)raw) & mask;
        u32 probes = 0;
        while (__ballot(pending) != 0ULL) {
            if (pending) {
                i32 cur = __hip_atomic_load(&slots[pos].head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == -1) {
                    cur = atomicCAS(&slots[pos].head, -1, kSlotBusy);
                    if (cur == -1) {  // claimed: key (and the chain's length while no second row shows: k_join_keyed_next counts then) first, then the position
                        __hip_atomic_store(&slots[pos].count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&slots[pos].key, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        atomicMax(&slots[pos].head, p);
                        slot_of[p] = (i32)pos;
                        pending = false;
                    }
                }
                if (pending && cur >= 0) {
                    if (__hip_atomic_load(&slots[pos].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == v) {
                        err[1] = 1;  // the key has a second row: positionLinks chains exist (idempotent plain store)
                        atomicMax(&slots[pos].head, p);
                        slot_of[p] = (i32)pos;
                        pending = false;
                    }
                    else {
                        pos = (pos + 1) & mask;
                        if (++probes > mask) {
                            pa_raise(err, PA_DEV_ERR_RESOURCES);
                            slot_of[p] = -1;
                            pending = false;
                        }
                    }
                }
                // (cur == kSlotBusy, or the CAS lost to a claim: look at the slot again in the next round)
            }
        }
    }
}

// ---- partitioned build of the keyed table --------------------------------------------------------------------------------------
// Large build sides.  The rows are first regrouped (scan_kernels' multisplit: coalesced reads and writes) by the PARTITION of
// their home slot -- the table is cut into partitions of kJoinPartSlots consecutive slots, and linear probing wraps around inside a
// partition -- then one workgroup per partition assembles its slots in LDS and writes them out once, as whole lines.  No table
// access ever goes to HBM alone: the build streams (keys twice, the table once) instead of opening a DRAM row per operation.
// Rows with a NULL key travel with row position -1 and are not inserted.
__global__ __launch_bounds__(256) void k_join_part_ids(JoinCol build_key, i32 n, u32 mask, i32* __restrict__ part, u64* __restrict__ keybits,
                                                       i32* __restrict__ rowpos)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const i32 p = (i32)i;
        const bool null = jcol_is_null(build_key, p);
        const u64 v = null ? 0ULL : join_key_bits(build_key, p);
        const u32 home = (u32)pa_murmur3_fmix((u64)pa_hash_bigint((i64)v)) & mask;
        part[p] = null ? (i32)((u32)p & (mask >> kJoinPartSlotsLog2)) : (i32)(home >> kJoinPartSlotsLog2);  // (NULL rows: spread, any partition)
        keybits[p] = v;
        rowpos[p] = null ? -1 : p;
    }
}
// err[1]: some key has several rows (their chains are built here too, see below); err[2]: a partition holds more rows than its
// slots take, or a key has more rows than the chain rounds below are good for -- the caller then builds the table the other way.
//
// Chains (ArrayPositionLinks, ArrayPositionLinks.java:45-50: the rows of a key from the highest position down).  A partition's
// rows sit in the registers of its workgroup -- seven per thread -- so a chain is put together in rounds, in LDS: the slot's head
// is the highest position (an atomic max while the rows are inserted); in every round each row not yet on its chain offers its
// position to the current tail's link (another atomic max), the one whose offer stands has become the new tail.  Rounds = the
// longest chain of the partition; no row reads another row's position from HBM, and the links leave the workgroup as one store
// per row.  (The table-wide way -- a lock-free sorted-list insertion through HBM, k_join_keyed_links -- took 1.9 ms for 8 M rows
// with five rows per key.)
constexpr int kJoinPartRowsMax = kJoinPartSlots - kJoinPartSlots / 8;  // rows of a partition (load <= 7/8)
constexpr int kJoinPartRowsPerThread = (kJoinPartRowsMax + 1023) / 1024;
constexpr int kJoinPartChainRounds = 256;
__global__ __launch_bounds__(1024) void k_join_part_build(const u64* __restrict__ keys, const i32* __restrict__ rows, const i64* __restrict__ first, u32 mask,
                                                          JoinKeySlot* __restrict__ slots, i32* __restrict__ links, i32* err)
{
    __shared__ JoinKeySlot tab[kJoinPartSlots];
    __shared__ i32 lnext[kJoinPartRowsMax];  // chain link of local row li, as a build position (-1: none)
    __shared__ int any_dup, more;
    constexpr u32 lmask = kJoinPartSlots - 1;
    for (int i = threadIdx.x; i < kJoinPartSlots; i += 1024) {
        JoinKeySlot e;
        e.key = 0ULL;
        e.head = -1;
        e.count = 0;
        tab[i] = e;
    }
    if (threadIdx.x == 0) any_dup = 0;
    const i64 b0 = first[blockIdx.x], b1 = first[blockIdx.x + 1];
    i32 my_p[kJoinPartRowsPerThread];
    u64 my_v[kJoinPartRowsPerThread];
    u32 my_slot[kJoinPartRowsPerThread];
    // the partition's rows first, all rounds of loads in flight together (while the table above is being cleared): round by round,
    // a workgroup -- the only one its CU holds, the table takes the LDS -- paid seven HBM round trips one after the other
    const bool fits = b1 - b0 <= (i64)kJoinPartRowsMax;
#pragma unroll
    for (int r = 0; r < kJoinPartRowsPerThread; r++) {
        const i64 i = b0 + (i64)r * 1024 + threadIdx.x;
        const bool in = fits && i < b1;
        my_p[r] = in ? rows[i] : -1;
        my_v[r] = in ? keys[i] : 0ULL;
    }
    __syncthreads();
    if (!fits) {
        if (threadIdx.x == 0) err[2] = 1;
    }
    else {
#pragma unroll
        for (int r = 0; r < kJoinPartRowsPerThread; r++) {
            if (b0 + (i64)r * 1024 >= b1) break;  // (uniform: whole rounds of 1024 rows)
            const i32 p = my_p[r];
            bool pending = p >= 0;
            const u64 v = my_v[r];
            u32 pos = (u32)pa_murmur3_fmix((u64)pa_hash_bigint((i64)v)) & lmask;   // (home & lmask: the partition is the slot's upper bits)
            while (__ballot(pending) != 0ULL) {  // same claim / publish protocol as k_join_keyed_build, on LDS
                if (pending) {
                    i32 cur = __hip_atomic_load(&tab[pos].head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (cur == -1) {
                        i32 expected = -1;
                        if (__hip_atomic_compare_exchange_strong(&tab[pos].head, &expected, kSlotBusy, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            __hip_atomic_store(&tab[pos].key, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            tab[pos].count = 1;  // (a partition without a key on several rows leaves as it is now)
                            __hip_atomic_store(&tab[pos].head, p, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            pending = false;
                        }
                        cur = kSlotBusy;  // (lost the claim: look again)
                    }
                    if (pending && cur >= 0) {
                        (void)__hip_atomic_load(&tab[pos].head, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (__hip_atomic_load(&tab[pos].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == v) {
                            any_dup = 1;
                            __hip_atomic_fetch_max(&tab[pos].head, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            pending = false;
                        }
                        else pos = (pos + 1) & lmask;  // (load <= 7/8: an empty slot exists)
                    }
                }
            }
            my_slot[r] = pos;
        }
    }
    __syncthreads();
    if (any_dup) {  // (uniform: LDS flag behind a barrier)
        if (threadIdx.x == 0) err[1] = 1;
        bool linked[kJoinPartRowsPerThread];
        int depth[kJoinPartRowsPerThread];  // place of the row on its chain: 0 = the head
        // the head of every chain is known (the slot's maximum): it is the first tail; while the chains are put together
        // tab[slot].count names the tail's local row (the chain lengths are written when they are done)
#pragma unroll
        for (int r = 0; r < kJoinPartRowsPerThread; r++) {
            linked[r] = true;
            depth[r] = 0;
            if (my_p[r] < 0) continue;
            const int li = r * 1024 + (int)threadIdx.x;
            lnext[li] = -1;
            linked[r] = tab[my_slot[r]].head == my_p[r];
            if (linked[r]) tab[my_slot[r]].count = li;
        }
        __syncthreads();
        int round = 0;
        for (; round < kJoinPartChainRounds; round++) {
            if (threadIdx.x == 0) more = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kJoinPartRowsPerThread; r++) {
                if (linked[r]) continue;
                __hip_atomic_fetch_max(&lnext[tab[my_slot[r]].count], my_p[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                more = 1;
            }
            __syncthreads();
            if (!more) break;
            bool tail_now[kJoinPartRowsPerThread];
#pragma unroll
            for (int r = 0; r < kJoinPartRowsPerThread; r++) {
                tail_now[r] = !linked[r] && lnext[tab[my_slot[r]].count] == my_p[r];
            }
            __syncthreads();  // (every row has read its chain's tail before any tail moves)
#pragma unroll
            for (int r = 0; r < kJoinPartRowsPerThread; r++) {
                if (!tail_now[r]) continue;
                linked[r] = true;
                depth[r] = round + 1;
                tab[my_slot[r]].count = r * 1024 + (int)threadIdx.x;
            }
            __syncthreads();
        }
        if (round == kJoinPartChainRounds && threadIdx.x == 0) err[2] = 1;  // a key with more rows than that: the caller's other way
        // the links leave for HBM; the slot gets its chain's length (JoinKeySlot::count): the place of its last row + 1
#pragma unroll
        for (int r = 0; r < kJoinPartRowsPerThread; r++) {
            if (my_p[r] >= 0) links[my_p[r]] = lnext[r * 1024 + (int)threadIdx.x];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kJoinPartRowsPerThread; r++) {
            if (my_p[r] >= 0 && depth[r] == 0) tab[my_slot[r]].count = 1;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kJoinPartRowsPerThread; r++) {
            if (my_p[r] >= 0 && depth[r] > 0) __hip_atomic_fetch_max(&tab[my_slot[r]].count, depth[r] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
    }
    const uint4* src = (const uint4*)tab;
    uint4* dst = (uint4*)(slots + (u64)blockIdx.x * kJoinPartSlots);
    for (int i = threadIdx.x; i < kJoinPartSlots; i += 1024) dst[i] = src[i];
}

// positionLinks for the keyed table: as k_join_build_links, the head of a row's chain is slots[slot].head
__global__ __launch_bounds__(256) void k_join_keyed_links(i32 n, const JoinKeySlot* __restrict__ slots, const i32* __restrict__ slot_of, i32* links)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const i32 p = (i32)i;
        const i32 slot = slot_of[p];
        if (slot < 0) continue;
        i32 a = slots[slot].head;
        if (a == p) continue;
        for (;;) {
            i32 nxt = __hip_atomic_load(&links[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (nxt > p) {
                a = nxt;
                continue;
            }
            __hip_atomic_store(&links[p], nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicCAS(&links[a], nxt, p) == nxt) break;
        }
    }
}
__global__ __launch_bounds__(256) void k_join_keyed_next(JoinKeySlot* __restrict__ slots, i64 size, const i32* __restrict__ links)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < size; i += (i64)gridDim.x * 256) {
        const i32 h = slots[i].head;
        if (h < 0) continue;
        i32 c = 1;
        for (i32 j = links[h]; j != -1; j = links[j]) c++;
        slots[i].count = c;
    }
}

__global__ __launch_bounds__(256) void k_join_key_bitmap(JoinCol build_key, i32 n, i64 min_key, u64 range, u64* __restrict__ bits)
{
    // Build sides often arrive clustered by the key (Q3's orders are in orderkey order): neighbouring lanes then set bits of the
    // same word.  The bits of a run of lanes with one word are OR-ed together along the wave and the run's last lane sends one
    // atomic -- scattered atomics are what bounds this kernel (they execute at the memory side, about 40 G/s).
    const int lane = threadIdx.x & 63;
    const i64 padded = ((i64)n + 63) & ~(i64)63;  // whole waves in every round
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < padded; i += (i64)gridDim.x * 256) {
        u64 word = ~0ULL, bit = 0ULL;
        if (i < n && !jcol_is_null(build_key, (i32)i)) {
            const u64 d = (u64)((i64)join_key_bits(build_key, (i32)i) - min_key);
            if (d <= range) {
                word = d >> 6;
                bit = 1ULL << (d & 63ULL);
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u64 w2 = (u64)__shfl_up((unsigned long long)word, off, 64);
            const u64 b2 = (u64)__shfl_up((unsigned long long)bit, off, 64);
            if (lane >= off && w2 == word) bit |= b2;
        }
        const u64 next = (u64)__shfl_down((unsigned long long)word, 1, 64);
        if (bit != 0ULL && (lane == 63 || next != word)) atomicOr((unsigned long long*)&bits[word], (unsigned long long)bit);
    }
}

// ---- the bitmap of a build side whose rows do NOT arrive in key order (behind an exchange, behind another join) ----
// One atomic per key at a random word of the bitmap is what the kernel above then costs (15 M keys: 0.56 ms, the rank -> row
// scatter as much again).  Instead: is the column out of order at all (k_join_key_stats), then the (key, row) pairs are
// regrouped by key RANGE -- partition = (key - min) >> shift, at most 4096 of them, one LDS-staged multisplit --, a workgroup
// ORs the bits of its partitions together in LDS and writes the words out with plain stores, and the rank -> row scatter walks
// the regrouped pairs: its stores stay inside one partition's slice of the array at a time.
// One pass over the build keys for everything the build decides on first: min and max of the non-NULL keys, whether there is one at
// all, and whether some key is smaller than the key of the row before it.  out[0] = max over the keys of ~image, out[1] = max of image
// (image = the key with its sign bit flipped: unsigned order = signed order), out[2] = 1 when some key is not NULL, out[3] = the number of
// pairs of neighbouring non-NULL keys that descend (0 is the identity of all four).
constexpr int kKeyStatsBlocks = 2048;
__global__ __launch_bounds__(256) void k_join_key_stats(JoinCol key, i32 n, u64* __restrict__ partials)
{
    __shared__ u64 s_lo[4], s_hi[4], s_off[4];
    __shared__ int s_flags[4];
    u64 lo = 0ULL, hi = 0ULL;   // max of ~image / of image
    bool any = false;
    u32 off = 0u;               // neighbouring pairs that descend
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        if (jcol_is_null(key, (i32)i)) continue;
        const i64 v = (i64)join_key_bits(key, (i32)i);
        const u64 img = (u64)v ^ 0x8000000000000000ULL;
        lo = any ? (~img > lo ? ~img : lo) : ~img;
        hi = any ? (img > hi ? img : hi) : img;
        any = true;
        if (i + 1 < n && !jcol_is_null(key, (i32)i + 1) && v > (i64)join_key_bits(key, (i32)i + 1)) off++;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u64 l2 = (u64)__shfl_xor((unsigned long long)lo, d, 64), h2 = (u64)__shfl_xor((unsigned long long)hi, d, 64);
        lo = l2 > lo ? l2 : lo;   // (a lane without keys holds 0 in both: the identity of max, and "any" travels on its own)
        hi = h2 > hi ? h2 : hi;
    }
    const int flags = __ballot(any) != 0ULL ? 1 : 0;
    const u64 offs = (u64)pa_wave_sum_i64((i64)off);
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
        s_off[threadIdx.x >> 6] = offs;
        s_flags[threadIdx.x >> 6] = flags;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int f = 0;
        for (int w = 0; w < 4; w++) {
            lo = s_lo[w] > lo ? s_lo[w] : lo;
            hi = s_hi[w] > hi ? s_hi[w] : hi;
            f |= s_flags[w];
        }
        // (one line of partials per workgroup, folded by k_join_key_stats_fold: 4096 atomic maxima on two words took 170 us)
        u64* p = partials + (u64)blockIdx.x * 4ULL;
        p[0] = lo;
        p[1] = hi;
        p[2] = (u64)(f & 1);
        p[3] = s_off[0] + s_off[1] + s_off[2] + s_off[3];
    }
}
__global__ __launch_bounds__(256) void k_join_key_stats_fold(const u64* __restrict__ partials, int blocks, u64* __restrict__ out)
{
    __shared__ u64 s_v[4][4];
    u64 v[4] = {0ULL, 0ULL, 0ULL, 0ULL};
    for (int b = threadIdx.x; b < blocks; b += 256) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u64 x = partials[(u64)b * 4ULL + k];
            v[k] = k == 3 ? v[k] + x : (x > v[k] ? x : v[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const u64 o = (u64)__shfl_xor((unsigned long long)v[k], d, 64);
            v[k] = k == 3 ? v[k] + o : (o > v[k] ? o : v[k]);
        }
        if ((threadIdx.x & 63) == 0) s_v[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        u64 r = 0ULL;
        for (int w = 0; w < 4; w++) r = threadIdx.x == 3 ? r + s_v[w][threadIdx.x] : (s_v[w][threadIdx.x] > r ? s_v[w][threadIdx.x] : r);
        out[threadIdx.x] = r;
    }
}
__global__ __launch_bounds__(256) void k_join_range_ids(JoinCol build_key, i32 n, i64 min_key, int shift, i32 partitions, i32* __restrict__ part,
                                                        u64* __restrict__ keybits, i32* __restrict__ rowpos)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const i32 p = (i32)i;
        const bool null = jcol_is_null(build_key, p);
        const u64 v = null ? 0ULL : join_key_bits(build_key, p);
        part[p] = null ? partitions : (i32)(((u64)((i64)v - min_key)) >> shift);   // (NULL keys: a partition of their own, never read)
        keybits[p] = v;
        rowpos[p] = p;
    }
}
constexpr int kJoinRangeWordsMax = 8192;   // 2^19 key values per partition at most (64 KB of LDS)
__global__ __launch_bounds__(1024) void k_join_range_bitmap(const u64* __restrict__ keys, const i64* __restrict__ first, i32 partitions, i64 min_key, int shift,
                                                            u64 range, u64* __restrict__ bits)
{
    __shared__ u64 words[kJoinRangeWordsMax];
    const int nw = 1 << (shift - 6);
    const i64 total_words = (i64)(range >> 6) + 1;
    for (i32 p = (i32)blockIdx.x; p < partitions; p += (i32)gridDim.x) {
        for (int i = threadIdx.x; i < nw; i += 1024) words[i] = 0ULL;
        __syncthreads();
        const u64 base = (u64)p << shift;
        for (i64 i = first[p] + threadIdx.x; i < first[p + 1]; i += 1024) {
            const u64 d = (u64)((i64)keys[i] - min_key) - base;   // < 2^shift by the partition rule
            __hip_atomic_fetch_or(&words[d >> 6], 1ULL << (d & 63ULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        const i64 w0 = (i64)(base >> 6);
        for (int i = threadIdx.x; i < nw; i += 1024) {
            if (w0 + i < total_words) bits[w0 + i] = words[i];
        }
        __syncthreads();
    }
}
// ---- key rank index (join_kernels.hpp) ----
// 16 consecutive words per thread, 1024 threads: a workgroup's tile is 16 384 words (2^20 key values).  First the tiles' bit counts,
// scanned by the caller (573 entries for Q3's 600 M-wide orderkey range: one small launch); then every tile again -- its words ranked
// inside the workgroup, plus the tile's offset -- writing {bits, bits below} once.  (Five launches before: counts per word, a three-launch
// scan of them, and the zip -- the count array was written, scanned and read again: 109 us for 9.4 M words, now the two passes below.)
constexpr int kRankTileThreads = 1024, kRankWordsPerThread = 16, kRankTileWords = kRankTileThreads * kRankWordsPerThread;
__device__ __forceinline__ i32 rank_thread_words(const u64* __restrict__ bits, i64 nwords, i64 first, u64 (&w)[kRankWordsPerThread])
{
    i32 c = 0;
    if (first + kRankWordsPerThread <= nwords) {
#pragma unroll
        for (int i = 0; i < kRankWordsPerThread; i += 2) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(bits + first + i);   // (first is a multiple of 16: 16-byte aligned)
            w[i] = v.x;
            w[i + 1] = v.y;
        }
    }
    else {
#pragma unroll
        for (int i = 0; i < kRankWordsPerThread; i++) w[i] = first + i < nwords ? bits[first + i] : 0ULL;
    }
#pragma unroll
    for (int i = 0; i < kRankWordsPerThread; i++) c += (i32)__popcll(w[i]);
    return c;
}
__global__ __launch_bounds__(kRankTileThreads) void k_join_rank_tile_sums(const u64* __restrict__ bits, i64 nwords, i32* __restrict__ sums)
{
    __shared__ i32 s_wave[kRankTileThreads / 64];
    u64 w[kRankWordsPerThread];
    i32 c = rank_thread_words(bits, nwords, (i64)blockIdx.x * kRankTileWords + (i64)threadIdx.x * kRankWordsPerThread, w);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        i32 t = 0;
        for (int i = 0; i < kRankTileThreads / 64; i++) t += s_wave[i];
        sums[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(kRankTileThreads) void k_join_rank_tile_zip(const u64* __restrict__ bits, const i32* __restrict__ tile_below, i64 nwords,
                                                                          JoinRankWord* __restrict__ words)
{
    __shared__ i32 s_wave[kRankTileThreads / 64];
    u64 w[kRankWordsPerThread];
    const i64 first = (i64)blockIdx.x * kRankTileWords + (i64)threadIdx.x * kRankWordsPerThread;
    const i32 mine = rank_thread_words(bits, nwords, first, w);
    // exclusive scan of the threads' counts: inside the wave by shuffles, across the 16 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const i32 o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    i32 below = tile_below[blockIdx.x] + incl - mine;
    for (int i = 0; i < wave; i++) below += s_wave[i];
#pragma unroll
    for (int i = 0; i < kRankWordsPerThread; i++) {
        if (first + i < nwords) ((uint4*)words)[first + i] = uint4{(u32)w[i], (u32)(w[i] >> 32), (u32)below, 0u};
        below += (i32)__popcll(w[i]);
    }
}
__device__ __forceinline__ i32 join_rank_of(const JoinRankWord* __restrict__ words, const u64 d)
{
    const uint4 w = ((const uint4*)words)[d >> 6];
    const u64 bits = ((u64)w.y << 32) | (u64)w.x;
    const u32 b = (u32)(d & 63ULL);
    if (((bits >> b) & 1ULL) == 0ULL) return -1;
    return (i32)w.z + (i32)__popcll(bits & ((1ULL << b) - 1ULL));
}
__global__ __launch_bounds__(256) void k_join_rank_rows_pairs(const u64* __restrict__ keys, const i32* __restrict__ rowpos, i64 n,
                                                              const JoinRankWord* __restrict__ words, i64 min_key, i32* __restrict__ rows, const i32* __restrict__ distinct)
{
    if (distinct && (i64)*distinct != n) return;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const i32 r = join_rank_of(words, (u64)((i64)keys[i] - min_key));
        if (r >= 0) rows[r] = rowpos[i];
    }
}

// ... or, partition by partition, through LDS: the ranks of a partition's keys are one contiguous range (its first key value's
// rank onwards), so the workgroup places the rows in LDS and writes the range out as whole lines
constexpr int kJoinRangeRowsMax = 32768;   // key values per partition this form takes (128 KB of LDS)
__global__ __launch_bounds__(1024) void k_join_range_rows(const u64* __restrict__ keys, const i32* __restrict__ rowpos, const i64* __restrict__ first,
                                                          i32 partitions, const JoinRankWord* __restrict__ words, i64 min_key, int shift, i64 n,
                                                          i32* __restrict__ rows, const i32* __restrict__ distinct)
{
    __shared__ i32 stage[kJoinRangeRowsMax];
    if (distinct && (i64)*distinct != n) return;  // some key has several rows: the index will be dropped, nothing to write
    for (i32 p = (i32)blockIdx.x; p < partitions; p += (i32)gridDim.x) {
        const i64 b0 = first[p], b1 = first[p + 1];
        if (b1 <= b0) continue;   // (uniform: every thread reads the same bounds)
        const i32 base = (i32)((const uint4*)words)[((u64)p << shift) >> 6].z;   // build keys below the partition's first key value
        const i64 cnt = b1 - b0;   // (one row per rank when no key repeats -- else the index is dropped and what lands here is not read)
        for (i64 i = threadIdx.x; i < cnt && i < kJoinRangeRowsMax; i += 1024) stage[i] = -1;
        __syncthreads();
        for (i64 i = b0 + threadIdx.x; i < b1; i += 1024) {
            const i32 r = join_rank_of(words, (u64)((i64)keys[i] - min_key)) - base;
            if (r >= 0 && r < kJoinRangeRowsMax) stage[r] = rowpos[i];
        }
        __syncthreads();
        for (i64 i = threadIdx.x; i < cnt && i < kJoinRangeRowsMax; i += 1024) {
            if ((i64)base + i < n) rows[(i64)base + i] = stage[i];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_join_rank_rows(JoinCol build_key, i32 n, const JoinRankWord* __restrict__ words, i64 min_key, i32* __restrict__ rows,
                                                        i32* __restrict__ unordered, const i32* __restrict__ distinct)
{
    if (distinct && *distinct != n) return;  // some key has several rows: the index will be dropped
    bool off = false;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const i32 r = join_rank_of(words, (u64)((i64)join_key_bits(build_key, (i32)i) - min_key));  // (every build key has its bit)
        if (r >= 0) rows[r] = (i32)i;
        off = off || r != (i32)i;
    }
    if (__ballot(off) != 0ULL && (threadIdx.x & 63) == 0) *unordered = 1;
}

__global__ __launch_bounds__(256) void k_join_probe_count_keyed(JoinCol probe_key, const i64* __restrict__ probe_hash, i32 n_probe,
                                                                const JoinKeySlot* __restrict__ slots, u32 mask, u32 wrap, const i32* __restrict__ links,
                                                                JoinKeyBitmap bitmap, JoinRankIndex rank, i32* __restrict__ head, i32* __restrict__ counts, int flags,
                                                                unsigned long long* __restrict__ total)
{
    i64 mine = 0;  // this thread's output rows: the page's total is summed on the way (one atomic per wave), not by a pass of its own
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n_probe; i += (i64)gridDim.x * 256) {
        const i32 r = (i32)i;
        i32 h = -1, rows_of_key = 0;
        bool may_match = !jcol_is_null(probe_key, r);  // JoinProbe.java:89-91
        if (rank.words) {  // (no duplicate keys: the key's rank names its only build row)
            if (may_match) {
                const u64 d = (u64)((i64)join_key_bits(probe_key, r) - rank.min_key);
                if (d <= rank.range) {
                    h = join_rank_of(rank.words, d);
                    if (h >= 0 && rank.rows) h = rank.rows[h];
                }
            }
            may_match = false;
        }
        if (may_match && bitmap.bits) {
            const u64 d = (u64)((i64)join_key_bits(probe_key, r) - bitmap.min_key);
            may_match = d <= bitmap.range && ((bitmap.bits[d >> 6] >> (d & 63ULL)) & 1ULL) != 0ULL;
        }
        if (may_match) {
            const u64 v = join_key_bits(probe_key, r);
            const i64 raw = probe_hash ? probe_hash[r] : pa_hash_bigint((i64)v);  // 31 * 0 + hash(value): one channel
            // linear probing, fetched a 64-byte line (4 slots) at a time: the wave waits for its longest probe sequence, and a
            // sequence of k slots costs ceil(k / 4) memory round trips instead of k
            u32 pos = (u32)pa_murmur3_fmix((u64)raw) & mask;
            const uint4* lines = (const uint4*)slots;
            bool done = false;
            for (u32 seen = 0; !done && seen <= wrap;) {  // (wrap: the probe sequence stays inside the slot's partition)
                const u32 base = pos & ~3u, first = pos & 3u;
                uint4 q[4];
#pragma unroll
                for (int k = 0; k < 4; k++) q[k] = lines[base + k];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (done || (u32)k < first) continue;
                    const i32 cur = (i32)q[k].z;
                    if (cur == -1) done = true;
                    else if ((((u64)q[k].y << 32) | (u64)q[k].x) == v) {
                        h = cur;
                        rows_of_key = (i32)q[k].w;
                        done = true;
                    }
                }
                seen += 4u - first;
                pos = (pos & ~wrap) | ((base + 4u) & wrap);
            }
        }
        head[r] = h;
        i32 c = 0;
        if (h != -1) c = ((flags & 2) || rank.words || rows_of_key < 1) ? 1 : rows_of_key;  // (the slot knows its chain's length: no walk here)
        counts[r] = (c == 0 && (flags & 1)) ? 1 : c;  // DefaultPageJoiner.outerJoinCurrentPosition: one NULL-extended row
        mine += (c == 0 && (flags & 1)) ? 1 : c;
    }
    if (total) {
        // one atomic per WORKGROUP, spread over 16 counters: same-address atomics retire one after the other at the memory side
        // (~8 ns each), and the grid's waves all arrive here together -- one per wave on one counter cost this kernel 100 us of tail
        __shared__ i64 wave_total[4];
        mine = pa_wave_sum_i64(mine);
        if ((threadIdx.x & 63) == 0) wave_total[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            const i64 all = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
            if (all != 0) atomicAdd(total + (blockIdx.x & 15u), (unsigned long long)all);   // 16 counters: the host adds them up
        }
    }
}


// partition 0 = build positions never visited (the rows LookupOuterOperator emits), 1 = visited
__global__ __launch_bounds__(256) void k_join_unvisited_flag(const u8* __restrict__ visited, i64 n, i32* __restrict__ partition)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) partition[i] = visited[i] ? 1 : 0;
}

void launch_join_tag_slots(const int32_t* key, int64_t hash_size, const int64_t* raw_hash, uint64_t* tagged, uint32_t tagged_mask, hipStream_t s)
{
    hipLaunchKernelGGL(k_join_tag_slots_clear, grid_for((int64_t)tagged_mask + 1), 256, 0, s, (u64*)tagged, (i64)tagged_mask + 1);
    hipLaunchKernelGGL(k_join_tag_slots, grid_for(hash_size), 256, 0, s, key, (i64)hash_size, (const i64*)raw_hash, (u64*)tagged, tagged_mask);
    PA_HIP(hipGetLastError());
}
void launch_join_probe_count(const JoinKeys& build, const JoinKeys& probe, const int64_t* probe_hash, int32_t n_probe, const uint64_t* tagged,
                             uint32_t mask, const int32_t* links, int32_t* head, int32_t* counts, int flags, hipStream_t s)
{
    if (n_probe <= 0) return;
    hipLaunchKernelGGL(k_join_probe_count, grid_for(n_probe), 256, 0, s, build, probe, (const i64*)probe_hash, n_probe, (const u64*)tagged, mask,
                       links, head, counts, flags);
    PA_HIP(hipGetLastError());
}
void launch_join_unvisited_flag(const uint8_t* visited, int64_t n, int32_t* partition, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_join_unvisited_flag, grid_for(n), 256, 0, s, visited, (i64)n, partition);
    PA_HIP(hipGetLastError());
}
void launch_join_probe_emit(const int32_t* head, const int32_t* offsets, int32_t n_probe, int32_t total, const int32_t* links, int32_t* probe_idx,
                            int32_t* build_pos, int flags, uint8_t* visited, hipStream_t s)
{
    if (n_probe <= 0) return;
    hipLaunchKernelGGL(k_join_probe_emit, grid_for(n_probe), 256, 0, s, head, offsets, n_probe, total, links, probe_idx, build_pos, flags, visited);
    PA_HIP(hipGetLastError());
}
void launch_join_key_slots(const int32_t* key, int64_t hash_size, const JoinCol& build_key, const int64_t* raw_hash, const int32_t* links,
                           JoinKeySlot* slots, uint32_t slots_mask, hipStream_t s)
{
    hipLaunchKernelGGL(k_join_key_slots_clear, grid_for((int64_t)slots_mask + 1), 256, 0, s, slots, (i64)slots_mask + 1);
    hipLaunchKernelGGL(k_join_key_slots, grid_for(hash_size), 256, 0, s, key, (i64)hash_size, build_key, (const i64*)raw_hash, links, slots, slots_mask);
    PA_HIP(hipGetLastError());
}
void launch_join_keyed_build(const JoinCol& build_key, const int64_t* raw_hash, int32_t n, JoinKeySlot* slots, uint32_t slots_mask, int32_t* slot_of,
                             int32_t* links, int32_t* err, hipStream_t s)
{
    hipLaunchKernelGGL(k_join_key_slots_clear, grid_for((int64_t)slots_mask + 1), 256, 0, s, slots, (i64)slots_mask + 1);
    if (n > 0) {
        launch_fill_i32(links, -1, n, s);
        hipLaunchKernelGGL(k_join_keyed_build, grid_for(n), 256, 0, s, build_key, (const i64*)raw_hash, n, slots, slots_mask, slot_of, err);
    }
    PA_HIP(hipGetLastError());
}
void launch_join_part_ids(const JoinCol& build_key, int32_t n, uint32_t slots_mask, int32_t* part, uint64_t* keybits, int32_t* rowpos, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_join_part_ids, grid_for(n), 256, 0, s, build_key, n, slots_mask, part, (u64*)keybits, rowpos);
    PA_HIP(hipGetLastError());
}
void launch_join_part_build(const uint64_t* keys, const int32_t* rows, const int64_t* first, int32_t partitions, uint32_t slots_mask, JoinKeySlot* slots,
                            int32_t* links, int32_t* err, hipStream_t s)
{
    hipLaunchKernelGGL(k_join_part_build, partitions, 1024, 0, s, (const u64*)keys, rows, (const i64*)first, slots_mask, slots, links, err);
    PA_HIP(hipGetLastError());
}
void launch_join_keyed_links(int32_t n, JoinKeySlot* slots, uint32_t slots_mask, const int32_t* slot_of, int32_t* links, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_join_keyed_links, grid_for(n), 256, 0, s, n, (const JoinKeySlot*)slots, (const i32*)slot_of, links);
    hipLaunchKernelGGL(k_join_keyed_next, grid_for((int64_t)slots_mask + 1), 256, 0, s, slots, (i64)slots_mask + 1, (const i32*)links);
    PA_HIP(hipGetLastError());
}
void launch_join_key_bitmap(const JoinCol& build_key, int32_t n, int64_t min_key, uint64_t range, uint64_t* bits, hipStream_t s)
{
    PA_HIP(hipMemsetAsync(bits, 0, (size_t)((range >> 6) + 1) * 8, s));
    if (n <= 0) return;
    hipLaunchKernelGGL(k_join_key_bitmap, grid_for(n), 256, 0, s, build_key, n, (i64)min_key, (u64)range, (u64*)bits);
    PA_HIP(hipGetLastError());
}
size_t join_key_stats_temp_bytes() { return (size_t)kKeyStatsBlocks * 32; }
void launch_join_key_stats(const JoinCol& key, int32_t n, uint64_t* out, void* temp, hipStream_t s)
{
    if (n <= 0) {
        PA_HIP(hipMemsetAsync(out, 0, 32, s));
        return;
    }
    const int blocks = (int)std::min<int64_t>(((int64_t)n + 255) / 256, kKeyStatsBlocks);
    hipLaunchKernelGGL(k_join_key_stats, blocks, 256, 0, s, key, n, (u64*)temp);
    hipLaunchKernelGGL(k_join_key_stats_fold, 1, 256, 0, s, (const u64*)temp, blocks, (u64*)out);
    PA_HIP(hipGetLastError());
}
int join_range_shift(uint64_t range)
{
    int shift = 15;   // 2^15 key values per partition: the rank -> row array then goes through LDS too
    while (((range >> shift) + 1) > 4096) shift++;
    return shift <= 19 ? shift : -1;   // (a wider key range: the atomic bitmap kernel)
}
void launch_join_range_ids(const JoinCol& build_key, int32_t n, int64_t min_key, int shift, int32_t partitions, int32_t* part, uint64_t* keybits,
                           int32_t* rowpos, hipStream_t s)
{
    hipLaunchKernelGGL(k_join_range_ids, grid_for(n), 256, 0, s, build_key, n, (i64)min_key, shift, partitions, part, (u64*)keybits, rowpos);
    PA_HIP(hipGetLastError());
}
void launch_join_range_bitmap(const uint64_t* keys, const int64_t* first, int32_t partitions, int64_t min_key, int shift, uint64_t range, uint64_t* bits,
                              hipStream_t s)
{
    hipLaunchKernelGGL(k_join_range_bitmap, std::min(partitions, 2048), 1024, 0, s, (const u64*)keys, (const i64*)first, partitions, (i64)min_key, shift,
                       (u64)range, (u64*)bits);
    PA_HIP(hipGetLastError());
}
void launch_join_rank_rows_pairs(const uint64_t* keys, const int32_t* rowpos, int64_t n, const JoinRankWord* words, int64_t min_key, int32_t* rows,
                                 hipStream_t s, const int64_t* first, int32_t partitions, int shift, const int32_t* distinct)
{
    if (n <= 0) return;
    if (first != nullptr && shift >= 6 && (1 << shift) <= kJoinRangeRowsMax) {
        hipLaunchKernelGGL(k_join_range_rows, std::min(partitions, 2048), 1024, 0, s, (const u64*)keys, rowpos, (const i64*)first, partitions, words, (i64)min_key,
                           shift, (i64)n, rows, distinct);
        PA_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(k_join_rank_rows_pairs, grid_for(n), 256, 0, s, (const u64*)keys, rowpos, (i64)n, words, (i64)min_key, rows, distinct);
    PA_HIP(hipGetLastError());
}
int64_t join_rank_tiles(int64_t nwords) { return (nwords + kRankTileWords - 1) / kRankTileWords; }
void launch_join_rank_words(const uint64_t* bits, int64_t nwords, JoinRankWord* words, int32_t* counts, void* temp, int32_t* total_out, hipStream_t s)
{
    if (nwords <= 0) return;
    const int64_t tiles = (nwords + kRankTileWords - 1) / kRankTileWords;
    hipLaunchKernelGGL(k_join_rank_tile_sums, (int)tiles, kRankTileThreads, 0, s, (const u64*)bits, (i64)nwords, counts);
    launch_exclusive_scan_i32(counts, counts, tiles, total_out, temp, s);
    hipLaunchKernelGGL(k_join_rank_tile_zip, (int)tiles, kRankTileThreads, 0, s, (const u64*)bits, (const i32*)counts, (i64)nwords, words);
    PA_HIP(hipGetLastError());
}
void launch_join_rank_rows(const JoinCol& build_key, int32_t n, const JoinRankWord* words, int64_t min_key, int32_t* rows, int32_t* unordered, hipStream_t s,
                           const int32_t* distinct)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_join_rank_rows, grid_for(n), 256, 0, s, build_key, n, words, (i64)min_key, rows, unordered, distinct);
    PA_HIP(hipGetLastError());
}
// The same over a lookup source WITHOUT duplicate keys, four consecutive probe rows per thread and step: their keys, then their
// bitmap / rank words, then their slots are loaded back to back and waited for once (pa_join_probe4, the probe of the generated
// fused kernels) -- row by row a wave waits for two to three dependent trips to memory per row and runs at a third of the rate.
// Every chain is one row long, so a row's count is whether it matched (or 1, for the NULL-extended row of a probe-outer join).
struct ProbeTable {
    const void* jslots;
    const u64* jbits;
    i64 jmin;
    u64 jrange;
    u32 jmask, jwrap;
    const pa_u32x4* jrank;
    const i32* jrank_rows;
};
// tile_totals (may be null): the output rows of every tile of kJoinProbeTileRows probe rows -- with them the pairs are emitted tile by
// tile (k_join_probe_emit_tiles: ranks inside the workgroup) and the exclusive scan runs over the tiles' totals, a thousandth of the rows
// kChains: keys may have several build rows -- the slot names the chain's length (JoinKeySlot::count), so counting still costs one
// slot per probe row and no step along positionLinks (flags & 2, outputSingleMatch: one output row per matching probe row)
template <bool kChains>
__global__ __launch_bounds__(256) void k_join_probe_count_keyed4(JoinCol probe_key, i32 n_probe, ProbeTable t, i32* __restrict__ head, i32* __restrict__ counts,
                                                                 int flags, unsigned long long* __restrict__ total, i32* __restrict__ tile_totals)
{
    __shared__ i64 wave_total[4];
    __shared__ i32 wave_tile[4];
    i64 mine = 0;
    const i64 quads = ((i64)n_probe + 3) >> 2;
    const i64 tiles = (quads + 255) >> 8;
    for (i64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const i64 q = tile * 256 + threadIdx.x;
        i32 here = 0;
        if (q < quads) {
            bool s[4];
            u64 k[4];
            i32 jb[4], jc[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const i64 i = 4 * q + r;
                s[r] = i < n_probe && !jcol_is_null(probe_key, (i32)i);  // JoinProbe.java:89-91
                k[r] = s[r] ? join_key_bits(probe_key, (i32)i) : 0ULL;
            }
            pa_join_probe4x<kChains>(t, s, k, jb, jc);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const i64 i = 4 * q + r;
                if (i >= n_probe) continue;
                const i32 h = s[r] ? jb[r] : -1;
                const i32 c = h != -1 ? ((kChains && !(flags & 2) && jc[r] > 1) ? jc[r] : 1) : ((flags & 1) ? 1 : 0);
                head[i] = h;
                counts[i] = c;
                here += c;
            }
        }
        mine += here;
        if (tile_totals) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) here += __shfl_xor(here, d, 64);
            if ((threadIdx.x & 63) == 0) wave_tile[threadIdx.x >> 6] = here;
            __syncthreads();
            if (threadIdx.x == 0) tile_totals[tile] = wave_tile[0] + wave_tile[1] + wave_tile[2] + wave_tile[3];
            __syncthreads();
        }
    }
    if (total) {
        mine = pa_wave_sum_i64(mine);
        if ((threadIdx.x & 63) == 0) wave_total[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            const i64 all = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
            if (all != 0) atomicAdd(total + (blockIdx.x & 15u), (unsigned long long)all);
        }
    }
}
// The pairs of a page whose probe rows emit at most one pair each (no build key on several rows), tile by tile: a row's place is its
// tile's offset (the scanned tile totals) plus the pairs of the rows before it in the tile.  (Measured and dropped: the same for chains
// of several build rows -- four consecutive probe rows per thread, each walking its chain: 1.4 M probe rows x 5 matches 0.19 -> 0.34 ms;
// those pages keep one probe row per thread and the scan over the rows' counts.)
__global__ __launch_bounds__(256) void k_join_probe_emit_tiles(const i32* __restrict__ head, const i32* __restrict__ tile_offsets, i32 n_probe, int probe_outer,
                                                               i32* __restrict__ probe_idx, i32* __restrict__ build_pos, u8* __restrict__ visited)
{
    __shared__ i32 wave_sum[4];
    const i64 quads = ((i64)n_probe + 3) >> 2;
    const i64 tiles = (quads + 255) >> 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (i64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const i64 q = tile * 256 + threadIdx.x;
        i32 h[4] = {-1, -1, -1, -1};
        bool out[4] = {false, false, false, false};
        i32 here = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const i64 i = 4 * q + r;
            if (i < n_probe) {
                h[r] = head[i];
                out[r] = h[r] != -1 || (probe_outer & 1);
                here += out[r] ? 1 : 0;
            }
        }
        i32 incl = here;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const i32 o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        i32 o = tile_offsets[tile] + incl - here;
        for (int w = 0; w < wave; w++) o += wave_sum[w];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (!out[r]) continue;
            probe_idx[o] = (i32)(4 * q + r);
            build_pos[o] = h[r];   // (-1: LookupJoinPageBuilder.appendNullForBuild)
            if (visited && h[r] >= 0) visited[h[r]] = 1;
            o++;
        }
    }
}

int64_t join_probe_tiles(int32_t n_probe) { return ((((int64_t)n_probe + 3) >> 2) + 255) >> 8; }
bool launch_join_probe_count_keyed(const JoinCol& probe_key, const int64_t* probe_hash, int32_t n_probe, const JoinKeySlot* slots, uint32_t mask,
                                   uint32_t wrap, const int32_t* links, const JoinKeyBitmap& bitmap, const JoinRankIndex& rank, int32_t* head, int32_t* counts,
                                   int flags, hipStream_t s, int64_t* total, bool unique_keys, int32_t* tile_totals)
{
    if (n_probe <= 0) return false;
    // (with a $hashvalue channel the home slot comes from the channel's value: the row-by-row kernel reads it)
    if (probe_hash == nullptr && (rank.words == nullptr || rank.min_key == bitmap.min_key || bitmap.bits == nullptr)) {
        ProbeTable t{};
        t.jslots = slots;
        t.jbits = rank.words ? nullptr : reinterpret_cast<const u64*>(bitmap.bits);
        t.jmin = rank.words ? rank.min_key : bitmap.min_key;
        t.jrange = rank.words ? rank.range : bitmap.range;
        t.jmask = mask;
        t.jwrap = wrap;
        t.jrank = reinterpret_cast<const pa_u32x4*>(rank.words);
        t.jrank_rows = rank.rows;
        if (unique_keys) {
            hipLaunchKernelGGL(k_join_probe_count_keyed4<false>, grid_for(((int64_t)n_probe + 3) / 4), 256, 0, s, probe_key, n_probe, t, head, counts, flags,
                               (unsigned long long*)total, tile_totals);
        }
        else {  // (the tiles' totals belong to the emission of one-row chains: not here)
            hipLaunchKernelGGL(k_join_probe_count_keyed4<true>, grid_for(((int64_t)n_probe + 3) / 4), 256, 0, s, probe_key, n_probe, t, head, counts, flags,
                               (unsigned long long*)total, (i32*)nullptr);
        }
        PA_HIP(hipGetLastError());
        return unique_keys && tile_totals != nullptr;
    }
    hipLaunchKernelGGL(k_join_probe_count_keyed, grid_for(n_probe), 256, 0, s, probe_key, (const i64*)probe_hash, n_probe, slots, mask, wrap, links, bitmap,
                       rank, head, counts, flags, (unsigned long long*)total);
    PA_HIP(hipGetLastError());
    return false;
}
void launch_join_probe_emit_tiles(const int32_t* head, const int32_t* tile_offsets, int32_t n_probe, int flags, int32_t* probe_idx, int32_t* build_pos,
                                  uint8_t* visited, hipStream_t s)
{
    if (n_probe <= 0) return;
    hipLaunchKernelGGL(k_join_probe_emit_tiles, grid_for(((int64_t)n_probe + 3) / 4), 256, 0, s, head, tile_offsets, n_probe, flags, probe_idx, build_pos, visited);
    PA_HIP(hipGetLastError());
}

// ---- JoinFilterFunction (JoinHash.isJoinPositionEligible, JoinHash.java:116-120; DefaultPageJoiner.joinCurrentPosition, :266-292) ----
// The candidates of a probe page -- every (probe row, build position of its chain), in emission order -- have been through the
// filter; `eligible` holds the indices of those it kept, ascending.  The kernels below turn them into the output pairs.
// keep[i] = 1 for the first eligible candidate of every probe row (outputSingleMatch), else 0
__global__ __launch_bounds__(256) void k_jf_first_of_row(const i32* __restrict__ eligible, i32 ne, const i32* __restrict__ cand_probe, i32* __restrict__ keep)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < ne; i += (i64)gridDim.x * 256) {
        keep[i] = (i == 0 || cand_probe[eligible[i]] != cand_probe[eligible[i - 1]]) ? 1 : 0;
    }
}
// eligible[i] -> out[pos[i]] for the kept ones (pos = exclusive scan of keep)
__global__ __launch_bounds__(256) void k_jf_compact(const i32* __restrict__ eligible, i32 ne, const i32* __restrict__ keep_scan, i32 kept, i32* __restrict__ out)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < ne; i += (i64)gridDim.x * 256) {
        const i32 at = keep_scan[i], next = i + 1 < ne ? keep_scan[i + 1] : kept;
        if (next != at) out[at] = eligible[i];
    }
}
// eligible candidates per probe row
__global__ __launch_bounds__(256) void k_jf_count_rows(const i32* __restrict__ eligible, i32 ne, const i32* __restrict__ cand_probe, i32* __restrict__ per_row)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < ne; i += (i64)gridDim.x * 256) atomicAdd(&per_row[cand_probe[eligible[i]]], 1);
}
// probe-outer: a probe row without an eligible match comes out once, NULL-extended (DefaultPageJoiner.outerJoinCurrentPosition,
// :296-303).  first[r] = exclusive scan of per_row (index of the row's first eligible candidate in `eligible`), at[r] = exclusive
// scan of max(per_row, 1) (the row's first output position)
__global__ __launch_bounds__(256) void k_jf_outer_rows(const i32* __restrict__ per_row, const i32* __restrict__ at, i32 rows, i32* __restrict__ out_probe,
                                                       i32* __restrict__ out_build)
{
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < rows; r += (i64)gridDim.x * 256) {
        if (per_row[r] == 0) {
            out_probe[at[r]] = (i32)r;
            out_build[at[r]] = -1;
        }
    }
}
__global__ __launch_bounds__(256) void k_jf_outer_matches(const i32* __restrict__ eligible, i32 ne, const i32* __restrict__ cand_probe, const i32* __restrict__ cand_build,
                                                          const i32* __restrict__ first, const i32* __restrict__ at, i32* __restrict__ out_probe,
                                                          i32* __restrict__ out_build)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < ne; i += (i64)gridDim.x * 256) {
        const i32 c = eligible[i], r = cand_probe[c];
        const i32 o = at[r] + ((i32)i - first[r]);
        out_probe[o] = r;
        out_build[o] = cand_build[c];
    }
}
__global__ __launch_bounds__(256) void k_jf_max1(const i32* __restrict__ in, i64 n, i32* __restrict__ out)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) out[i] = in[i] > 0 ? in[i] : 1;
}
__global__ __launch_bounds__(256) void k_jf_mark_visited(const i32* __restrict__ build_pos, i64 n, u8* __restrict__ visited)
{
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        if (build_pos[i] >= 0) visited[build_pos[i]] = 1;
    }
}
void launch_jf_first_of_row(const int32_t* eligible, int32_t ne, const int32_t* cand_probe, int32_t* keep, hipStream_t s)
{
    if (ne <= 0) return;
    hipLaunchKernelGGL(k_jf_first_of_row, grid_for(ne), 256, 0, s, eligible, ne, cand_probe, keep);
    PA_HIP(hipGetLastError());
}
void launch_jf_compact(const int32_t* eligible, int32_t ne, const int32_t* keep_scan, int32_t kept, int32_t* out, hipStream_t s)
{
    if (ne <= 0) return;
    hipLaunchKernelGGL(k_jf_compact, grid_for(ne), 256, 0, s, eligible, ne, keep_scan, kept, out);
    PA_HIP(hipGetLastError());
}
void launch_jf_count_rows(const int32_t* eligible, int32_t ne, const int32_t* cand_probe, int32_t* per_row, hipStream_t s)
{
    if (ne <= 0) return;
    hipLaunchKernelGGL(k_jf_count_rows, grid_for(ne), 256, 0, s, eligible, ne, cand_probe, per_row);
    PA_HIP(hipGetLastError());
}
void launch_jf_outer(const int32_t* eligible, int32_t ne, const int32_t* cand_probe, const int32_t* cand_build, const int32_t* per_row, const int32_t* first,
                     const int32_t* at, int32_t rows, int32_t* out_probe, int32_t* out_build, hipStream_t s)
{
    if (rows > 0) hipLaunchKernelGGL(k_jf_outer_rows, grid_for(rows), 256, 0, s, per_row, at, rows, out_probe, out_build);
    if (ne > 0) hipLaunchKernelGGL(k_jf_outer_matches, grid_for(ne), 256, 0, s, eligible, ne, cand_probe, cand_build, first, at, out_probe, out_build);
    PA_HIP(hipGetLastError());
}
void launch_jf_max1(const int32_t* in, int64_t n, int32_t* out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_jf_max1, grid_for(n), 256, 0, s, in, (i64)n, out);
    PA_HIP(hipGetLastError());
}
void launch_jf_mark_visited(const int32_t* build_pos, int64_t n, uint8_t* visited, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_jf_mark_visited, grid_for(n), 256, 0, s, build_pos, (i64)n, visited);
    PA_HIP(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_sum_i32_i64(const i32* __restrict__ v, i64 n, unsigned long long* __restrict__ out)
{
    __shared__ i64 part[4];
    i64 acc = 0;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) acc += v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    // one atomic per workgroup: atomics on ONE address retire at about 10 ns each on this part (they execute at the memory side), so
    // the 4096 of a one-per-wave reduction were 40 of this kernel's 65 us over 13 M counts
    if (threadIdx.x == 0 && (acc = part[0] + part[1] + part[2] + part[3]) != 0) atomicAdd(out, (unsigned long long)acc);
}
void launch_sum_i32_i64(const int32_t* v, int64_t n, int64_t* out, hipStream_t s)
{
    PA_HIP(hipMemsetAsync(out, 0, 8, s));
    if (n <= 0) return;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 512));
    hipLaunchKernelGGL(k_sum_i32_i64, grid, 256, 0, s, v, (i64)n, reinterpret_cast<unsigned long long*>(out));
    PA_HIP(hipGetLastError());
}

}  // namespace pa
