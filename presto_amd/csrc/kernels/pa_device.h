// pa_device.h -- device-side building blocks for gfx950 (CDNA4, wave64) shared by the statically
// compiled kernels and by the kernels generated per query (exprgen.cpp + hiprtc).
//
// Everything here is integer / f64 streaming work bounded by HBM: no MFMA.  Wave width is
// hard-coded to 64 (cdna_hip_programming.md section 1).
#pragma once

typedef long long i64;
typedef unsigned long long u64;
typedef int i32;
typedef unsigned int u32;
typedef unsigned char u8;

#define PA_WAVE 64

// pa_status values raised from device code (include/presto_amd.h)
#define PA_DEV_ERR_OUT_OF_RANGE (-4)
#define PA_DEV_ERR_DIV_ZERO (-5)
#define PA_DEV_ERR_RESOURCES (-6)

typedef i32 pa_i32x4 __attribute__((ext_vector_type(4)));
typedef double pa_f64x2 __attribute__((ext_vector_type(2)));
typedef float pa_f32x4 __attribute__((ext_vector_type(4)));
typedef i64 pa_i64x2 __attribute__((ext_vector_type(2)));
typedef u32 pa_u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// hash arithmetic -- bit-exact with the reference
// ---------------------------------------------------------------------------------------------
#define PA_P1 0x9E3779B185EBCA87ULL
#define PA_P2 0xC2B2AE3D27D4EB4FULL
#define PA_P3 0x165667B19E3779F9ULL
#define PA_P4 0x85EBCA77C2B2AE63ULL
#define PA_P5 0x27D4EB2F165667C5ULL

__device__ __forceinline__ u64 pa_rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }

// AbstractLongType.hash (core/trino-spi/.../type/AbstractLongType.java:126-130)
__device__ __forceinline__ i64 pa_hash_bigint(i64 v) { return (i64)(pa_rotl64((u64)v * PA_P2, 31) * PA_P1); }
// DoubleType.hashCodeOperator (core/trino-spi/.../type/DoubleType.java:163-170)
__device__ __forceinline__ i64 pa_hash_double(double v)
{
    if (v == 0) v = 0;
    i64 bits = (v != v) ? 0x7ff8000000000000LL : __double_as_longlong(v);
    return pa_hash_bigint(bits);
}
// RealType.hashCodeOperator (core/trino-spi/.../type/RealType.java:107-115): hash(floatToIntBits(v == 0 ? 0 : v)); the canonical bits
// (+0 for both zeros, one NaN) are also what a REAL group key is compared by (IS NOT DISTINCT FROM, RealType.java:127-140)
__device__ __forceinline__ u32 pa_real_key_bits(float v) { return v == 0.0f ? 0u : ((v != v) ? 0x7fc00000u : __float_as_uint(v)); }
__device__ __forceinline__ i64 pa_hash_real(float v) { return pa_hash_bigint((i64)(i32)pa_real_key_bits(v)); }
// fastutil HashCommon.murmurHash3 == PagesHash.getHashPosition mix (…/operator/join/PagesHash.java:225-241)
__device__ __forceinline__ u64 pa_murmur3_fmix(u64 h)
{
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return h;
}
// CombineHashFunction.getHash (…/operator/scalar/CombineHashFunction.java:26-29)
__device__ __forceinline__ i64 pa_combine_hash(i64 prev, i64 v) { return (i64)(31ULL * (u64)prev + (u64)v); }

__device__ __forceinline__ u64 pa_xxh_round(u64 acc, u64 in) { return pa_rotl64(acc + in * PA_P2, 31) * PA_P1; }
__device__ __forceinline__ u64 pa_xxh_merge(u64 h, u64 v) { return (h ^ pa_xxh_round(0, v)) * PA_P1 + PA_P4; }
// little-endian words of a byte string at any alignment: one unaligned global load each (gfx950 allows them), not 8 / 4 byte loads
__device__ __forceinline__ u64 pa_rd64(const u8* p)
{
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ u32 pa_rd32(const u8* p)
{
    u32 v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ u64 pa_xxh_avalanche(u64 h)
{
    h ^= h >> 33;
    h *= PA_P2;
    h ^= h >> 29;
    h *= PA_P3;
    h ^= h >> 32;
    return h;
}
// XXH64 seed 0 of a byte string == io.airlift.slice.XxHash64.hash(Slice) (VARCHAR hash,
// core/trino-spi/.../block/AbstractVariableWidthBlock.java:92-96)
__device__ inline u64 pa_xxh64(const u8* p, i32 len)
{
    const u8* end = p + len;
    u64 h;
    if (len >= 32) {
        u64 v1 = PA_P1 + PA_P2, v2 = PA_P2, v3 = 0, v4 = 0ULL - PA_P1;
        do {
            v1 = pa_xxh_round(v1, pa_rd64(p));
            v2 = pa_xxh_round(v2, pa_rd64(p + 8));
            v3 = pa_xxh_round(v3, pa_rd64(p + 16));
            v4 = pa_xxh_round(v4, pa_rd64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = pa_rotl64(v1, 1) + pa_rotl64(v2, 7) + pa_rotl64(v3, 12) + pa_rotl64(v4, 18);
        h = pa_xxh_merge(h, v1);
        h = pa_xxh_merge(h, v2);
        h = pa_xxh_merge(h, v3);
        h = pa_xxh_merge(h, v4);
    }
    else {
        h = PA_P5;
    }
    h += (u64)len;
    while (p + 8 <= end) {
        h ^= pa_xxh_round(0, pa_rd64(p));
        h = pa_rotl64(h, 27) * PA_P1 + PA_P4;
        p += 8;
    }
    if (p + 4 <= end) {
        h ^= (u64)pa_rd32(p) * PA_P1;
        h = pa_rotl64(h, 23) * PA_P2 + PA_P3;
        p += 4;
    }
    while (p < end) {
        h ^= (u64)(*p) * PA_P5;
        h = pa_rotl64(h, 11) * PA_P1;
        p++;
    }
    return pa_xxh_avalanche(h);
}
// XxHash64.hash(long): XXH64 of the 8 little-endian bytes
__device__ __forceinline__ u64 pa_xxh64_long(u64 v)
{
    u64 h = PA_P5 + 8;
    h ^= pa_xxh_round(0, v);
    h = pa_rotl64(h, 27) * PA_P1 + PA_P4;
    return pa_xxh_avalanche(h);
}

// ---------------------------------------------------------------------------------------------
// VARCHAR helpers (Slice.equals / Slice.compareTo: unsigned bytes, then length)
// ---------------------------------------------------------------------------------------------
__device__ inline bool pa_str_eq(const u8* a, i32 alen, const u8* b, i32 blen)
{
    if (alen != blen) return false;
    i32 i = 0;
    for (; i + 8 <= alen; i += 8)
        if (pa_rd64(a + i) != pa_rd64(b + i)) return false;
    for (; i < alen; i++)
        if (a[i] != b[i]) return false;
    return true;
}
__device__ inline int pa_str_cmp(const u8* a, i32 alen, const u8* b, i32 blen)
{
    i32 n = alen < blen ? alen : blen;
    for (i32 i = 0; i < n; i++) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return (alen > blen) - (alen < blen);
}

// ---------------------------------------------------------------------------------------------
// exact integer arithmetic (BigintOperators.java:47-121 / IntegerOperators.java): overflow -> *err
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pa_raise(i32* err, i32 code)
{
    if (*err == 0) atomicCAS(err, 0, code);
}
__device__ __forceinline__ i64 pa_add_exact(i64 a, i64 b, i32* err)
{
    i64 r;
    if (__builtin_add_overflow(a, b, &r)) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return r;
}
__device__ __forceinline__ i64 pa_sub_exact(i64 a, i64 b, i32* err)
{
    i64 r;
    if (__builtin_sub_overflow(a, b, &r)) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return r;
}
__device__ __forceinline__ i64 pa_mul_exact(i64 a, i64 b, i32* err)
{
    i64 r;
    if (__builtin_mul_overflow(a, b, &r)) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return r;
}
__device__ __forceinline__ i64 pa_div_exact(i64 a, i64 b, i32* err)
{
    if (b == 0) { pa_raise(err, PA_DEV_ERR_DIV_ZERO); return 0; }
    if (a == (-9223372036854775807LL - 1) && b == -1) { pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE); return 0; }
    return a / b;
}
__device__ __forceinline__ i64 pa_mod_exact(i64 a, i64 b, i32* err)
{
    if (b == 0) { pa_raise(err, PA_DEV_ERR_DIV_ZERO); return 0; }
    if (b == -1) return 0;
    return a % b;
}
__device__ __forceinline__ i64 pa_int_range(i64 v, i32* err)
{
    if (v > 2147483647LL || v < -2147483648LL) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return v;
}

// ---------------------------------------------------------------------------------------------
// wave64 / workgroup reductions in a fixed order (bitwise reproducible run to run)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double pa_wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ i64 pa_wave_sum_i64(i64 v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ u64 pa_wave_max_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u64 o = ((u64)(u32)__shfl_xor((int)(u32)(v >> 32), off, 64) << 32) | (u64)(u32)__shfl_xor((int)(u32)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ i64 pa_wave_sum_i64_exact(i64 v, i32* err)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        i64 o = __shfl_xor(v, off, 64);
        i64 r;
        if (__builtin_add_overflow(v, o, &r)) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
        v = r;
    }
    return v;
}
__device__ __forceinline__ double pa_wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { double o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ double pa_wave_max_f64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { double o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ i64 pa_wave_min_i64(i64 v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { i64 o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ i64 pa_wave_max_i64(i64 v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { i64 o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// exclusive prefix count of `pred` among the lanes of the wave below this one, plus the wave total
__device__ __forceinline__ u32 pa_wave_prefix(bool pred, u32* total)
{
    u64 mask = __ballot(pred);
    u32 lane = threadIdx.x & 63;
    *total = (u32)__popcll(mask);
    return (u32)__popcll(mask & ((1ULL << lane) - 1ULL));
}

// packed-key hashing for the group tables (any good mix; not part of results parity)
// (the halves are folded AFTER a 64-bit multiply: folding the key's own halves made every pair of packed keys (a, b), (a ^ d, b ^ d)
// collide in all 32 bits -- (day, priority) = (9172, 1) and (9173, 0) -- and equal hashes are equal tags in the tables)
__device__ __forceinline__ u32 pa_mix32(u64 k)
{
    k *= 0x9E3779B97F4A7C15ULL;
    u32 x = (u32)k ^ (u32)(k >> 32);
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}

// ---------------------------------------------------------------------------------------------
// Group tables.  Keys are packed into W 64-bit words (exprgen); equality is word equality.
// ---------------------------------------------------------------------------------------------
#define PA_MAX_CHANNELS 32
#define PA_MAX_BUILD_CHANNELS 8

// Kernel argument block of the fused scan-filter-project-aggregate kernels (op_fused.cpp keeps the
// host mirror `FusedArgs` in sync).
struct PaFusedArgs {
    const void* v[PA_MAX_CHANNELS];   // column values (FLAT) / bytes (VARWIDTH)
    const i32* o[PA_MAX_CHANNELS];    // VARWIDTH offsets
    const u8* nl[PA_MAX_CHANNELS];    // nulls (1 B / row) or nullptr
    i64 n;                            // rows in this page
    i32 vec;                          // 1: every used buffer is 16-B aligned -> 4 rows / lane / step
    i32 pad;
    u64* slab;                        // per-workgroup partial states of this launch
    u64* gt_tag;                      // global group table: 0 empty | hash<<2|1 busy | hash<<2|3 ready
    u64* gt_keys;                     // [capacity][W]
    u64* gt_words;                    // [NW][capacity] accumulator words
    u32 gt_mask;                      // capacity - 1
    i32 gt_max_fill;
    i32* gt_count;                    // groups in the table
    i32* err;                         // first device-raised pa_status
    u64* overflow_rows;               // rows that missed the LDS table (drives mode escalation)
    const i32* row_list;              // GT variant: process these rows (of this launch's range) instead of all n
    i64 n_list;
    i32* spill_rows;                  // GT variant: rows whose group did not fit the table (redone after a rehash)
    u32* spill_count;
    // Replicas of the group table: workgroup b works on replica b & gt_rep_mask (each a full table of gt_mask + 1 slots,
    // laid out one after the other in gt_tag / gt_keys / gt_words).  Atomics of many rows on few addresses retire at
    // ~0.15 M/s per address on this part; R replicas divide the pressure per address by R.  The host folds the
    // replicas into one table before the result is read.
    u32 gt_rep_mask;
    u32 part_mask;                    // hash-partitioning pass: partitions - 1 (partition id part_mask + 1 = row filtered out)
    i32* gt_rep_count;                // groups of replica r >= 1 at [r]; replica 0 counts in gt_count
    i32* part_ids;                    // hash-partitioning pass: partition of every row of the launch
    i32 list_blocked;                 // row_list is cut into one contiguous slice per workgroup (partition-ordered lists)
    i32 pad3;
    // Partition-owned tables (LDSP variant): workgroup p aggregates the rows [part_first[p], part_first[p + 1]) of the
    // partition-ordered columns into an LDS table it LOADS from and STORES to sub_*[p] -- the groups of partition p live nowhere
    // else, so neither the row loop nor the hand-over to HBM needs an atomic on HBM.  Layout per partition: PA_LC tags,
    // PA_LC x PA_KW key words, PA_LC x PA_NW accumulator words (slot-major), one group count.
    u64* sub_tag;
    u64* sub_keys;
    u64* sub_words;
    i32* sub_count;
    const i64* part_first;
    // Probe stage (fused FilterAndProject -> LookupJoin -> aggregation over a lookup source with ONE integer key and no
    // duplicate keys): the keyed probe-side table of join_kernels.hpp (16 B slots: key, head, next), the build side's key
    // existence bitmap (null = none) and the build columns the aggregation reads, indexed by build position.
    const void* jslots;
    const u64* jbits;
    i64 jmin;
    u64 jrange;
    u32 jmask;
    i32 jrows;                        // build positions (the build-row table of the BROW variant has this many slots)
    u32 jwrap;                        // probe sequences wrap inside (pos & ~jwrap): jmask, or the partition size - 1 of a partitioned build
    u32 jpad;
    const void* bv[PA_MAX_BUILD_CHANNELS];
    const u8* bn[PA_MAX_BUILD_CHANNELS];
    // key rank index (join_kernels.hpp JoinRankIndex; null = none, the slot table answers): 16-byte words {64 key bits, build keys
    // below the word} over [jmin, jmin + jrange], and rank -> build position (null: the rank is the position)
    const pa_u32x4* jrank;
    const i32* jrank_rows;
    // pa_fused_ranges: a table of row ranges taken in place (entry layout: op_fused.cpp range_entry_words)
    const u64* ranges;
    i64 n_ranges;
};

// JoinProbe.getCurrentJoinPosition for a keyed lookup source without duplicate keys (…/operator/join/JoinProbe.java:87-117,
// PagesHash.getAddressIndex, PagesHash.java:158-170): the build position of key `v`, or -1.  Same walk as
// k_join_probe_count_keyed (join_kernels.hip): bitmap first -- most probe rows of a selective join miss, and a clustered probe
// side reads the bitmap almost sequentially -- then linear probing over 64-byte lines of four slots.
// (A = PaFusedArgs or PaFpArgs: both carry the probe-side table as jslots / jmask / jwrap / jbits / jmin / jrange)
// (rows_of_key: the rows of the key on the build side -- the slot's chain length, join_kernels.hpp JoinKeySlot::count -- when found)
template <class A>
__device__ __forceinline__ i32 pa_join_probe_from(const A& a, const u64 v, u32 pos, i32* rows_of_key = nullptr)
{
    const pa_u32x4* lines = (const pa_u32x4*)a.jslots;
    for (u32 seen = 0; seen <= a.jwrap;) {
        const u32 base = pos & ~3u, first = pos & 3u;
        pa_u32x4 q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = lines[base + k];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((u32)k < first) continue;
            const i32 cur = (i32)q[k].z;
            if (cur == -1) return -1;
            if ((((u64)q[k].y << 32) | (u64)q[k].x) == v) {
                if (rows_of_key) *rows_of_key = (i32)q[k].w;
                return cur;
            }
        }
        seen += 4u - first;
        pos = (pos & ~a.jwrap) | ((base + 4u) & a.jwrap);
    }
    return -1;
}
template <class A>
__device__ __forceinline__ u32 pa_join_home(const A& a, const u64 v) { return (u32)pa_murmur3_fmix((u64)pa_hash_bigint((i64)v)) & a.jmask; }
// rank of the key at distance d from the bitmap's first key, from its rank word; -1 = not a build key
__device__ __forceinline__ i32 pa_join_rank(const pa_u32x4 w, const u64 d)
{
    const u64 bits = ((u64)w.y << 32) | (u64)w.x;
    const u32 b = (u32)(d & 63ULL);
    if (((bits >> b) & 1ULL) == 0ULL) return -1;
    return (i32)w.z + (i32)__popcll(bits & ((1ULL << b) - 1ULL));
}
template <class A>
__device__ __forceinline__ i32 pa_join_probe_keyed(const A& a, const u64 v)
{
    if (a.jrank) {  // the key's rank among the build keys names its build row (no duplicate keys)
        const u64 d = (u64)((i64)v - a.jmin);
        if (d > a.jrange) return -1;
        const i32 r = pa_join_rank(a.jrank[d >> 6], d);
        return (r >= 0 && a.jrank_rows) ? a.jrank_rows[r] : r;
    }
    if (a.jbits) {
        const u64 d = (u64)((i64)v - a.jmin);
        if (d > a.jrange || ((a.jbits[d >> 6] >> (d & 63ULL)) & 1ULL) == 0ULL) return -1;
    }
    return pa_join_probe_from(a, v, pa_join_home(a, v));
}
// The key rank index in two halves, for row loops that keep several quads in flight: pa_join_rank4_issue asks for the rank words of
// the four keys and pa_join_rank4_read, an iteration later, turns them into build positions.  The loads are UNCONDITIONAL (a row that
// does not probe reads word 0): a load inside a divergent branch may or may not have been issued, so the compiler could no longer count
// how many younger loads may stay in flight when it waits for an older one, and would wait for all of them.
struct PaRank4 {
    pa_u32x4 lo, hi, below;   // per row of the quad: the 64 key bits of the word and the build keys below it
                              // (vectors, not arrays: the words live in registers across the loop's back edge)
};
template <class A>
__device__ __forceinline__ void pa_join_rank4_issue(const A& a, const bool (&s)[4], const u64 (&k)[4], PaRank4& w)
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const bool dup = r > 0 && s[r] && s[r - 1] && k[r] == k[r - 1];
        const u64 d = (u64)((i64)k[r] - a.jmin);
        const u32* p = (const u32*)&a.jrank[(s[r] && !dup && d <= a.jrange) ? d >> 6 : 0ULL];
        w.lo[r] = p[0];
        w.hi[r] = p[1];
        w.below[r] = p[2];
    }
}
template <class A>
__device__ __forceinline__ void pa_join_rank4_read(const A& a, const bool (&s)[4], const u64 (&k)[4], const PaRank4& w, i32 (&jb)[4])
{
    bool dup[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        dup[r] = r > 0 && s[r] && s[r - 1] && k[r] == k[r - 1];
        const u64 d = (u64)((i64)k[r] - a.jmin);
        jb[r] = (s[r] && !dup[r] && d <= a.jrange) ? pa_join_rank(pa_u32x4{w.lo[r], w.hi[r], w.below[r], 0u}, d) : -1;
    }
    if (a.jrank_rows) {
        i32 p[4];
#pragma unroll
        for (int r = 0; r < 4; r++) p[r] = jb[r] >= 0 ? a.jrank_rows[jb[r]] : -1;
#pragma unroll
        for (int r = 0; r < 4; r++) jb[r] = p[r];
    }
#pragma unroll
    for (int r = 1; r < 4; r++) {
        if (dup[r]) jb[r] = jb[r - 1];
    }
}
// The same for the four consecutive rows a thread of the vector loops holds, stage by stage: the four bitmap words, then the four
// pairs of slots, are loaded back to back and waited for once.  Row by row, a row costs two to three DEPENDENT trips to HBM
// (bitmap word, slot, then the columns only matches read), and a wave of the row loop is then bound by latency, not bandwidth:
// Q3's lineitem pages ran at 1.7 ms per 2^28 rows that way, three times the time of the filter alone.  A key equal to its
// predecessor's (a probe side clustered by the key) reuses the predecessor's answer.  s[r]: row r probes; k[r]: its key.
// kCount: jc[r] = the rows of the key on the build side (1 over a key rank index: it stands on unique keys), 0 without a match
template <bool kCount, class A>
__device__ __forceinline__ void pa_join_probe4x(const A& a, const bool (&s)[4], const u64 (&k)[4], i32 (&jb)[4], i32 (&jc)[4])
{
    bool dup[4], need[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        dup[r] = r > 0 && s[r] && s[r - 1] && k[r] == k[r - 1];
        need[r] = s[r] && !dup[r];
    }
    if (a.jrank) {  // one 16-byte load per row: does the key exist, and its rank = its build row
        u64 d[4];
        pa_u32x4 w[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            d[r] = (u64)((i64)k[r] - a.jmin);
            need[r] = need[r] && d[r] <= a.jrange;
            w[r] = pa_u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (need[r]) w[r] = a.jrank[d[r] >> 6];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) jb[r] = need[r] ? pa_join_rank(w[r], d[r]) : -1;
        if (a.jrank_rows) {
            i32 p[4];
#pragma unroll
            for (int r = 0; r < 4; r++) p[r] = jb[r] >= 0 ? a.jrank_rows[jb[r]] : -1;
#pragma unroll
            for (int r = 0; r < 4; r++) jb[r] = p[r];
        }
#pragma unroll
        for (int r = 1; r < 4; r++) {
            if (dup[r]) jb[r] = jb[r - 1];
        }
        if (kCount) {
#pragma unroll
            for (int r = 0; r < 4; r++) jc[r] = jb[r] >= 0 ? 1 : 0;
        }
        return;
    }
    if (a.jbits) {
        u64 d[4], w[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            d[r] = (u64)((i64)k[r] - a.jmin);
            need[r] = need[r] && d[r] <= a.jrange;
            w[r] = 0ULL;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (need[r]) w[r] = a.jbits[d[r] >> 6];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) need[r] = need[r] && ((w[r] >> (d[r] & 63ULL)) & 1ULL) != 0ULL;
    }
    const pa_u32x4* slots = (const pa_u32x4*)a.jslots;
    u32 pos[4];
    pa_u32x4 s0[4], s1[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        pos[r] = pa_join_home(a, k[r]);
        s0[r] = pa_u32x4{0u, 0u, 0xffffffffu, 0u};
        s1[r] = s0[r];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (need[r]) {
            s0[r] = slots[pos[r]];
            s1[r] = slots[(pos[r] & ~a.jwrap) | ((pos[r] + 1u) & a.jwrap)];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        i32 res = -1, rows = 0;
        if (need[r] && (i32)s0[r].z != -1) {
            if ((((u64)s0[r].y << 32) | (u64)s0[r].x) == k[r]) {
                res = (i32)s0[r].z;
                rows = (i32)s0[r].w;
            }
            else if ((i32)s1[r].z != -1) {
                if ((((u64)s1[r].y << 32) | (u64)s1[r].x) == k[r]) {
                    res = (i32)s1[r].z;
                    rows = (i32)s1[r].w;
                }
                else res = pa_join_probe_from(a, k[r], (pos[r] & ~a.jwrap) | ((pos[r] + 2u) & a.jwrap), kCount ? &rows : nullptr);  // a probe sequence longer than two slots
            }
        }
        jb[r] = res;
        if (kCount) jc[r] = res >= 0 ? rows : 0;
    }
#pragma unroll
    for (int r = 1; r < 4; r++) {
        if (dup[r]) {
            jb[r] = jb[r - 1];
            if (kCount) jc[r] = jc[r - 1];
        }
    }
}
template <class A>
__device__ __forceinline__ void pa_join_probe4(const A& a, const bool (&s)[4], const u64 (&k)[4], i32 (&jb)[4])
{
    i32 unused[4];
    pa_join_probe4x<false>(a, s, k, jb, unused);
}

// Does the key find a build row at all?  The key bitmap holds exactly the non-NULL keys of the build side (every one sets its bit,
// the range is their min .. max), so where it exists it IS the answer and the slot table is not touched: a join whose output carries
// no build column -- and the counting pass of any join -- needs no build position.
template <class A>
__device__ __forceinline__ bool pa_join_exists_keyed(const A& a, const u64 v)
{
    if (a.jbits) {
        const u64 d = (u64)((i64)v - a.jmin);
        return d <= a.jrange && ((a.jbits[d >> 6] >> (d & 63ULL)) & 1ULL) != 0ULL;
    }
    return pa_join_probe_from(a, v, pa_join_home(a, v)) >= 0;
}
template <class A>
__device__ __forceinline__ void pa_join_exists4(const A& a, const bool (&s)[4], const u64 (&k)[4], bool (&hit)[4])
{
    if (a.jbits) {
        u64 d[4], w[4];
        bool need[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            d[r] = (u64)((i64)k[r] - a.jmin);
            need[r] = s[r] && d[r] <= a.jrange;
            w[r] = 0ULL;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (need[r]) w[r] = a.jbits[d[r] >> 6];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) hit[r] = need[r] && ((w[r] >> (d[r] & 63ULL)) & 1ULL) != 0ULL;
        return;
    }
    i32 jb[4];
    pa_join_probe4(a, s, k, jb);
#pragma unroll
    for (int r = 0; r < 4; r++) hit[r] = jb[r] >= 0;
}

// the replica of the group table this workgroup works on
struct PaGtView {
    u64* tag;
    u64* keys;
    u64* words;
    i32* count;
};
__device__ __forceinline__ PaGtView pa_gt_view(const PaFusedArgs& a, const int W, const int NW)
{
    const u32 r = blockIdx.x & a.gt_rep_mask;
    const u64 cap = (u64)a.gt_mask + 1ULL;
    PaGtView v;
    v.tag = a.gt_tag + (u64)r * cap;
    v.keys = a.gt_keys + (u64)r * cap * (u64)W;
    v.words = a.gt_words + (u64)r * cap * (u64)NW;
    v.count = r ? a.gt_rep_count + r : a.gt_count;
    return v;
}

// 64-bit value of lane `lane` (wave-uniform) broadcast to the wave
__device__ __forceinline__ u64 pa_readlane_u64(u64 v, int lane)
{
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
    u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
    return ((u64)hi << 32) | (u64)lo;
}

// order-preserving image of a string of at most 7 bytes: its bytes from the top of the word down, the length in the lowest byte
// (Slice.compareTo: unsigned bytes, then length -- a proper prefix sorts first: equal upper bytes, smaller length)
__device__ __forceinline__ u64 pa_img_str7(const u8* p, i32 len, i32* err)
{
    if (len > 7) {
        pa_raise(err, -3);
        len = 7;
    }
    u64 v = (u64)len;
    for (i32 b = 0; b < len; b++) v |= (u64)p[b] << (56 - 8 * b);
    return v;
}

// bytes of a short VARCHAR (declared bound <= 7) packed little-endian into one word
__device__ __forceinline__ u64 pa_short_bytes(const u8* p, i32 len, i32 bound, i32* err)
{
    if (len > bound) {
        pa_raise(err, -3);
        len = bound;
    }
    u64 v = 0;
    for (i32 b = 0; b < len; b++) v |= (u64)p[b] << (8 * b);
    return v;
}

// Global (HBM) group table upsert.  Table memory is only ever touched with agent-scope atomic
// accesses (sc1: coherent across the 8 XCD L2s); key words are published by: sc1 stores ->
// s_waitcnt vmcnt(0) -> sc1 tag store (MI355X_MICROARCH.md "handoff-flag", drained-sc1 form).
// A lane that finds the slot busy with its own hash re-polls it in the next loop iteration; the
// claiming lane finishes its publication inside the iteration of the claim, so lanes of one wave
// cannot wait on each other.
// Group counting without a hot address: a kernel reads the table's group count once (base), every wave keeps its own
// tally of the slots it claimed and adds it to the global counter once, at the end of the kernel (pa_gt_ctr_flush).
// The fill check inside a launch uses the estimate base + (claims this lane has seen in its wave) x (waves of the
// grid): loads and atomics on ONE address from every insert retire at well under 1 G/s on this part and used to bound
// launches that create millions of groups.  The estimate only decides when rows start to be spilled; a probe-length
// bound backs it up, and the exact count is known to the host after every launch.
#define PA_GT_KEY_CLEAR 0xA5A5A5A5A5A5A5A5ULL  // what hipMemset(0xA5) leaves in the key words of a new table
struct PaGtCtr {
    i32 base;   // groups in the table when the kernel started
    i32 scale;  // waves in the grid
    i32 seen;   // claims made by this wave while this lane was searching
    i32 mine;   // claims this lane accounts for (each claim is counted by exactly one lane)
    u32 max_probes;
};
// spilling = true: the caller can spill a row that finds no room (fill estimate and probe bound apply);
// false: the host sized the table for everything the kernel inserts (merge, rehash): only a full table fails.
__device__ __forceinline__ PaGtCtr pa_gt_ctr_init(const i32* count, bool spilling, u32 replicas = 1)
{
    PaGtCtr c;
    c.base = *count;
    c.scale = spilling ? (i32)((gridDim.x * (blockDim.x >> 6) + replicas - 1) / replicas) : 0;  // waves working on this table
    c.seen = 0;
    c.mine = 0;
    c.max_probes = spilling ? 512u : 0xffffffffu;
    return c;
}
__device__ __forceinline__ int pa_gt_upsert_n(u64* tag, u64* keys, u32 mask, u32 h, const u64* k, const int W, PaGtCtr& ctr,
                                              i32 max_fill, i32* err)
{
    const u64 busy = ((u64)h << 2) | 1ULL, ready = ((u64)h << 2) | 3ULL;
    u32 i = h & mask;
    u32 probes = 0;
    int spins = 0;
    int result = -2;  // -2 = still searching, -1 = no room, >= 0 = slot
    // Fast path for groups that already exist (the common case once a table is warm): ordinary cached loads.
    // Tags only ever go empty -> busy -> ready and key words are written once, before the tag turns ready, so a
    // (possibly stale) cached view can at worst miss a group that exists -- never match a wrong one -- PROVIDED the key words
    // of an unclaimed slot cannot look like a key: the key arrays are cleared to PA_GT_KEY_CLEAR when a table is made (a
    // recycled block holds the keys of its previous life: a stale line then matched a key whose hash collided with the
    // slot's new owner, and its rows went to that group), and a key that IS the clear pattern skips this path.  Hot groups
    // are then served from L1/L2 instead of hammering one memory channel with device-scope loads.
    bool clear_pattern = true;
    for (int w = 0; w < W; w++) clear_pattern = clear_pattern && k[w] == PA_GT_KEY_CLEAR;
    if (!clear_pattern) {
        u32 j = i;
        for (int probe = 0; probe < 8; probe++) {
            const u64 t = tag[j];
            if (t == 0ULL) break;
            if (t == ready) {
                bool eq = true;
                for (int w = 0; w < W; w++) eq = eq && (keys[(u64)j * W + w] == k[w]);
                if (eq) return (int)j;
            }
            j = (j + 1) & mask;
        }
    }
    // Wave-uniform loop: every lane of the wave stays in the loop until all of them are done, so a lane that
    // claims a slot publishes its key INSIDE the iteration of the claim.  (With an early `return` the compiler
    // may sink the publication behind the loop, and lanes of the same wave waiting for that very slot would
    // spin until their bound: SIMT forward-progress hazard.)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    while (__ballot(result == -2) != 0ULL) {
        const bool searching = result == -2;
        u64 t = 1ULL;
        if (searching) t = __hip_atomic_load(&tag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const i32 filled = ctr.base + ctr.seen * ctr.scale;
        bool claimed = false;
        if (searching) {
            bool advance = false;
            if (t == 0ULL) {
                // a new group: refuse it once the table holds max_fill groups (the caller spills the row and the
                // host rehashes into a larger table); concurrent inserts may overshoot by the lanes in flight
                if (filled >= max_fill) {
                    result = -1;
                }
                else {
                    u64 old = atomicCAS(&tag[i], 0ULL, busy);
                    if (old == 0ULL) {
                        for (int w = 0; w < W; w++)
                            __hip_atomic_store(&keys[(u64)i * W + w], k[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(&tag[i], ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        claimed = true;
                        result = (int)i;
                    }
                    // else: someone else claimed it between the load and the CAS: look at it again next iteration
                }
            }
            else if ((t | 2ULL) == ready) {
                if (t == busy) {
                    // being published by another lane (possibly of this wave, which finishes inside its iteration)
                    if (++spins > (1 << 20)) result = -1;  // bounded: a stuck publisher surfaces as a spilled row
                }
                else {
                    bool eq = true;
                    for (int w = 0; w < W; w++)
                        eq = eq && (__hip_atomic_load(&keys[(u64)i * W + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k[w]);
                    if (eq) result = (int)i;
                    else advance = true;
                }
            }
            else {
                advance = true;
            }
            if (advance) {
                i = (i + 1) & mask;
                if (++probes > ctr.max_probes || probes > mask) result = -1;  // a cluster this long means the table is (nearly) full
            }
        }
        const u64 claims = __ballot(claimed);
        if (claims != 0ULL) {
            const i32 c = (i32)__popcll(claims);
            ctr.seen += c;
            if (lane == __ffsll((long long)claims) - 1) ctr.mine += c;
        }
    }
    return result;
}
template <int W>
__device__ __forceinline__ int pa_gt_upsert(u64* tag, u64* keys, u32 mask, u32 h, const u64 (&k)[W], PaGtCtr& ctr,
                                            i32 max_fill, i32* err)
{
    return pa_gt_upsert_n(tag, keys, mask, h, k, W, ctr, max_fill, err);
}

// end of kernel, all lanes of the wave: one atomic per wave publishes its claims
__device__ __forceinline__ void pa_gt_ctr_flush(const PaGtCtr& ctr, i32* count)
{
    const i64 total = pa_wave_sum_i64((i64)ctr.mine);
    if ((threadIdx.x & 63) == 0 && total != 0) atomicAdd(count, (i32)total);
}

// hash of a packed key (identical in the generated kernels and in the merge / rehash kernels)
__device__ __forceinline__ u32 pa_key_hash(const u64* k, const int W)
{
    u64 hk = k[0];
    for (int i = 1; i < W; i++) hk = (hk ^ (hk >> 29)) * 0x9E3779B97F4A7C15ULL + k[i];
    return pa_mix32(hk);
}

// accumulator word kinds (op_fused.cpp)
#define PA_W_CNT 0
#define PA_W_SUMF 1
#define PA_W_SUMI 2
#define PA_W_MAXU 3   // min / max: unsigned maximum of an order-preserving image of the value (identity 0)

// Order-preserving 64-bit images (a < b in the type's COMPARISON order <=> image(a) < image(b)): min / max become one
// unsigned maximum (min over x = max over ~image(x)) whose identity is the zero the tables are cleared to.
__device__ __forceinline__ u64 pa_img_i64(i64 v) { return (u64)v ^ 0x8000000000000000ULL; }
__device__ __forceinline__ u64 pa_img_f64(double d)
{
    // Double.compare (DoubleType.java:194-198): -0.0 < 0.0, one NaN above everything
    const u64 b = d != d ? 0x7ff8000000000000ULL : (u64)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ i64 pa_unimg_i64(u64 img) { return (i64)(img ^ 0x8000000000000000ULL); }
__device__ __forceinline__ u64 pa_unimg_f64_bits(u64 img) { return (img >> 63) ? (img & 0x7fffffffffffffffULL) : ~img; }

__device__ __forceinline__ void pa_gt_add_f64(u64* words, u64 idx, double v)
{
    unsafeAtomicAdd((double*)&words[idx], v);  // global_atomic_add_f64
}
__device__ __forceinline__ void pa_gt_add_u64(u64* words, u64 idx, u64 v)
{
    atomicAdd((unsigned long long*)&words[idx], (unsigned long long)v);
}
__device__ __forceinline__ void pa_gt_max_u64(u64* words, u64 idx, u64 v)
{
    atomicMax((unsigned long long*)&words[idx], (unsigned long long)v);
}
__device__ __forceinline__ void pa_gt_add_i64_exact(u64* words, u64 idx, i64 v, i32* err)
{
    i64 old = (i64)atomicAdd((unsigned long long*)&words[idx], (unsigned long long)v);
    i64 r;
    if (__builtin_add_overflow(old, v, &r)) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
}

// ---------------------------------------------------------------------------------------------
// DECIMAL (ShortDecimalType: an unscaled i64; LongDecimalType: 16 bytes, the low 64 bits of the magnitude then the high 63
// bits with the sign in the top bit -- UnscaledDecimal128Arithmetic.java).  Inside expressions a long decimal is a two's
// complement i128; no 128-bit division happens on the device.
// ---------------------------------------------------------------------------------------------
typedef __int128 i128;
typedef unsigned __int128 u128;
__device__ __forceinline__ i128 pa_i128_of(u64 hi, u64 lo) { return (i128)(((u128)hi << 64) | (u128)lo); }
__device__ __forceinline__ i128 pa_ld_from(i64 lo, i64 hi)   // the block's two longs -> value
{
    const u128 mag = ((u128)((u64)hi & 0x7fffffffffffffffULL) << 64) | (u128)(u64)lo;
    return hi < 0 ? -(i128)mag : (i128)mag;
}
__device__ __forceinline__ i128 pa_ld_read(const u64* p) { return pa_ld_from((i64)p[0], (i64)p[1]); }
__device__ __forceinline__ void pa_ld_write(u64* p, i128 v)
{
    const u128 mag = v < 0 ? (u128)(-v) : (u128)v;
    p[0] = (u64)mag;
    p[1] = (u64)(mag >> 64) | (v < 0 ? 0x8000000000000000ULL : 0ULL);
}
// 10^38 = 0x4B3B4CA85A86C47A_098A224000000000
__device__ __forceinline__ i128 pa_ten38() { return pa_i128_of(0x4B3B4CA85A86C47AULL, 0x098A224000000000ULL); }
// UnscaledDecimal128Arithmetic.throwIfOverflows: a magnitude of 10^38 or more
__device__ __forceinline__ i128 pa_dec_check38(i128 v, i32* err)
{
    if (v >= pa_ten38() || v <= -pa_ten38()) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return v;
}
// |v| < bound (10^precision) or NUMERIC_VALUE_OUT_OF_RANGE: casts to DECIMAL(p, s)
__device__ __forceinline__ i128 pa_dec_check_bound(i128 v, i128 bound, i32* err)
{
    if (v >= bound || v <= -bound) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return v;
}
// a * b, NUMERIC_VALUE_OUT_OF_RANGE when the product does not fit 127 bits (the caller then checks 10^38): 64-bit limbs, no
// compiler-rt (__muloti4 does not exist on the device)
__device__ __forceinline__ i128 pa_dec_mul(i128 a, i128 b, i32* err)
{
    const bool neg = (a < 0) != (b < 0);
    const u128 x = a < 0 ? (u128)(-a) : (u128)a, y = b < 0 ? (u128)(-b) : (u128)b;
    const u64 x0 = (u64)x, x1 = (u64)(x >> 64), y0 = (u64)y, y1 = (u64)(y >> 64);
    bool ovf = x1 != 0 && y1 != 0;
    const u128 low = (u128)x0 * y0;
    const u128 c1 = (u128)x1 * y0, c2 = (u128)x0 * y1;
    ovf = ovf || (c1 >> 64) != 0 || (c2 >> 64) != 0;
    const u128 cross = (u128)(u64)c1 + (u128)(u64)c2 + (low >> 64);
    ovf = ovf || (cross >> 63) != 0;
    const u128 mag = (cross << 64) | (u128)(u64)low;
    if (ovf) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return neg ? -(i128)mag : (i128)mag;
}
__device__ __forceinline__ i128 pa_dec_add(i128 a, i128 b, i32* err)
{
    i128 r;
    if (__builtin_add_overflow(a, b, &r)) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return r;
}
__device__ __forceinline__ i128 pa_dec_sub(i128 a, i128 b, i32* err)
{
    i128 r;
    if (__builtin_sub_overflow(a, b, &r)) pa_raise(err, PA_DEV_ERR_OUT_OF_RANGE);
    return r;
}
// x / d rounded half up (away from zero), d > 0: rescaling a SHORT decimal to a smaller scale (DecimalConversions)
__device__ __forceinline__ i64 pa_dec_div_round(i64 x, i64 d)
{
    const i64 q = x / d, r = x % d, half = d / 2;
    return q + (r >= half && d > 1 ? 1 : (r <= -half && d > 1 ? -1 : 0));
}
// Limb k of a decimal value for the sum accumulators: 30 bits per limb, the top limb signed.  A sum of decimals is kept as
// independent i64 sums of the limbs (every word of a group's state is then an ordinary integer sum: merges, folds, replicas and
// PARTIAL states treat it like any other), exact for 2^33 rows per group; the value is put together again at output.
#define PA_DEC_LIMB_BITS 30
__device__ __forceinline__ i64 pa_dec_limb(i128 v, int k, int last)
{
    const i128 s = v >> (PA_DEC_LIMB_BITS * k);
    return k == last ? (i64)s : (i64)((u64)s & ((1ULL << PA_DEC_LIMB_BITS) - 1ULL));
}

// ---------------------------------------------------------------------------------------------
// workgroup (256 threads) exclusive scan of small per-thread counts: wave shuffles + LDS
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ i32 pa_block_exclusive_scan_256(i32 v, i32* total)
{
    __shared__ i32 pa_scan_wave_sums[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        i32 o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) pa_scan_wave_sums[wave] = inc;
    __syncthreads();
    i32 base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        i32 s = pa_scan_wave_sums[w];
        if (w < wave) base += s;
        all += s;
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

// Kernel argument block of the generated FilterAndProject kernels (op_filter_project.cpp mirrors it).
struct PaFpArgs {
    const void* v[PA_MAX_CHANNELS];
    const i32* o[PA_MAX_CHANNELS];
    const u8* nl[PA_MAX_CHANNELS];
    void* out_v[PA_MAX_CHANNELS];   // per projection: output values (worst case n entries)
    u8* out_nl[PA_MAX_CHANNELS];    // per projection: output nulls or nullptr
    i64 n;
    i32 vec;
    i32 pad;
    u8* sel4;                       // 4 selection bits per row quad (filter result, PageFilter.filter)
    i32* tile_counts;               // selected rows per 1024-row tile
    const i32* tile_offsets;        // exclusive scan of tile_counts
    i32* positions;                 // SelectedPositions.positions (ascending), worst case n entries
    i32* err;
    const u64* dyn_bits;            // dynamic filter from a join's build side: existence bitmap over [dyn_min, dyn_min + dyn_range]
    i64 dyn_min;
    u64 dyn_range;
    // probe stage (FilterAndProject -> LookupJoin in one pass, op_filter_project.cpp): the keyed probe-side table, its key bitmap,
    // and the build columns the output carries, as in PaFusedArgs
    const void* jslots;
    const u64* jbits;
    i64 jmin;
    u64 jrange;
    u32 jmask;
    u32 jwrap;
    const void* bv[PA_MAX_BUILD_CHANNELS];
    const u8* bn[PA_MAX_BUILD_CHANNELS];
    const pa_u32x4* jrank;          // key rank index, as in PaFusedArgs
    const i32* jrank_rows;
};
// can a probe row with this key match any build row?  (exact inside the bitmap's range; NULL keys never match)
__device__ __forceinline__ bool pa_dyn_test(const PaFpArgs& a, const i64 key)
{
    const u64 d = (u64)(key - a.dyn_min);
    return d <= a.dyn_range && ((a.dyn_bits[d >> 6] >> (d & 63ULL)) & 1ULL) != 0ULL;
}
