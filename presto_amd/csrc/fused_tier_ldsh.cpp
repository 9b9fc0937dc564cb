// fused_tier_ldsh.cpp -- LDSH: medium cardinality.  The workgroup aggregates into an open-addressing table in LDS and adds it to
// the HBM table once, at the end of the launch; rows whose group finds no room there go to the HBM table directly (the GT tier's
// accumulation, fused_tier_gt.cpp).  The table machinery (pa_lt_upsert, its sizing) is shared with LDSP (fused_tier_ldsp.cpp),
// whose tables belong to hash partitions instead of launches.
#include "decimal_host.hpp"
#include "fused_codegen.hpp"
#include "scan_kernels.hpp"

namespace pa {
namespace fused {

void FusedGen::lds_table_size()
{
    // one table per workgroup: tag + key words + accumulator words per slot.  A 1024-thread workgroup with 150 of the CU's
    // 160 KB of LDS (4096 slots for a one-word key and two accumulator words) against two 512-thread workgroups with 62 KB
    // each, measured over 64 M rows: 300 groups 64 -> 77 G rows/s, 1 K 26 -> 61 G, 2 K 21 -> 45 G (these now fit the
    // table without the hash-partitioning passes), 20 K 15 -> 18 G, 100 K 13 -> 15 G (fewer, denser partitions)
    const size_t slot_bytes = 8 * (size_t)(1 + std::max(k.w, 1) + k.nw);
    static const size_t budget = [] {
        const char* e = getenv("PRESTO_AMD_LDSH_KB");
        return (size_t)(e ? atoi(e) : 150) * 1024;
    }();
    int lc = 4096;
    while (lc > 32 && (size_t)lc * slot_bytes > budget) lc >>= 1;
    PA_REQUIRE((size_t)lc * slot_bytes <= budget, PA_ERR_NOT_SUPPORTED, "group state too wide for the LDS-table variant");
    k.lc = lc;
    k.block = budget > 64 * 1024 ? 1024 : 512;
}

void FusedGen::lds_table_declarations()
{
    // Medium cardinality: the workgroup aggregates into an open-addressing table in LDS (ds_cmpst / ds_add: no HBM
    // atomics in the row loop -- atomics of many rows on a few HBM addresses retire at ~16 M/s per address on this
    // part), and adds its table to the HBM table once, at the end of the kernel.  A row whose group finds no room
    // in the LDS table (more than PA_LC / 2 groups seen by the workgroup) goes to the HBM table directly.
    int lc_bits = 0;
    while ((1 << lc_bits) < k.lc) lc_bits++;
    // the LDS table is indexed by the TOP bits of the 32-bit key hash: the low bits choose the partition (hash-partitioned
    // path) and the HBM-table slot, so rows of one partition would otherwise share their home slots
    // rows whose group finds the table this full go to the HBM table: half of it when the table is the workgroup's own for one
    // launch (its groups are flushed into HBM, which must have room), three quarters when it is a partition's table for good
    src << "#define PA_LC " << k.lc << "\n#define PA_LC_SHIFT " << (32 - lc_bits) << "\n#define PA_LT_LIMIT " << (variant == V_LDSP ? "(PA_LC * 3 / 4)" : "(PA_LC / 2)") << "\n";
    src << "__shared__ u64 pa_lt_tag[PA_LC];\n__shared__ u64 pa_lt_key[PA_LC * PA_KW];\n__shared__ u64 pa_lt_acc[PA_LC * PA_NW];\n"
           "__shared__ i32 pa_lt_count;\n";
    src << "struct PaAcc { PaGtView tv; PaGtCtr gt; PaGtCtr flush; i64 fell; };\n";
    // same claim / publish protocol as pa_gt_upsert_n, on LDS: tag 0 -> busy -> ready, wave-uniform loop so that a
    // lane waiting for a slot another lane of its wave is publishing cannot starve it
    src << "__device__ __forceinline__ int pa_lt_upsert(const u32 h, const u64 (&k)[PA_KW])\n{\n"
           "    const u64 busy = ((u64)h << 2) | 1ULL, ready = ((u64)h << 2) | 3ULL;\n"
           "    u32 i = h >> PA_LC_SHIFT;\n    u32 probes = 0;\n    int spins = 0;\n    int result = -2;\n"
           "    while (__ballot(result == -2) != 0ULL) {\n        if (result == -2) {\n"
           "            const u64 t = __hip_atomic_load(&pa_lt_tag[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
           "            bool advance = false;\n"
           "            if (t == 0ULL) {\n"
           "                if (__hip_atomic_load(&pa_lt_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= PA_LT_LIMIT) result = -1;\n"
           "                else {\n"
           "                    u64 expected = 0ULL;\n"
           "                    if (__hip_atomic_compare_exchange_strong(&pa_lt_tag[i], &expected, busy, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {\n"
           "#pragma unroll\n                        for (int w = 0; w < PA_KW; w++) pa_lt_key[i * PA_KW + w] = k[w];\n"
           "                        __hip_atomic_store(&pa_lt_tag[i], ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
           "                        __hip_atomic_fetch_add(&pa_lt_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
           "                        result = (int)i;\n                    }\n                }\n            }\n"
           "            else if ((t | 2ULL) == ready) {\n"
           "                if (t == busy) { if (++spins > (1 << 20)) result = -1; }\n"
           "                else {\n                    bool eq = true;\n#pragma unroll\n"
           "                    for (int w = 0; w < PA_KW; w++) eq = eq && (pa_lt_key[i * PA_KW + w] == k[w]);\n"
           "                    if (eq) result = (int)i; else advance = true;\n                }\n            }\n"
           "            else advance = true;\n"
           "            if (advance) { i = (i + 1) & (PA_LC - 1); if (++probes >= PA_LC) result = -1; }\n"
           "        }\n    }\n    return result;\n}\n";
}

// a row into the workgroup's table; `else`: it falls through to the HBM table (table_accumulate goes on from there)
void FusedGen::lds_table_accumulate_begin()
{
    src << "  const int ls = pa_lt_upsert(h, key);\n  if (ls >= 0) {\n";
    for (int w = 0; w < k.nw; w++) {
        std::string idx = "pa_lt_acc[ls * PA_NW + " + std::to_string(w) + "]";
        if (words[w].kind == W_SUMF) {
            src << "    if (u" << w << ") __hip_atomic_fetch_add((double*)&" << idx << ", x" << w << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
        }
        else if (words[w].kind == W_SUMI) {
            src << "    if (u" << w << ") { i64 o = (i64)__hip_atomic_fetch_add(&" << idx << ", (u64)x" << w
                << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); i64 r; if (__builtin_add_overflow(o, x" << w
                << ", &r)) pa_raise(a.err, PA_DEV_ERR_OUT_OF_RANGE); }\n";
        }
        else if (words[w].kind == W_MAXU) {
            src << "    if (u" << w << ") __hip_atomic_fetch_max(&" << idx << ", x" << w << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
        }
        else {
            src << "    if (u" << w << ") __hip_atomic_fetch_add(&" << idx << ", " << ("(u64)x" + std::to_string(w))
                << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
        }
    }
    src << "  } else {\n  acc.fell++;\n";
}

void FusedGen::ldsh_kernel_begin()
{
    src << "    for (int i = threadIdx.x; i < PA_LC; i += " << B << ") pa_lt_tag[i] = 0ULL;\n";
    src << "    for (int i = threadIdx.x; i < PA_LC * PA_NW; i += " << B << ") pa_lt_acc[i] = 0ULL;\n";
    src << "    if (threadIdx.x == 0) pa_lt_count = 0;\n    __syncthreads();\n";
    src << "    PaAcc acc; acc.tv = pa_gt_view(a, PA_KW, PA_NW); acc.gt = pa_gt_ctr_init(acc.tv.count, true, a.gt_rep_mask + 1u);\n"
           "    acc.flush = pa_gt_ctr_init(acc.tv.count, false); acc.fell = 0;\n";
}

void FusedGen::ldsh_kernel_end()
{
    // the workgroup's table -> HBM table: one upsert and PA_NW atomics per group and workgroup
    src << "    __syncthreads();\n    const u64 cap = (u64)a.gt_mask + 1ULL;\n";
    // (every workgroup starts somewhere else in its table: a group sits at the same place in all of them -- the slot is a function
    // of the key's hash -- and 256 workgroups flushing slot after slot in step would meet on one HBM address after the other)
    src << "    for (int s0 = threadIdx.x; s0 < PA_LC; s0 += " << B << ") {\n        const int sl = (s0 + (int)(blockIdx.x * 1297u)) & (PA_LC - 1);\n"
           "        if (pa_lt_tag[sl] == 0ULL) continue;\n"
           "        u64 fk[PA_KW];\n#pragma unroll\n        for (int w = 0; w < PA_KW; w++) fk[w] = pa_lt_key[sl * PA_KW + w];\n"
           "        const int g = pa_gt_upsert<PA_KW>(acc.tv.tag, acc.tv.keys, a.gt_mask, pa_key_hash(fk, PA_KW), fk, acc.flush, 0x7fffffff, a.err);\n"
           "        if (g < 0) { pa_raise(a.err, PA_DEV_ERR_RESOURCES); continue; }\n";
    for (int w = 0; w < k.nw; w++) {
        std::string idx = std::to_string(w) + "ULL * cap + (u64)g";
        std::string v = "pa_lt_acc[sl * PA_NW + " + std::to_string(w) + "]";
        if (words[w].kind == W_SUMF) src << "        pa_gt_add_f64(acc.tv.words, " << idx << ", __longlong_as_double((i64)" << v << "));\n";
        else if (words[w].kind == W_SUMI) src << "        pa_gt_add_i64_exact(acc.tv.words, " << idx << ", (i64)" << v << ", a.err);\n";
        else if (words[w].kind == W_MAXU) src << "        pa_gt_max_u64(acc.tv.words, " << idx << ", " << v << ");\n";
        else src << "        pa_gt_add_u64(acc.tv.words, " << idx << ", " << v << ");\n";
    }
    src << "    }\n    pa_gt_ctr_flush(acc.flush, acc.tv.count);\n";
    src << "    { const i64 f = pa_wave_sum_i64(acc.fell); if ((threadIdx.x & 63) == 0 && f != 0) atomicAdd((unsigned long long*)a.overflow_rows, (unsigned long long)f); }\n";
}

}  // namespace fused
}  // namespace pa
