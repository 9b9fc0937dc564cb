// fused_tier_ldsp.cpp -- high cardinality: HASH, the pass that computes every row's hash partition and histograms the tiles of the
// multisplit behind it (scan_kernels.hpp), and LDSP, the LDS-table tier over partition-ordered rows whose tables belong to the
// partitions: a workgroup loads its partition's table, aggregates the partition's rows and stores it back -- no atomics on HBM.
#include "decimal_host.hpp"
#include "fused_codegen.hpp"
#include "scan_kernels.hpp"

namespace pa {
namespace fused {

void FusedGen::hash_declarations()
{
    // Hash-partitioning pass in front of the LDS-table variant at medium cardinality (hundreds to ~10^5 groups): it only
    // computes every row's partition = hash(key) mod P (P + 1 for rows the filter drops).  The rows are then taken in
    // partition order, a contiguous slice per workgroup, so that a workgroup's LDS table meets a few partitions' groups only.
    // (round 3) ... and histograms every 8192-row tile on the way (the tiles of the multisplit behind it, scan_kernels.hpp): the
    // multisplit's own counting pass read the ids a second time
    src << "struct PaAcc { int unused; };\n__shared__ i32 pa_hist[4097];\n";
}

void FusedGen::hash_accumulate_row()
{
    src << "if (live) { const i32 pid = sel ? (i32)(pa_key_hash(key, PA_KW) & a.part_mask) : (i32)(a.part_mask + 1u); a.part_ids[row] = pid; "
           "__hip_atomic_fetch_add(&pa_hist[pid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }\n";
}

void FusedGen::hash_kernel_begin()
{
    src << "    PaAcc acc; acc.unused = 0;\n";
}

void FusedGen::hash_tile_loop()
{
    std::string args[4];
    // tile by tile (a.sub_count: the tile x partition counts of the multisplit, tile-major)
    src << "    const i64 nq = a.vec ? (a.n >> 2) : 0;\n    const i64 tiles = (a.n + " << (kMsplitTileRows - 1) << ") / " << kMsplitTileRows << ";\n"
           "    const i32 hp = (i32)a.part_mask + 2;\n"
           "    for (i64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {\n"
           "      for (i32 i = threadIdx.x; i < hp; i += " << B << ") pa_hist[i] = 0;\n      __syncthreads();\n"
           "      const i64 r0 = tile * " << kMsplitTileRows << ", r1 = r0 + " << kMsplitTileRows << " < a.n ? r0 + " << kMsplitTileRows << " : a.n;\n"
           "      const i64 q1 = (r1 >> 2) < nq ? (r1 >> 2) : nq;\n"
           "      for (i64 q = (r0 >> 2) + threadIdx.x; q < q1; q += " << B << ") {\n";
    emit_vector_loads(ri, layout, src, args);
    emit_quad(args);
    src << "      }\n"
           "      for (i64 r = ((q1 << 2) > r0 ? (q1 << 2) : r0) + threadIdx.x; r < r1; r += " << B << ") {\n        pa_row(a, acc, true, (i32)r" << scalar_args(ri, layout) << ");\n      }\n"
           "      __syncthreads();\n"
           "      for (i32 i = threadIdx.x; i < hp; i += " << B << ") a.sub_count[tile * hp + i] = pa_hist[i];\n      __syncthreads();\n"
           "    }\n";
}

void FusedGen::ldsp_kernel_begin()
{
    // the partition's table comes from HBM as the last launch left it (zeroes at first) ...
    src << "    const u64 sp = (u64)blockIdx.x * PA_LC;\n";
    // (accumulator words are word-major in HBM, [word][slot over all partitions] -- the layout of the HBM group table, so
    // that the partitions' tables can be emitted, or folded, as one table of gridDim.x * PA_LC slots)
    src << "    const u64 ts = (u64)gridDim.x * PA_LC;\n";
    // a.pad3: the first launch on these tables -- they are empty by definition, nothing to load (and the host cleared nothing)
    src << "    if (a.pad3) {\n";
    src << "      for (int i = threadIdx.x; i < PA_LC; i += " << B << ") pa_lt_tag[i] = 0ULL;\n";
    src << "      for (int i = threadIdx.x; i < PA_LC * PA_NW; i += " << B << ") pa_lt_acc[i] = 0ULL;\n";
    src << "      if (threadIdx.x == 0) pa_lt_count = 0;\n";
    src << "    } else {\n";
    src << "      for (int i = threadIdx.x; i < PA_LC; i += " << B << ") pa_lt_tag[i] = a.sub_tag[sp + i];\n";
    src << "      for (int i = threadIdx.x; i < PA_LC * PA_KW; i += " << B << ") pa_lt_key[i] = a.sub_keys[sp * PA_KW + i];\n";
    src << "      for (int i = threadIdx.x; i < PA_LC * PA_NW; i += " << B << ") { const int w = i / PA_LC, sl = i % PA_LC; pa_lt_acc[sl * PA_NW + w] = a.sub_words[(u64)w * ts + sp + sl]; }\n";
    src << "      if (threadIdx.x == 0) pa_lt_count = a.sub_count[blockIdx.x];\n";
    src << "    }\n    __syncthreads();\n";
    src << "    PaAcc acc; acc.tv = pa_gt_view(a, PA_KW, PA_NW); acc.gt = pa_gt_ctr_init(acc.tv.count, true, a.gt_rep_mask + 1u);\n"
           "    acc.flush = pa_gt_ctr_init(acc.tv.count, false); acc.fell = 0;\n";
}

void FusedGen::ldsp_partition_loop()
{
    // ... the workgroup walks the rows of its partition (the columns are partition-ordered), four rows per thread and step with
    // their loads issued together: a partition is a few dozen rows per thread, and a row's way through the LDS table (tag, key
    // compare, atomics) would otherwise wait for one HBM round trip per row (rows beyond the partition ride along, not live)
    src << "    {\n        const i64 b0 = a.part_first[blockIdx.x], b1 = a.part_first[blockIdx.x + 1];\n"
           "        for (i64 r0 = b0 + threadIdx.x; r0 < b1; r0 += " << 4 * B << ") {\n";
    std::string rows[4];
    for (int u = 0; u < 4; u++) {
        const std::string U = std::to_string(u);
        if (u > 0) src << "            const bool v" << U << " = r0 + " << u * B << " < b1; const i64 r" << U << " = v" << U << " ? r0 + " << u * B << " : r0;\n";
        std::ostringstream decl;
        rows[u] = scalar_loads(ri, layout, "r" + U, "_" + U, decl);
        src << "            " << decl.str() << "\n";
    }
    for (int u = 0; u < 4; u++) {
        const std::string U = std::to_string(u);
        src << "            pa_row(a, acc, " << (u == 0 ? std::string("true") : "v" + U) << ", (i32)r" << U << rows[u] << ");\n";
    }
    src << "        }\n    }\n";
    // ... and the table goes back (plain coalesced stores: nobody else touches this partition)
    src << "    __syncthreads();\n";
    src << "    for (int i = threadIdx.x; i < PA_LC; i += " << B << ") a.sub_tag[sp + i] = pa_lt_tag[i];\n";
    src << "    for (int i = threadIdx.x; i < PA_LC * PA_KW; i += " << B << ") a.sub_keys[sp * PA_KW + i] = pa_lt_key[i];\n";
    src << "    for (int i = threadIdx.x; i < PA_LC * PA_NW; i += " << B << ") { const int w = i / PA_LC, sl = i % PA_LC; a.sub_words[(u64)w * ts + sp + sl] = pa_lt_acc[sl * PA_NW + w]; }\n";
    src << "    if (threadIdx.x == 0) a.sub_count[blockIdx.x] = pa_lt_count;\n";
    src << "    { const i64 f = pa_wave_sum_i64(acc.fell); if ((threadIdx.x & 63) == 0 && f != 0) atomicAdd((unsigned long long*)a.overflow_rows, (unsigned long long)f); }\n";
}

}  // namespace fused
}  // namespace pa
