// fused_tier_gt.cpp -- GT: any cardinality.  Open-addressing table in HBM (pa_gt_upsert: claim / publish on the tag word, agent-scope
// atomics on word-major accumulator arrays), rows whose group does not fit are spilled to a list the host replays after a rehash.
// In front of the table a thread combines consecutive rows with equal keys in registers (clustered inputs).  The accumulation into
// the HBM table is also where the LDS-table tiers send what their tables have no room for.
// (InMemoryHashAggregationBuilder.processPage, …/aggregation/builder/InMemoryHashAggregationBuilder.java:139-155.)
#include "decimal_host.hpp"
#include "fused_codegen.hpp"
#include "scan_kernels.hpp"

namespace pa {
namespace fused {

void FusedGen::gt_declarations()
{
    // pending run of the thread: consecutive selected rows with equal keys are combined before they touch the table
    src << "struct PaAcc { PaGtView tv; PaGtCtr gt; i32 pn; i32 prow; u64 pk[PA_KW];";
    for (int w = 0; w < k.nw; w++) src << " bool pu" << w << "; " << (words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")) << " px" << w << ";";
    src << " };\n";
}

void FusedGen::table_accumulate()
{
    // accumulation of one row into the workgroup's LDS table / the HBM table
    src << "__device__ __forceinline__ void " << (gt_like ? "pa_acc_now" : "pa_acc")
        << "(const PaFusedArgs& a, PaAcc& acc, const bool sel, const i32 row, " << (gt_like ? "const i32 nrows, " : "") << "const u64 (&key)[PA_KW]";
    for (int w = 0; w < k.nw; w++) src << ", const bool u" << w << ", const " << (words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")) << " x" << w;
    src << ")\n{\n";
    src << "if (sel) {\n  const u32 h = pa_key_hash(key, PA_KW);\n";
    if (lds_table) lds_table_accumulate_begin();
    src << "  int g = pa_gt_upsert<PA_KW>(acc.tv.tag, acc.tv.keys, a.gt_mask, h, key, acc.gt, a.gt_max_fill, a.err);\n";
    src << "  if (g >= 0) {\n    const u64 cap = (u64)a.gt_mask + 1ULL;\n";
    for (int w = 0; w < k.nw; w++) {
        std::string idx = std::to_string(w) + "ULL * cap + (u64)g";
        if (words[w].kind == W_SUMF) src << "    if (u" << w << ") pa_gt_add_f64(acc.tv.words, " << idx << ", x" << w << ");\n";
        else if (words[w].kind == W_SUMI) src << "    if (u" << w << ") pa_gt_add_i64_exact(acc.tv.words, " << idx << ", x" << w << ", a.err);\n";
        else if (words[w].kind == W_MAXU) src << "    if (u" << w << ") pa_gt_max_u64(acc.tv.words, " << idx << ", x" << w << ");\n";
        else src << "    if (u" << w << ") pa_gt_add_u64(acc.tv.words, " << idx << ", (u64)x" << w << ");\n";
    }
    // no room for this row's group: spill the row; the host rehashes and replays the spilled rows
    if (variant == V_GT) {
        src << "  } else {\n    const u32 sb = atomicAdd(a.spill_count, (u32)nrows);\n    for (i32 i = 0; i < nrows; i++) a.spill_rows[sb + (u32)i] = row + i;\n  }\n";
    }
    else {
        src << "  } else {\n    a.spill_rows[atomicAdd(a.spill_count, 1u)] = row;\n  }\n";
    }
    if (lds_table) src << "  }\n";
    src << "}\n";
    src << "}\n\n";
}

void FusedGen::gt_run_combining()
{
    // Run combining.  A thread of the vector loop walks 4 consecutive rows; when their keys repeat (clustered inputs: a
    // fact table joined on its own key order) the rows are combined in registers and reach the table once -- one probe and
    // one atomic per word for the run.  pa_flush ends the pending run; the loops call it after every quad (every row
    // in the scalar / list loops), so a run is always a range of consecutive rows, which is what a spill records.
    src << "__device__ __forceinline__ void pa_flush(const PaFusedArgs& a, PaAcc& acc, const bool doit)\n{\n"
           "  pa_acc_now(a, acc, doit && acc.pn > 0, acc.prow, acc.pn, acc.pk";
    for (int w = 0; w < k.nw; w++) src << ", acc.pu" << w << ", acc.px" << w;
    src << ");\n  if (doit) acc.pn = 0;\n}\n";
    src << "__device__ __forceinline__ void pa_acc(const PaFusedArgs& a, PaAcc& acc, const bool sel, const i32 row, const u64 (&key)[PA_KW]";
    for (int w = 0; w < k.nw; w++) src << ", const bool u" << w << ", const " << (words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")) << " x" << w;
    src << ")\n{\n  bool same = sel && acc.pn > 0;\n#pragma unroll\n  for (int w = 0; w < PA_KW; w++) same = same && key[w] == acc.pk[w];\n"
           "  pa_flush(a, acc, !same);\n  if (sel) {\n    if (acc.pn == 0) {\n#pragma unroll\n      for (int w = 0; w < PA_KW; w++) acc.pk[w] = key[w];\n"
           "      acc.prow = row;\n      acc.pn = 1;\n";
    for (int w = 0; w < k.nw; w++) src << "      acc.pu" << w << " = u" << w << "; acc.px" << w << " = x" << w << ";\n";
    src << "    } else {\n      acc.pn = row - acc.prow + 1;\n";
    for (int w = 0; w < k.nw; w++) {
        const std::string P = "acc.px" + std::to_string(w), U = "acc.pu" + std::to_string(w), X = "x" + std::to_string(w);
        std::string comb;
        if (words[w].kind == W_SUMF || words[w].kind == W_CNT) comb = P + " + " + X;
        else if (words[w].kind == W_SUMI) comb = "pa_add_exact(" + P + ", " + X + ", a.err)";
        else comb = "(" + X + " > " + P + " ? " + X + " : " + P + ")";
        src << "      if (u" << w << ") { " << P << " = " << U << " ? " << comb << " : " << X << "; " << U << " = true; }\n";
    }
    src << "    }\n  }\n}\n\n";
}

void FusedGen::table_accumulate_row()
{
    src << "pa_acc(a, acc, sel, row, key";
    for (int w = 0; w < k.nw; w++) src << ", u" << w << ", x" << w;
    src << ");\n";
}

void FusedGen::gt_kernel_begin()
{
    src << "    PaAcc acc; acc.tv = pa_gt_view(a, PA_KW, PA_NW); acc.gt = pa_gt_ctr_init(acc.tv.count, true, a.gt_rep_mask + 1u); acc.pn = 0;\n";
}

void FusedGen::list_loops()
{
    // rows given by a list: grid-stride (spill replays), or one contiguous slice per workgroup (partition-ordered lists:
    // the workgroup's LDS table then meets the groups of a few partitions only)
    src << "    if (a.list_blocked) {\n        const i64 per = (a.n_list + gridDim.x - 1) / gridDim.x;\n"
           "        const i64 b0 = (i64)blockIdx.x * per, b1 = b0 + per < a.n_list ? b0 + per : a.n_list;\n"
           "        for (i64 i = b0 + threadIdx.x; i < b1; i += " << B << ") {\n            const i64 r = a.list_blocked == 2 ? i : (i64)a.row_list[i];\n            pa_row(a, acc, true, (i32)r"
        << scalar_args(ri, layout) << ");" << flush << "\n        }\n    } else {\n";
    src << "    for (i64 i = t; i < a.n_list; i += T) {\n        const i64 r = a.row_list[i];\n        pa_row(a, acc, true, (i32)r" << scalar_args(ri, layout)
        << ");" << flush << "\n    }\n    }\n";
}

}  // namespace fused
}  // namespace pa
