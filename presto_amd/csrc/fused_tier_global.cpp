// fused_tier_global.cpp -- GLOBAL: no group keys.  Per-lane register accumulators -> wave shuffle -> LDS -> one partial state per
// workgroup in a slab -> fixed-order merge kernel (bitwise reproducible).  (AggregationOperator.addInput,
// …/operator/AggregationOperator.java:145-160, behind the page's filter and projections.)
#include "decimal_host.hpp"
#include "fused_codegen.hpp"
#include "scan_kernels.hpp"

namespace pa {
namespace fused {

void FusedGen::global_declarations()
{
    src << "struct PaAcc {";
    for (int w = 0; w < k.nw; w++) src << (words[w].kind == W_SUMF ? " double" : (words[w].kind == W_MAXU ? " u64" : " i64")) << " w" << w << ";";
    src << " };\n";
}

void FusedGen::global_accumulate_row()
{
    src << "if (sel) {\n";
    for (int w = 0; w < k.nw; w++) {
        if (words[w].kind == W_SUMF) src << "if (u" << w << ") acc.w" << w << " = acc.w" << w << " + x" << w << ";\n";
        else if (words[w].kind == W_SUMI) src << "if (u" << w << ") acc.w" << w << " = pa_add_exact(acc.w" << w << ", x" << w << ", a.err);\n";
        else if (words[w].kind == W_MAXU) src << "if (u" << w << ") acc.w" << w << " = x" << w << " > acc.w" << w << " ? x" << w << " : acc.w" << w << ";\n";
        else src << "if (u" << w << ") acc.w" << w << " += x" << w << ";\n";
    }
    src << "}\n";
}

void FusedGen::global_kernel_begin()
{
    src << "    PaAcc acc;\n";
    for (int w = 0; w < k.nw; w++) src << "    acc.w" << w << " = 0;\n";
}

void FusedGen::global_thread_ids()
{
    // XCD-aware block -> tile mapping: consecutive workgroup ids go round-robin to the 8 XCDs, so give the
    // workgroups of one XCD consecutive tiles (each XCD's L2 / TLB then walks one contiguous eighth of every grid
    // stride).  Measured on Q6: 0.75 -> 0.79 of the HBM peak; neutral for the one-wave workgroups of the LDS variant.
    src << "    const u32 bsw = (gridDim.x & 7u) == 0u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;\n"
           "    const i64 t = (i64)bsw * " << B << " + threadIdx.x, T = (i64)gridDim.x * " << B << ";\n";
}

void FusedGen::global_kernel_end()
{
    src << "    __shared__ u64 red[" << (B / 64) << " * PA_NW];\n    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;\n";
    for (int w = 0; w < k.nw; w++) {
        if (words[w].kind == W_SUMF) src << "    { double v = pa_wave_sum_f64(acc.w" << w << "); if (lane == 0) red[wave * PA_NW + " << w << "] = (u64)__double_as_longlong(v); }\n";
        else if (words[w].kind == W_SUMI) src << "    { i64 v = pa_wave_sum_i64_exact(acc.w" << w << ", a.err); if (lane == 0) red[wave * PA_NW + " << w << "] = (u64)v; }\n";
        else if (words[w].kind == W_MAXU) src << "    { u64 v = pa_wave_max_u64(acc.w" << w << "); if (lane == 0) red[wave * PA_NW + " << w << "] = v; }\n";
        else src << "    { i64 v = pa_wave_sum_i64(acc.w" << w << "); if (lane == 0) red[wave * PA_NW + " << w << "] = (u64)v; }\n";
    }
    src << "    __syncthreads();\n    if (threadIdx.x < PA_NW) {\n        const int w = threadIdx.x;\n        u64 r = red[w];\n";
    src << "        for (int i = 1; i < " << (B / 64) << "; i++) {\n            u64 o = red[i * PA_NW + w];\n";
    src << "            switch (w) {\n";
    for (int w = 0; w < k.nw; w++) {
        src << "                case " << w << ": ";
        if (words[w].kind == W_SUMF) src << "r = (u64)__double_as_longlong(__longlong_as_double((i64)r) + __longlong_as_double((i64)o)); break;\n";
        else if (words[w].kind == W_SUMI) src << "r = (u64)pa_add_exact((i64)r, (i64)o, a.err); break;\n";
        else if (words[w].kind == W_MAXU) src << "r = o > r ? o : r; break;\n";
        else src << "r = r + o; break;\n";
    }
    src << "            }\n        }\n        a.slab[(u64)blockIdx.x * PA_NW + w] = r;\n    }\n";
}

}  // namespace fused
}  // namespace pa
