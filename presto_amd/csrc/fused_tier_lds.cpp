// fused_tier_lds.cpp -- LDS: at most PA_C groups.  One wave per workgroup; the wave's key table is wave-uniform state in (scalar)
// registers, the accumulators are lane-private in LDS (no atomics, no barriers); per-wave partial tables go to a slab that a merge
// kernel folds into the HBM table.  A wave that meets more groups marks its table as overflowed and the launch is redone on the
// next tier.  Every lane must take part in every pa_row call: `pa_fused` (mode 1) takes the leading multiple of 256 rows,
// `pa_fused_tail` (mode 2) the rest row by row with a wave-uniform trip count, `pa_fused_ranges` (mode 3) both per range.
#include "decimal_host.hpp"
#include "fused_codegen.hpp"
#include "scan_kernels.hpp"

namespace pa {
namespace fused {

void FusedGen::lds_check_capacity()
{
    PA_REQUIRE((size_t)k.nw * kLdsSlots * 64 * 8 <= 64 * 1024, PA_ERR_NOT_SUPPORTED, "too many accumulator words for the LDS variant");
}

void FusedGen::lds_declarations()
{
    // key table of the wave in (scalar) registers; accumulators lane-private in LDS: word w of group g
    // of lane l lives at pa_accw[(w * C + g) * 64 + l], so no two lanes ever share an address
    src << "__shared__ u64 pa_accw[PA_NW * PA_C * 64];\n";
    src << "struct PaAcc { u64 tk[PA_C][PA_KW]; int tcount; u32 lane; };\n";
}

void FusedGen::lds_accumulate_row()
{
    src << "int g = -1;\nif (sel) {\n#pragma unroll\n  for (int s = 0; s < PA_C; s++) {\n    bool eq = s < acc.tcount;\n#pragma unroll\n"
           "    for (int w = 0; w < PA_KW; w++) eq = eq && (key[w] == acc.tk[s][w]);\n    if (eq) g = s;\n  }\n}\n";
    // first occurrences: append the missing keys to the wave's table one at a time (wave-uniform loop)
    src << "u64 miss = __ballot(sel && g < 0);\nwhile (miss != 0ULL) {\n  const int src_lane = __builtin_amdgcn_readfirstlane(__ffsll((long long)miss) - 1);\n"
           "  u64 nk[PA_KW];\n#pragma unroll\n  for (int w = 0; w < PA_KW; w++) nk[w] = pa_readlane_u64(key[w], src_lane);\n"
           "  const int slot = acc.tcount;\n  if (slot < PA_C) {\n#pragma unroll\n    for (int s = 0; s < PA_C; s++) {\n      if (s == slot) {\n#pragma unroll\n"
           "        for (int w = 0; w < PA_KW; w++) acc.tk[s][w] = nk[w];\n      }\n    }\n    acc.tcount = slot + 1;\n  } else {\n    acc.tcount = PA_C + 1;\n  }\n"
           "  bool mine = sel && g == -1;\n#pragma unroll\n  for (int w = 0; w < PA_KW; w++) mine = mine && (key[w] == nk[w]);\n"
           "  if (mine) g = slot < PA_C ? slot : -2;\n  miss = __ballot(sel && g == -1);\n}\n";
    src << "if (sel) {\n  if (g >= 0) {\n";
    for (int w = 0; w < k.nw; w++) {
        std::string idx = "pa_accw[(" + std::to_string(w) + " * PA_C + g) * 64 + acc.lane]";
        if (words[w].kind == W_SUMF) {
            src << "    if (u" << w << ") __hip_atomic_fetch_add((double*)&" << idx << ", x" << w << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
        }
        else if (words[w].kind == W_SUMI) {
            src << "    if (u" << w << ") { i64* p = (i64*)&" << idx << "; *p = pa_add_exact(*p, x" << w << ", a.err); }\n";
        }
        else if (words[w].kind == W_MAXU) {
            src << "    if (u" << w << ") { u64* p = &" << idx << "; if (x" << w << " > *p) *p = x" << w << "; }\n";
        }
        else {
            src << "    if (u" << w << ") __hip_atomic_fetch_add(&" << idx << ", " << (words[w].val == "1" ? std::string("1ULL") : "(u64)x" + std::to_string(w))
                << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
        }
    }
    // a row whose group found no slot: the wave has marked its table as overflowed (tcount = PA_C + 1, set in the loop
    // above) and reports once at the end of the kernel; the launch is discarded as a whole.  (A per-row atomic on the
    // one overflow counter would serialise the useless pass on a single address: 3 ms instead of 0.25 ms per 64 M rows.)
    src << "  }\n}\n";
}

void FusedGen::lds_kernel_begin()
{
    src << "    for (int i = threadIdx.x; i < PA_NW * PA_C * 64; i += 64) pa_accw[i] = 0ULL;\n";
    src << "    __syncthreads();\n";
    src << "    PaAcc acc; acc.tcount = 0; acc.lane = threadIdx.x;\n";
    src << "#pragma unroll\n    for (int s = 0; s < PA_C; s++) {\n#pragma unroll\n        for (int w = 0; w < PA_KW; w++) acc.tk[s][w] = 0ULL;\n    }\n";
}

void FusedGen::lds_head_loop()
{
    std::string args[4];
    src << "    const i64 nq = a.n >> 2;  // the host passes a multiple of 256 rows\n";
    src << "    for (i64 q = t; q < nq; q += T) {\n";
    emit_vector_loads(ri, layout, src, args);
    emit_quad(args);
    src << "    }\n";
}

void FusedGen::lds_tail_loop()
{
    src << "    for (i64 rb = (i64)blockIdx.x * 64; rb < a.n; rb += T) {\n        const bool live = rb + threadIdx.x < a.n;\n"
           "        const i64 r = live ? rb + threadIdx.x : a.n - 1;\n        pa_row(a, acc, live, (i32)r" << scalar_args(ri, layout) << ");\n    }\n";
}

void FusedGen::lds_kernel_end()
{
    // per-wave partial table -> slab, field-major: field f of entry e = blockIdx.x * C + i at slab[f * E + e]
    src << "    __syncthreads();\n";
    src << "    if (acc.tcount > PA_C && threadIdx.x == 0) atomicAdd((unsigned long long*)a.overflow_rows, 1ULL);\n";
    src << "    const u64 E = (u64)gridDim.x * PA_C;\n";
    src << "#pragma unroll\n    for (int i = 0; i < PA_C; i++) {\n        u64* e = a.slab + (u64)blockIdx.x * PA_C + i;\n";
    src << "        const bool occ = i < acc.tcount;\n        if (threadIdx.x == 0) e[0] = occ ? 1ULL : 0ULL;\n        if (occ) {\n";
    src << "#pragma unroll\n            for (int w = 0; w < PA_KW; w++) { if (threadIdx.x == 0) e[(u64)(1 + w) * E] = acc.tk[i][w]; }\n";
    for (int w = 0; w < k.nw; w++) {
        std::string idx = "pa_accw[(" + std::to_string(w) + " * PA_C + i) * 64 + threadIdx.x]";
        std::string dst = "e[(u64)(1 + PA_KW + " + std::to_string(w) + ") * E]";
        if (words[w].kind == W_SUMF) src << "            { double v = pa_wave_sum_f64(__longlong_as_double((i64)" << idx << ")); if (threadIdx.x == 0) " << dst << " = (u64)__double_as_longlong(v); }\n";
        else if (words[w].kind == W_SUMI) src << "            { i64 v = pa_wave_sum_i64_exact((i64)" << idx << ", a.err); if (threadIdx.x == 0) " << dst << " = (u64)v; }\n";
        else if (words[w].kind == W_MAXU) src << "            { u64 v = pa_wave_max_u64(" << idx << "); if (threadIdx.x == 0) " << dst << " = v; }\n";
        else src << "            { i64 v = pa_wave_sum_i64((i64)" << idx << "); if (threadIdx.x == 0) " << dst << " = (u64)v; }\n";
    }
    src << "        }\n    }\n";
}

}  // namespace fused
}  // namespace pa
