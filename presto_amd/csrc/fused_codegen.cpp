// fused_codegen.cpp -- the generator of the fused operator's kernels: what every tier shares (see fused_codegen.hpp for the map).
//
// The generated translation unit: #defines of the state's shape, the tier's declarations (PaAcc, LDS arrays, helpers), the row
// function -- pa_row(a, acc, live, row, <the row's column values>): filter, projections of the selected rows, key words,
// accumulator inputs, then the tier's accumulation -- and the kernels, whose loops hand rows to it four at a time (16-byte loads)
// or one by one.
#include "decimal_host.hpp"
#include "fused_codegen.hpp"
#include "scan_kernels.hpp"

namespace pa {
namespace fused {

int range_entry_words(const Spec& s, const std::vector<ChannelLayout>& layout)
{
    int words = 1;
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used_channel[c]) continue;
        words += 1 + (layout[c].type == PA_VARCHAR ? 1 : 0) + (layout[c].nullable ? 1 : 0);
    }
    return words;
}

static std::vector<ChannelLayout> extended_layout(const Spec& s, const std::vector<ChannelLayout>& layout)
{
    // the page's channels, then the build columns of the probe stage as channels n_in + v (`layout` may already hold them:
    // their nullability is the lookup source's)
    std::vector<ChannelLayout> ext(layout.begin(), layout.begin() + s.n_in);
    if (s.join) {
        for (size_t v = 0; v < s.join->build_cols.size(); v++) {
            ChannelLayout cl;
            cl.type = s.join->build_types[v];
            cl.nullable = (size_t)s.n_in + v < layout.size() ? layout[(size_t)s.n_in + v].nullable : true;
            ext.push_back(cl);
        }
    }
    return ext;
}

FusedGen::FusedGen(const Spec& spec, const std::vector<ChannelLayout>& page_layout, int requested_variant)
    : s(spec), layout(page_layout), variant(requested_variant), ranged(requested_variant == V_GLOBAL_R || requested_variant == V_LDS_R),
      ext(extended_layout(spec, page_layout)), gen(ext, "a.err")
{
}

KernelInfo generate(const Spec& s, const std::vector<ChannelLayout>& layout, int variant)
{
    FusedGen g(s, layout, variant);
    return g.run();
}

KernelInfo FusedGen::run()
{
    if (ranged) {
        PA_REQUIRE(!s.join, PA_ERR_NOT_SUPPORTED, "no range-table variant behind a probe stage");
        variant = variant == V_GLOBAL_R ? V_GLOBAL : V_LDS;
    }
    k.variant = variant;
    k.ranged = ranged;
    // entry names carry the tier (and "probe" behind a probe stage); the translation unit appends the first 8 hex digits of the
    // code object's key (PA_K, jit.cpp), so that a kernel trace tells the plans apart: pa_fused_lds_<key8>, pa_fused_probe_brow_<key8> ...
    {
        static const char* const kTierName[] = {"global", "lds", "gt", "ldsh", "hash", "ldsp", "brow"};
        k.entry = std::string("pa_fused_") + (s.join ? "probe_" : "") + kTierName[variant] + (ranged ? "_ranges" : "");
    }
    k.block = variant == V_LDS ? 64 : 256;
    k.c = variant == V_LDS ? kLdsSlots : 0;

    PA_REQUIRE(variant != V_BROW || (s.join && !s.join->brow_group_proj.empty()), PA_ERR_NOT_SUPPORTED, "no build-row variant for this plan");
    PA_REQUIRE(!s.join || (variant != V_HASH && variant != V_LDSP), PA_ERR_NOT_SUPPORTED, "no hash-partitioned variants behind a probe stage");
    ri.n_in = s.n_in;
    ri.used = s.used_channel;
    for (int c = 0; c < s.n_in; c++) ri.used[c] = s.used_channel[c] && !s.lazy_channel[c];
    ri.short_bound = s.short_bound;
    brow = variant == V_BROW;
    lds_table = variant == V_LDSH || variant == V_LDSP;
    gt_like = variant == V_GT;  // a thread-private pending run in front of the table

    row_filter();         // 1. filter (and, behind it, the probe key of a probe stage)
    group_keys();         // 2. + 3. group keys -> bit-packed words, over the projections they read
    accumulator_words();  // 4. accumulator words, shared between aggregates over the same (input, mask)
    if (brow) probe_occupancy_word();
    state_layout_id();
    if (variant == V_LDS) lds_check_capacity();
    if (lds_table) lds_table_size();
    B = k.block;
    // V_GT: the pending run of every thread ends after a quad of consecutive rows / after every row of the other loops
    flush = gt_like || brow ? " pa_flush(a, acc, true);" : "";

    // ---- assemble the translation unit ----
    // (BROW: the row loop's "key" is the build position, one word; PA_TW = key words of the table, written by pa_brow_keys)
    src << "#define PA_NW " << k.nw << "\n#define PA_KW " << (brow ? 1 : (k.w > 0 ? k.w : 1)) << "\n#define PA_TW " << (k.w > 0 ? k.w : 1) << "\n#define PA_C "
        << (k.c > 0 ? k.c : 1) << "\n";
    if (variant == V_GLOBAL) global_declarations();
    else if (variant == V_LDS) lds_declarations();
    else if (lds_table) lds_table_declarations();
    else if (variant == V_HASH) hash_declarations();
    else if (brow) brow_declarations();
    else gt_declarations();
    if (gt_like || lds_table) table_accumulate();
    if (gt_like) gt_run_combining();
    if (s.join) probe_build_loads();
    row_function();
    if (s.join) probe_row_composition();

    // kernels.  mode 0: one kernel walks the page (GLOBAL / GT).  LDS variant: the wave's key table is wave-uniform
    // state, so every lane must take part in every pa_row call; the host splits the page and `pa_fused` (mode 1) takes the
    // leading multiple of 256 rows -- whole groups of 64 quads per wave, all lanes live, 16-byte loads -- while
    // `pa_fused_tail` (mode 2) takes the remaining < 256 rows (or everything when a buffer is unaligned) row by row with a
    // wave-uniform trip count, finished lanes riding along with live == false on a clamped row.  Two entry points keep the
    // tail's code out of the hot loop's register allocation.
    if (ranged) {
        emit_kernel(k.entry, 3);
    }
    else if (variant == V_LDS) {
        emit_kernel(k.entry, 1);
        emit_kernel(k.entry + "_tail", 2);
    }
    else {
        emit_kernel(k.entry, 0);
    }
    if (brow) brow_keys_kernel();
    k.source = src.str();
    return k;
}

void FusedGen::row_filter()
{
    std::string sel = "true";
    if (s.has_filter) {
        GenValue f = gen.emit(s.filter, body);
        sel = f.nullable() ? "(!" + f.n + " && " + f.v + ")" : f.v;  // PageFunctionCompiler.java:539-542
    }
    if (s.join) {
        probe_filter_and_key(sel);
    }
    else {
        body << "const bool sel = live && " << sel << ";\n";
    }
}

// projections used downstream, evaluated once, only for selected rows
const GenValue& FusedGen::proj_value(int j)
{
    auto it = pv.find(j);
    if (it == pv.end()) it = pv.emplace(j, gen.emit(s.proj[j], inner)).first;
    return it->second;
}

const GenValue& FusedGen::key_value(int j)
{
    if (!brow) return proj_value(j);
    auto it = kpv.find(j);
    if (it == kpv.end()) it = kpv.emplace(j, gen.emit(s.proj[j], key_os)).first;
    return it->second;
}

void FusedGen::add_term(int w, const std::string& term)
{
    if ((int)word_terms.size() <= w) word_terms.resize(w + 1);
    word_terms[w].push_back(term);
}

void FusedGen::group_keys()
{
    // BROW: the key words are not computed per row -- the slot is the build position -- but once per group, by pa_brow_keys, from
    // the build columns alone
    const std::vector<int>& gp = brow ? s.join->brow_group_proj : s.group_proj;
    std::ostringstream& kinner = brow ? key_os : inner;
    for (size_t gi = 0; gi < gp.size(); gi++) {
        const GenValue& kv = key_value(gp[gi]);
        const OwnedExpr& pe = s.proj[gp[gi]];
        KeyPart part;
        part.type = kv.type;
        std::string value;  // u64 expression already confined to `bits` bits
        switch (kv.type) {
            case PA_BIGINT:
            case PA_DECIMAL:  // ShortDecimalType: equal values are equal longs
                part.bits = 64;
                value = "(u64)" + kv.v;
                break;
            case PA_INTEGER:
            case PA_DATE:
                part.bits = 32;
                value = "(u64)(u32)(i32)" + kv.v;
                break;
            case PA_REAL:
                part.bits = 32;  // the key's canonical bits: -0 == +0, NaN == NaN (RealType.java:127-140), hashed as RealType hashes them
                value = "(u64)pa_real_key_bits(" + kv.v + ")";
                break;
            case PA_BOOLEAN:
                part.bits = 1;
                value = "(" + kv.v + " ? 1ULL : 0ULL)";
                break;
            case PA_DOUBLE:
                // IS NOT DISTINCT semantics of the group key: -0 == +0, NaN == NaN (DoubleType.java:163-184)
                part.bits = 64;
                value = "((" + kv.v + " == 0.0) ? 0ULL : ((" + kv.v + " != " + kv.v + ") ? 0x7ff8000000000000ULL : (u64)__double_as_longlong(" +
                        kv.v + ")))";
                break;
            case PA_VARCHAR: {
                int ch = pe.is_input_ref() ? pe.node(pe.root).channel : -1;
                if (ch >= 0 && ch < s.n_in && s.short_bound[ch] > 0) {
                    part.bound = s.short_bound[ch];
                    part.bits = 8 * part.bound + 4;
                    value = "(cs" + std::to_string(ch) + " | ((u64)" + kv.len + " << " + std::to_string(8 * part.bound) + "))";
                }
                else {
                    part.bits = 128;
                }
                break;
            }
            default:
                throw Error(PA_ERR_NOT_SUPPORTED, "group key type not supported on device");
        }
        std::string guard = kv.nullable() ? "(" + kv.n + ") ? 0ULL : " : "";
        if (part.bits == 128) {
            // up to 15 bytes in two dedicated words, length in the top byte of the second
            int sh;
            part.word = packer.place(64, &sh);
            int w2 = packer.place(64, &sh);
            PA_REQUIRE(w2 == part.word + 1, PA_ERR_NOT_SUPPORTED, "internal: long VARCHAR key words not adjacent");
            std::string id = "ks" + std::to_string(gi);
            kinner << "u64 " << id << "a = 0, " << id << "b = 0;\n";
            kinner << "if (" << (kv.nullable() ? "!" + kv.n : "true") << ") {\n";
            kinner << "  if (" << kv.len << " > 15) pa_raise(a.err, -3);\n";
            kinner << "  for (i32 b = 0; b < " << kv.len << " && b < 15; b++) {\n";
            kinner << "    if (b < 8) " << id << "a |= (u64)" << kv.v << "[b] << (8 * b); else " << id << "b |= (u64)" << kv.v
                  << "[b] << (8 * (b - 8));\n  }\n";
            kinner << "  " << id << "b |= (u64)" << kv.len << " << 56;\n}\n";
            add_term(part.word, id + "a");
            add_term(part.word + 1, id + "b");
        }
        else {
            part.word = packer.place(part.bits, &part.shift);
            add_term(part.word, "((" + guard + value + ") << " + std::to_string(part.shift) + ")");
        }
        if (kv.nullable()) {
            part.null_word = packer.place(1, &part.null_shift);
            add_term(part.null_word, "((" + kv.n + ") ? " + std::to_string(1ULL << part.null_shift) + "ULL : 0ULL)");
        }
        k.keys.push_back(part);
    }
    k.w = (int)packer.used.size();
    PA_REQUIRE(k.w <= 8, PA_ERR_NOT_SUPPORTED, "group key wider than 8 words");
}

int FusedGen::word(int kind, const std::string& cond, const std::string& val, const std::string& key)
{
    auto it = word_index.find(key);
    if (it != word_index.end()) return it->second;
    words.push_back({kind, cond, val});
    word_index[key] = (int)words.size() - 1;
    return (int)words.size() - 1;
}

// min / max: u64 maximum of an order-preserving image (pa_img_*; min takes the complement), see pa_device.h
std::string FusedGen::minmax_image(const GenValue& x, bool is_min)
{
    std::string img;
    switch (x.type) {
        case PA_BIGINT:
        case PA_INTEGER:
        case PA_DECIMAL:  // ShortDecimalType's comparison is the longs' (one scale)
        case PA_DATE: img = "pa_img_i64((i64)" + x.v + ")"; break;
        case PA_DOUBLE: img = "pa_img_f64(" + x.v + ")"; break;
        case PA_REAL: img = "pa_img_f64((double)" + x.v + ")"; break;  // (float order = order of the widened values)
        case PA_BOOLEAN: img = "(" + x.v + " ? 1ULL : 0ULL)"; break;
        case PA_VARCHAR: img = "pa_img_str7(" + x.v + ", " + x.len + ", a.err)"; break;
        default: throw Error(PA_ERR_NOT_SUPPORTED, "min/max input type not supported on device");
    }
    return is_min ? "(~" + img + ")" : img;
}

std::string FusedGen::wtype(int w) const
{
    return words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64");
}

void FusedGen::accumulator_words()
{
    for (const auto& ag : s.aggs) {
        if (s.step == PA_STEP_FINAL) {
            // combine functions (DoubleSumAggregation.combine, AverageAggregations.combine, CountAggregation.combine,
            // LongSumAggregation.combine: SURVEY a15): counts and sums of the partial states add up
            const GenValue& c = proj_value(ag.input_channel);
            std::string ch = std::to_string(ag.input_channel);
            std::string ccond = c.nullable() ? "(!" + c.n + ")" : "true";
            int cw = word(W_CNT, ccond, c.v, "fcnt|" + ch);
            int vw = -1;
            if (ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_AVG) {
                const GenValue& v = proj_value(ag.input_channel + 1);
                std::string vcond = v.nullable() ? "(" + ccond + " && !" + v.n + ")" : ccond;
                if (v.type == PA_LONG_DECIMAL) {  // the sum half of a DECIMAL state: limbs again (combine = add)
                    const int limbs = decimal_limbs_for_bits(128);
                    for (int l = 0; l < limbs; l++) {
                        const int w = word(W_SUMI, vcond, "pa_dec_limb(" + v.v + ", " + std::to_string(l) + ", " + std::to_string(limbs - 1) + ")", "fdec" + std::to_string(l) + "|" + ch);
                        if (l == 0) vw = w;
                        PA_REQUIRE(w == vw + l, PA_ERR_NOT_SUPPORTED, "internal: the limb words of a DECIMAL sum are not adjacent");
                    }
                    k.agg_words.emplace_back(cw, vw);
                    k.agg_limbs.resize(k.agg_words.size(), 0);
                    k.agg_limbs.back() = limbs;
                    continue;
                }
                vw = word(v.type == PA_DOUBLE ? W_SUMF : W_SUMI, vcond, v.v, "fsum|" + ch);
            }
            else if (ag.fn == PA_AGG_MIN || ag.fn == PA_AGG_MAX) {
                // AbstractMinMaxAggregationFunction.combine: compare-and-update with the other state's value
                const GenValue& v = proj_value(ag.input_channel + 1);
                std::string vcond = v.nullable() ? "(" + ccond + " && !" + v.n + ")" : ccond;
                vw = word(W_MAXU, vcond, minmax_image(v, ag.fn == PA_AGG_MIN), std::string(ag.fn == PA_AGG_MIN ? "fmin|" : "fmax|") + ch);
            }
            k.agg_words.emplace_back(cw, vw);
            continue;
        }
        std::string cond = "true", ckey = "m" + std::to_string(ag.mask_channel);
        if (ag.mask_channel >= 0) {
            const GenValue& m = proj_value(ag.mask_channel);
            PA_REQUIRE(m.type == PA_BOOLEAN, PA_ERR_INVALID_ARGUMENT, "mask channel must be BOOLEAN");
            cond = m.nullable() ? "(!" + m.n + " && " + m.v + ")" : "(" + m.v + ")";  // CompilerOperations.java:65-74
        }
        if (ag.fn == PA_AGG_COUNT_STAR) {
            k.agg_words.emplace_back(word(W_CNT, cond, "1", "cnt|*|" + ckey), -1);
            continue;
        }
        const GenValue& x = proj_value(ag.input_channel);
        std::string xkey = s.proj[ag.input_channel].fingerprint();
        std::string ccond = cond, cntkey = "cnt|*|" + ckey;
        if (x.nullable()) {
            ccond = "(" + cond + " && !" + x.n + ")";
            cntkey = "cnt|" + xkey + "|" + ckey;
        }
        // sum / min / max only ask "was there any input?" -- and a GROUP exists because a row created it: with a non-null
        // input and no mask the answer is always yes, so the group needs no count word (one HBM atomic less per row on the
        // table tier: Q3's sum(revenue) keeps ONE accumulator word).  -1 = "counts as 1" for every consumer of agg_words.
        // (Step.PARTIAL keeps the real count: its [count, value] state channels are part of the boundary, include/presto_amd.h)
        const bool implicit_count = s.step == PA_STEP_SINGLE && !s.group_proj.empty() && ag.mask_channel < 0 && !x.nullable() &&
                                    (ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_MIN || ag.fn == PA_AGG_MAX);
        int cw = implicit_count ? -1 : word(W_CNT, ccond, "1", cntkey);
        int vw = -1;
        if ((ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_AVG) && (x.type == PA_DECIMAL || x.type == PA_LONG_DECIMAL)) {
            // DecimalSumAggregation / DecimalAverageAggregation: the exact sum as limb words, shared between sum(x) and avg(x)
            // (as many limbs as the TYPE's precision needs: |x| < 10^p by the planner's type derivation; the top limb is signed and
            // takes whatever is left of a value that breaks it, up to 63 bits)
            const int limbs = decimal_limbs_for_bits(std::min(x.type == PA_DECIMAL ? 64 : 128, decimal_bits_for_precision(PA_DECIMAL_PRECISION(x.param))));
            for (int l = 0; l < limbs; l++) {
                const int w = word(W_SUMI, ccond, "pa_dec_limb((i128)" + x.v + ", " + std::to_string(l) + ", " + std::to_string(limbs - 1) + ")",
                                   "dec" + std::to_string(l) + "/" + std::to_string(limbs) + "|" + xkey + "|" + ckey);
                if (l == 0) vw = w;
                PA_REQUIRE(w == vw + l, PA_ERR_NOT_SUPPORTED, "internal: the limb words of a DECIMAL sum are not adjacent");
            }
            k.agg_words.emplace_back(cw, vw);
            k.agg_limbs.resize(k.agg_words.size(), 0);
            k.agg_limbs.back() = limbs;
            continue;
        }
        // (REAL inputs: RealSumAggregation / RealAverageAggregation keep a DOUBLE sum of the widened floats -- the same accumulator
        // words as for DOUBLE; the output functions narrow the result)
        if (ag.fn == PA_AGG_SUM && x.type != PA_DOUBLE && x.type != PA_REAL) {
            vw = word(W_SUMI, ccond, x.v, "sumi|" + xkey + "|" + ckey);
        }
        else if (ag.fn == PA_AGG_SUM || ag.fn == PA_AGG_AVG) {
            std::string v = x.type == PA_DOUBLE ? x.v : "((double)" + x.v + ")";  // AverageAggregations.java:34-39
            vw = word(W_SUMF, ccond, v, std::string("sumf|") + (x.type == PA_DOUBLE ? "d|" : "i|") + xkey + "|" + ckey);
        }
        else if (ag.fn == PA_AGG_MIN || ag.fn == PA_AGG_MAX) {
            vw = word(W_MAXU, ccond, minmax_image(x, ag.fn == PA_AGG_MIN), std::string(ag.fn == PA_AGG_MIN ? "min|" : "max|") + xkey + "|" + ckey);
        }
        k.agg_words.emplace_back(cw, vw);
    }
    k.nw = (int)words.size();
    k.agg_limbs.resize(k.agg_words.size(), 0);
    PA_REQUIRE(k.nw > 0 || k.w > 0, PA_ERR_INVALID_ARGUMENT, "aggregation without aggregates and keys");
    if (k.nw == 0) {  // DISTINCT-style group by without aggregates: keep a row count so the kernels stay uniform
        words.push_back({W_CNT, "true", "1"});
        k.nw = 1;
    }
    for (const auto& w : words) k.word_kind.push_back(w.kind);
}

// identity of the state layout (KernelInfo::layout_id)
void FusedGen::state_layout_id()
{
    std::vector<std::string> names(words.size(), "rows");
    for (const auto& kv : word_index) names[(size_t)kv.second] = kv.first;
    std::ostringstream id;
    for (const KeyPart& kp : k.keys) {
        id << kp.type << ',' << kp.word << ',' << kp.shift << ',' << kp.bits << ',' << kp.bound << ',' << kp.null_word << ',' << kp.null_shift << ';';
    }
    id << '#';
    for (size_t w = 0; w < words.size(); w++) id << words[w].kind << ':' << names[w] << ';';
    k.layout_id = id.str();
}

// The row function: pa_row(a, acc, live, row, <column values>) -- behind a probe stage pa_pre / pa_post, and pa_row their
// composition (fused_tier_probe.cpp).
void FusedGen::row_function()
{
    if (s.join) {
        probe_row_frames();
    }
    else {
        src << "__device__ __forceinline__ void pa_row(const PaFusedArgs& a, PaAcc& acc, const bool live, const i32 row" << row_params(ri, layout) << ")\n{\n";
    }
    src << body.str();
    // values needed after the selected-only block are declared up front
    for (int w = 0; w < k.nw; w++) {
        src << "bool u" << w << " = false; " << (words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")) << " x" << w << " = 0;\n";
    }
    if (k.w > 0) src << "u64 key[PA_KW];\n#pragma unroll\nfor (int i = 0; i < PA_KW; i++) key[i] = 0;\n";
    src << "if (sel) {\n" << build_loads.str() << inner.str();
    for (int w = 0; w < k.nw; w++) src << "u" << w << " = " << words[w].cond << "; x" << w << " = " << words[w].val << ";\n";
    if (brow) {
        src << "key[0] = (u64)(u32)jb;\n";
    }
    else {
        for (int i = 0; i < k.w; i++) {
            src << "key[" << i << "] = ";
            for (size_t t = 0; t < word_terms[i].size(); t++) src << (t ? " | " : "") << word_terms[i][t];
            src << ";\n";
        }
    }
    src << "}\n";
    if (variant == V_HASH) hash_accumulate_row();
    else if (variant == V_GLOBAL) global_accumulate_row();
    else if (variant == V_LDS) lds_accumulate_row();
    else if (brow) brow_accumulate_row();
    else table_accumulate_row();
    src << "}\n\n";
}

// the four rows 4q .. 4q + 3 of a thread of the vector loops
void FusedGen::emit_quad(const std::string (&args)[4])
{
    if (s.join) {
        probe_quad(args);
        return;
    }
    for (int r = 0; r < 4; r++) src << "        pa_row(a, acc, true, (i32)(4 * q + " << r << ")" << args[r] << ");\n";
}

void FusedGen::emit_kernel(const std::string& name, int mode)
{
    std::string occupancy;
    if (brow) {  // (measurement switch: ask the register allocator for this many waves per SIMD)
        if (const char* e = getenv("PRESTO_AMD_BROW_WAVES")) occupancy = " __attribute__((amdgpu_waves_per_eu(" + std::to_string(atoi(e)) + ")))";
    }
    src << "extern \"C\" __global__ __launch_bounds__(" << B << ")" << occupancy << " void PA_K(" << name << ")(PaFusedArgs a)\n{\n";
    if (variant == V_GLOBAL) global_kernel_begin();
    else if (variant == V_LDS) lds_kernel_begin();
    else if (variant == V_LDSP) ldsp_kernel_begin();
    else if (variant == V_LDSH) ldsh_kernel_begin();
    else if (variant == V_HASH) hash_kernel_begin();
    else if (brow) brow_kernel_begin();
    else gt_kernel_begin();
    if (mode != 2 && mode != 3) emit_prologue(ri, layout, src);
    if (mode == 3) {
        ranges_loop();
    }
    else if (variant == V_GLOBAL) {
        global_thread_ids();
    }
    else {
        src << "    const i64 t = (i64)blockIdx.x * " << B << " + threadIdx.x, T = (i64)gridDim.x * " << B << ";\n";
    }
    if (mode == 3) {
        // (the loop above)
    }
    else if (mode == 1) lds_head_loop();
    else if (mode == 2) lds_tail_loop();
    else if (variant == V_HASH) hash_tile_loop();
    else page_loop();
    if (variant == V_LDSP) ldsp_partition_loop();
    if (variant == V_GT || variant == V_LDSH) list_loops();
    if (variant == V_LDSH) ldsh_kernel_end();
    if (brow) brow_kernel_end();
    if (gt_like || brow || lds_table) table_counter_flush();
    if (variant == V_GLOBAL) global_kernel_end();
    else if (variant == V_LDS) lds_kernel_end();
    src << "}\n\n";
}

// a table of row ranges, one workgroup per entry at a time (entries of one XCD's workgroups next to each other, as in the page loop
// of the GLOBAL tier)
void FusedGen::ranges_loop()
{
    ColumnNames rn;
    rn.ranged = true;
    const int rw = range_entry_words(s, layout);
    src << "    const u32 bsw = (gridDim.x & 7u) == 0u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;\n"
           "    for (i64 ri = bsw; ri < a.n_ranges; ri += gridDim.x) {\n"
           "      const u64* RT = a.ranges + (u64)ri * " << rw << "ULL;\n";
    int at = 0;
    for (int c = 0; c < s.n_in; c++) {
        if (!ri.used[c]) continue;
        src << "      const void* const RV" << c << " = (const void*)RT[" << at++ << "];\n";
        if (layout[c].type == PA_VARCHAR) src << "      const i32* const RO" << c << " = (const i32*)RT[" << at++ << "];\n";
        if (layout[c].nullable) src << "      const u8* const RNL" << c << " = (const u8*)RT[" << at++ << "];\n";
    }
    src << "      const i64 RN = (i64)(RT[" << at << "] & 0xffffffffULL);\n"
           "      const bool rvec = (RT[" << at << "] >> 32) != 0ULL;\n";
    {
        std::ostringstream pro;
        emit_prologue(ri, layout, pro, rn);
        src << pro.str();
    }
    // whole groups of 64 quads for the wave-level table of the LDS variant (every lane takes part in every row call)
    if (variant == V_LDS) src << "      const i64 nq = rvec ? (RN >> 8) << 6 : 0;\n";
    else src << "      const i64 nq = rvec ? RN >> 2 : 0;\n";
    src << "      for (i64 q = threadIdx.x; q < nq; q += " << B << ") {\n";
    std::string rargs[4];
    emit_vector_loads(ri, layout, src, rargs, rn);
    emit_quad(rargs);
    src << "      }\n";
    if (variant == V_LDS) {
        src << "      for (i64 rb = nq << 2; rb < RN; rb += 64) {\n        const bool live = rb + threadIdx.x < RN;\n"
               "        const i64 r = live ? rb + threadIdx.x : RN - 1;\n        pa_row(a, acc, live, (i32)r" << scalar_args(ri, layout, rn) << ");\n      }\n";
    }
    else {
        src << "      for (i64 r = (nq << 2) + threadIdx.x; r < RN; r += " << B << ") {\n        pa_row(a, acc, true, (i32)r" << scalar_args(ri, layout, rn) << ");\n      }\n";
    }
    src << "    }\n";
}

// one page: quads of rows grid-strided over the threads, then the rows behind the last whole quad one by one
void FusedGen::page_loop()
{
    std::string args[4];
    src << "    const i64 nq = a.vec ? (a.n >> 2) : 0;\n";
    const std::string tail_rows = "    for (i64 r = (nq << 2) + t; r < a.n; r += T) {\n        pa_row(a, acc, true, (i32)r" + scalar_args(ri, layout) + ");" + flush + "\n    }\n";
    bool two_loops = false;
    if (brow) {
        int level = 3;
        if (const char* e = getenv("PRESTO_AMD_BROW_PIPE")) level = atoi(e);  // (measurement switch: 0 = one quad at a time)
        std::vector<VectorVar> vars;
        const bool can = level > 0 && s.join && vector_load_vars(ri, layout, vars);
        if (can && level < 3) {
            brow_pipelined_loop(level);
            src << tail_rows;
            return;
        }
        if (can) {  // the software pipeline works on the key rank index; a lookup source without one takes the plain loop
            src << "    if (a.jrank) {\n";
            brow_pipelined_loop(level);
            src << "    } else {\n";
            two_loops = true;
        }
        brow_wave_loop_head();
    }
    else {
        src << "    for (i64 q = t; q < nq; q += T) {\n";
    }
    emit_vector_loads(ri, layout, src, args);
    emit_quad(args);
    src << "       " << flush << "\n    }\n";
    if (two_loops) src << "    }\n";
    src << tail_rows;
}

void FusedGen::table_counter_flush()
{
    src << "    pa_gt_ctr_flush(acc.gt, acc.tv.count);\n";
}

}  // namespace fused
}  // namespace pa
