// op_filter_project.cpp -- FilterAndProjectOperator on device: one input Page -> at most one output Page
// holding the projections of the selected rows in input order.
//
// Reference path replaced:
//   FilterAndProjectOperator (…/operator/FilterAndProjectOperator.java:37-71)
//   PageProcessor.createWorkProcessor / ProjectSelectedPositions (…/operator/project/PageProcessor.java:111-137, 180-263)
//   generated PageFilter.filter + PageFilter.positionsArrayToSelectedPositions (…/sql/gen/PageFunctionCompiler.java:459-544,
//     …/operator/project/PageFilter.java:27-50)
//   generated PageProjectionWork.process (…/sql/gen/PageFunctionCompiler.java:283-320), InputPageProjection.project
//     (…/operator/project/InputPageProjection.java:57-70: identity = getRegion / copyPositions)
// Batch boundaries of the reference (adaptive 1..8192-row batches, MergePages re-chunking) are not part of the
// result: the device operator emits the whole page's output at once.
//
// Kernels (generated per expression set and column-layout signature):
//   pa_fp_count    filter per row -> 4 selection bits per row quad + selected rows per 1024-row tile
//   (scan)         exclusive scan of the tile counts (scan_kernels.hip)
//   pa_fp_scatter  ranks inside the tile by workgroup scan, evaluates the projections of the selected rows and
//                  writes them (and the ascending positions list) compacted
// Both are HBM-bound: filter columns are read once, projection inputs once, outputs written once.
#include <map>
#include <memory>
#include <mutex>
#include <sstream>

#include "exprgen.hpp"
#include "jit.hpp"
#include "join_source.hpp"
#include "operator.hpp"
#include "rowgen.hpp"
#include "scan_kernels.hpp"

namespace pa {
namespace {

constexpr int kMaxChannels = 32;
constexpr int kMaxBuildChannels = 8;  // PA_MAX_BUILD_CHANNELS
// A workgroup of 256 threads handles kTileQuads x 256 row quads.  More than one quad per thread (fewer, longer workgroups,
// all loads in flight before the first block scan) was measured SLOWER on MI355X -- page -> page over 2^27 rows, 1 / 2 / 4 / 8
// quads: Q6 filter 155 / 146 / 147 / 147 G rows/s, Q1 filter (96 % pass) 106 / 92 / 80 / 79 G, Q3 per step 27.8 / 28.5 / 29.1 /
// 29.0 ms -- so the default stays 1; PRESTO_AMD_FP_QUADS overrides it for experiments.  Also measured and dropped: counting and
// ranking per WAVE (256 rows) with shuffles only, no block-wide scan and no barrier in either kernel -- same box, same run:
// Q6 filter 149 vs 152 G rows/s, Q1 filter 97 vs 100 G, Q3 even; the four times longer count array and its scan cost more
// than the two barriers per tile.
static const int kTileQuads = [] {
    const char* e = getenv("PRESTO_AMD_FP_QUADS");
    const int q = e ? atoi(e) : 1;
    return q >= 1 && q <= 8 ? q : 1;
}();
static const int kTileRows = 1024 * kTileQuads;
// Also measured and dropped: filter (+ probe), ranks and projections in ONE pass over the page -- tiles handed out by a ticket, a
// chained scan with look-back between them (64 predecessors per round), the probe run once and the build positions kept in
// registers.  Q3's orders pipeline (150 M rows, one row in ten selected, probe inside): 3.95 / 3.35 / 3.23 / 3.06 ms with 1 / 2 / 4 /
// 8 row quads per thread against 2.71 ms for count -> scan -> scatter; the customer pipeline 0.73 vs 0.56 ms.  The ticket is a
// same-address device-scope atomic (about 8 ns each, executed at the memory side) and three of four waves idle during the
// look-back; re-reading the selected quads and probing them a second time costs less.

struct FpArgs {  // host mirror of PaFpArgs
    const void* v[kMaxChannels];
    const int32_t* o[kMaxChannels];
    const uint8_t* nl[kMaxChannels];
    void* out_v[kMaxChannels];
    uint8_t* out_nl[kMaxChannels];
    int64_t n;
    int32_t vec;
    int32_t pad;
    uint8_t* sel4;
    int32_t* tile_counts;
    const int32_t* tile_offsets;
    int32_t* positions;
    int32_t* err;
    const uint64_t* dyn_bits;
    int64_t dyn_min;
    uint64_t dyn_range;
    const void* jslots;
    const uint64_t* jbits;
    int64_t jmin;
    uint64_t jrange;
    uint32_t jmask;
    uint32_t jwrap;
    const void* bv[kMaxBuildChannels];
    const uint8_t* bn[kMaxBuildChannels];
    const void* jrank;
    const int32_t* jrank_rows;
};

// Probe stage (FilterAndProject -> LookupJoin (INNER) in one pass; pa_fused_join_create): a row is selected when the filter keeps it
// AND its key finds a build row -- the lookup source has one integer key without duplicates, so at most one -- and the output page
// is [probe output channels (projections), build output channels]: the build columns are "virtual" channels n_in + v of the
// generated code, read at the build position (as in op_fused.hpp's probe stage).
struct FpJoin {
    std::shared_ptr<LookupSourceImpl> ls;
    OwnedExpr key;                  // the probe join key (a projection of the FilterAndProject)
    std::vector<int> build_cols;    // virtual channel n_in + v = ls->cols[build_cols[v]]
    std::vector<int32_t> build_types;
};

struct FpSpec {
    std::shared_ptr<FpJoin> join;
    int n_in = 0;
    std::vector<int32_t> in_types;
    bool has_filter = false;
    OwnedExpr filter;
    std::vector<OwnedExpr> proj;
    int output_mem = PA_MEM_HOST;
    bool output_handover = false;  // pa_filter_project_desc.output_handover
    std::vector<bool> used_channel;
    // the selection comes from outside (dictionary-aware filter): pa_fp_scatter reads sel4, the filter is not evaluated
    bool filter_external = false;
    // channel tested against the existence bitmap of a join's build keys (pa_filter_project_set_dynamic_filter), or -1
    int dyn_channel = -1;
};

struct FpKernelInfo {
    std::string source;
    std::vector<bool> proj_nullable;
};

FpSpec make_fp_spec(const pa_filter_project_desc* d)
{
    PA_REQUIRE(d != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    FpSpec s;
    PA_REQUIRE(d->input_channel_count >= 0 && d->input_channel_count <= kMaxChannels, PA_ERR_NOT_SUPPORTED, "at most 32 input channels");
    PA_REQUIRE(d->projection_count >= 0 && d->projection_count <= kMaxChannels, PA_ERR_NOT_SUPPORTED, "at most 32 projections");
    s.n_in = d->input_channel_count;
    if (s.n_in) s.in_types.assign(d->input_types, d->input_types + s.n_in);
    s.has_filter = d->filter != nullptr;
    if (s.has_filter) {
        s.filter = OwnedExpr::copy(*d->filter);
        PA_REQUIRE(s.filter.root_type() == PA_BOOLEAN, PA_ERR_INVALID_ARGUMENT, "filter must be BOOLEAN");
    }
    for (int32_t j = 0; j < d->projection_count; j++) {
        s.proj.push_back(OwnedExpr::copy(d->projections[j]));
        const OwnedExpr& e = s.proj.back();
        PA_REQUIRE(e.root_type() != PA_VARCHAR || e.is_input_ref(), PA_ERR_NOT_SUPPORTED,
                   "VARCHAR projections other than plain input references are not on the device path");
    }
    s.output_mem = d->output_mem;
    s.output_handover = d->output_handover != 0 && d->output_mem == PA_MEM_DEVICE;
    std::set<int32_t> used;
    if (s.has_filter) s.filter.collect_channels(&used);
    for (const auto& e : s.proj) e.collect_channels(&used);
    s.used_channel.assign(s.n_in, false);
    for (int32_t c : used) {
        PA_REQUIRE(c >= 0 && c < s.n_in, PA_ERR_INVALID_ARGUMENT, "expression references a channel outside the page");
        s.used_channel[c] = true;
    }
    return s;
}

// FilterAndProject(fp) -> LookupJoin(jd over `bridge`): the output page's projections are the probe output channels of the
// FilterAndProject's projections, then the build output channels
FpSpec make_fp_join_spec(const pa_filter_project_desc* fp, const pa_lookup_join_desc* jd, pa_lookup_source* bridge)
{
    PA_REQUIRE(fp != nullptr && jd != nullptr && bridge != nullptr && bridge->impl != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    FpSpec base = make_fp_spec(fp);
    PA_REQUIRE(jd->join_type == PA_JOIN_INNER && jd->filter == nullptr, PA_ERR_NOT_SUPPORTED, "the fused probe is an inner join without a filter function");
    auto js = std::make_shared<FpJoin>();
    js->ls = bridge->impl;
    const LookupSourceImpl& ls = *js->ls;
    PA_REQUIRE(jd->probe_channel_count == (int32_t)base.proj.size(), PA_ERR_INVALID_ARGUMENT, "the probe page is the projection output");
    PA_REQUIRE(jd->join_channel_count == 1 && ls.join_channels.size() == 1, PA_ERR_NOT_SUPPORTED, "the fused probe takes one join key");
    const int kc = jd->probe_join_channels[0];
    PA_REQUIRE(kc >= 0 && kc < (int)base.proj.size(), PA_ERR_INVALID_ARGUMENT, "probe join channel out of range");
    const int32_t kt = base.proj[(size_t)kc].root_type();
    PA_REQUIRE(kt == PA_BIGINT || kt == PA_INTEGER || kt == PA_DATE, PA_ERR_NOT_SUPPORTED, "the fused probe takes a BIGINT / INTEGER / DATE key");
    PA_REQUIRE(kt == ls.cols[(size_t)ls.join_channels[0]].type, PA_ERR_INVALID_ARGUMENT, "probe / build join key types differ");
    js->key = base.proj[(size_t)kc];
    FpSpec s = base;
    s.proj.clear();
    for (int32_t i = 0; i < jd->probe_output_channel_count; i++) {
        const int c = jd->probe_output_channels[i];
        PA_REQUIRE(c >= 0 && c < (int)base.proj.size(), PA_ERR_INVALID_ARGUMENT, "probe output channel out of range");
        s.proj.push_back(base.proj[(size_t)c]);
    }
    for (int col : ls.output_channels) {
        const int32_t t = ls.cols[(size_t)col].type;
        PA_REQUIRE(t != PA_VARCHAR, PA_ERR_NOT_SUPPORTED, "VARCHAR build columns are not carried by the fused probe");
        PA_REQUIRE((int)js->build_cols.size() < kMaxBuildChannels, PA_ERR_NOT_SUPPORTED, "the fused probe carries at most 8 build columns");
        OwnedExpr e;
        pa_expr_node node{};
        node.kind = PA_EXPR_INPUT_REF;
        node.type = t;
        node.channel = s.n_in + (int)js->build_cols.size();
        e.nodes.push_back(node);
        e.strings.emplace_back();
        e.root = 0;
        s.proj.push_back(std::move(e));
        js->build_cols.push_back(col);
        js->build_types.push_back(t);
    }
    PA_REQUIRE(s.proj.size() <= (size_t)kMaxChannels, PA_ERR_NOT_SUPPORTED, "at most 32 output channels");
    s.has_filter = true;  // the probe selects
    s.join = js;
    std::set<int32_t> used;
    if (s.filter.root >= 0 && fp->filter != nullptr) s.filter.collect_channels(&used);
    js->key.collect_channels(&used);
    for (const auto& e : s.proj) e.collect_channels(&used);
    s.used_channel.assign(s.n_in, false);
    for (int32_t c : used) {
        if (c < s.n_in) s.used_channel[(size_t)c] = true;
    }
    return s;
}

FpKernelInfo generate_fp(const FpSpec& s, const std::vector<ChannelLayout>& layout_in)
{
    // (probe stage: the build columns as channels n_in + v; the caller may have appended them already, with their nullability)
    std::vector<ChannelLayout> layout(layout_in.begin(), layout_in.begin() + s.n_in);
    if (s.join) {
        for (size_t v = 0; v < s.join->build_cols.size(); v++) {
            ChannelLayout cl;
            cl.type = s.join->build_types[v];
            cl.nullable = (size_t)s.n_in + v < layout_in.size() ? layout_in[(size_t)s.n_in + v].nullable : true;
            layout.push_back(cl);
        }
    }
    FpKernelInfo k;
    RowInputs ri;
    ri.n_in = s.n_in;
    ri.used = s.used_channel;
    ri.short_bound.assign(s.n_in, 0);
    std::ostringstream src;
    const std::string params = row_params(ri, layout);

    // filter: bool pa_sel(args, row...)
    // probe stage: a kept row is selected when its key finds a build row (a NULL key matches nothing, JoinProbe.java:89-91); the key's
    // bitmap is tested first, inside the probe.  The filter and the key are their own function (pa_keep), so that the kernels can
    // run the probes of a row quad side by side (pa_join_probe4) instead of one dependent chain of loads per row.
    const bool probing = s.join && !s.filter_external;
    src << "__device__ __forceinline__ bool " << (probing ? "pa_keep(const PaFpArgs& a, u64& pa_key" : "pa_sel(const PaFpArgs& a") << params << ")\n{\n";
    if (probing) src << "pa_key = 0ULL;\n";
    if (s.has_filter && !s.filter_external && s.filter.root >= 0) {
        RowCodegen gen(layout, "a.err");
        std::ostringstream body;
        GenValue f = gen.emit(s.filter, body);
        src << body.str() << "const bool keep = " << (f.nullable() ? "(!" + f.n + " && " + f.v + ")" : f.v);  // PageFunctionCompiler.java:539-542
    }
    else {
        src << "const bool keep = true";
    }
    if (s.dyn_channel >= 0 && !s.filter_external) {
        const std::string C = std::to_string(s.dyn_channel);
        src << " && " << (layout[s.dyn_channel].nullable ? "!cn" + C + " && " : "") << "pa_dyn_test(a, (i64)c" << C << ")";
    }
    src << ";\n";
    std::ostringstream key_code;
    std::string key_expr;
    if (s.join) {
        RowCodegen gen(layout, "a.err");
        GenValue kv = gen.emit(s.join->key, key_code);
        key_expr = "(u64)(i64)" + kv.v;
        if (probing) {
            src << "if (!keep) return false;\n" << key_code.str();
            if (kv.nullable()) src << "if (" << kv.n << ") return false;\n";
            src << "pa_key = " << key_expr << ";\nreturn true;\n";
        }
        else {
            src << "return keep;\n";
        }
    }
    else {
        src << "return keep;\n";
    }
    src << "}\n";
    const std::string names = row_param_names(ri, layout);
    if (probing) {
        src << "__device__ __forceinline__ bool pa_sel(const PaFpArgs& a" << params << ")\n{\nu64 pa_key;\n"
            << "if (!pa_keep(a, pa_key" << names << ")) return false;\nreturn pa_join_exists_keyed(a, pa_key);\n}\n";
    }
    if (s.join) {
        // a selected row's key once more (pa_fp_scatter: one row in ten is selected -- cheaper than carrying 4 B per row between the
        // two kernels)
        src << "__device__ __forceinline__ u64 pa_key_of(const PaFpArgs& a" << params << ")\n{\n" << key_code.str() << "return " << key_expr << ";\n}\n";
    }

    // projections of one selected row written at output position `rank`
    src << "__device__ __forceinline__ void pa_out(const PaFpArgs& a, i64 rank, i32 row" << (s.join ? ", i32 jb" : "") << params << ")\n{\n";
    src << "if (a.positions) a.positions[rank] = row;\n";
    if (s.join) {
        // the build columns of the output at the row's build position
        for (size_t v = 0; v < s.join->build_cols.size(); v++) {
            const std::string id = std::to_string(s.n_in + (int)v), V = std::to_string(v);
            const int32_t t = s.join->build_types[v];
            src << "const " << RowCodegen::ctype(t) << " c" << id << " = ";
            if (t == PA_BIGINT) src << "((const i64*)a.bv[" << V << "])[jb];\n";
            else if (t == PA_INTEGER || t == PA_DATE) src << "(i64)((const i32*)a.bv[" << V << "])[jb];\n";
            else if (t == PA_DOUBLE) src << "((const double*)a.bv[" << V << "])[jb];\n";
            else if (t == PA_REAL) src << "((const float*)a.bv[" << V << "])[jb];\n";
            else if (t == PA_BOOLEAN) src << "((const u8*)a.bv[" << V << "])[jb] != 0;\n";
            else throw Error(PA_ERR_NOT_SUPPORTED, "build column type not carried by the fused probe");
            if (layout[(size_t)s.n_in + v].nullable) src << "const bool cn" << id << " = a.bn[" << V << "] != nullptr && a.bn[" << V << "][jb] != 0;\n";
        }
    }
    {
        RowCodegen gen(layout, "a.err");
        std::ostringstream body;
        for (size_t j = 0; j < s.proj.size(); j++) {
            const OwnedExpr& e = s.proj[j];
            if (e.root_type() == PA_VARCHAR) {  // gathered afterwards through the positions list
                k.proj_nullable.push_back(layout[e.node(e.root).channel].nullable);
                continue;
            }
            GenValue v = gen.emit(e, body);
            const char* ct = nullptr;
            std::string val = v.v;
            if (e.root_type() == PA_LONG_DECIMAL) {  // 16 bytes per position in the reference's layout; NULL rows store zero
                body << "pa_ld_write((u64*)a.out_v[" << j << "] + 2 * (i64)rank, " << (v.nullable() ? "(" + v.n + ") ? (i128)0 : " : "") << val << ");\n";
                if (v.nullable()) body << "a.out_nl[" << j << "][rank] = (" << v.n << ") ? (u8)1 : (u8)0;\n";
                k.proj_nullable.push_back(v.nullable());
                continue;
            }
            switch (e.root_type()) {
                case PA_BIGINT:
                case PA_DECIMAL: ct = "i64"; break;
                case PA_INTEGER:
                case PA_DATE: ct = "i32"; val = "(i32)" + val; break;
                case PA_DOUBLE: ct = "double"; break;
                case PA_REAL: ct = "float"; break;
                case PA_BOOLEAN: ct = "u8"; val = "(" + val + " ? (u8)1 : (u8)0)"; break;
                default: throw Error(PA_ERR_NOT_SUPPORTED, "projection type not supported on device");
            }
            // NULL rows store a zero value, as BlockBuilder.appendNull does
            std::string stored = v.nullable() ? "(" + v.n + ") ? (" + ct + ")0 : (" + ct + ")(" + val + ")" : val;
            body << "((" << ct << "*)a.out_v[" << j << "])[rank] = " << stored << ";\n";
            if (v.nullable()) body << "a.out_nl[" << j << "][rank] = (" << v.n << ") ? (u8)1 : (u8)0;\n";
            k.proj_nullable.push_back(v.nullable());
        }
        src << body.str();
    }
    src << "}\n\n";

    std::string vargs[4];
    std::ostringstream vloads, prologue;
    emit_vector_loads(ri, layout, vloads, vargs);
    emit_prologue(ri, layout, prologue);
    const std::string sargs = scalar_args(ri, layout);
    // tile handled by this workgroup: consecutive workgroup ids go round-robin to the 8 XCDs, so inside every window of
    // 2048 tiles the workgroups of one XCD take 256 consecutive tiles (+2-3 % on the page -> page path)
    const std::string tile_decl =
        "    const u32 pa_w = blockIdx.x & ~2047u, pa_i = blockIdx.x & 2047u;\n"
        "    const u32 pa_tile = (pa_w + 2048u <= gridDim.x) ? pa_w + ((pa_i & 7u) << 8) + (pa_i >> 3) : blockIdx.x;\n";

    const std::string J = std::to_string(kTileQuads);
    if (s.has_filter) {
        src << "extern \"C\" __global__ __launch_bounds__(256) void PA_K(pa_fp_count)(PaFpArgs a)\n{\n";
        src << prologue.str();
        src << tile_decl << "    i32 mine = 0;\n";
        src << "#pragma unroll\n    for (int pa_j = 0; pa_j < " << J << "; pa_j++) {\n";
        src << "    const i64 q = ((i64)pa_tile * " << J << " + pa_j) * 256 + threadIdx.x;\n    const i64 row0 = q << 2;\n    u32 bits = 0;\n";
        src << "    if (a.vec && row0 + 4 <= a.n) {\n" << vloads.str();
        if (probing) {
            src << "        bool pa_s[4], pa_hit[4]; u64 pa_k[4];\n";
            for (int r = 0; r < 4; r++) src << "        pa_s[" << r << "] = pa_keep(a, pa_k[" << r << "]" << vargs[r] << ");\n";
            src << "        pa_join_exists4(a, pa_s, pa_k, pa_hit);\n";
            for (int r = 0; r < 4; r++) src << "        if (pa_hit[" << r << "]) bits |= " << (1 << r) << "u;\n";
        }
        else {
            for (int r = 0; r < 4; r++) src << "        if (pa_sel(a" << vargs[r] << ")) bits |= " << (1 << r) << "u;\n";
        }
        src << "    } else {\n        for (int i = 0; i < 4; i++) {\n            const i64 r = row0 + i;\n"
               "            if (r < a.n) { if (pa_sel(a" << sargs << ")) bits |= 1u << i; }\n        }\n    }\n";
        src << "    if (row0 < a.n) a.sel4[q] = (u8)bits;\n    mine += (i32)__popc(bits);\n    }\n";
        src << "    i32 total;\n    (void)pa_block_exclusive_scan_256(mine, &total);\n";
        src << "    if (threadIdx.x == 0) a.tile_counts[pa_tile] = total;\n}\n\n";
    }
    src << "extern \"C\" __global__ __launch_bounds__(256) void PA_K(pa_fp_scatter)(PaFpArgs a)\n{\n";
    src << prologue.str();
    src << tile_decl;
    if (s.has_filter) {
        // the selection bits of all quads first (one load each), then quad by quad: rank inside the tile by a block scan
        src << "    u32 allbits[" << J << "];\n#pragma unroll\n    for (int pa_j = 0; pa_j < " << J << "; pa_j++) {\n"
               "        const i64 q = ((i64)pa_tile * " << J << " + pa_j) * 256 + threadIdx.x;\n"
               "        allbits[pa_j] = (q << 2) < a.n ? (u32)a.sel4[q] : 0u;\n    }\n";
        src << "    i64 pa_base = (i64)a.tile_offsets[pa_tile];\n";
    }
    src << "#pragma unroll\n    for (int pa_j = 0; pa_j < " << J << "; pa_j++) {\n";
    src << "    const i64 q = ((i64)pa_tile * " << J << " + pa_j) * 256 + threadIdx.x;\n    const i64 row0 = q << 2;\n";
    if (s.has_filter) {
        src << "    const u32 bits = allbits[pa_j];\n";
        src << "    i32 total;\n    i64 rank = pa_base + pa_block_exclusive_scan_256((i32)__popc(bits), &total);\n    pa_base += total;\n";
        src << "    if (bits != 0u) {\n";
    }
    else {
        src << "    if (row0 < a.n) {\n    const i64 left = a.n - row0;\n    const u32 bits = left >= 4 ? 15u : ((1u << left) - 1u);\n    i64 rank = row0;\n";
    }
    src << "    if (a.vec && row0 + 4 <= a.n) {\n" << vloads.str();
    const bool carries = s.join && !s.join->build_cols.empty();  // (no build column in the output: the selected rows are the join's output)
    if (s.join && !carries) src << "        const i32 pa_jb[4] = {-1, -1, -1, -1};\n";
    if (carries) {
        src << "        bool pa_s[4]; u64 pa_k[4]; i32 pa_jb[4];\n";
        for (int r = 0; r < 4; r++) {
            src << "        pa_s[" << r << "] = (bits & " << (1 << r) << "u) != 0u;\n        pa_k[" << r << "] = pa_s[" << r << "] ? pa_key_of(a" << vargs[r] << ") : 0ULL;\n";
        }
        src << "        pa_join_probe4(a, pa_s, pa_k, pa_jb);\n";
    }
    for (int r = 0; r < 4; r++) {
        src << "        if (bits & " << (1 << r) << "u) { pa_out(a, rank, (i32)(row0 + " << r << ")" << (s.join ? ", pa_jb[" + std::to_string(r) + "]" : std::string()) << vargs[r] << "); rank++; }\n";
    }
    src << "    } else {\n        for (int i = 0; i < 4; i++) {\n            const i64 r = row0 + i;\n"
           "            if (r < a.n && (bits & (1u << i))) { pa_out(a, rank, (i32)r" << (carries ? ", pa_join_probe_keyed(a, pa_key_of(a" + sargs + "))" : (s.join ? std::string(", -1") : std::string())) << sargs << "); rank++; }\n        }\n    }\n    }\n    }\n}\n";
    k.source = src.str();
    return k;
}

class FilterProjectOperator : public pa_operator {
public:
    explicit FilterProjectOperator(const pa_filter_project_desc* d) : FilterProjectOperator(d, make_fp_spec(d)) {}
    // the probe-stage form: FilterAndProject -> LookupJoin (make_fp_join_spec)
    FilterProjectOperator(const pa_filter_project_desc* d, FpSpec spec) : spec_(std::move(spec)), stream_(d->stream)
    {
        require_device();
        ctl_ = static_cast<int32_t*>(ctl_buf_.ensure(64));  // [0] err [1] selected count
        PA_HIP(hipMemsetAsync(ctl_, 0, 64, stream_.get()));
        h_ctl_ = static_cast<int32_t*>(h_ctl_buf_.ensure(64));
        out_cols_.resize(spec_.proj.size());
        var_first_.assign(spec_.proj.size(), 0);
        var_bytes_.assign(spec_.proj.size(), 0);
        PA_REQUIRE(d->min_output_page_bytes >= 0 && d->min_output_page_rows >= 0 && d->max_output_page_bytes >= 0, PA_ERR_INVALID_ARGUMENT,
                   "MergePages thresholds must not be negative");  // MergePages.java:101-105
        merge_min_bytes_ = d->min_output_page_bytes;
        merge_min_rows_ = d->min_output_page_rows;
        if (d->max_output_page_bytes > 0) merge_max_bytes_ = d->max_output_page_bytes;
        PA_REQUIRE(merge_max_bytes_ >= merge_min_bytes_, PA_ERR_INVALID_ARGUMENT, "maxPageSizeInBytes must be greater or equal than minPageSizeInBytes");
        merging_ = merge_min_bytes_ > 0 || merge_min_rows_ > 0;
        // DictionaryAwarePageFilter (PageFunctionCompiler wraps a filter over a single input channel in it): two more
        // kernel families -- the filter alone, run over a dictionary, and the projections under a selection made elsewhere
        if (spec_.has_filter && !spec_.join && spec_.filter.root >= 0) {
            std::set<int32_t> fc;
            spec_.filter.collect_channels(&fc);
            if (fc.size() == 1) {
                dict_channel_ = *fc.begin();
                dict_spec_ = spec_;
                dict_spec_.proj.clear();
                dict_spec_.used_channel.assign(spec_.n_in, false);
                dict_spec_.used_channel[dict_channel_] = true;
                ext_spec_ = spec_;
                ext_spec_.filter_external = true;
                std::set<int32_t> used;
                for (const auto& e : ext_spec_.proj) e.collect_channels(&used);
                ext_spec_.used_channel.assign(spec_.n_in, false);
                for (int32_t c : used) ext_spec_.used_channel[c] = true;
            }
        }
    }
    ~FilterProjectOperator() override { (void)hipStreamSynchronize(stream_.get()); }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    // probe stage: no page before the build side has published its lookup source (LookupJoinOperator.java:63, 100)
    bool lookup_source_ready() const { return !spec_.join || spec_.join->ls->built.load(); }
    bool needs_input() override { return !finishing_ && !pending_ && !big_queued_ && lookup_source_ready(); }
    bool is_blocked() override { return !finishing_ && !lookup_source_ready(); }

    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(!pending_, PA_ERR_ILLEGAL_STATE, "Operator still holds an output page");
        PA_REQUIRE(page != nullptr, PA_ERR_INVALID_ARGUMENT, "page is null");
        PA_REQUIRE(page->channel_count == spec_.n_in, PA_ERR_INVALID_ARGUMENT, "page channel count does not match the operator's input types");
        if (page->position_count == 0) return;  // PageProcessor.java:113-115
        hipStream_t s = stream_.get();
        const int64_t n = page->position_count;
        const int64_t tiles = (n + kTileRows - 1) / kTileRows;
        // a dictionary / RLE block under a single-channel filter: the filter runs over the dictionary, the rows look it up
        const bool external = dict_channel_ >= 0 && dictionary_filter(page, n, tiles, s);
        const FpSpec& sp = external ? ext_spec_ : spec_;
        in_ = stager_.stage(page, &sp.used_channel, s);
        in_device_ = page->mem == PA_MEM_DEVICE;
        std::vector<ChannelLayout> layout(sp.n_in);
        std::string sig;
        bool vec = true;
        FpArgs a;
        memset(&a, 0, sizeof a);
        bind_inputs(sp, in_, &layout, &sig, &vec, &a);
        if (spec_.join) {
            const LookupSourceImpl& ls = *spec_.join->ls;
            PA_REQUIRE(ls.built.load(), PA_ERR_ILLEGAL_STATE, "probe page before the lookup source was built");
            if (int32_t e = ls.error.load()) throw Error(e, "hash build failed on device");
            PA_REQUIRE(ls.keyed && !ls.has_duplicates, PA_ERR_ILLEGAL_STATE, "internal: fused probe over a lookup source with duplicate keys");
            a.jslots = ls.key_slots.ptr();
            a.jmask = ls.probe_mask;
            a.jwrap = ls.probe_wrap;
            a.jbits = ls.bitmap.bits;
            a.jmin = ls.bitmap.min_key;
            a.jrange = ls.bitmap.range;
            a.jrank = ls.rank.words;
            a.jrank_rows = ls.rank.rows;
            for (size_t v = 0; v < spec_.join->build_cols.size(); v++) {
                const BuildColumn& bc = ls.cols[(size_t)spec_.join->build_cols[v]];
                a.bv[v] = bc.values.ptr();
                a.bn[v] = bc.has_nulls ? bc.nulls.as<uint8_t>() : nullptr;
                ChannelLayout cl;
                cl.type = spec_.join->build_types[v];
                cl.nullable = bc.has_nulls;
                layout.push_back(cl);
                sig += cl.nullable ? 'N' : '_';
            }
        }
        const Compiled& ck = kernel_for(external ? 2 : 0, sig, layout);
        cur_ = &ck;
        a.n = n;
        a.vec = vec ? 1 : 0;
        a.err = ctl_;
        a.dyn_bits = dyn_bits_;
        a.dyn_min = dyn_min_;
        a.dyn_range = dyn_range_;
        need_positions_ = false;
        for (size_t j = 0; j < spec_.proj.size(); j++) {
            OutColumn& oc = out_cols_[j];
            const OwnedExpr& e = spec_.proj[j];
            oc.type = e.root_type();
            oc.varwidth = oc.type == PA_VARCHAR;
            oc.has_nulls = ck.info.proj_nullable[j];
            oc.is_view = false;
            oc.host_ready = false;
            if (oc.varwidth) {
                need_positions_ = need_positions_ || spec_.has_filter;
                continue;
            }
            a.out_v[j] = oc.values.ensure((size_t)n * type_width(oc.type));
            if (oc.has_nulls) a.out_nl[j] = static_cast<uint8_t*>(oc.nulls.ensure((size_t)n));
        }
        if (spec_.has_filter) {
            a.sel4 = static_cast<uint8_t*>(sel4_.ensure((size_t)(n + 3) / 4));
            a.tile_counts = static_cast<int32_t*>(tile_counts_.ensure((size_t)tiles * 4));
            a.positions = static_cast<int32_t*>(positions_.ensure((size_t)n * 4));
            a.tile_offsets = a.tile_counts;  // scanned in place
            void* params[] = {&a};
            timer.set_name(ck.name);
            timer.begin(s);
            if (!external) PA_HIP(hipModuleLaunchKernel(ck.count_fn, (unsigned)tiles, 1, 1, 256, 1, 1, 0, s, params, nullptr));
            launch_exclusive_scan_i32(a.tile_counts, a.tile_counts, tiles, ctl_ + 1, scan_temp_.ensure(scan_temp_bytes(tiles)), s);
            PA_HIP(hipModuleLaunchKernel(ck.scatter_fn, (unsigned)tiles, 1, 1, 256, 1, 1, 0, s, params, nullptr));
            timer.end(s);
        }
        else {
            void* params[] = {&a};
            timer.begin(s);
            if (!spec_.proj.empty()) PA_HIP(hipModuleLaunchKernel(ck.scatter_fn, (unsigned)tiles, 1, 1, 256, 1, 1, 0, s, params, nullptr));
            timer.end(s);
        }
        PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 8, hipMemcpyDeviceToHost, s));
        pending_ = true;
    }

    // MergePages.MergePagesTransformation.process (MergePages.java:112-153) around the page the PageProcessor made
    bool get_output(pa_page* out) override
    {
        hipStream_t s = stream_.get();
        if (big_queued_) {  // :114-119 the big page that followed a flush
            big_queued_ = false;
            publish_output(out_cols_, big_count_, spec_.output_mem, s, out, out_storage_);
            return true;
        }
        if (pending_) {
            int32_t count = 0;
            const bool have = process_pending(&count);
            if (have && !merging_) {
                publish_output(out_cols_, count, spec_.output_mem, s, out, out_storage_);
                if (spec_.output_handover) hand_over(out);
                return true;
            }
            if (have) {
                const int64_t bytes = page_size_in_bytes(count);
                if (count >= merge_min_rows_ || bytes >= merge_min_bytes_) {  // :128
                    if (m_rows_ == 0) {
                        publish_output(out_cols_, count, spec_.output_mem, s, out, out_storage_);
                        return true;
                    }
                    big_queued_ = true;  // :133-138
                    big_count_ = count;
                    return flush_merged(out);
                }
                append_merged(count, bytes);  // :141
                if (m_size_ >= merge_max_bytes_) return flush_merged(out);  // :143-145
            }
        }
        if (finishing_ && m_rows_ > 0) return flush_merged(out);  // :121-124
        return false;
    }

    // pa_filter_project_desc.output_handover: the page's buffers leave the operator with the page (released by whoever took it)
    struct HandedOver {
        std::vector<DevBuf> bufs;
    };
    void hand_over(pa_page* out)
    {
        if (out->mem != PA_MEM_DEVICE) return;
        for (const OutColumn& oc : out_cols_) {
            if (oc.is_view) return;  // a zero-copy view of the input page is the input page's owner's to keep alive: lent as before
        }
        auto* h = new HandedOver;
        for (OutColumn& oc : out_cols_) {
            h->bufs.push_back(std::move(oc.values));
            h->bufs.push_back(std::move(oc.offsets));
            h->bufs.push_back(std::move(oc.nulls));
        }
        out->flags |= PA_PAGE_RETAINED;
        out->release = [](void* ctx) { delete static_cast<HandedOver*>(ctx); };
        out->release_ctx = h;
    }

    // the PageProcessor part: fills out_cols_ for the page added last; false when no row was selected
    bool process_pending(int32_t* out_count)
    {
        pending_ = false;
        hipStream_t s = stream_.get();
        PA_HIP(hipStreamSynchronize(s));
        const int32_t err = h_ctl_[0];
        if (err != 0) {
            PA_HIP(hipMemsetAsync(ctl_, 0, 4, s));
            switch (err) {
                case PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE: throw Error(err, "numeric value out of range (bigint/integer arithmetic overflow)");
                case PA_ERR_DIVISION_BY_ZERO: throw Error(err, "Division by zero");
                default: throw Error(err, "device-side error");
            }
        }
        const int32_t n = in_.n;
        const int32_t count = spec_.has_filter ? h_ctl_[1] : n;
        last_count_ = count;
        last_is_list_ = count != 0 && count != n;  // PageFilter.java:37-39: range when none or all rows pass
        if (count == 0) return false;  // PageProcessor.java:127-129
        for (size_t j = 0; j < spec_.proj.size(); j++) {
            OutColumn& oc = out_cols_[j];
            const OwnedExpr& e = spec_.proj[j];
            const bool identity = e.is_input_ref() && e.node(e.root).channel < spec_.n_in;  // (not a build column of the probe stage)
            if (identity && count == n) {
                // positionsRange(0, n): InputPageProjection returns block.getRegion -> zero copy
                const DevColumn& src = in_.cols[e.node(e.root).channel];
                oc.is_view = true;
                oc.view_values = src.values;
                oc.view_offsets = src.offsets;
                oc.view_nulls = src.nulls;
                oc.has_nulls = src.nulls != nullptr;
                continue;
            }
            if (oc.varwidth) {
                // Block.copyPositions for a VariableWidthBlock: lengths -> exclusive scan -> byte copy
                const DevColumn& src = in_.cols[e.node(e.root).channel];
                int32_t* lens = static_cast<int32_t*>(oc.offsets.ensure((size_t)(count + 1) * 4));
                launch_varwidth_lengths(positions_.as<int32_t>(), count, src.offsets, src.nulls, lens, s);
                launch_exclusive_scan_i32(lens, lens, count, ctl_ + 2, scan_temp_.ensure(scan_temp_bytes(count)), s);
                PA_HIP(hipMemcpyAsync(h_ctl_ + 2, ctl_ + 2, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
                int32_t total = h_ctl_[2];
                uint8_t* bytes = static_cast<uint8_t*>(oc.values.ensure((size_t)(total > 0 ? total : 1)));
                launch_varwidth_copy(positions_.as<int32_t>(), count, src.offsets, static_cast<const uint8_t*>(src.values), src.nulls, lens, bytes,
                                     ctl_ + 2, s);
                if (src.nulls) {
                    launch_gather_nulls(src.nulls, positions_.as<int32_t>(), count, static_cast<uint8_t*>(oc.nulls.ensure((size_t)count)), s);
                    oc.has_nulls = true;
                }
                else {
                    oc.has_nulls = false;
                }
            }
        }
        *out_count = count;
        return true;
    }

    // Page.getSizeInBytes of the output page with the reference's block accounting (see include/presto_amd.h)
    int64_t page_size_in_bytes(int32_t count)
    {
        hipStream_t s = stream_.get();
        int64_t bytes = 0;
        for (size_t j = 0; j < out_cols_.size(); j++) {
            OutColumn& oc = out_cols_[j];
            if (!oc.varwidth) {
                bytes += (int64_t)(type_width(oc.type) + 1) * count;
                continue;
            }
            const int32_t* off = oc.is_view ? oc.view_offsets : oc.offsets.as<int32_t>();
            int32_t ends[2];
            PA_HIP(hipMemcpyAsync(&ends[0], off, 4, hipMemcpyDeviceToHost, s));
            PA_HIP(hipMemcpyAsync(&ends[1], off + count, 4, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            var_first_[j] = ends[0];
            var_bytes_[j] = ends[1] - ends[0];
            bytes += (int64_t)var_bytes_[j] + 5LL * count;
        }
        return bytes;
    }

    // PageBuilder.appendPage (MergePages.java:155-164): the page's blocks behind the buffered ones, on device
    void append_merged(int32_t count, int64_t bytes)
    {
        hipStream_t s = stream_.get();
        if (merge_cols_.size() != out_cols_.size()) {
            merge_cols_.clear();
            merge_cols_.resize(out_cols_.size());
            m_var_bytes_.assign(out_cols_.size(), 0);
        }
        for (size_t j = 0; j < out_cols_.size(); j++) {
            OutColumn& oc = out_cols_[j];
            OutColumn& mc = merge_cols_[j];
            const void* sv = oc.is_view ? oc.view_values : oc.values.ptr();
            const uint8_t* sn = oc.is_view ? oc.view_nulls : (oc.has_nulls ? oc.nulls.as<uint8_t>() : nullptr);
            mc.type = oc.type;
            mc.varwidth = oc.varwidth;
            if (oc.varwidth) {
                const int32_t* so = oc.is_view ? oc.view_offsets : oc.offsets.as<int32_t>();
                int32_t* mo = static_cast<int32_t*>(mc.offsets.reserve_keep((size_t)(m_rows_ + count + 1) * 4, (size_t)(m_rows_ ? m_rows_ + 1 : 0) * 4, s));
                launch_offsets_append(so, count, (int32_t)m_var_bytes_[j], mo + m_rows_, m_rows_ == 0, s);
                char* mv = static_cast<char*>(mc.values.reserve_keep((size_t)(m_var_bytes_[j] + var_bytes_[j] + 1), (size_t)m_var_bytes_[j], s));
                if (var_bytes_[j] > 0)
                    PA_HIP(hipMemcpyAsync(mv + m_var_bytes_[j], static_cast<const char*>(sv) + var_first_[j], (size_t)var_bytes_[j], hipMemcpyDeviceToDevice, s));
                m_var_bytes_[j] += var_bytes_[j];
            }
            else {
                const size_t w = (size_t)type_width(oc.type);
                char* mv = static_cast<char*>(mc.values.reserve_keep((size_t)(m_rows_ + count) * w, (size_t)m_rows_ * w, s));
                PA_HIP(hipMemcpyAsync(mv + (size_t)m_rows_ * w, sv, (size_t)count * w, hipMemcpyDeviceToDevice, s));
            }
            if (sn || mc.has_nulls) {
                uint8_t* mn = static_cast<uint8_t*>(mc.nulls.reserve_keep((size_t)(m_rows_ + count), mc.has_nulls ? (size_t)m_rows_ : 0, s));
                if (!mc.has_nulls && m_rows_ > 0) PA_HIP(hipMemsetAsync(mn, 0, (size_t)m_rows_, s));  // earlier pages had no NULL
                if (sn) PA_HIP(hipMemcpyAsync(mn + m_rows_, sn, (size_t)count, hipMemcpyDeviceToDevice, s));
                else PA_HIP(hipMemsetAsync(mn + m_rows_, 0, (size_t)count, s));
                mc.has_nulls = true;
            }
        }
        m_rows_ += count;
        m_size_ += bytes;
    }

    // pageBuilder.build() + reset(): the buffered rows leave as one page (valid until the next add_input / get_output)
    bool flush_merged(pa_page* out)
    {
        const int32_t n = (int32_t)m_rows_;
        for (auto& mc : merge_cols_) {
            mc.is_view = false;
            mc.host_ready = false;
        }
        publish_output(merge_cols_, n, spec_.output_mem, stream_.get(), out, merge_storage_);
        m_rows_ = 0;
        m_size_ = 0;
        for (auto& mc : merge_cols_) mc.has_nulls = false;  // the pointers of the published page are already taken
        std::fill(m_var_bytes_.begin(), m_var_bytes_.end(), 0);
        return true;
    }

    void finish() override { finishing_ = true; }
    bool is_finished() override { return finishing_ && !pending_ && !big_queued_ && m_rows_ == 0; }
    int64_t memory_bytes() override { return (int64_t)(stager_.bytes() + sel4_.capacity() + positions_.capacity() + tile_counts_.capacity()); }

    // SelectedPositions of the last processed page (after its get_output)
    // The dynamic filter of a join whose probe side this operator feeds (DynamicFilter.getCurrentPredicate as the page source of
    // ScanFilterAndProjectOperator applies it): rows whose `channel` value matches no build key are dropped with the filter.
    void set_dynamic_filter(int channel, const uint64_t* bits, int64_t min_key, uint64_t range, std::shared_ptr<void> keep)
    {
        PA_REQUIRE(!pending_ && compiled_.empty(), PA_ERR_ILLEGAL_STATE, "a dynamic filter must be set before the first page");
        PA_REQUIRE(channel >= 0 && channel < spec_.n_in, PA_ERR_INVALID_ARGUMENT, "dynamic filter channel out of range");
        const int32_t t = spec_.in_types[channel];
        PA_REQUIRE(t == PA_BIGINT || t == PA_INTEGER || t == PA_DATE, PA_ERR_NOT_SUPPORTED, "dynamic filters apply to BIGINT / INTEGER / DATE channels");
        spec_.dyn_channel = channel;
        spec_.has_filter = true;
        spec_.used_channel[channel] = true;
        dict_channel_ = -1;  // (the dictionary-aware path evaluates the filter on dictionaries: not combined with a dynamic filter)
        dyn_bits_ = bits;
        dyn_min_ = min_key;
        dyn_range_ = range;
        dyn_keep_ = std::move(keep);
    }

    void last_positions(const int32_t** dev_positions, int32_t* count, int32_t* is_list) const
    {
        *dev_positions = positions_.as<int32_t>();
        *count = last_count_;
        *is_list = last_is_list_ ? 1 : 0;
    }

private:
    struct Compiled {
        FpKernelInfo info;
        hipFunction_t count_fn = nullptr, scatter_fn = nullptr;
        std::string name;  // of the kernel that reads the page first (pa_op_kernel_name)
    };

    // One code object per (plan, column-layout signature, variant, device), shared by every operator instance of the process: an operator
    // lives for one query (OperatorFactory.createOperator) while the generated code of its plan node does not change -- generating and
    // hashing the source again for every instance cost 80 us per pipeline of Q3 (as in op_fused.hpp).
    static std::mutex& shared_mutex()
    {
        static std::mutex* m = new std::mutex();
        return *m;
    }
    static std::map<std::string, std::shared_ptr<const Compiled>>& shared_cache()
    {
        static auto* c = new std::map<std::string, std::shared_ptr<const Compiled>>();  // leaked: HIP may be gone at exit
        return *c;
    }
    // everything of a spec the generated code depends on (generate_fp)
    static std::string plan_fingerprint(const FpSpec& sp)
    {
        std::ostringstream f;
        f << sp.n_in << '|';
        for (int32_t t : sp.in_types) f << t << ',';
        // (behind a probe stage "has a filter" may mean the probe alone: no expression then)
        f << '|' << (sp.has_filter ? (sp.filter.root >= 0 ? sp.filter.fingerprint() : std::string("+")) : std::string("-")) << '|' << (sp.filter_external ? 'x' : 'i')
          << '|' << sp.dyn_channel << '|';
        for (const OwnedExpr& e : sp.proj) f << e.fingerprint() << '#';
        f << '|';
        for (bool u : sp.used_channel) f << (u ? '1' : '0');
        if (sp.join) {
            f << "|J" << sp.join->key.fingerprint() << ':';
            for (size_t v = 0; v < sp.join->build_cols.size(); v++) f << sp.join->build_cols[v] << '/' << sp.join->build_types[v] << ',';
        }
        f << "|q" << kTileQuads;
        return f.str();
    }

    // variant 0: the operator as described; 1: the filter alone (over a dictionary); 2: the projections under an external selection
    const Compiled& kernel_for(int variant, const std::string& sig, const std::vector<ChannelLayout>& layout)
    {
        const FpSpec& sp = variant == 0 ? spec_ : (variant == 1 ? dict_spec_ : ext_spec_);
        // (the dynamic filter's channel is set after the operator was made: it is part of the key)
        const std::string key = std::to_string(variant) + sig + "|" + std::to_string(sp.dyn_channel);
        auto it = compiled_.find(key);
        if (it != compiled_.end()) return *it->second;
        int dev = 0;
        PA_HIP(hipGetDevice(&dev));
        const std::string shared_key = std::to_string(dev) + "|" + key + "|" + plan_fingerprint(sp);
        {
            std::lock_guard<std::mutex> lock(shared_mutex());
            auto hit = shared_cache().find(shared_key);
            if (hit != shared_cache().end()) {
                compiled_[key] = hit->second;
                return *hit->second;
            }
        }
        auto c = std::make_shared<Compiled>();
        c->info = generate_fp(sp, layout);
        if (variant != 1) {
            JitKernel k = jit_get(c->info.source, "pa_fp_scatter");
            c->scatter_fn = k.fn;
            c->name = k.name;
        }
        if (sp.has_filter && !sp.filter_external) {
            JitKernel k = jit_get(c->info.source, "pa_fp_count");
            c->count_fn = k.fn;
            c->name = k.name;
        }
        {
            std::lock_guard<std::mutex> lock(shared_mutex());
            shared_cache()[shared_key] = c;
        }
        compiled_[key] = c;
        return *c;
    }

    // column pointers, layout signature and alignment of the channels `sp` reads
    static void bind_inputs(const FpSpec& sp, const DevPage& in, std::vector<ChannelLayout>* layout, std::string* sig, bool* vec, FpArgs* a)
    {
        for (int c = 0; c < sp.n_in; c++) {
            ChannelLayout& l = (*layout)[c];
            l.type = sp.used_channel[c] ? in.cols[c].type : sp.in_types[c];
            l.nullable = sp.used_channel[c] && in.cols[c].nulls != nullptr;
            if (sp.used_channel[c]) {
                PA_REQUIRE(in.cols[c].type == sp.in_types[c], PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared input type");
                *vec = *vec && ((uintptr_t)in.cols[c].values % 16 == 0) && ((uintptr_t)in.cols[c].offsets % 16 == 0) &&
                       ((uintptr_t)in.cols[c].nulls % 4 == 0);
                a->v[c] = in.cols[c].values;
                a->o[c] = in.cols[c].offsets;
                a->nl[c] = in.cols[c].nulls;
            }
            *sig += l.nullable ? 'n' : '-';
        }
    }

    // DictionaryAwarePageFilter.filter (DictionaryAwarePageFilter.java:57-83): evaluates the filter on the dictionary (or the
    // RLE value) and maps the verdicts through the ids into sel4_ / tile_counts_ of the page.  False = ordinary processing:
    // not a dictionary block, a dictionary larger than the page, or the dictionary pass raised an error (:105-111 -- an
    // entry no row uses may fail; the ordinary pass then reports the error only if a row really hits it).
    bool dictionary_filter(const pa_page* page, int64_t n, int64_t tiles, hipStream_t s)
    {
        const pa_column& col = page->columns[dict_channel_];
        if ((col.encoding != PA_DICTIONARY && col.encoding != PA_RLE) || col.dictionary == nullptr) return false;
        const pa_column& dict = *col.dictionary;
        if (dict.encoding != PA_FLAT && dict.encoding != PA_VARWIDTH) return false;
        const int64_t dn = col.encoding == PA_RLE ? 1 : col.dictionary_size;
        if (dn <= 0 || dn > n || (col.encoding == PA_DICTIONARY && col.ids == nullptr)) return false;
        std::vector<pa_column> cols((size_t)spec_.n_in);
        cols[dict_channel_] = dict;
        pa_page dpage{};
        dpage.position_count = (int32_t)dn;
        dpage.channel_count = spec_.n_in;
        dpage.columns = cols.data();
        dpage.mem = page->mem;
        DevPage d = dict_stager_.stage(&dpage, &dict_spec_.used_channel, s);
        std::vector<ChannelLayout> layout(spec_.n_in);
        std::string sig;
        bool vec = true;
        FpArgs a;
        memset(&a, 0, sizeof a);
        bind_inputs(dict_spec_, d, &layout, &sig, &vec, &a);
        const Compiled& dk = kernel_for(1, sig, layout);
        const int64_t dtiles = (dn + kTileRows - 1) / kTileRows;
        a.n = dn;
        a.vec = vec ? 1 : 0;
        a.err = ctl_;
        a.sel4 = static_cast<uint8_t*>(dict_sel4_.ensure((size_t)(dn + 3) / 4));
        a.tile_counts = static_cast<int32_t*>(dict_tiles_.ensure((size_t)dtiles * 4));
        void* params[] = {&a};
        PA_HIP(hipModuleLaunchKernel(dk.count_fn, (unsigned)dtiles, 1, 1, 256, 1, 1, 0, s, params, nullptr));
        PA_HIP(hipMemcpyAsync(h_ctl_, ctl_, 4, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        if (h_ctl_[0] != 0) {
            PA_HIP(hipMemsetAsync(ctl_, 0, 4, s));
            return false;
        }
        const int32_t* ids = nullptr;
        if (col.encoding == PA_DICTIONARY) {
            if (page->mem == PA_MEM_DEVICE) ids = col.ids;
            else {
                int32_t* dev = static_cast<int32_t*>(ids_.ensure((size_t)n * 4));
                PA_HIP(hipMemcpyAsync(dev, col.ids, (size_t)n * 4, hipMemcpyHostToDevice, s));
                ids = dev;
            }
        }
        launch_dict_filter_sel(ids, a.sel4, n, static_cast<uint8_t*>(sel4_.ensure((size_t)(n + 3) / 4)),
                               static_cast<int32_t*>(tile_counts_.ensure((size_t)tiles * 4)), kTileQuads, s);
        return true;
    }

    const uint64_t* dyn_bits_ = nullptr;
    int64_t dyn_min_ = 0;
    uint64_t dyn_range_ = 0;
    std::shared_ptr<void> dyn_keep_;  // the lookup source that owns the bitmap
    FpSpec spec_, dict_spec_, ext_spec_;
    int dict_channel_ = -1;  // the filter's only input channel, or -1
    Stream stream_;
    PageStager stager_, dict_stager_;
    DevBuf dict_sel4_, dict_tiles_, ids_;
    std::map<std::string, std::shared_ptr<const Compiled>> compiled_;
    const Compiled* cur_ = nullptr;
    DevPage in_;
    bool in_device_ = false, finishing_ = false, pending_ = false, need_positions_ = false;
    DevBuf ctl_buf_, sel4_, tile_counts_, positions_, scan_temp_;
    PinnedBuf h_ctl_buf_;
    int32_t* ctl_ = nullptr;
    int32_t* h_ctl_ = nullptr;
    int32_t last_count_ = 0;
    bool last_is_list_ = false;
    std::vector<OutColumn> out_cols_;
    std::vector<pa_column> out_storage_;
    // MergePages state
    bool merging_ = false, big_queued_ = false;
    int32_t merge_min_rows_ = 0, big_count_ = 0;
    int64_t merge_min_bytes_ = 0, merge_max_bytes_ = 1 << 20, m_rows_ = 0, m_size_ = 0;
    std::vector<OutColumn> merge_cols_;
    std::vector<pa_column> merge_storage_;
    std::vector<int64_t> m_var_bytes_;
    std::vector<int32_t> var_first_, var_bytes_;
};

}  // namespace

void filter_project_set_dynamic_filter(pa_operator* op, int channel, const uint64_t* bits, int64_t min_key, uint64_t range, std::shared_ptr<void> keep)
{
    auto* fp = dynamic_cast<FilterProjectOperator*>(op);
    PA_REQUIRE(fp != nullptr, PA_ERR_INVALID_ARGUMENT, "not a FilterAndProject operator");
    fp->set_dynamic_filter(channel, bits, min_key, range, std::move(keep));
}

pa_operator* make_filter_project(const pa_filter_project_desc* desc)
{
    return new FilterProjectOperator(desc);
}

pa_operator* make_filter_project_probe(const pa_filter_project_desc* fp, const pa_lookup_join_desc* join, pa_lookup_source* bridge)
{
    PA_REQUIRE(fp != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    return new FilterProjectOperator(fp, make_fp_join_spec(fp, join, bridge));
}

std::string filter_project_source_for_desc(const pa_filter_project_desc* desc, std::string* entry)
{
    FpSpec s = make_fp_spec(desc);
    std::vector<ChannelLayout> layout(s.n_in);
    for (int c = 0; c < s.n_in; c++) layout[c].type = s.in_types[c];
    if (entry) *entry = "pa_fp_scatter";
    return generate_fp(s, layout).source;
}

// the FilterAndProject kernels with the probe inside, over a lookup source shaped like `build` (no device needed)
std::string filter_project_probe_source_for_desc(const pa_fused_join_desc* desc, const pa_hash_builder_desc* build)
{
    PA_REQUIRE(desc != nullptr && build != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
    pa_lookup_source bridge;
    lookup_source_shape_for_desc(build, &bridge);
    FpSpec s = make_fp_join_spec(&desc->filter_project, &desc->join, &bridge);
    std::vector<ChannelLayout> layout(s.n_in);
    for (int c = 0; c < s.n_in; c++) layout[c].type = s.in_types[c];
    return generate_fp(s, layout).source;
}

int32_t filter_project_last_positions(pa_operator* op, const int32_t** dev_positions, int32_t* count, int32_t* is_list)
{
    auto* fp = dynamic_cast<FilterProjectOperator*>(op);
    PA_REQUIRE(fp != nullptr, PA_ERR_INVALID_ARGUMENT, "not a FilterAndProject operator");
    fp->last_positions(dev_positions, count, is_list);
    return PA_OK;
}

}  // namespace pa
