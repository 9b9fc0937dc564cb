// fused_tier_probe.cpp -- the probe stage: FilterAndProjectOperator -> LookupJoinOperator (INNER) -> aggregation as ONE row function
// (JoinProbe.getCurrentJoinPosition, JoinProbe.java:87-117; DefaultPageJoiner.joinCurrentPosition, DefaultPageJoiner.java:236-320)
// in front of any tier -- pa_pre (filter + probe key), the keyed probe (four per quad: pa_join_probe4), pa_post (build columns at
// the match, lazy channels, accumulation) -- and BROW, the tier whose group IS the build row: accumulators indexed by build
// position, no hashing, a per-wave window of build positions in LDS in front of the table.  BROW's page loop over a key rank index is
// a four-stage software pipeline (brow_pipelined_loop): columns, rank words, lazy channels and accumulation of four quads in flight.
#include "decimal_host.hpp"
#include "fused_codegen.hpp"
#include "scan_kernels.hpp"

namespace pa {
namespace fused {

void FusedGen::probe_filter_and_key(const std::string& sel)
{
    // 1b. rows the filter keeps look their key up; a NULL key matches nothing (JoinProbe.java:89-91)
    pre << body.str() << "sel0 = live && " << sel << ";\njk = 0ULL;\nif (sel0) {\n";
    GenValue pk = gen.emit(s.proj[s.join->key_proj], pre);
    if (pk.nullable()) pre << "if (" << pk.n << ") sel0 = false; else ";
    pre << "jk = (u64)(i64)" << pk.v << ";\n}\n";
    body.str("");
    // what follows (pa_post) runs per row with `jb`, the build position of the match -- the lookup source has no duplicate
    // keys, so it is the only one -- or -1
    body << "const bool sel = jb >= 0;\n";
}

void FusedGen::probe_occupancy_word()
{
    // a word every row of a group updates tells whether the build row has a group: a count is then > 0; a DOUBLE sum that
    // starts at -0.0 and only ever takes values canonicalised by + 0.0 (x + 0.0 is x, except that -0.0 becomes +0.0) is then
    // anything but -0.0 -- and equals what the reference computes, whose sum starts at +0.0 (0.0 + -0.0 = 0.0)
    for (size_t w = 0; w < words.size() && k.occ_word < 0; w++) {
        if (words[w].cond != "true" || (words[w].kind != W_CNT && words[w].kind != W_SUMF)) continue;
        k.occ_word = (int)w;
        k.occ_empty = words[w].kind == W_SUMF ? 0x8000000000000000ULL : 0ULL;
    }
    if (getenv("PRESTO_AMD_BROW_TAGS")) k.occ_word = -1;  // (test switch: the tag-storing form)
}

void FusedGen::probe_build_loads()
{
    for (size_t v = 0; v < s.join->build_cols.size(); v++) {
        const std::string id = std::to_string(s.n_in + (int)v), V = std::to_string(v);
        const int32_t t = s.join->build_types[v];
        const std::string ct = RowCodegen::ctype(t);
        build_loads << "const " << ct << " c" << id << " = ";
        if (t == PA_BIGINT) build_loads << "((const i64*)a.bv[" << V << "])[jb];\n";
        else if (t == PA_INTEGER || t == PA_DATE) build_loads << "(i64)((const i32*)a.bv[" << V << "])[jb];\n";
        else if (t == PA_DOUBLE) build_loads << "((const double*)a.bv[" << V << "])[jb];\n";
        else if (t == PA_BOOLEAN) build_loads << "((const u8*)a.bv[" << V << "])[jb] != 0;\n";
        else throw Error(PA_ERR_NOT_SUPPORTED, "build column type not read by the fused probe");
        if (ext[(size_t)s.n_in + v].nullable) build_loads << "const bool cn" << id << " = a.bn[" << V << "] != nullptr && a.bn[" << V << "][jb] != 0;\n";
    }
}

// loads of the lazy channels of one row into the variables c<C><suffix> (cn<C><suffix>)
std::string FusedGen::lazy_assign(const std::string& suffix, const std::string& row)
{
    std::ostringstream o;
    for (int c = 0; c < s.n_in && s.join; c++) {
        if (!s.lazy_channel[c]) continue;
        const std::string C = std::to_string(c);
        const int32_t t = layout[c].type;
        o << "c" << C << suffix << " = ";
        if (t == PA_BIGINT) o << "((const i64*)a.v[" << C << "])[" << row << "]; ";
        else if (t == PA_INTEGER || t == PA_DATE) o << "(i64)((const i32*)a.v[" << C << "])[" << row << "]; ";
        else if (t == PA_DOUBLE) o << "((const double*)a.v[" << C << "])[" << row << "]; ";
        else if (t == PA_BOOLEAN) o << "((const u8*)a.v[" << C << "])[" << row << "] != 0; ";
        else throw Error(PA_ERR_NOT_SUPPORTED, "column type not supported on device");
        if (layout[c].nullable) o << "cn" << C << suffix << " = a.nl[" << C << "] != nullptr && a.nl[" << C << "][" << row << "] != 0; ";
    }
    return o.str();
}

std::string FusedGen::lazy_declare(const std::string& suffix)
{
    std::string d;
    for (int c = 0; c < s.n_in && s.join; c++) {
        if (!s.lazy_channel[c]) continue;
        const std::string C = std::to_string(c);
        d += RowCodegen::ctype(layout[c].type) + " c" + C + suffix + " = 0; ";
        if (layout[c].nullable) d += "bool cn" + C + suffix + " = false; ";
    }
    return d;
}

    // probe stage: pa_pre (filter + key) and pa_post (everything behind the probe) are separate functions, so that the vector loops
    // can probe the four rows of a quad together (pa_join_probe4); pa_row, their row-by-row composition, serves the scalar loops
void FusedGen::probe_row_frames()
{
    for (int c = 0; c < s.n_in; c++) {
        if (!s.lazy_channel[c]) continue;
        const std::string C = std::to_string(c), ct = RowCodegen::ctype(layout[c].type);
        lazy_params += ", const " + ct + " c" + C;
        lazy_names += ", c" + C;
        if (layout[c].nullable) {
            lazy_params += ", const bool cn" + C;
            lazy_names += ", cn" + C;
        }
    }
    src << "__device__ __forceinline__ void pa_pre(const PaFusedArgs& a, const bool live, const i32 row" << row_params(ri, layout)
        << ", bool& sel0, u64& jk)\n{\n" << pre.str() << "}\n\n";
    src << "__device__ __forceinline__ void pa_post(const PaFusedArgs& a, PaAcc& acc, const int slot, const i32 row, const i32 jb" << row_params(ri, layout)
        << lazy_params << ")\n{\n";
}

void FusedGen::probe_row_composition()
{
    src << "__device__ __forceinline__ void pa_row(const PaFusedArgs& a, PaAcc& acc, const bool live, const i32 row" << row_params(ri, layout) << ")\n{\n"
        << "bool sel0; u64 jk;\npa_pre(a, live, row" << row_param_names(ri, layout) << ", sel0, jk);\n"
        << "i32 jb = -1;\nif (sel0) jb = pa_join_probe_keyed(a, jk);\n"
        << lazy_declare("") << "\nif (jb >= 0) { " << lazy_assign("", "row") << "}\n"
        << "pa_post(a, acc, 0, row, jb" << row_param_names(ri, layout) << lazy_names << ");\n}\n\n";
}

void FusedGen::probe_quad(const std::string (&args)[4])
{
    // probe stage: filter and key of the four rows, ONE staged probe for all of them, the lazy channels of the matches
    // (again four loads in flight), then the rows one by one
    src << "        bool js[4]; u64 jk[4]; i32 jb[4];\n";
    for (int r = 0; r < 4; r++) src << "        pa_pre(a, true, (i32)(4 * q + " << r << ")" << args[r] << ", js[" << r << "], jk[" << r << "]);\n";
    src << "        pa_join_probe4(a, js, jk, jb);\n";
    for (int r = 0; r < 4; r++) src << "        " << lazy_declare("_" + std::to_string(r)) << "\n";
    for (int r = 0; r < 4; r++) {
        const std::string R = std::to_string(r);
        src << "        if (jb[" << R << "] >= 0) { " << lazy_assign("_" + R, "4 * q + " + R) << "}\n";
    }
    for (int r = 0; r < 4; r++) {
        const std::string R = std::to_string(r);
        std::string names;
        for (int c = 0; c < s.n_in; c++) {
            if (!s.lazy_channel[c]) continue;
            names += ", c" + std::to_string(c) + "_" + R;
            if (layout[c].nullable) names += ", cn" + std::to_string(c) + "_" + R;
        }
        src << "        pa_post(a, acc, " << R << ", (i32)(4 * q + " << R << "), jb[" << R << "]" << args[r] << names << ");\n";
    }
}

void FusedGen::brow_declarations()
{
    // Build-row table.  Every scattered store / atomic INSTRUCTION of a wave costs the CU on the order of 100 ns whatever the
    // number of active lanes (measured on Q3's lineitem pages: a flush wherever a thread's key changes -- up to five divergent
    // tag-store + atomic sequences per quad -- 1.77 ms per 2^28-row page; one sequence per quad 1.15 ms).  So the rows of a quad
    // are only NOTED (slot r of the thread: build position, flags, values; a row continuing its predecessor's build position
    // takes that one's values over), and at the end of the quad the wave's noted rows -- a dozen of its 256 when 5 % match --
    // are compacted through LDS and go out together: one tag store and one atomic per accumulator word for up to 64 of them.
    // The slot is the build position: nothing to search, nothing to claim.  The tag only says "this build row has a group"
    // -- a plain store into its own array: every writer stores the same value, so the XCD L2s need not agree on the line before
    // the kernel ends; the key words are written once per group by pa_brow_keys.  (One record [tag, words] per build position
    // instead of word-major arrays was 3 x slower: the memory-side atomics of neighbouring build rows share 64-byte requests
    // only while the words of a kind lie side by side.)
    auto wtype = [&](int w) { return std::string(words[w].kind == W_SUMF ? "double" : (words[w].kind == W_MAXU ? "u64" : "i64")); };
    src << "struct PaAcc { PaGtView tv; PaGtCtr gt; bool ev[4]; u32 eg[4];";
    for (int w = 0; w < k.nw; w++) src << " bool eu" << w << "[4]; " << wtype(w) << " ex" << w << "[4];";
    src << " };\n";
    // ... and they do not go out one by one (round 3).  Every wave walks ONE contiguous row range of the page, so when the probe
    // side is clustered by the join key -- a fact table ordered by the key of its dimension, lineitem by orderkey -- the build
    // positions a wave meets rise with its rows.  The wave keeps a WINDOW of PA_WIN consecutive build positions in LDS
    // (accumulator words + one touched bit per position): a noted row inside the window is an LDS atomic (ds_add_f64 / ds_add_u64
    // / ds_max_u64) -- no HBM traffic, no waiting --, a row beyond it first flushes the window and moves it there.  A flush
    // hands the touched positions to the table 64 consecutive positions per instruction: the memory-side atomics of one
    // instruction share a 64-byte request when their addresses are neighbours, so eight build rows go out per request where the
    // sorted drains of round 2 (128 noted rows, bitonic sort, one atomic per distinct position) reached about two -- and the
    // sort is gone.  Windows of different waves overlap only where their row ranges meet, and the flush is atomic, so nothing
    // here depends on the clustering for correctness: rows in no particular order move the window at most PA_WIN_MOVES times
    // per quad and then go to the table directly, one atomic each.
    int win = 256;
    while (win > 64 && (size_t)win * 8 * (size_t)k.nw * 4 > 48 * 1024) win >>= 1;
    src << "#define PA_WIN " << win << "u\n#define PA_WIN_MOVES 2\n";
    src << "__shared__ u64 pa_win[4][PA_NW][PA_WIN];\n__shared__ u64 pa_wtouch[4][PA_WIN / 64u];\n__shared__ u32 pa_sfill[4];\n__shared__ u32 pa_wbase[4];\n";
    src << "#define PA_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); "
           "__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\"); } while (0)\n";
    // noting row `slot` of the quad (a literal at every call site: the arrays stay in registers)
    src << "__device__ __forceinline__ void pa_acc(const PaFusedArgs& a, PaAcc& acc, const int slot, const bool sel, const u64 (&key)[PA_KW]";
    for (int w = 0; w < k.nw; w++) src << ", const bool u" << w << ", const " << wtype(w) << " x" << w;
    src << ")\n{\n  acc.ev[slot] = sel;\n  acc.eg[slot] = (u32)key[0];\n";
    for (int w = 0; w < k.nw; w++) src << "  acc.eu" << w << "[slot] = u" << w << "; acc.ex" << w << "[slot] = x" << w << ";\n";
    src << "  if (slot > 0 && sel && acc.ev[slot > 0 ? slot - 1 : 0] && acc.eg[slot > 0 ? slot - 1 : 0] == acc.eg[slot]) {\n    const int p = slot > 0 ? slot - 1 : 0;\n";
    for (int w = 0; w < k.nw; w++) {
        const std::string P = "acc.ex" + std::to_string(w) + "[p]", X = "acc.ex" + std::to_string(w) + "[slot]";
        std::string comb;
        if (words[w].kind == W_SUMF || words[w].kind == W_CNT) comb = P + " + " + X;
        else if (words[w].kind == W_SUMI) comb = "pa_add_exact(" + P + ", " + X + ", a.err)";
        else comb = "(" + X + " > " + P + " ? " + X + " : " + P + ")";
        src << "    if (acc.eu" << w << "[p]) { " << X << " = acc.eu" << w << "[slot] ? " << comb << " : " << P << "; acc.eu" << w << "[slot] = true; }\n";
    }
    src << "    acc.ev[p] = false;\n  }\n}\n";
    // one value for build position g -> table (the direct route, and the window's flush)
    auto emit_issue = [&](const std::string& ind, const std::string& g, const std::function<std::string(int)>& cond, const std::function<std::string(int)>& val) {
        if (k.occ_word < 0) src << ind << "acc.tv.tag[" << g << "] = 3ULL;\n";
        for (int w = 0; w < k.nw; w++) {
            const std::string W = std::to_string(w), idx = W + "ULL * cap + " + g, v = val(w), c = cond(w);
            src << ind << (c.empty() ? std::string() : "if (" + c + ") ");
            if (words[w].kind == W_SUMF) src << "pa_gt_add_f64(acc.tv.words, " << idx << ", " << v << (w == k.occ_word ? " + 0.0" : "") << ");\n";
            else if (words[w].kind == W_SUMI) src << "pa_gt_add_i64_exact(acc.tv.words, " << idx << ", " << v << ", a.err);\n";
            else if (words[w].kind == W_MAXU) src << "pa_gt_max_u64(acc.tv.words, " << idx << ", " << v << ");\n";
            else src << "pa_gt_add_u64(acc.tv.words, " << idx << ", (u64)" << v << ");\n";
        }
    };
    // the window -> table: lane l takes positions base + 64 k + l; only touched positions issue (and are reset)
    // (the issuing lanes are the ACTIVE ones, by rank: lanes that have left the row loop issue nothing, and the window is
    // complete all the same)
    src << "__device__ __forceinline__ void pa_window_flush(const PaFusedArgs& a, PaAcc& acc)\n{\n"
           "  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;\n  const u64 cap = (u64)a.gt_mask + 1ULL;\n"
           "  const u64 act = __ballot(true);\n  const u32 nact = (u32)__popcll(act), rank = (u32)__popcll(act & ((1ULL << lane) - 1ULL));\n"
           "  PA_WAVE_SYNC();\n"
           "  const u32 wbase = pa_wbase[wave];  // (in LDS: a lane that was not active when the window moved must see where it is)\n"
           "  for (u32 at = rank; at < PA_WIN; at += nact) {\n    const u64 touch = pa_wtouch[wave][at >> 6];\n"
           "    if ((touch >> (at & 63u)) & 1ULL) {\n      const u64 g = (u64)wbase + at;\n";
    emit_issue("      ", "g", [](int) { return std::string(); }, [&](int w) {
        const std::string X = "pa_win[wave][" + std::to_string(w) + "][at]";
        return words[w].kind == W_SUMF ? "__longlong_as_double((i64)" + X + ")" : (words[w].kind == W_MAXU ? X : "(i64)" + X);
    });
    for (int w = 0; w < k.nw; w++) src << "      pa_win[wave][" << w << "][at] = 0ULL;\n";
    src << "    }\n  }\n  PA_WAVE_SYNC();\n  for (u32 i = rank; i < PA_WIN / 64u; i += nact) pa_wtouch[wave][i] = 0ULL;\n  PA_WAVE_SYNC();\n}\n";
    src << "__device__ __forceinline__ void pa_drain(const PaFusedArgs& a, PaAcc& acc, const u32)\n{\n  pa_window_flush(a, acc);\n}\n";
    // end of a quad: the wave's noted rows go into the window, which moves on when they lie beyond it
    src << "__device__ __forceinline__ void pa_flush(const PaFusedArgs& a, PaAcc& acc, const bool)\n{\n"
           "  const u32 wave = threadIdx.x >> 6;\n  const u64 cap = (u64)a.gt_mask + 1ULL;\n"
           "  if (__ballot(acc.ev[0] || acc.ev[1] || acc.ev[2] || acc.ev[3]) == 0ULL) return;\n"
           "  u32 wbase = pa_wbase[wave];\n"
           "  for (int moves = 0;; moves++) {\n"
           "#pragma unroll\n    for (int e = 0; e < 4; e++) {\n      const u32 at = acc.eg[e] - wbase;\n      if (acc.ev[e] && at < PA_WIN) {\n";
    for (int w = 0; w < k.nw; w++) {
        const std::string W = std::to_string(w), L = "pa_win[wave][" + W + "][at]", X = "acc.ex" + W + "[e]";
        src << "        if (acc.eu" << W << "[e]) ";
        if (words[w].kind == W_SUMF) src << "__hip_atomic_fetch_add((double*)&" << L << ", " << X << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
        else if (words[w].kind == W_SUMI) {
            src << "{ const i64 o = (i64)__hip_atomic_fetch_add(&" << L << ", (u64)" << X << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); i64 r; "
                   "if (__builtin_add_overflow(o, " << X << ", &r)) pa_raise(a.err, PA_DEV_ERR_OUT_OF_RANGE); }\n";
        }
        else if (words[w].kind == W_MAXU) src << "__hip_atomic_fetch_max(&" << L << ", " << X << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
        else src << "__hip_atomic_fetch_add(&" << L << ", (u64)" << X << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
    }
    src << "        __hip_atomic_fetch_or(&pa_wtouch[wave][at >> 6], 1ULL << (at & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
           "        acc.ev[e] = false;\n      }\n    }\n"
           "    const bool left = acc.ev[0] || acc.ev[1] || acc.ev[2] || acc.ev[3];\n"
           "    if (__ballot(left) == 0ULL) break;\n"
           "    if (moves >= PA_WIN_MOVES) {\n"
           // rows in no particular order: the rest of the quad goes to the table directly
           "#pragma unroll\n      for (int e = 0; e < 4; e++) {\n        if (!acc.ev[e]) continue;\n        const u64 g = (u64)acc.eg[e];\n";
    emit_issue("        ", "g", [](int w) { return "acc.eu" + std::to_string(w) + "[e]"; }, [&](int w) { return "acc.ex" + std::to_string(w) + "[e]"; });
    src << "        acc.ev[e] = false;\n      }\n      break;\n    }\n"
           // the window moves to the smallest position still waiting (down to a 64-byte line of the table's word arrays)
           // (through LDS: a shuffle would read the registers of lanes that have left the loop)
           "    u32 gmin = 0xffffffffu;\n"
           "#pragma unroll\n    for (int e = 0; e < 4; e++) { if (acc.ev[e] && acc.eg[e] < gmin) gmin = acc.eg[e]; }\n"
           "    pa_sfill[wave] = 0xffffffffu;\n    PA_WAVE_SYNC();\n"
           "    if (left) __hip_atomic_fetch_min(&pa_sfill[wave], gmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
           "    PA_WAVE_SYNC();\n    gmin = pa_sfill[wave];\n"
           "    pa_window_flush(a, acc);\n    wbase = gmin & ~7u;\n    pa_wbase[wave] = wbase;\n    PA_WAVE_SYNC();\n  }\n}\n\n";
}

void FusedGen::brow_accumulate_row()
{
    src << "pa_acc(a, acc, slot, sel, key";
    for (int w = 0; w < k.nw; w++) src << ", u" << w << ", x" << w;
    src << ");\n";
}

void FusedGen::brow_kernel_begin()
{
    src << "    PaAcc acc; acc.tv = pa_gt_view(a, PA_KW, PA_NW); acc.gt = pa_gt_ctr_init(acc.tv.count, true, a.gt_rep_mask + 1u);\n"
           "#pragma unroll\n    for (int e = 0; e < 4; e++) acc.ev[e] = false;\n"
           "    if ((threadIdx.x & 63u) == 0u) pa_wbase[threadIdx.x >> 6] = 0u;\n"
           "    for (u32 i = threadIdx.x & 63u; i < PA_NW * PA_WIN; i += 64u) (&pa_win[threadIdx.x >> 6][0][0])[i] = 0ULL;\n"
           "    if ((threadIdx.x & 63u) < PA_WIN / 64u) pa_wtouch[threadIdx.x >> 6][threadIdx.x & 63u] = 0ULL;\n"
           "    if ((threadIdx.x & 63u) == 0u) pa_sfill[threadIdx.x >> 6] = 0u;\n    PA_WAVE_SYNC();\n";
}

void FusedGen::brow_wave_loop_head()
{
    // every wave walks ONE contiguous range of the page (its loads stay coalesced: 64 lanes x 16 B per instruction).
    // When the probe side is clustered by the join key, the rows a wave notes then belong to neighbouring build rows,
    // and a drained buffer reaches the table as a few dense 64-byte requests -- tags as whole lines, eight adds per atomic
    // request -- instead of one read-modify-write in HBM per group
    src << "    const i64 gw = (i64)blockIdx.x * " << (B / 64) << " + (threadIdx.x >> 6), nwv = (i64)gridDim.x * " << (B / 64) << ";\n"
           "    const i64 per = (((nq + nwv - 1) / nwv) + 63) & ~(i64)63;\n"
           "    const i64 q1 = (gw + 1) * per < nq ? (gw + 1) * per : nq;\n"
           "    for (i64 q = gw * per + (threadIdx.x & 63); q < q1; q += 64) {\n";
}

// The BROW page loop as a software pipeline (round 4).  One iteration of the plain loop is a chain of dependent round trips -- the
// quad's columns (HBM), the rank words of its keys (L2), the lazy channels of its matches (HBM again: a few scattered lines per wave
// and iteration, but some lane matches in every iteration), the window in LDS -- and at four waves per SIMD the chain, not the
// bandwidth, was the kernel's time: waves waited 77 % of their cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES) at 3.1 TB/s on Q3's lineitem pages.
// Here a lane has four quads in flight, one per stage, and an iteration runs the stages youngest data last:
//   S3  accumulates quad i - 2 (pa_post, pa_flush) -- its lazy channels were asked for an iteration ago;
//   S2  reads the rank words of quad i - 1 (asked for an iteration ago) and asks for the lazy channels of its matches;
//   S1  takes the columns of quad i (asked for an iteration ago), filters, and asks for the rank words of its keys;
//   S0  asks for the columns of quad i + 1.
// Every load in the loop is unconditional (a row without a match reads the lazy channels of the range's first row, a row that does
// not probe reads rank word 0 -- one hot line each): loads come back in issue order and the compiler counts them, so each stage waits
// for exactly what was issued an iteration ago and leaves the younger loads in flight.  Quads beyond the range are loaded from its
// last quad and take part as rows the filter drops, so the loop has no prologue or epilogue: it just runs two iterations longer.
// `level` 1: only the columns of the next quad are loaded ahead (the rows are accumulated in their own iteration).
bool FusedGen::brow_pipelined_loop(int level)
{
    std::vector<VectorVar> vars;
    if (!s.join || !vector_load_vars(ri, layout, vars)) return false;
    auto lazy_vars = [&]() {
        std::vector<std::string> names;
        for (int c = 0; c < s.n_in; c++) {
            if (!s.lazy_channel[c]) continue;
            names.push_back("c" + std::to_string(c));
            if (layout[c].nullable) names.push_back("cn" + std::to_string(c));
        }
        return names;
    };
    auto lazy_names_of = [&](const std::string& suffix) {
        std::string names;
        for (const std::string& n : lazy_vars()) names += ", " + n + suffix;
        return names;
    };
    src << "    const i64 gw = (i64)blockIdx.x * " << (B / 64) << " + (threadIdx.x >> 6), nwv = (i64)gridDim.x * " << (B / 64) << ";\n"
           "    const i64 per = (((nq + nwv - 1) / nwv) + 63) & ~(i64)63;\n"
           "    const i64 q1 = (gw + 1) * per < nq ? (gw + 1) * per : nq;\n"
           "    if (gw * per < q1) {\n"   // (wave-uniform)
           "      i64 q = gw * per + (threadIdx.x & 63);\n"
           "      const i64 qf = q < q1 ? q : q1 - 1;\n";
    for (const VectorVar& v : vars) src << "      " << v.type << " n" << v.name << " = " << v.load("qf") << ";\n";
    if (level < 3) {
        src << "      for (; q < q1; q += 64) {\n";
        for (const VectorVar& v : vars) src << "        const " << v.type << " " << v.name << " = n" << v.name << ";\n";
        src << "        bool js[4]; u64 jk[4]; i32 jb[4];\n";
        for (int r = 0; r < 4; r++) {
            src << "        pa_pre(a, true, (i32)(4 * q + " << r << ")" << vector_var_args(ri, layout, "", r) << ", js[" << r << "], jk[" << r << "]);\n";
        }
        src << "        {\n          const i64 qn = q + 64 < q1 ? q + 64 : q;\n";
        for (const VectorVar& v : vars) src << "          n" << v.name << " = " << v.load("qn") << ";\n";
        src << "        }\n        pa_join_probe4(a, js, jk, jb);\n";
        for (int r = 0; r < 4; r++) src << "        " << lazy_declare("_" + std::to_string(r)) << "\n";
        for (int r = 0; r < 4; r++) {
            const std::string R = std::to_string(r);
            src << "        if (jb[" << R << "] >= 0) { " << lazy_assign("_" + R, "4 * q + " + R) << "}\n";
        }
        for (int r = 0; r < 4; r++) {
            const std::string R = std::to_string(r);
            src << "        pa_post(a, acc, " << R << ", (i32)(4 * q + " << R << "), jb[" << R << "]" << vector_var_args(ri, layout, "", r) << lazy_names_of("_" + R) << ");\n";
        }
        src << "        pa_flush(a, acc, true);\n      }\n    }\n";
        return true;
    }
    // state between the stages: d = the quad whose rank words are in flight, p = the quad whose lazy channels are
    src << "      const i64 n_it = (q1 - gw * per + 63) >> 6;\n"
           "      const i64 rs = 4 * (gw * per);\n"   // (the row whose lazy channels stand in for rows without a match)
           "      bool djs[4] = {false, false, false, false}; u64 djk[4] = {0ULL, 0ULL, 0ULL, 0ULL};\n"
           "      PaRank4 dw; dw.lo = pa_u32x4{0u, 0u, 0u, 0u}; dw.hi = dw.lo; dw.below = dw.lo;\n"
           "      i64 dq = qf, pq = qf; i32 pjb0 = -1, pjb1 = -1, pjb2 = -1, pjb3 = -1;\n";
    for (const VectorVar& v : vars) src << "      " << v.type << " d" << v.name << " = n" << v.name << ", p" << v.name << " = n" << v.name << ";\n";
    for (int r = 0; r < 4; r++) src << "      " << lazy_declare("_p" + std::to_string(r)) << "\n";
    src << "      for (i64 it = 0; it < n_it + 2; it++, q += 64) {\n";
    // S3
    for (int r = 0; r < 4; r++) {
        const std::string R = std::to_string(r);
        src << "        pa_post(a, acc, " << R << ", (i32)(4 * pq + " << R << "), pjb" << R << vector_var_args(ri, layout, "p", r) << lazy_names_of("_p" + R) << ");\n";
    }
    src << "        pa_flush(a, acc, true);\n";
    // S2
    src << "        {\n          i32 jb[4];\n          pa_join_rank4_read(a, djs, djk, dw, jb);\n"
           "          pjb0 = jb[0]; pjb1 = jb[1]; pjb2 = jb[2]; pjb3 = jb[3]; pq = dq;\n";
    for (const VectorVar& v : vars) src << "          p" << v.name << " = d" << v.name << ";\n";
    for (int r = 0; r < 4; r++) {
        const std::string R = std::to_string(r);
        src << "          " << lazy_assign("_p" + R, "(pjb" + R + " >= 0 ? 4 * pq + " + R + " : rs)") << "\n";
    }
    src << "        }\n";
    // S1
    src << "        {\n";
    for (const VectorVar& v : vars) src << "          const " << v.type << " " << v.name << " = n" << v.name << ";\n";
    src << "          const bool live = q < q1;\n";
    for (int r = 0; r < 4; r++) {
        src << "          pa_pre(a, live, (i32)(4 * q + " << r << ")" << vector_var_args(ri, layout, "", r) << ", djs[" << r << "], djk[" << r << "]);\n";
    }
    src << "          pa_join_rank4_issue(a, djs, djk, dw);\n          dq = live ? q : qf;\n";
    for (const VectorVar& v : vars) src << "          d" << v.name << " = " << v.name << ";\n";
    src << "        }\n";
    // S0
    src << "        {\n          const i64 qn = q + 64 < q1 ? q + 64 : q1 - 1;\n";
    for (const VectorVar& v : vars) src << "          n" << v.name << " = " << v.load("qn") << ";\n";
    src << "        }\n      }\n    }\n";
    return true;
}

void FusedGen::brow_kernel_end()
{
    src << "    pa_drain(a, acc, pa_sfill[threadIdx.x >> 6]);\n";  // (all lanes are back together behind the row loops)
}

void FusedGen::brow_keys_kernel()
{
    // key words of the groups, once per group: build row b has a group when its tag is set; its key is a function of the
    // build columns (the probe join key equals the build key column on every match)
    src << "extern \"C\" __global__ __launch_bounds__(256) void PA_K(pa_brow_keys)(PaFusedArgs a)\n{\n"
           "    const i64 cap = (i64)a.gt_mask + 1;\n"
           "    i64 found = 0;\n"
           "    for (i64 b = (i64)blockIdx.x * 256 + threadIdx.x; b < cap; b += (i64)gridDim.x * 256) {\n"
        << (k.occ_word < 0 ? std::string("        if (a.gt_tag[b] == 0ULL) continue;\n")
                           : "        if (a.gt_words[" + std::to_string(k.occ_word) + "ULL * (u64)cap + (u64)b] == " + std::to_string(k.occ_empty) + "ULL) continue;\n")
        << "        found++;\n        const i32 jb = (i32)b;\n";
    src << build_loads.str() << key_os.str();
    for (int i = 0; i < k.w; i++) {
        src << "        a.gt_keys[(u64)b * PA_TW + " << i << "] = ";
        for (size_t t = 0; t < word_terms[i].size(); t++) src << (t ? " | " : "") << word_terms[i][t];
        src << ";\n";
    }
    // (the groups are counted on the way: one atomic per wave on the table's group counter)
    src << "    }\n    found = pa_wave_sum_i64(found);\n    if ((threadIdx.x & 63) == 0 && found != 0) atomicAdd(a.gt_count, (i32)found);\n}\n\n";
}

}  // namespace fused
}  // namespace pa
