// fused_codegen.hpp -- the generator of the fused operator's kernels, as ONE object whose methods live in one translation unit per
// kernel tier:
//   fused_codegen.cpp      what every tier shares: filter, projections, key packing, accumulator words, the row function's frame,
//                          the row loops (page, list, range table) and the kernel frame
//   fused_tier_global.cpp  GLOBAL  no group keys: register accumulators -> one partial state per workgroup
//   fused_tier_lds.cpp     LDS     a handful of groups: wave-level key table, lane-private accumulators in LDS
//   fused_tier_ldsh.cpp    LDSH    the workgroup's open-addressing table in LDS, flushed into the HBM table at the end of the launch
//   fused_tier_ldsp.cpp    LDSP    partition-owned LDS tables, and HASH, the partition-id / histogram pass in front of them
//   fused_tier_gt.cpp      GT      the HBM table: upsert + atomics, thread-private run combining, spill lists
//   fused_tier_probe.cpp   the probe stage in front of any tier (pa_pre / pa_post, lazy channels, four probes per quad) and BROW,
//                          the tier whose table slot is the build position
// A tier's file holds the text only that tier's kernels contain; tests/test_codegen_tiers.py compiles every tier x nullability
// signature x key layout for gfx950 without a GPU, scripts/dump_codegen.py writes the sources out (a refactoring changes none).
#pragma once

#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "fused_plan.hpp"
#include "rowgen.hpp"

namespace pa {
namespace fused {

// first-fit bit packing of the key parts into 64-bit words
struct KeyPacker {
    std::vector<int> used;  // bits used per word
    int place(int bits, int* shift)
    {
        for (size_t w = 0; w < used.size(); w++) {
            if (used[w] + bits <= 64) {
                *shift = used[w];
                used[w] += bits;
                return (int)w;
            }
        }
        used.push_back(bits);
        *shift = 0;
        return (int)used.size() - 1;
    }
};

// one accumulator word of the group state: its kind, when a row updates it, with what
struct WordDef {
    int kind;
    std::string cond;
    std::string val;
};

struct FusedGen {
    FusedGen(const Spec& spec, const std::vector<ChannelLayout>& page_layout, int requested_variant);
    KernelInfo run();

    // ---- state (names as the generated text's comments use them) ----
    const Spec& s;
    const std::vector<ChannelLayout>& layout;
    int variant;                       // the BASE tier (V_GLOBAL_R / V_LDS_R: V_GLOBAL / V_LDS with `ranged`)
    bool ranged;
    KernelInfo k;
    RowInputs ri;
    std::vector<ChannelLayout> ext;    // the page's channels, then the build columns of the probe stage as channels n_in + v
    RowCodegen gen;
    std::ostringstream body;           // inside pa_row (pa_post behind a probe stage)
    std::ostringstream pre;            // probe stage: body of pa_pre (filter, then the probe key of the rows it keeps)
    std::ostringstream inner;          // projections used downstream, evaluated once, only for selected rows
    std::ostringstream key_os;         // BROW: the key expressions, evaluated once per group by pa_brow_keys
    std::ostringstream build_loads;    // probe stage: the build columns at the match
    std::ostringstream src;            // the translation unit
    std::map<int, GenValue> pv, kpv;
    KeyPacker packer;
    std::vector<std::vector<std::string>> word_terms;  // per key word: OR-ed terms
    std::vector<WordDef> words;
    std::map<std::string, int> word_index;
    bool brow = false, lds_table = false, gt_like = false;
    std::string lazy_params, lazy_names;
    int B = 256;                       // workgroup size
    std::string flush;                 // what ends a thread's pending run behind a quad / a row (GT, BROW), or nothing

    // ---- shared (fused_codegen.cpp) ----
    const GenValue& proj_value(int j);
    const GenValue& key_value(int j);
    void add_term(int word, const std::string& term);
    int word(int kind, const std::string& cond, const std::string& val, const std::string& key);
    std::string minmax_image(const GenValue& x, bool is_min);
    std::string wtype(int w) const;   // C type of accumulator word w
    void row_filter();
    void group_keys();
    void accumulator_words();
    void state_layout_id();
    void row_function();
    void emit_quad(const std::string (&args)[4]);
    void emit_kernel(const std::string& name, int mode);
    void ranges_loop();
    void page_loop();
    void table_counter_flush();

    // ---- GLOBAL (fused_tier_global.cpp) ----
    void global_declarations();
    void global_accumulate_row();
    void global_kernel_begin();
    void global_thread_ids();
    void global_kernel_end();

    // ---- LDS (fused_tier_lds.cpp) ----
    void lds_check_capacity();
    void lds_declarations();
    void lds_accumulate_row();
    void lds_kernel_begin();
    void lds_head_loop();
    void lds_tail_loop();
    void lds_kernel_end();

    // ---- LDSH and what it shares with LDSP: the workgroup's table in LDS (fused_tier_ldsh.cpp) ----
    void lds_table_size();
    void lds_table_declarations();
    void lds_table_accumulate_begin();
    void ldsh_kernel_begin();
    void ldsh_kernel_end();

    // ---- LDSP + HASH (fused_tier_ldsp.cpp) ----
    void hash_declarations();
    void hash_accumulate_row();
    void hash_kernel_begin();
    void hash_tile_loop();
    void ldsp_kernel_begin();
    void ldsp_partition_loop();

    // ---- GT (fused_tier_gt.cpp) ----
    void gt_declarations();
    void table_accumulate();
    void gt_run_combining();
    void table_accumulate_row();
    void gt_kernel_begin();
    void list_loops();

    // ---- probe stage + BROW (fused_tier_probe.cpp) ----
    void probe_filter_and_key(const std::string& sel);
    void probe_occupancy_word();
    void probe_build_loads();
    std::string lazy_assign(const std::string& suffix, const std::string& row);
    std::string lazy_declare(const std::string& suffix);
    void probe_row_frames();
    void probe_row_composition();
    void probe_quad(const std::string (&args)[4]);
    void brow_declarations();
    void brow_accumulate_row();
    void brow_kernel_begin();
    void brow_wave_loop_head();
    bool brow_pipelined_loop(int level);
    void brow_kernel_end();
    void brow_keys_kernel();
};

}  // namespace fused
}  // namespace pa
