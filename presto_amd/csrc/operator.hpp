// operator.hpp -- the Operator protocol of the reference, as seen from behind the C ABI.
// (core/trino-main/src/main/java/io/trino/operator/Operator.java:21-103; call order as driven by
//  Driver.processInternal, core/trino-main/src/main/java/io/trino/operator/Driver.java:355-457)
#pragma once

#include <memory>

#include "common.hpp"
#include "device_page.hpp"

struct pa_operator {
    virtual ~pa_operator() = default;
    virtual bool needs_input() = 0;
    virtual void add_input(const pa_page* page) = 0;
    virtual bool get_output(pa_page* out) = 0;
    virtual void finish() = 0;
    virtual bool is_finished() = 0;
    virtual bool is_blocked() { return false; }
    virtual int64_t memory_bytes() { return 0; }
    virtual void close() {}
    // the operator's stream when the library owns it (desc.stream == NULL), else nullptr: nobody outside can order work
    // against an owned stream, so device output pages are completed before pa_op_get_output returns them
    virtual hipStream_t private_stream() { return nullptr; }
    // the stream the operator enqueues its work on: pooled HBM blocks released inside one of its calls are tagged with it and
    // re-granted to another stream only once this one has drained (pool.cpp)
    virtual hipStream_t main_stream() { return nullptr; }
    // PA_PAGE_RETAINED pages: true = the operator keeps reading such a page after add_input returns and calls its release itself, once
    // nothing reads it any more (and, at the latest, when it is destroyed); false = pa_op_add_input releases the page when the call
    // returns, after the operator's own stream has drained
    virtual bool takes_retained() { return false; }
    // pa_aggregation_set_output_topn_hint: the only consumer of this operator's output is a TopN(n; sort channels / orders over the
    // output channels); false = the hint is not taken (everything is emitted)
    virtual bool set_output_topn(int64_t, const int32_t*, const int32_t*, int32_t) { return false; }
    // HIP device the operator was created on: every C-ABI entry rebinds the calling thread to it (Trino's Driver threads never
    // call pa_init; a handle created on one thread may be driven and closed on others)
    int device = -1;
    pa_operator() { if (hipGetDevice(&device) != hipSuccess) device = -1; }
    pa::KernelTimer timer;
    // the stopwatch pa_op_kernel_time reads (an operator made of operators names the one of its dominant kernel)
    virtual pa::KernelTimer& kernel_timer() { return timer; }
};

namespace pa {

pa_operator* make_fused_aggregation(const pa_fused_aggregation_desc* desc);
// (Hash)AggregationOperator over plain channels (abi.cpp: the fused operator with identity projections, or the reference-state adapter)
pa_operator* make_hash_aggregation(const pa_hash_aggregation_desc* desc);
// FilterAndProject -> LookupJoin -> aggregation (op_fused_join.cpp); make_fused_probe_aggregation is its one-kernel form
// (op_fused.hpp), valid only for keyed lookup sources without duplicate keys -- probe_source_is_unique, once built
pa_operator* make_fused_join_aggregation(const pa_fused_join_aggregation_desc* desc, pa_lookup_source* bridge);
pa_operator* make_fused_probe_aggregation(const pa_fused_join_aggregation_desc* desc, pa_lookup_source* bridge);
// FilterAndProject -> LookupJoin (op_fused_join.cpp); make_filter_project_probe is its one-pass form (op_filter_project.cpp), valid
// under the same condition
pa_operator* make_fused_join(const pa_fused_join_desc* desc, pa_lookup_source* bridge);
pa_operator* make_filter_project_probe(const pa_filter_project_desc* fp, const pa_lookup_join_desc* join, pa_lookup_source* bridge);
bool lookup_source_built(pa_lookup_source* ls);
bool lookup_source_unique_keyed(pa_lookup_source* ls);
pa_operator* make_filter_project(const pa_filter_project_desc* desc);
pa_operator* make_scan_filter_project(const pa_filter_project_desc* desc, const pa_page_source* source);
void scan_stats(pa_operator* op, int64_t* rows, int64_t* bytes, int64_t* loaded, int64_t* skipped);
pa_operator* make_hash_builder(const pa_hash_builder_desc* desc, pa_lookup_source* bridge);
pa_operator* make_lookup_join(const pa_lookup_join_desc* desc, pa_lookup_source* bridge);
pa_operator* make_topn(const pa_topn_desc* desc);
pa_operator* make_order_by(const pa_order_by_desc* desc);
pa_operator* make_lookup_outer(const pa_lookup_join_desc* desc, pa_lookup_source* bridge);
// the consumer of an aggregation's output is a TopN over it: groups that cannot be among its n best rows may be left out (op_fused.hpp);
// false: the operator does not take the hint (it emits everything)
bool aggregation_set_output_topn(pa_operator* op, int64_t n, const int32_t* sort_channels, const int32_t* sort_orders, int32_t count);
// join-side dynamic filter: the existence bitmap of a built lookup source (false: none -- not built, not a single integer key,
// or keys too sparse) and its application in a FilterAndProject operator upstream of the probe
bool lookup_source_key_bitmap(pa_lookup_source* ls, const uint64_t** bits, int64_t* min_key, uint64_t* range, std::shared_ptr<void>* keep);
int32_t lookup_source_position_count(pa_lookup_source* ls);
bool lookup_source_key_range(pa_lookup_source* ls, int64_t* min_key, int64_t* max_key);
void lookup_source_fill_bitmap(pa_lookup_source* ls, int64_t min_key, uint64_t range, uint64_t* bits, hipStream_t s);
void filter_project_set_dynamic_filter(pa_operator* op, int channel, const uint64_t* bits, int64_t min_key, uint64_t range, std::shared_ptr<void> keep);
pa_operator* make_dynamic_filter_source(const pa_dynamic_filter_source_desc* desc);
int32_t dynamic_filter_poll(pa_operator* op, int32_t* is_all, pa_domain* domains, int32_t capacity);

// (Hash)AggregationOperator over plain channels with the reference's intermediate-state format at its PARTIAL output / FINAL
// input: the flat-format operator `make_flat` builds, inside an adapter (op_states.cpp)
pa_operator* make_aggregation_with_reference_states(const pa_hash_aggregation_desc* agg, pa_operator* (*make_flat)(const pa_hash_aggregation_desc*));
// HashAggregationOperator's default output rows when no input arrived (op_default_output.cpp): takes ownership of `inner`
pa_operator* make_default_output_aggregation(pa_operator* inner, const pa_hash_aggregation_desc* desc, pa_operator* (*make_flat)(const pa_hash_aggregation_desc*));
// partitioned exchange (op_exchange.cpp)
pa_exchange* exchange_new(const pa_exchange_desc* desc, pa_comm* comm);
void exchange_delete(pa_exchange* ex);
void exchange_stats(pa_exchange* ex, int64_t* rows_sent, int64_t* rows_received, int64_t* bytes_remote, double* transfer_ms);
pa_operator* make_partitioned_output(pa_exchange* ex, void* stream);
pa_operator* make_exchange_source(pa_exchange* ex, int32_t output_mem, void* stream);
// the build-side existence bitmap of a partitioned join over the union key range of all ranks (op_join.cpp)
bool lookup_source_shared_bitmap(pa_lookup_source* ls, pa_comm* comm, bool partitioned_by_key, hipStream_t s, const uint64_t** bits, int64_t* min_key,
                                 uint64_t* range);

// page wire format (page_serde.cpp)
int64_t serialize_page(const pa_page* page, void* out_host, int64_t capacity, hipStream_t s, bool compress);
pa_page_buffer* deserialize_page(const void* bytes, int64_t size, hipStream_t s, const int32_t* expected_types, int32_t expected_count);
void page_buffer_page(pa_page_buffer* buffer, pa_page* out);
void page_buffer_free(pa_page_buffer* buffer);

// code-object source for a fused descriptor under the "no nulls, aligned" layout; used by build() to
// pre-compile the TPC-H shapes and by the CPU-side codegen tests
std::string fused_source_for_desc(const pa_fused_aggregation_desc* desc, int variant, std::string* entry);
// ... under any nullability signature (bit c: channel c carries NULL flags) and for every tier
std::string fused_source_for_layout(const pa_fused_aggregation_desc* desc, int variant, uint64_t nullable_channels);
void lookup_source_shape_for_desc(const pa_hash_builder_desc* build, pa_lookup_source* bridge);
std::string filter_project_probe_source_for_desc(const pa_fused_join_desc* desc, const pa_hash_builder_desc* build);
std::string fused_join_source_for_desc(const pa_fused_join_aggregation_desc* desc, const pa_hash_builder_desc* build, int variant, std::string* entry);
std::string filter_project_source_for_desc(const pa_filter_project_desc* desc, std::string* entry);

}  // namespace pa
