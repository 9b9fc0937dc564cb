// abi.cpp -- the extern "C" surface declared in include/presto_amd.h.  Thin: argument checks, the
// exception -> status translation, and dispatch to the operator objects.  No arithmetic lives here.
#include <chrono>
#include <mutex>
#include <mutex>

#include <atomic>

#include "comm.hpp"
#include "exprgen.hpp"
#include "jit.hpp"
#include "operator.hpp"
#include "scan_kernels.hpp"
#include "static_kernels.hpp"

namespace pa {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

static int g_cu_count = 0;
static std::once_flag g_device_once;
static int32_t g_device_status = PA_OK;
static std::string g_device_message;

static void probe_device()
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_device_status = PA_ERR_NO_DEVICE;
        g_device_message = "no HIP device visible: libpresto_amd.so needs an MI355X (gfx950); there is no CPU fallback";
        return;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        g_device_status = PA_ERR_NO_DEVICE;
        g_device_message = "hipGetDeviceProperties failed";
        return;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_device_status = PA_ERR_NO_DEVICE;
        g_device_message = std::string("device is ") + prop.gcnArchName + ", but this library is built for gfx950 only";
        return;
    }
    g_cu_count = prop.multiProcessorCount;
}

void require_device()
{
    std::call_once(g_device_once, probe_device);
    if (g_device_status != PA_OK) throw Error(g_device_status, g_device_message);
}

int device_cu_count()
{
    require_device();
    return g_cu_count > 0 ? g_cu_count : 256;
}

// The device pa_init chose, process-wide: Trino's Driver / TaskExecutor threads never call pa_init themselves, and a HIP
// thread that has not chosen a device works on device 0.  Every C-ABI entry binds a thread that has no binding of its own
// to it; operator handles carry their device and rebind per call (OpScope).
static std::atomic<int> g_default_device{-1};
static thread_local bool t_device_bound = false;

static void bind_default_device()
{
    if (t_device_bound) return;
    const int d = g_default_device.load(std::memory_order_relaxed);
    if (d >= 0) {
        (void)hipSetDevice(d);
        t_device_bound = true;
    }
}

// An operator call: the thread works on the operator's device, and HBM blocks the operator releases meanwhile are tagged with
// its stream (pool.cpp).
struct OpScope {
    explicit OpScope(pa_operator* op)
    {
        if (op->device >= 0) {
            int cur = -1;
            if (hipGetDevice(&cur) != hipSuccess || cur != op->device) (void)hipSetDevice(op->device);
            t_device_bound = true;
        }
        prev_ = pool_scope_stream(op->main_stream());
    }
    ~OpScope() { (void)pool_scope_stream(prev_); }
    hipStream_t prev_;
};

// PRESTO_AMD_HOST_TRACE=<file>: wall-clock start and duration of every C-ABI call (and of the phases inside an operator that
// carry a HostTraceScope), appended to <file> when the process ends (start us since the first call, duration us, name) -- the host
// side of a kernel timeline (scripts/kernel_timeline.py)
struct HostTrace {
    struct Rec {
        double start_us, us;
        const char* name;
    };
    std::vector<Rec> recs;
    std::mutex mu;
    const char* path = getenv("PRESTO_AMD_HOST_TRACE");
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~HostTrace()
    {
        if (!path || recs.empty()) return;
        if (FILE* f = fopen(path, "a")) {
            for (const Rec& r : recs) fprintf(f, "%12.1f %9.1f %s\n", r.start_us, r.us, r.name);
            fclose(f);
        }
    }
};
static HostTrace g_host_trace;
HostTraceScope::HostTraceScope(const char* n) : name(n), on(g_host_trace.path != nullptr)
{
    if (on) t = std::chrono::steady_clock::now();
}
HostTraceScope::~HostTraceScope()
{
    if (!on) return;
    const auto e = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lock(g_host_trace.mu);
    g_host_trace.recs.push_back({std::chrono::duration<double, std::micro>(t - g_host_trace.t0).count(), std::chrono::duration<double, std::micro>(e - t).count(), name});
}

template <typename F>
static int32_t guarded(F&& f, const char* entry = __builtin_FUNCTION())
{
    HostTraceScope trace(entry);
    try {
        bind_default_device();
        return f();
    }
    catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    }
    catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return PA_ERR_INSUFFICIENT_RESOURCES;
    }
    catch (const std::exception& e) {
        set_last_error(e.what());
        return PA_ERR_DEVICE;
    }
    catch (...) {
        set_last_error("unknown failure");
        return PA_ERR_DEVICE;
    }
}

}  // namespace pa

namespace pa {
int32_t filter_project_last_positions(pa_operator* op, const int32_t** dev_positions, int32_t* count, int32_t* is_list);
}
using namespace pa;

namespace pa {
pa_lookup_source* lookup_source_new();
void lookup_source_delete(pa_lookup_source* ls);
int32_t lookup_join_last_pairs(pa_operator* op, const int32_t** probe_idx, const int32_t** build_pos, int32_t* count);
int32_t lookup_source_tables(pa_lookup_source* ls, const int32_t** key, int32_t* hash_size, const int32_t** links, int32_t* positions);
}

extern "C" {

int32_t pa_abi_version(void) { return PA_ABI_VERSION; }

int32_t pa_init(int32_t device)
{
    return guarded([&]() -> int32_t {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            throw Error(PA_ERR_NO_DEVICE, "no HIP device visible: libpresto_amd.so needs an MI355X (gfx950); there is no CPU fallback");
        }
        if (device >= 0) {
            PA_REQUIRE(device < count, PA_ERR_INVALID_ARGUMENT, "device ordinal out of range");
            PA_HIP(hipSetDevice(device));
            t_device_bound = true;
            g_default_device.store(device, std::memory_order_relaxed);
        }
        require_device();
        return PA_OK;
    });
}

int32_t pa_shutdown(void) { return PA_OK; }

const char* pa_last_error(void) { return g_last_error.c_str(); }

int32_t pa_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int32_t pa_device_malloc(void** ptr, int64_t bytes)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(ptr != nullptr && bytes >= 0, PA_ERR_INVALID_ARGUMENT, "bad arguments");
        require_device();
        PA_HIP(hipMalloc(ptr, (size_t)(bytes > 0 ? bytes : 1)));
        return PA_OK;
    });
}
int32_t pa_device_free(void* ptr)
{
    return guarded([&]() -> int32_t {
        if (ptr) PA_HIP(hipFree(ptr));
        return PA_OK;
    });
}
int32_t pa_host_malloc_pinned(void** ptr, int64_t bytes)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(ptr != nullptr && bytes >= 0, PA_ERR_INVALID_ARGUMENT, "bad arguments");
        require_device();
        PA_HIP(hipHostMalloc(ptr, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault));
        return PA_OK;
    });
}
int32_t pa_host_free_pinned(void* ptr)
{
    return guarded([&]() -> int32_t {
        if (ptr) PA_HIP(hipHostFree(ptr));
        return PA_OK;
    });
}
int32_t pa_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream)
{
    return guarded([&]() -> int32_t {
        if (bytes > 0) PA_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        PA_HIP(hipStreamSynchronize((hipStream_t)stream));
        return PA_OK;
    });
}
int32_t pa_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream)
{
    return guarded([&]() -> int32_t {
        if (bytes > 0) PA_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        PA_HIP(hipStreamSynchronize((hipStream_t)stream));
        return PA_OK;
    });
}
int32_t pa_stream_synchronize(void* stream)
{
    return guarded([&]() -> int32_t {
        PA_HIP(hipStreamSynchronize((hipStream_t)stream));
        return PA_OK;
    });
}

int32_t pa_device_synchronize(void)
{
    return guarded([&]() -> int32_t {
        require_device();
        PA_HIP(hipDeviceSynchronize());
        return PA_OK;
    });
}

int32_t pa_memory_set_limit(int64_t bytes)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(bytes >= 0, PA_ERR_INVALID_ARGUMENT, "negative limit");
        pool_set_limit(bytes);
        return PA_OK;
    });
}
int32_t pa_memory_stats(int64_t* in_use, int64_t* cached, int64_t* limit)
{
    return guarded([&]() -> int32_t {
        pool_stats(in_use, cached, limit);
        return PA_OK;
    });
}

int32_t pa_stream_create(void** stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(stream != nullptr, PA_ERR_INVALID_ARGUMENT, "stream is null");
        require_device();
        hipStream_t s;
        PA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        *stream = s;
        return PA_OK;
    });
}
int32_t pa_stream_destroy(void* stream)
{
    return guarded([&]() -> int32_t {
        if (stream) {
            PA_HIP(hipStreamSynchronize((hipStream_t)stream));
            pool_forget_stream((hipStream_t)stream);
            PA_HIP(hipStreamDestroy((hipStream_t)stream));
        }
        return PA_OK;
    });
}

// ---- factories ----
int32_t pa_fused_aggregation_create(const pa_fused_aggregation_desc* desc, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "out is null");
        *out = make_fused_aggregation(desc);
        return PA_OK;
    });
}

// AggregationOperator / HashAggregationOperator on their own = the fused operator with an empty filter
// and identity projections over the input channels.
static pa_operator* plain_aggregation(const pa_hash_aggregation_desc* agg);
static int32_t create_plain_aggregation(const pa_hash_aggregation_desc* agg, pa_operator** out)
{
    PA_REQUIRE(agg != nullptr && out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    if (agg->state_format == PA_STATES_REFERENCE && agg->step != PA_STEP_SINGLE) {
        *out = make_aggregation_with_reference_states(agg, &plain_aggregation);
        return PA_OK;
    }
    PA_REQUIRE(agg->state_format == PA_STATES_FLAT || agg->state_format == PA_STATES_REFERENCE, PA_ERR_INVALID_ARGUMENT, "unknown state format");
    *out = plain_aggregation(agg);
    return PA_OK;
}
static pa_operator* plain_aggregation(const pa_hash_aggregation_desc* agg)
{
    int n = agg->input_channel_count;
    PA_REQUIRE(n > 0, PA_ERR_INVALID_ARGUMENT, "aggregation needs input channels");
    std::vector<pa_expr_node> nodes(n);
    std::vector<pa_expr> exprs(n);
    for (int c = 0; c < n; c++) {
        memset(&nodes[c], 0, sizeof(pa_expr_node));
        nodes[c].kind = PA_EXPR_INPUT_REF;
        nodes[c].type = agg->input_types[c];
        nodes[c].channel = c;
        // DECIMAL channels: their type parameter, or -- for a descriptor that carries none (pa_aggregation_desc) -- the widest
        // precision of the block type; the aggregates only take the value's width from it, never the scale
        if (agg->input_types[c] == PA_DECIMAL || agg->input_types[c] == PA_LONG_DECIMAL) {
            const int32_t given = agg->input_type_params ? agg->input_type_params[c] : 0;
            nodes[c].type_param = given ? given : (agg->input_types[c] == PA_DECIMAL ? PA_DECIMAL_PARAM(18, 0) : PA_DECIMAL_PARAM(38, 0));
        }
        exprs[c].node_count = 1;
        exprs[c].root = 0;
        exprs[c].nodes = &nodes[c];
        exprs[c].arg_count = 0;
        exprs[c].args = nullptr;
    }
    pa_fused_aggregation_desc d;
    memset(&d, 0, sizeof d);
    d.filter_project.input_channel_count = n;
    d.filter_project.input_types = agg->input_types;
    d.filter_project.input_type_params = agg->input_type_params;
    d.filter_project.filter = nullptr;
    d.filter_project.projection_count = n;
    d.filter_project.projections = exprs.data();
    d.filter_project.output_mem = agg->output_mem;
    d.filter_project.stream = agg->stream;
    d.aggregation = *agg;
    d.aggregation.state_format = PA_STATES_FLAT;
    pa_operator* op = make_fused_aggregation(&d);
    // produceDefaultOutput (HashAggregationOperator.java:486-492): the rows of the global grouping sets when no input arrives
    if (agg->produce_default_output && agg->group_by_count > 0) return make_default_output_aggregation(op, agg, &plain_aggregation);
    return op;
}

int32_t pa_hash_aggregation_create(const pa_hash_aggregation_desc* desc, pa_operator** out)
{
    return guarded([&]() -> int32_t { return create_plain_aggregation(desc, out); });
}

int32_t pa_fused_join_create(const pa_fused_join_desc* desc, pa_lookup_source* bridge, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(desc != nullptr && out != nullptr && bridge != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_fused_join(desc, bridge);
        return PA_OK;
    });
}

int32_t pa_fused_join_aggregation_create(const pa_fused_join_aggregation_desc* desc, pa_lookup_source* bridge, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(desc != nullptr && out != nullptr && bridge != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_fused_join_aggregation(desc, bridge);
        return PA_OK;
    });
}

int32_t pa_aggregation_create(const pa_aggregation_desc* desc, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(desc != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
        pa_hash_aggregation_desc h;
        memset(&h, 0, sizeof h);
        h.input_channel_count = desc->input_channel_count;
        h.input_types = desc->input_types;
        h.group_by_count = 0;
        h.hash_channel = -1;
        h.step = desc->step;
        h.aggregate_count = desc->aggregate_count;
        h.aggregates = desc->aggregates;
        h.output_mem = desc->output_mem;
        h.stream = desc->stream;
        h.state_format = desc->state_format;
        return create_plain_aggregation(&h, out);
    });
}

int32_t pa_filter_project_create(const pa_filter_project_desc* desc, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "out is null");
        *out = make_filter_project(desc);
        return PA_OK;
    });
}

int32_t pa_scan_filter_project_create(const pa_filter_project_desc* desc, const pa_page_source* source, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "out is null");
        *out = make_scan_filter_project(desc, source);
        return PA_OK;
    });
}
int32_t pa_scan_stats(pa_operator* op, int64_t* processed_positions, int64_t* materialized_bytes, int64_t* blocks_loaded, int64_t* blocks_skipped)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr, PA_ERR_INVALID_ARGUMENT, "operator is null");
        scan_stats(op, processed_positions, materialized_bytes, blocks_loaded, blocks_skipped);
        return PA_OK;
    });
}

int32_t pa_lookup_source_create(pa_lookup_source** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "out is null");
        *out = lookup_source_new();
        return PA_OK;
    });
}
int32_t pa_lookup_source_destroy(pa_lookup_source* ls)
{
    return guarded([&]() -> int32_t {
        if (ls) lookup_source_delete(ls);
        return PA_OK;
    });
}
int32_t pa_lookup_join_match_pairs(pa_operator* op, const int32_t** probe_positions, const int32_t** build_positions, int32_t* count)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op && probe_positions && build_positions && count, PA_ERR_INVALID_ARGUMENT, "null argument");
        return lookup_join_last_pairs(op, probe_positions, build_positions, count);
    });
}
int32_t pa_lookup_source_tables(pa_lookup_source* ls, const int32_t** key, int32_t* hash_size, const int32_t** position_links, int32_t* positions)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(ls && key && hash_size && position_links && positions, PA_ERR_INVALID_ARGUMENT, "null argument");
        return lookup_source_tables(ls, key, hash_size, position_links, positions);
    });
}
int32_t pa_topn_create(const pa_topn_desc* desc, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_topn(desc);
        return PA_OK;
    });
}
int32_t pa_dynamic_filter_source_create(const pa_dynamic_filter_source_desc* desc, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_dynamic_filter_source(desc);
        return PA_OK;
    });
}
int32_t pa_dynamic_filter_poll(pa_operator* op, int32_t* is_all, pa_domain* domains, int32_t domain_capacity)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr && is_all != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        return dynamic_filter_poll(op, is_all, domains, domain_capacity);
    });
}
int32_t pa_filter_project_set_dynamic_filter(pa_operator* op, int32_t channel, pa_lookup_source* source)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr && source != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        const uint64_t* bits = nullptr;
        int64_t min_key = 0;
        uint64_t range = 0;
        std::shared_ptr<void> keep;
        if (!lookup_source_key_bitmap(source, &bits, &min_key, &range, &keep)) return 0;
        filter_project_set_dynamic_filter(op, channel, bits, min_key, range, std::move(keep));
        return 1;
    });
}
int32_t pa_aggregation_set_output_topn_hint(pa_operator* op, int64_t n, int32_t sort_channel_count, const int32_t* sort_channels, const int32_t* sort_orders)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr && sort_channel_count > 0 && sort_channels != nullptr && sort_orders != nullptr && n > 0, PA_ERR_INVALID_ARGUMENT, "null argument");
        for (int i = 0; i < sort_channel_count; i++) {
            PA_REQUIRE(sort_channels[i] >= 0 && sort_orders[i] >= 0 && sort_orders[i] <= 3, PA_ERR_INVALID_ARGUMENT, "bad sort channel / order");
        }
        return aggregation_set_output_topn(op, n, sort_channels, sort_orders, sort_channel_count) ? 1 : 0;
    });
}
int32_t pa_lookup_source_position_count(pa_lookup_source* source)
{
    return guarded([&]() -> int32_t { return lookup_source_position_count(source); });
}
int32_t pa_lookup_source_key_range(pa_lookup_source* source, int64_t* min_key, int64_t* max_key)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(source != nullptr && min_key != nullptr && max_key != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        return lookup_source_key_range(source, min_key, max_key) ? 1 : 0;
    });
}
int32_t pa_lookup_source_key_bitmap(pa_lookup_source* source, int64_t min_key, uint64_t range, uint64_t* bits, void* stream)
{
    return guarded([&]() -> int32_t {
        lookup_source_fill_bitmap(source, min_key, range, bits, (hipStream_t)stream);
        return PA_OK;
    });
}
int32_t pa_filter_project_set_dynamic_filter_bitmap(pa_operator* op, int32_t channel, const uint64_t* bits, int64_t min_key, uint64_t range)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr && bits != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        filter_project_set_dynamic_filter(op, channel, bits, min_key, range, nullptr);
        return PA_OK;
    });
}
int32_t pa_order_by_create(const pa_order_by_desc* desc, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_order_by(desc);
        return PA_OK;
    });
}
int32_t pa_hash_builder_create(const pa_hash_builder_desc* desc, pa_lookup_source* bridge, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr && bridge != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_hash_builder(desc, bridge);
        return PA_OK;
    });
}
int32_t pa_lookup_join_create(const pa_lookup_join_desc* desc, pa_lookup_source* bridge, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr && bridge != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_lookup_join(desc, bridge);
        return PA_OK;
    });
}

int32_t pa_lookup_outer_create(const pa_lookup_join_desc* desc, pa_lookup_source* bridge, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr && bridge != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_lookup_outer(desc, bridge);
        return PA_OK;
    });
}

// ---- partitioned exchange ----
int32_t pa_comm_unique_id(void* id_out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(id_out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        comm_unique_id(id_out);
        return PA_OK;
    });
}
int32_t pa_comm_create(const void* unique_id, int32_t rank, int32_t world, pa_comm** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = comm_create_rccl(unique_id, rank, world);
        return PA_OK;
    });
}
int32_t pa_comm_create_host(const pa_host_transport* transport, int32_t rank, int32_t world, pa_comm** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = comm_create_host(transport, rank, world);
        return PA_OK;
    });
}
int32_t pa_comm_destroy(pa_comm* comm)
{
    return guarded([&]() -> int32_t {
        delete comm;
        return PA_OK;
    });
}
int32_t pa_comm_rank(pa_comm* comm)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(comm != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        return comm->rank;
    });
}
int32_t pa_comm_world(pa_comm* comm)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(comm != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        return comm->world;
    });
}
int32_t pa_comm_all_reduce_i64(pa_comm* comm, int64_t* values, int32_t count, int32_t op, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(comm != nullptr && values != nullptr && count > 0 && op >= 0 && op <= 2, PA_ERR_INVALID_ARGUMENT, "bad arguments");
        comm_all_reduce_i64(comm, values, count, op, (hipStream_t)stream);
        return PA_OK;
    });
}
int32_t pa_comm_all_gather_i64(pa_comm* comm, const int64_t* send, int64_t* recv, int32_t count, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(comm != nullptr && send != nullptr && recv != nullptr && count > 0, PA_ERR_INVALID_ARGUMENT, "bad arguments");
        comm_all_gather_i64(comm, send, recv, count, (hipStream_t)stream);
        return PA_OK;
    });
}
int32_t pa_comm_preflight(pa_comm* comm, int64_t bytes_per_peer, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(comm != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        comm_preflight(comm, bytes_per_peer, (hipStream_t)stream);
        return PA_OK;
    });
}
int32_t pa_exchange_create(const pa_exchange_desc* desc, pa_comm* comm, pa_exchange** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = exchange_new(desc, comm);
        return PA_OK;
    });
}
int32_t pa_exchange_destroy(pa_exchange* exchange)
{
    return guarded([&]() -> int32_t {
        exchange_delete(exchange);
        return PA_OK;
    });
}
int32_t pa_partitioned_output_create(pa_exchange* exchange, void* stream, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_partitioned_output(exchange, stream);
        return PA_OK;
    });
}
int32_t pa_exchange_source_create(pa_exchange* exchange, int32_t output_mem, void* stream, pa_operator** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = make_exchange_source(exchange, output_mem, stream);
        return PA_OK;
    });
}
int32_t pa_exchange_stats(pa_exchange* exchange, int64_t* rows_sent, int64_t* rows_received, int64_t* bytes_sent_remote, double* transfer_ms)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(exchange != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        exchange_stats(exchange, rows_sent, rows_received, bytes_sent_remote, transfer_ms);
        return PA_OK;
    });
}
int32_t pa_lookup_source_shared_key_bitmap(pa_lookup_source* source, pa_comm* comm, int32_t partitioned_by_key, void* stream,
                                           const uint64_t** bits, int64_t* min_key, uint64_t* range)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(source && comm && bits && min_key && range, PA_ERR_INVALID_ARGUMENT, "null argument");
        return lookup_source_shared_bitmap(source, comm, partitioned_by_key != 0, (hipStream_t)stream, bits, min_key, range) ? 1 : 0;
    });
}

// ---- page wire format ----
int64_t pa_page_serialize(const pa_page* page, void* out_host, int64_t capacity, void* stream)
{
    int64_t written = 0;
    int32_t rc = guarded([&]() -> int32_t {
        written = serialize_page(page, out_host, capacity, static_cast<hipStream_t>(stream), false);
        return PA_OK;
    });
    return rc < 0 ? (int64_t)rc : written;
}
int64_t pa_page_serialize_lz4(const pa_page* page, void* out_host, int64_t capacity, void* stream)
{
    int64_t written = 0;
    int32_t rc = guarded([&]() -> int32_t {
        written = serialize_page(page, out_host, capacity, static_cast<hipStream_t>(stream), true);
        return PA_OK;
    });
    return rc < 0 ? (int64_t)rc : written;
}
int32_t pa_page_deserialize_typed(const void* bytes_host, int64_t size, const int32_t* expected_types, int32_t channel_count, void* stream,
                                  pa_page_buffer** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr && expected_types != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = deserialize_page(bytes_host, size, static_cast<hipStream_t>(stream), expected_types, channel_count);
        return PA_OK;
    });
}
int32_t pa_page_deserialize(const void* bytes_host, int64_t size, void* stream, pa_page_buffer** out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        *out = deserialize_page(bytes_host, size, static_cast<hipStream_t>(stream), nullptr, 0);
        return PA_OK;
    });
}
int32_t pa_page_buffer_page(pa_page_buffer* buffer, pa_page* out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(buffer != nullptr && out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        page_buffer_page(buffer, out);
        return PA_OK;
    });
}
int32_t pa_page_buffer_free(pa_page_buffer* buffer)
{
    return guarded([&]() -> int32_t {
        page_buffer_free(buffer);
        return PA_OK;
    });
}

// ---- Operator protocol ----
int32_t pa_op_needs_input(pa_operator* op)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr, PA_ERR_INVALID_ARGUMENT, "operator is null");
        OpScope scope(op);
        return op->needs_input() ? 1 : 0;
    });
}
int32_t pa_op_add_input(pa_operator* op, const pa_page* page)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr, PA_ERR_INVALID_ARGUMENT, "operator is null");
        OpScope scope(op);
        PA_REQUIRE(op->needs_input(), PA_ERR_ILLEGAL_STATE, "Operator does not need input");
        const bool retained = page != nullptr && (page->flags & PA_PAGE_RETAINED) != 0 && page->release != nullptr;
        if (retained && op->takes_retained()) {
            op->add_input(page);  // the operator owns the release from here on (it is called even when this call throws: at close at the latest)
            return PA_OK;
        }
        if (retained) {
            // an operator that does not hold on to pages: the page is taken as a plain one and handed back when the call returns
            pa_page plain = *page;
            plain.flags &= ~PA_PAGE_RETAINED;
            struct Releaser {
                const pa_page* p;
                ~Releaser() { p->release(p->release_ctx); }
            };
            try {
                op->add_input(&plain);
                // device buffers and pinned host buffers are read by the device itself: the reads the operator enqueued must have run
                // (pageable host buffers were staged by the runtime before hipMemcpyAsync returned)
                if (plain.mem == PA_MEM_DEVICE || (plain.flags & PA_PAGE_PINNED) != 0) {
                    hipStream_t m = op->main_stream() ? op->main_stream() : op->private_stream();
                    if (m) PA_HIP(hipStreamSynchronize(m));
                    else PA_HIP(hipDeviceSynchronize());
                }
            }
            catch (...) {
                Releaser r{page};
                throw;
            }
            Releaser r{page};
            return PA_OK;
        }
        op->add_input(page);
        // A device page some operator returned is that operator's again with its next call.  Work this operator has enqueued on the
        // page is ordered before that call when both run on the caller's stream; on a stream of the operator's own nothing orders it
        // (an aggregation copies a small page into its arena with a kernel of its own stream and returns), so that stream is drained.
        if (page != nullptr && page->mem == PA_MEM_DEVICE && (page->flags & PA_PAGE_STABLE) == 0) {
            if (hipStream_t s = op->private_stream()) PA_HIP(hipStreamSynchronize(s));
        }
        return PA_OK;
    });
}
int32_t pa_op_get_output(pa_operator* op, pa_page* out)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr && out != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        OpScope scope(op);
        if (!op->get_output(out)) return 0;
        if (out->mem == PA_MEM_DEVICE) {
            if (hipStream_t s = op->private_stream()) PA_HIP(hipStreamSynchronize(s));
        }
        return 1;
    });
}
int32_t pa_op_finish(pa_operator* op)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr, PA_ERR_INVALID_ARGUMENT, "operator is null");
        OpScope scope(op);
        op->finish();
        return PA_OK;
    });
}
int32_t pa_op_is_finished(pa_operator* op)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr, PA_ERR_INVALID_ARGUMENT, "operator is null");
        OpScope scope(op);
        return op->is_finished() ? 1 : 0;
    });
}
int32_t pa_op_is_blocked(pa_operator* op)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr, PA_ERR_INVALID_ARGUMENT, "operator is null");
        OpScope scope(op);
        return op->is_blocked() ? 1 : 0;
    });
}
int64_t pa_op_memory_bytes(pa_operator* op)
{
    int64_t bytes = 0;
    int32_t rc = guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr, PA_ERR_INVALID_ARGUMENT, "operator is null");
        OpScope scope(op);
        bytes = op->memory_bytes();
        return PA_OK;
    });
    return rc < 0 ? rc : bytes;
}
int32_t pa_op_close(pa_operator* op)
{
    return guarded([&]() -> int32_t {
        if (op) {
            OpScope scope(op);
            op->close();
            delete op;
        }
        return PA_OK;
    });
}
int32_t pa_op_kernel_time(pa_operator* op, double* total_ms, int64_t* launches)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr, PA_ERR_INVALID_ARGUMENT, "operator is null");
        OpScope scope(op);
        pa::KernelTimer& timer = op->kernel_timer();
        timer.drain();
        if (total_ms) *total_ms = timer.total_ms();
        if (launches) *launches = timer.launches();
        return PA_OK;
    });
}
int32_t pa_op_kernel_name(pa_operator* op, char* buf, int32_t buf_size)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr && buf != nullptr && buf_size > 0, PA_ERR_INVALID_ARGUMENT, "null argument");
        OpScope scope(op);
        const std::string& n = op->kernel_timer().name();
        snprintf(buf, (size_t)buf_size, "%s", n.c_str());
        return PA_OK;
    });
}

// ---- stand-alone kernels ----
int32_t pa_hash_page(const pa_page* page, int32_t channel_count, const int32_t* channels, int64_t* out_raw_hash, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(page != nullptr && channels != nullptr && out_raw_hash != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        PA_REQUIRE(page->mem == PA_MEM_DEVICE, PA_ERR_INVALID_ARGUMENT, "pa_hash_page takes a device-resident page");
        PA_REQUIRE(channel_count >= 0 && channel_count <= 16, PA_ERR_NOT_SUPPORTED, "at most 16 hash channels");
        require_device();
        HashPageArgs a;
        memset(&a, 0, sizeof a);
        for (int32_t i = 0; i < channel_count; i++) {
            int32_t c = channels[i];
            PA_REQUIRE(c >= 0 && c < page->channel_count, PA_ERR_INVALID_ARGUMENT, "hash channel out of range");
            const pa_column& col = page->columns[c];
            PA_REQUIRE(col.encoding == PA_FLAT || col.encoding == PA_VARWIDTH, PA_ERR_NOT_SUPPORTED, "hash of an encoded block");
            PA_REQUIRE(col.type != PA_ROW, PA_ERR_NOT_SUPPORTED, "hash of a ROW channel is not on the device path");
            a.col[i].values = col.values;
            a.col[i].offsets = col.offsets;
            a.col[i].nulls = col.nulls;
            a.col[i].type = col.type;
        }
        a.ncols = channel_count;
        a.n = page->position_count;
        a.out = out_raw_hash;
        launch_hash_page(a, (hipStream_t)stream);
        return PA_OK;
    });
}

int32_t pa_partition_ids(const int64_t* raw_hash, int32_t position_count, int32_t partition_count, int32_t local,
                         int32_t* out_partition, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(raw_hash != nullptr && out_partition != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        PA_REQUIRE(partition_count > 0, PA_ERR_INVALID_ARGUMENT, "partitionCount must be positive");
        PA_REQUIRE(!local || (partition_count & (partition_count - 1)) == 0, PA_ERR_INVALID_ARGUMENT, "partitionCount must be a power of 2");
        require_device();
        launch_partition_ids(raw_hash, position_count, partition_count, local, out_partition, (hipStream_t)stream);
        return PA_OK;
    });
}

int32_t pa_gather_flat(const void* src, int32_t elem_bytes, const int32_t* positions, int32_t count, void* dst, void* stream)
{
    return guarded([&]() -> int32_t {
        require_device();
        launch_gather_flat(src, elem_bytes, positions, count, dst, (hipStream_t)stream);
        return PA_OK;
    });
}

namespace {
// exclusive scan of n lengths already sitting in offs[0..n) into offs[0..n], total to the host
int64_t scan_lengths_in_place(int32_t* offs, int32_t n, hipStream_t s)
{
    pa::DevBuf temp, total;
    int32_t* t = static_cast<int32_t*>(total.ensure(64));
    PA_HIP(hipMemsetAsync(t, 0, 4, s));
    int32_t h = 0;
    if (n > 0) {
        pa::launch_exclusive_scan_i32(offs, offs, n, t, temp.ensure(pa::scan_temp_bytes(n)), s);
        PA_HIP(hipMemcpyAsync(offs + n, t, 4, hipMemcpyDeviceToDevice, s));
        PA_HIP(hipMemcpyAsync(&h, t, 4, hipMemcpyDeviceToHost, s));
    }
    else {
        PA_HIP(hipMemsetAsync(offs, 0, 4, s));
    }
    PA_HIP(hipStreamSynchronize(s));  // temp / total go back to the pool
    return h;
}
}  // namespace

int32_t pa_varwidth_gather_offsets(const int32_t* offsets, const int32_t* positions, int32_t count, int32_t* out_lengths,
                                   int32_t* out_offsets, int64_t* total_bytes_host, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(offsets && out_offsets && total_bytes_host && count >= 0, PA_ERR_INVALID_ARGUMENT, "bad arguments");
        require_device();
        hipStream_t s = (hipStream_t)stream;
        if (count > 0) {
            launch_varwidth_lengths(positions, count, offsets, nullptr, out_offsets, s);
            if (out_lengths) PA_HIP(hipMemcpyAsync(out_lengths, out_offsets, (size_t)count * 4, hipMemcpyDeviceToDevice, s));
        }
        *total_bytes_host = scan_lengths_in_place(out_offsets, count, s);
        return PA_OK;
    });
}
int32_t pa_varwidth_gather_bytes(const void* values, const int32_t* offsets, const int32_t* positions, int32_t count,
                                 const int32_t* out_offsets, void* out_values, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(offsets && out_offsets && count >= 0, PA_ERR_INVALID_ARGUMENT, "bad arguments");
        require_device();
        if (count == 0) return PA_OK;
        pa::DevBuf total;
        int32_t* t = static_cast<int32_t*>(total.ensure(64));
        hipStream_t s = (hipStream_t)stream;
        PA_HIP(hipMemcpyAsync(t, out_offsets + count, 4, hipMemcpyDeviceToDevice, s));
        launch_varwidth_copy(positions, count, offsets, static_cast<const uint8_t*>(values), nullptr, const_cast<int32_t*>(out_offsets),
                             static_cast<uint8_t*>(out_values), t, s);
        PA_HIP(hipStreamSynchronize(s));
        return PA_OK;
    });
}
int32_t pa_offsets_from_lengths(const int32_t* lengths, int32_t count, int32_t* out_offsets, int64_t* total_bytes_host, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(out_offsets && total_bytes_host && count >= 0 && (lengths || count == 0), PA_ERR_INVALID_ARGUMENT, "bad arguments");
        require_device();
        hipStream_t s = (hipStream_t)stream;
        if (count > 0) PA_HIP(hipMemcpyAsync(out_offsets, lengths, (size_t)count * 4, hipMemcpyDeviceToDevice, s));
        *total_bytes_host = scan_lengths_in_place(out_offsets, count, s);
        return PA_OK;
    });
}

int32_t pa_tpch_generate(int32_t column, double scale_factor, int64_t first_row, int64_t row_count, uint64_t seed, void* values,
                         int32_t* offsets, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(values != nullptr && row_count >= 0, PA_ERR_INVALID_ARGUMENT, "bad arguments");
        require_device();
        launch_tpch(column, scale_factor, first_row, row_count, seed, values, offsets, (hipStream_t)stream);
        return PA_OK;
    });
}

int32_t pa_filter_project_selected_positions(pa_operator* op, const int32_t** dev_positions, int32_t* count, int32_t* is_list)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(op != nullptr && dev_positions != nullptr && count != nullptr && is_list != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        return filter_project_last_positions(op, dev_positions, count, is_list);
    });
}

int32_t pa_partition_positions(const int32_t* partition, int32_t position_count, int32_t partition_count, int32_t* out_positions,
                               int64_t* out_counts_host, void* stream)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(partition != nullptr && out_positions != nullptr && out_counts_host != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
        require_device();
        hipStream_t s = (hipStream_t)stream;
        DevBuf temp, counts;
        temp.ensure(partition_temp_bytes(position_count, partition_count));
        counts.ensure((size_t)partition_count * 8);
        launch_partition_positions(partition, position_count, partition_count, out_positions, counts.as<int64_t>(), temp.ptr(), s);
        PA_HIP(hipMemcpyAsync(out_counts_host, counts.ptr(), (size_t)partition_count * 8, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        return PA_OK;
    });
}

static int32_t partition_columns(const int32_t* partition, int32_t position_count, int32_t partition_count, const void* const* in_columns,
                                 void* const* out_columns, const int32_t* elem_bytes, int32_t column_count, int64_t* out_counts_host, void* stream,
                                 bool stable);
int32_t pa_partition_columns(const int32_t* partition, int32_t position_count, int32_t partition_count, const void* const* in_columns,
                             void* const* out_columns, const int32_t* elem_bytes, int32_t column_count, int64_t* out_counts_host, void* stream)
{
    return partition_columns(partition, position_count, partition_count, in_columns, out_columns, elem_bytes, column_count, out_counts_host, stream, false);
}
int32_t pa_partition_columns_stable(const int32_t* partition, int32_t position_count, int32_t partition_count, const void* const* in_columns,
                                    void* const* out_columns, const int32_t* elem_bytes, int32_t column_count, int64_t* out_counts_host, void* stream)
{
    return partition_columns(partition, position_count, partition_count, in_columns, out_columns, elem_bytes, column_count, out_counts_host, stream, true);
}
static int32_t partition_columns(const int32_t* partition, int32_t position_count, int32_t partition_count, const void* const* in_columns,
                                 void* const* out_columns, const int32_t* elem_bytes, int32_t column_count, int64_t* out_counts_host, void* stream,
                                 bool stable)
{
    return guarded([&]() -> int32_t {
        PA_REQUIRE(partition != nullptr && out_counts_host != nullptr && position_count >= 0, PA_ERR_INVALID_ARGUMENT, "null argument");
        PA_REQUIRE(column_count >= 0 && column_count <= kMsplitMaxCols && (column_count == 0 || (in_columns && out_columns && elem_bytes)),
                   PA_ERR_INVALID_ARGUMENT, "bad column arguments");
        require_device();
        hipStream_t s = (hipStream_t)stream;
        std::vector<MsplitCol> cols((size_t)column_count);
        for (int32_t c = 0; c < column_count; c++) {
            cols[c].in = in_columns[c];
            cols[c].out = out_columns[c];
            cols[c].width = elem_bytes[c];
        }
        DevBuf temp, counts;
        temp.ensure(msplit_temp_bytes(position_count, partition_count));
        counts.ensure((size_t)partition_count * 8);
        launch_msplit(partition, position_count, partition_count, cols.data(), column_count, counts.as<int64_t>(), temp.ptr(), s, stable);
        PA_HIP(hipMemcpyAsync(out_counts_host, counts.ptr(), (size_t)partition_count * 8, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        return PA_OK;
    });
}

// ---- code generation without a device (build(), CPU-side tests) ----
// Writes the generated translation unit of a fused descriptor into buf (NUL terminated) and its cache
// key into key[17]; returns the needed buffer size.  variant: -1 default, 0 GLOBAL, 1 LDS, 2 GT.
int64_t pa_codegen_fused(const pa_fused_aggregation_desc* desc, int32_t variant, char* buf, int64_t buf_size, char* key)
{
    int64_t need = 0;
    int32_t rc = guarded([&]() -> int32_t {
        std::string entry;
        std::string src = fused_source_for_desc(desc, variant, &entry);
        std::string tu = jit_translation_unit(src);
        need = (int64_t)tu.size() + 1;
        if (buf && buf_size >= need) memcpy(buf, tu.c_str(), (size_t)need);
        if (key) {
            std::string k = jit_key(src);
            memcpy(key, k.c_str(), k.size() + 1);
        }
        return PA_OK;
    });
    return rc < 0 ? rc : need;
}

int64_t pa_codegen_fused_layout(const pa_fused_aggregation_desc* desc, int32_t variant, uint64_t nullable_channels, int32_t compile, char* buf,
                                int64_t buf_size)
{
    int64_t need = 0;
    int32_t rc = guarded([&]() -> int32_t {
        std::string src = fused_source_for_layout(desc, variant, nullable_channels);
        if (compile) {
            need = (int64_t)jit_compile_only(src).size();
            return PA_OK;
        }
        std::string tu = jit_translation_unit(src);
        need = (int64_t)tu.size() + 1;
        if (buf && buf_size >= need) memcpy(buf, tu.c_str(), (size_t)need);
        return PA_OK;
    });
    return rc < 0 ? rc : need;
}

// Compiles a fused descriptor's kernel with hiprtc for gfx950 (no device needed); returns the code
// object size or a negative status.
int64_t pa_codegen_compile_fused(const pa_fused_aggregation_desc* desc, int32_t variant)
{
    int64_t size = 0;
    int32_t rc = guarded([&]() -> int32_t {
        std::string entry;
        std::string src = fused_source_for_desc(desc, variant, &entry);
        size = (int64_t)jit_compile_only(src).size();
        return PA_OK;
    });
    return rc < 0 ? rc : size;
}

int64_t pa_codegen_fused_join(const pa_fused_join_aggregation_desc* desc, const pa_hash_builder_desc* build, int32_t variant, char* buf, int64_t buf_size)
{
    int64_t need = 0;
    int32_t rc = guarded([&]() -> int32_t {
        std::string entry;
        std::string tu = jit_translation_unit(fused_join_source_for_desc(desc, build, variant, &entry));
        need = (int64_t)tu.size() + 1;
        if (buf && buf_size >= need) memcpy(buf, tu.c_str(), (size_t)need);
        return PA_OK;
    });
    return rc < 0 ? rc : need;
}
int64_t pa_codegen_compile_fused_join(const pa_fused_join_aggregation_desc* desc, const pa_hash_builder_desc* build, int32_t variant)
{
    int64_t size = 0;
    int32_t rc = guarded([&]() -> int32_t {
        std::string entry;
        size = (int64_t)jit_compile_only(fused_join_source_for_desc(desc, build, variant, &entry)).size();
        return PA_OK;
    });
    return rc < 0 ? rc : size;
}

int64_t pa_codegen_fused_join_probe(const pa_fused_join_desc* desc, const pa_hash_builder_desc* build, char* buf, int64_t buf_size)
{
    int64_t need = 0;
    int32_t rc = guarded([&]() -> int32_t {
        std::string tu = jit_translation_unit(filter_project_probe_source_for_desc(desc, build));
        need = (int64_t)tu.size() + 1;
        if (buf && buf_size >= need) memcpy(buf, tu.c_str(), (size_t)need);
        return PA_OK;
    });
    return rc < 0 ? rc : need;
}
int64_t pa_codegen_compile_fused_join_probe(const pa_fused_join_desc* desc, const pa_hash_builder_desc* build)
{
    int64_t size = 0;
    int32_t rc = guarded([&]() -> int32_t {
        size = (int64_t)jit_compile_only(filter_project_probe_source_for_desc(desc, build)).size();
        return PA_OK;
    });
    return rc < 0 ? rc : size;
}

int64_t pa_codegen_filter_project(const pa_filter_project_desc* desc, char* buf, int64_t buf_size, char* key)
{
    int64_t need = 0;
    int32_t rc = guarded([&]() -> int32_t {
        std::string entry;
        std::string src = filter_project_source_for_desc(desc, &entry);
        std::string tu = jit_translation_unit(src);
        need = (int64_t)tu.size() + 1;
        if (buf && buf_size >= need) memcpy(buf, tu.c_str(), (size_t)need);
        if (key) {
            std::string k = jit_key(src);
            memcpy(key, k.c_str(), k.size() + 1);
        }
        return PA_OK;
    });
    return rc < 0 ? rc : need;
}

int64_t pa_codegen_compile_filter_project(const pa_filter_project_desc* desc)
{
    int64_t size = 0;
    int32_t rc = guarded([&]() -> int32_t {
        std::string entry;
        std::string src = filter_project_source_for_desc(desc, &entry);
        size = (int64_t)jit_compile_only(src).size();
        return PA_OK;
    });
    return rc < 0 ? rc : size;
}

}  // extern "C"

namespace pa {
pa_operator* make_hash_aggregation(const pa_hash_aggregation_desc* desc)
{
    pa_operator* op = nullptr;
    create_plain_aggregation(desc, &op);
    return op;
}
}  // namespace pa
