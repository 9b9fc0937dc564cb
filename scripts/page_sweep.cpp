// page_sweep.cpp -- Q1 + Q6 through the C ABI at the page sizes an unmodified Driver delivers, with no Python in the loop:
// a Driver thread's view (one needsInput + one addInput native call per page, as the JNI shim of INTEGRATION.md makes them).
//
//   page_sweep [--sf 100] [--steps 5] [--layout table|shuffled|separate|retained|host] [--rows N,N,...] [--shared-stream]
//   --shared-stream  the operators run on a stream the caller made (what a pipeline of device operators does: pages change hands in
//                    stream order, nobody waits); without it the operator owns its stream and drains it after every page that is
//                    not PA_PAGE_STABLE, because the caller may recycle the page's buffers once add_input has returned
//
//   table     pages are consecutive row ranges of resident columns, PA_PAGE_STABLE, handed over in table order
//   shuffled  the same pages in a seeded random order: no page continues its predecessor
//   separate  every page's columns were copied to buffers of their own (not stable: what a device operator upstream hands
//             over); bounded to --max-separate-gb of copies
//   retained  the same separate buffers, handed over as PA_PAGE_RETAINED with a release callback: the producer (an upstream device
//             operator handing its output buffers over, the staging pool of the JNI shim) keeps a page's buffers until the operator
//             says it has read them -- the sweep counts the releases: every page exactly once per pass, none missing at close
//   host      PA_MEM_HOST pages in pinned memory kept until the operator is closed (what the JNI shim's PinnedPagePool stages:
//             PA_PAGE_STABLE | PA_PAGE_PINNED), bounded sample
//   hostcopy  the same buffers without the flags: the library copies every block array with hipMemcpyAsync
// Prints one JSON line per (layout, page rows).  Built by scripts/collect_profiles.sh (g++, links libpresto_amd.so).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "presto_amd.hpp"

using namespace presto_amd;
using Clock = std::chrono::steady_clock;

static const int64_t kLineitemRowsPerSf = 6001215;

struct Column {
    int32_t column, type;
    void* values = nullptr;
    int32_t* offsets = nullptr;
    int width;
};

static Column make_column(int32_t column, int32_t type, double sf, int64_t rows)
{
    Column c;
    c.column = column;
    c.type = type;
    c.width = type == PA_VARCHAR ? 1 : (type == PA_DATE ? 4 : 8);
    check(pa_device_malloc(&c.values, rows * c.width + 64));
    if (type == PA_VARCHAR) check(pa_device_malloc(reinterpret_cast<void**>(&c.offsets), (rows + 1) * 4 + 64));
    check(pa_tpch_generate(column, sf, 0, rows, 0x5EED0000ULL, c.values, c.offsets, nullptr));
    return c;
}

struct PageSet {
    std::vector<pa_page> pages;
    std::vector<std::vector<pa_column>> cols;
};

static PageSet table_pages(const std::vector<Column>& columns, int64_t rows, int64_t page_rows, bool stable)
{
    PageSet ps;
    page_rows = std::max<int64_t>(4, page_rows - page_rows % 4);
    for (int64_t first = 0; first < rows; first += page_rows) {
        const int64_t n = std::min(page_rows, rows - first);
        std::vector<pa_column> cols(columns.size());
        for (size_t i = 0; i < columns.size(); i++) {
            const Column& c = columns[i];
            pa_column& col = cols[i];
            memset(&col, 0, sizeof col);
            col.type = c.type;
            if (c.type == PA_VARCHAR) {
                col.encoding = PA_VARWIDTH;
                col.values = c.values;
                col.offsets = c.offsets + first;
            }
            else {
                col.encoding = PA_FLAT;
                col.values = static_cast<const char*>(c.values) + first * c.width;
            }
        }
        ps.cols.push_back(std::move(cols));
        pa_page p{};
        p.position_count = (int32_t)n;
        p.channel_count = (int32_t)columns.size();
        p.mem = PA_MEM_DEVICE;
        p.flags = stable ? PA_PAGE_STABLE : 0;
        ps.pages.push_back(p);
    }
    for (size_t i = 0; i < ps.pages.size(); i++) ps.pages[i].columns = ps.cols[i].data();
    return ps;
}

struct Query {
    const char* name;
    std::vector<int32_t> types, params;
    std::vector<int32_t> tpch_columns;
    Expr filter;
    std::vector<Expr> projections;
    std::vector<int32_t> group_by;
    std::vector<pa_aggregate> aggregates;
    int bytes_per_row;
};

static Expr cmp(int32_t op, Expr a, Expr b) { return call(op, PA_BOOLEAN, {a, b}); }

static Query q6()
{
    Query q;
    q.name = "q6";
    q.types = {PA_DATE, PA_DOUBLE, PA_DOUBLE, PA_DOUBLE};
    q.params = {0, 0, 0, 0};
    q.tpch_columns = {PA_L_SHIPDATE, PA_L_DISCOUNT, PA_L_QUANTITY, PA_L_EXTENDEDPRICE};
    Expr shipdate = field(0, PA_DATE), discount = field(1, PA_DOUBLE), quantity = field(2, PA_DOUBLE);
    q.filter = specialForm(PA_FORM_AND, PA_BOOLEAN,
                           {cmp(PA_OP_GREATER_THAN_OR_EQUAL, shipdate, constantLong(8766, PA_DATE)), cmp(PA_OP_LESS_THAN, shipdate, constantLong(9131, PA_DATE)),
                            cmp(PA_OP_GREATER_THAN_OR_EQUAL, discount, constantDouble(0.05)), cmp(PA_OP_LESS_THAN_OR_EQUAL, discount, constantDouble(0.07)),
                            cmp(PA_OP_LESS_THAN, quantity, constantDouble(24.0))});
    q.projections = {call(PA_OP_MULTIPLY, PA_DOUBLE, {field(3, PA_DOUBLE), field(1, PA_DOUBLE)})};
    q.aggregates = {pa_aggregate{PA_AGG_SUM, 0, -1, PA_DOUBLE}};
    q.bytes_per_row = 28;
    return q;
}

static Query q1()
{
    Query q;
    q.name = "q1";
    q.types = {PA_VARCHAR, PA_VARCHAR, PA_DOUBLE, PA_DOUBLE, PA_DOUBLE, PA_DOUBLE, PA_DATE};
    q.params = {1, 1, 0, 0, 0, 0, 0};
    q.tpch_columns = {PA_L_RETURNFLAG, PA_L_LINESTATUS, PA_L_QUANTITY, PA_L_EXTENDEDPRICE, PA_L_DISCOUNT, PA_L_TAX, PA_L_SHIPDATE};
    q.filter = cmp(PA_OP_LESS_THAN_OR_EQUAL, field(6, PA_DATE), constantLong(10471, PA_DATE));
    Expr qty = field(2, PA_DOUBLE), price = field(3, PA_DOUBLE), disc = field(4, PA_DOUBLE), tax = field(5, PA_DOUBLE), one = constantDouble(1.0);
    Expr disc_price = call(PA_OP_MULTIPLY, PA_DOUBLE, {price, call(PA_OP_SUBTRACT, PA_DOUBLE, {one, disc})});
    Expr charge = call(PA_OP_MULTIPLY, PA_DOUBLE, {call(PA_OP_MULTIPLY, PA_DOUBLE, {price, call(PA_OP_SUBTRACT, PA_DOUBLE, {one, disc})}),
                                                   call(PA_OP_ADD, PA_DOUBLE, {one, tax})});
    q.projections = {field(0, PA_VARCHAR), field(1, PA_VARCHAR), qty, price, disc_price, charge, disc};
    q.group_by = {0, 1};
    q.aggregates = {pa_aggregate{PA_AGG_SUM, 2, -1, PA_DOUBLE}, pa_aggregate{PA_AGG_SUM, 3, -1, PA_DOUBLE}, pa_aggregate{PA_AGG_SUM, 4, -1, PA_DOUBLE},
                    pa_aggregate{PA_AGG_SUM, 5, -1, PA_DOUBLE}, pa_aggregate{PA_AGG_AVG, 2, -1, PA_DOUBLE}, pa_aggregate{PA_AGG_AVG, 3, -1, PA_DOUBLE},
                    pa_aggregate{PA_AGG_AVG, 6, -1, PA_DOUBLE}, pa_aggregate{PA_AGG_COUNT_STAR, -1, -1, 0}};
    q.bytes_per_row = 46;
    return q;
}

// the planner's part, once per plan: the serialised descriptor; createOperator() then only hands it to the native factory
struct Factory {
    explicit Factory(const Query& q) : query(q)
    {
        if (q.filter) filter = std::make_unique<SerializedExpression>(q.filter);
        for (const auto& p : q.projections) {
            projections.push_back(std::make_unique<SerializedExpression>(p));
            pexprs.push_back(*projections.back()->get());
            ptypes.push_back(p->node.type);
        }
        memset(&desc, 0, sizeof desc);
        desc.filter_project.input_channel_count = (int32_t)q.types.size();
        desc.filter_project.input_types = query.types.data();
        desc.filter_project.input_type_params = query.params.data();
        desc.filter_project.filter = filter ? filter->get() : nullptr;
        desc.filter_project.projection_count = (int32_t)pexprs.size();
        desc.filter_project.projections = pexprs.data();
        desc.filter_project.output_mem = PA_MEM_HOST;
        desc.aggregation.input_channel_count = (int32_t)ptypes.size();
        desc.aggregation.input_types = ptypes.data();
        desc.aggregation.group_by_count = (int32_t)query.group_by.size();
        desc.aggregation.group_by_channels = query.group_by.data();
        desc.aggregation.hash_channel = -1;
        desc.aggregation.step = PA_STEP_SINGLE;
        desc.aggregation.aggregate_count = (int32_t)query.aggregates.size();
        desc.aggregation.aggregates = query.aggregates.data();
        desc.aggregation.expected_groups = 10000;
        desc.aggregation.output_mem = PA_MEM_HOST;
    }
    pa_operator* create()
    {
        pa_operator* h = nullptr;
        check(pa_fused_aggregation_create(&desc, &h));
        return h;
    }
    Query query;
    std::unique_ptr<SerializedExpression> filter;
    std::vector<std::unique_ptr<SerializedExpression>> projections;
    std::vector<pa_expr> pexprs;
    std::vector<int32_t> ptypes;
    pa_fused_aggregation_desc desc;
};

// retained layout: what the producer's release callback does here -- count (a staging pool would put the slot back on its free list)
struct ReleaseCounter {
    std::vector<int32_t> per_page;
    int64_t total = 0;
};
struct ReleaseCtx {
    ReleaseCounter* counter;
    size_t page;
};
static void on_release(void* ctx)
{
    ReleaseCtx* c = static_cast<ReleaseCtx*>(ctx);
    c->counter->per_page[c->page]++;
    c->counter->total++;
}

// one pass of a query over its pages with the Driver's call protocol (Driver.java:355-457); returns the spins spent waiting
static int64_t run_pass(Factory& f, const std::vector<pa_page>& pages, double* first_value, ReleaseCounter* releases = nullptr)
{
    pa_operator* op = f.create();
    int64_t spins = 0;
    if (releases) {
        releases->per_page.assign(pages.size(), 0);
        releases->total = 0;
    }
    for (const pa_page& p : pages) {
        while (check(pa_op_needs_input(op)) == 0) {
            // the Driver would yield on isBlocked's future; this thread has nothing else to do
            // (the launch may finish between the two calls: ask again before complaining)
            if (check(pa_op_is_blocked(op)) == 0 && check(pa_op_needs_input(op)) == 0) throw std::runtime_error("operator refuses input without being blocked");
            spins++;
        }
        check(pa_op_add_input(op, &p));
    }
    check(pa_op_finish(op));
    pa_page out{};
    while (check(pa_op_is_finished(op)) == 0) {
        if (check(pa_op_get_output(op, &out)) == 1 && out.position_count > 0 && first_value) {
            const pa_column& c = out.columns[out.channel_count - 1];
            memcpy(first_value, c.values, 8);
        }
    }
    check(pa_op_close(op));
    if (releases) {
        for (size_t i = 0; i < pages.size(); i++) {
            if (releases->per_page[i] != 1) throw std::runtime_error("retained page " + std::to_string(i) + " released " + std::to_string(releases->per_page[i]) + " times");
        }
    }
    return spins;
}

int main(int argc, char** argv)
{
    double sf = 100.0;
    int steps = 5;
    std::string layout = "table";
    std::vector<int64_t> sizes = {1LL << 28, 1LL << 22, 1LL << 20, 1LL << 16, 8192};
    double max_copy_gb = 4.0;
    bool shared_stream = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--shared-stream") shared_stream = true;
        if (a == "--sf" && i + 1 < argc) sf = atof(argv[++i]);
        else if (a == "--steps" && i + 1 < argc) steps = atoi(argv[++i]);
        else if (a == "--layout" && i + 1 < argc) layout = argv[++i];
        else if (a == "--max-copy-gb" && i + 1 < argc) max_copy_gb = atof(argv[++i]);
        else if (a == "--rows" && i + 1 < argc) {
            sizes.clear();
            char* s = argv[++i];
            for (char* t = strtok(s, ","); t; t = strtok(nullptr, ",")) sizes.push_back(atoll(t));
        }
    }
    check(pa_init(0));
    void* stream = nullptr;
    if (shared_stream) check(pa_stream_create(&stream));
    int64_t rows = (int64_t)(kLineitemRowsPerSf * sf);
    const bool host_layout = layout == "host" || layout == "hostcopy";
    const bool retained = layout == "retained";
    const bool bounded = layout == "separate" || retained || host_layout;
    if (bounded) rows = std::min<int64_t>(rows, (int64_t)(max_copy_gb * 1e9 / 46));
    std::vector<Query> queries = {q6(), q1()};
    // resident columns (the union of both queries' channels is generated once per query here: they share nothing but HBM)
    struct Loaded { Factory* f; std::vector<Column> cols; };
    std::vector<Factory*> factories;
    std::vector<std::vector<Column>> columns;
    for (const Query& q : queries) {
        factories.push_back(new Factory(q));
        factories.back()->desc.filter_project.stream = stream;
        factories.back()->desc.aggregation.stream = stream;
        std::vector<Column> cs;
        for (size_t i = 0; i < q.tpch_columns.size(); i++) cs.push_back(make_column(q.tpch_columns[i], q.types[i], sf, rows));
        columns.push_back(std::move(cs));
    }
    check(pa_stream_synchronize(nullptr));
    for (int64_t page_rows : sizes) {
        std::vector<PageSet> sets;
        std::vector<std::vector<void*>> owned_dev, owned_host;
        std::vector<ReleaseCounter> counters(queries.size());
        std::vector<std::vector<ReleaseCtx>> release_ctx(queries.size());
        for (size_t qi = 0; qi < queries.size(); qi++) {
            PageSet ps = table_pages(columns[qi], rows, page_rows, layout == "table" || layout == "shuffled");
            if (layout == "shuffled") {
                std::mt19937_64 rng(0x5EED);
                std::vector<size_t> order(ps.pages.size());
                for (size_t i = 0; i < order.size(); i++) order[i] = i;
                std::shuffle(order.begin(), order.end(), rng);
                std::vector<pa_page> shuffled;
                for (size_t i : order) shuffled.push_back(ps.pages[i]);
                ps.pages = shuffled;
            }
            if (bounded) {
                // every page's blocks in buffers of their own (device: one allocation per page and column, so that no page
                // continues another; host: pinned, as the JNI shim stages them)
                owned_dev.emplace_back();
                owned_host.emplace_back();
                for (size_t pi = 0; pi < ps.pages.size(); pi++) {
                    const int64_t n = ps.pages[pi].position_count;
                    for (size_t c = 0; c < ps.cols[pi].size(); c++) {
                        pa_column& col = ps.cols[pi][c];
                        const Column& src = columns[qi][c];
                        if (col.encoding == PA_VARWIDTH) {
                            // VARCHAR(1) flags: one byte per row, offsets rebased to the page
                            std::vector<int32_t> off((size_t)n + 1);
                            for (int64_t r = 0; r <= n; r++) off[(size_t)r] = (int32_t)r;
                            const int64_t first = col.offsets - src.offsets;
                            void *dv = nullptr, *dof = nullptr;
                            if (host_layout) {
                                check(pa_host_malloc_pinned(&dv, n + 16));
                                check(pa_host_malloc_pinned(&dof, (n + 1) * 4));
                                check(pa_memcpy_d2h(dv, static_cast<const char*>(src.values) + first, n, nullptr));
                                memcpy(dof, off.data(), (size_t)(n + 1) * 4);
                                owned_host.back().push_back(dv);
                                owned_host.back().push_back(dof);
                            }
                            else {
                                check(pa_device_malloc(&dv, n + 16));
                                check(pa_device_malloc(&dof, (n + 1) * 4));
                                std::vector<char> tmp((size_t)n);
                                check(pa_memcpy_d2h(tmp.data(), static_cast<const char*>(src.values) + first, n, nullptr));
                                check(pa_memcpy_h2d(dv, tmp.data(), n, nullptr));
                                check(pa_memcpy_h2d(dof, off.data(), (n + 1) * 4, nullptr));
                                owned_dev.back().push_back(dv);
                                owned_dev.back().push_back(dof);
                            }
                            col.values = dv;
                            col.offsets = static_cast<const int32_t*>(dof);
                        }
                        else {
                            void* dv = nullptr;
                            const int64_t bytes = n * src.width;
                            if (host_layout) {
                                check(pa_host_malloc_pinned(&dv, bytes + 16));
                                check(pa_memcpy_d2h(dv, col.values, bytes, nullptr));
                                owned_host.back().push_back(dv);
                            }
                            else {
                                check(pa_device_malloc(&dv, bytes + 16));
                                std::vector<char> tmp((size_t)bytes);
                                check(pa_memcpy_d2h(tmp.data(), col.values, bytes, nullptr));
                                check(pa_memcpy_h2d(dv, tmp.data(), bytes, nullptr));
                                owned_dev.back().push_back(dv);
                            }
                            col.values = dv;
                        }
                    }
                    ps.pages[pi].mem = host_layout ? PA_MEM_HOST : PA_MEM_DEVICE;
                    ps.pages[pi].flags = layout == "host" ? (PA_PAGE_STABLE | PA_PAGE_PINNED) : (retained ? PA_PAGE_RETAINED : 0);
                }
                if (retained) {
                    release_ctx[qi].resize(ps.pages.size());
                    for (size_t pi = 0; pi < ps.pages.size(); pi++) {
                        release_ctx[qi][pi] = ReleaseCtx{&counters[qi], pi};
                        ps.pages[pi].release = &on_release;
                        ps.pages[pi].release_ctx = &release_ctx[qi][pi];
                    }
                }
            }
            sets.push_back(std::move(ps));
        }
        for (size_t qi = 0; qi < sets.size(); qi++) {
            for (size_t i = 0; i < sets[qi].pages.size(); i++) {
                // (the shuffle moved the page structs: their column arrays are still the ones of the table order)
            }
        }
        double v6 = 0, v1 = 0;
        int64_t spins = 0;
        for (size_t qi = 0; qi < queries.size(); qi++) run_pass(*factories[qi], sets[qi].pages, nullptr, retained ? &counters[qi] : nullptr);  // warm-up (JIT cache, pools)
        check(pa_stream_synchronize(nullptr));
        const auto t0 = Clock::now();
        double per_query_s[2] = {0, 0};
        for (int s = 0; s < steps; s++) {
            for (size_t qi = 0; qi < queries.size(); qi++) {
                const auto q0 = Clock::now();
                spins += run_pass(*factories[qi], sets[qi].pages, qi == 0 ? &v6 : &v1, retained ? &counters[qi] : nullptr);
                per_query_s[qi] += std::chrono::duration<double>(Clock::now() - q0).count();
            }
        }
        const double dt = std::chrono::duration<double>(Clock::now() - t0).count();
        const double rps = 2.0 * rows * steps / dt;
        printf("{\"layout\": \"%s%s\", \"page_rows\": %lld, \"pages_per_query\": %zu, \"rows\": %lld, \"steps\": %d, \"ms_per_step\": %.3f, "
               "\"rows_per_s\": %.4g, \"q6_rows_per_s\": %.4g, \"q1_rows_per_s\": %.4g, \"q6_GBps\": %.1f, \"q1_GBps\": %.1f, "
               "\"needs_input_spins_per_step\": %.1f, \"q6_revenue\": %.6f, \"q1_last_count\": %lld}\n",
               layout.c_str(), shared_stream ? ", caller's stream" : "", (long long)page_rows, sets[0].pages.size(), (long long)rows, steps, dt / steps * 1e3, rps,
               rows * steps / per_query_s[0], rows * steps / per_query_s[1], rows * steps / per_query_s[0] * 28 / 1e9,
               rows * steps / per_query_s[1] * 46 / 1e9, (double)spins / steps, v6, (long long)*reinterpret_cast<int64_t*>(&v1));
        fflush(stdout);
        for (auto& v : owned_dev)
            for (void* p : v) pa_device_free(p);
        for (auto& v : owned_host)
            for (void* p : v) pa_host_free_pinned(p);
    }
    return 0;
}
