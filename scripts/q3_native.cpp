// q3_native.cpp -- TPC-H Q3 through the C ABI with a C++ Driver loop (include/presto_amd.hpp's runDriver): the three pipelines of
// presto_amd/q3.py (one rank, fused probes) with no Python between the operator calls -- what the host side costs when it is
// compiled code, as the reference's Driver is.  (presto_amd.hpp's Page / runDriver hold host pages; the pipelines here move
// device pages, so the same loop runs over pa_page views.)
//
//   q3_native [--sf 100] [--steps 10] [--warmup 2] [--page-rows 268435456]
//
//   pipeline 1  customer: FilterAndProject(mktsegment = 'BUILDING' -> custkey)        -> HashBuilder(b1; key custkey)
//   pipeline 2  orders:   FilterAndProject(orderdate < 1995-03-15) -> LookupJoin(b1)   -> HashBuilder(b2; key orderkey)   [pa_fused_join_create]
//   pipeline 3  lineitem: FilterAndProject(shipdate > 1995-03-15; revenue) -> LookupJoin(b2) -> HashAggregation(orderkey, orderdate,
//               shippriority; sum(revenue)) [pa_fused_join_aggregation_create] -> TopN(10; revenue DESC, orderdate ASC)
// Prints one JSON line: ms per step, input rows/s, per-pipeline ms and the ten result rows.  Built by __graft_entry__.build_native_q3().
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "presto_amd.hpp"

using namespace presto_amd;
using Clock = std::chrono::steady_clock;

struct Column {
    int32_t type;
    void* values = nullptr;
    int32_t* offsets = nullptr;
    int width;
};

static Column make_column(int32_t column, int32_t type, double sf, int64_t rows, int max_bytes = 1)
{
    Column c;
    c.type = type;
    c.width = type == PA_VARCHAR ? 1 : ((type == PA_DATE || type == PA_INTEGER) ? 4 : 8);
    check(pa_device_malloc(&c.values, rows * (type == PA_VARCHAR ? max_bytes : c.width) + 64));
    if (type == PA_VARCHAR) check(pa_device_malloc(reinterpret_cast<void**>(&c.offsets), (rows + 1) * 4 + 64));
    check(pa_tpch_generate(column, sf, 0, rows, 0x5EED0000ULL, c.values, c.offsets, nullptr));
    return c;
}

// the table's pages: consecutive row ranges of the resident columns (Page.getRegion), PA_PAGE_STABLE
struct PageSet {
    std::vector<pa_page> pages;
    std::vector<std::vector<pa_column>> cols;
};
static PageSet table_pages(const std::vector<Column>& columns, int64_t rows, int64_t page_rows)
{
    PageSet ps;
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
        p.flags = PA_PAGE_STABLE;
        ps.pages.push_back(p);
    }
    for (size_t i = 0; i < ps.pages.size(); i++) ps.pages[i].columns = ps.cols[i].data();
    return ps;
}

// the planner's part: descriptors made once (OperatorFactory), operators created from them per step
struct FilterProject {
    std::vector<int32_t> types;
    std::unique_ptr<SerializedExpression> filter;
    std::vector<std::unique_ptr<SerializedExpression>> projections;
    std::vector<pa_expr> pexprs;
    pa_filter_project_desc desc{};
    FilterProject(std::vector<int32_t> t, const Expr& f, const std::vector<Expr>& ps, void* stream) : types(std::move(t))
    {
        if (f) filter = std::make_unique<SerializedExpression>(f);
        for (const auto& p : ps) {
            projections.push_back(std::make_unique<SerializedExpression>(p));
            pexprs.push_back(*projections.back()->get());
        }
        desc.input_channel_count = (int32_t)types.size();
        desc.input_types = types.data();
        desc.filter = filter ? filter->get() : nullptr;
        desc.projection_count = (int32_t)pexprs.size();
        desc.projections = pexprs.data();
        desc.output_mem = PA_MEM_DEVICE;
        desc.stream = stream;
    }
};
struct LookupJoin {
    std::vector<int32_t> types, join_channels, output_channels;
    pa_lookup_join_desc desc{};
    LookupJoin(std::vector<int32_t> t, std::vector<int32_t> jc, std::vector<int32_t> oc, void* stream)
        : types(std::move(t)), join_channels(std::move(jc)), output_channels(std::move(oc))
    {
        desc.probe_channel_count = (int32_t)types.size();
        desc.probe_types = types.data();
        desc.join_channel_count = (int32_t)join_channels.size();
        desc.probe_join_channels = join_channels.data();
        desc.probe_hash_channel = -1;
        desc.probe_output_channel_count = (int32_t)output_channels.size();
        desc.probe_output_channels = output_channels.data();
        desc.output_mem = PA_MEM_DEVICE;
        desc.join_type = PA_JOIN_INNER;
        desc.stream = stream;
    }
};
struct HashBuilder {
    std::vector<int32_t> types, join_channels, output_channels;
    pa_hash_builder_desc desc{};
    HashBuilder(std::vector<int32_t> t, std::vector<int32_t> jc, std::vector<int32_t> oc, void* stream)
        : types(std::move(t)), join_channels(std::move(jc)), output_channels(std::move(oc))
    {
        desc.input_channel_count = (int32_t)types.size();
        desc.input_types = types.data();
        desc.join_channel_count = (int32_t)join_channels.size();
        desc.join_channels = join_channels.data();
        desc.hash_channel = -1;
        desc.output_channel_count = (int32_t)output_channels.size();
        desc.output_channels = output_channels.data();
        desc.stream = stream;
    }
    std::unique_ptr<Operator> create(LookupSourceFactory& bridge)
    {
        pa_operator* h = nullptr;
        check(pa_hash_builder_create(&desc, bridge.handle(), &h));
        return std::make_unique<Operator>(h);
    }
};

// Driver.processInternal (Driver.java:355-457) over native handles and pa_page views: device pages move between neighbours without a
// copy (a page an operator returns stays valid until the next call on that operator: it is handed on at once); the pages of the last
// operator -- PA_MEM_HOST here -- are copied out.
static std::vector<Page> run_driver(const std::vector<pa_page>& source, const std::vector<pa_operator*>& ops)
{
    std::vector<Page> output;
    size_t next = 0;
    bool source_finished = false;
    for (int guard = 0; guard < (1 << 22); guard++) {
        if (next < source.size() && check(pa_op_needs_input(ops[0])) == 1) {
            if (source[next].position_count > 0) check(pa_op_add_input(ops[0], &source[next]));
            next++;
        }
        if (next == source.size() && !source_finished) {
            source_finished = true;
            check(pa_op_finish(ops[0]));
        }
        for (size_t i = 0; i + 1 < ops.size(); i++) {
            const bool finished = check(pa_op_is_finished(ops[i])) == 1;
            if (!finished && check(pa_op_needs_input(ops[i + 1])) == 1) {
                pa_page page{};
                if (check(pa_op_get_output(ops[i], &page)) == 1 && page.position_count > 0) check(pa_op_add_input(ops[i + 1], &page));
            }
            if (check(pa_op_is_finished(ops[i])) == 1) check(pa_op_finish(ops[i + 1]));
        }
        pa_page page{};
        if (check(pa_op_get_output(ops.back(), &page)) == 1 && page.position_count > 0 && page.mem == PA_MEM_HOST) output.push_back(Page::fromNative(page));
        if (check(pa_op_is_finished(ops.back())) == 1) return output;
    }
    throw std::runtime_error("pipeline did not finish");
}

int main(int argc, char** argv)
{
    double sf = 100.0;
    int steps = 10, warmup = 2;
    int64_t page_rows = 1 << 28;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        if (a == "--sf") sf = atof(argv[i + 1]);
        else if (a == "--steps") steps = atoi(argv[i + 1]);
        else if (a == "--warmup") warmup = atoi(argv[i + 1]);
        else if (a == "--page-rows") page_rows = atoll(argv[i + 1]);
    }
    check(pa_init(0));
    int64_t nc = (int64_t)(150000 * sf);
    nc -= nc % 5;
    const int64_t no = (int64_t)(1500000 * sf), nl = (int64_t)(6001215 * sf);
    std::vector<Column> customer = {make_column(PA_C_CUSTKEY, PA_BIGINT, sf, nc), make_column(PA_C_MKTSEGMENT, PA_VARCHAR, sf, nc, 10)};
    std::vector<Column> orders = {make_column(PA_O_ORDERKEY, PA_BIGINT, sf, no), make_column(PA_O_CUSTKEY, PA_BIGINT, sf, no),
                                  make_column(PA_O_ORDERDATE, PA_DATE, sf, no), make_column(PA_O_SHIPPRIORITY, PA_INTEGER, sf, no)};
    std::vector<Column> lineitem = {make_column(PA_L_ORDERKEY, PA_BIGINT, sf, nl), make_column(PA_L_EXTENDEDPRICE, PA_DOUBLE, sf, nl),
                                    make_column(PA_L_DISCOUNT, PA_DOUBLE, sf, nl), make_column(PA_L_SHIPDATE, PA_DATE, sf, nl)};
    const PageSet cps = table_pages(customer, nc, page_rows - page_rows % 20), ops = table_pages(orders, no, page_rows),
                  lps = table_pages(lineitem, nl, page_rows);
    void* stream = nullptr;
    check(pa_stream_create(&stream));

    // ---- the plan ----
    auto varchar = [](const char* s) {
        auto e = std::make_shared<RowExpression>();
        e->node.kind = PA_EXPR_CONSTANT;
        e->node.type = PA_VARCHAR;
        e->text = s;
        return e;
    };
    auto cmp = [](int32_t op, Expr a, Expr b) { return call(op, PA_BOOLEAN, {a, b}); };
    FilterProject customer_fp({PA_BIGINT, PA_VARCHAR}, cmp(PA_OP_EQUAL, field(1, PA_VARCHAR), varchar("BUILDING")), {field(0, PA_BIGINT)}, stream);
    customer_fp.desc.output_handover = 1;  // (the build sides leave their FilterAndProject with their buffers: the HashBuilders read them in place)
    HashBuilder build1({PA_BIGINT}, {0}, {}, stream);
    const std::vector<int32_t> orders_types = {PA_BIGINT, PA_BIGINT, PA_DATE, PA_INTEGER};
    FilterProject orders_fp(orders_types, cmp(PA_OP_LESS_THAN, field(2, PA_DATE), constantLong(9204, PA_DATE)),
                            {field(0, PA_BIGINT), field(1, PA_BIGINT), field(2, PA_DATE), field(3, PA_INTEGER)}, stream);
    LookupJoin orders_join(orders_types, {1}, {0, 2, 3}, stream);
    orders_fp.desc.output_handover = 1;
    pa_fused_join_desc orders_desc{};
    orders_desc.filter_project = orders_fp.desc;
    orders_desc.join = orders_join.desc;
    HashBuilder build2({PA_BIGINT, PA_DATE, PA_INTEGER}, {0}, {1, 2}, stream);
    FilterProject lineitem_fp({PA_BIGINT, PA_DOUBLE, PA_DOUBLE, PA_DATE}, cmp(PA_OP_GREATER_THAN, field(3, PA_DATE), constantLong(9204, PA_DATE)),
                              {field(0, PA_BIGINT), call(PA_OP_MULTIPLY, PA_DOUBLE,
                                                         {field(1, PA_DOUBLE), call(PA_OP_SUBTRACT, PA_DOUBLE, {constantDouble(1.0), field(2, PA_DOUBLE)})})},
                              stream);
    LookupJoin lineitem_join({PA_BIGINT, PA_DOUBLE}, {0}, {0, 1}, stream);
    const std::vector<int32_t> agg_types = {PA_BIGINT, PA_DOUBLE, PA_DATE, PA_INTEGER}, group_by = {0, 2, 3};
    const std::vector<pa_aggregate> aggregates = {pa_aggregate{PA_AGG_SUM, 1, -1, PA_DOUBLE}};
    pa_fused_join_aggregation_desc lineitem_desc{};
    lineitem_desc.filter_project = lineitem_fp.desc;
    lineitem_desc.join = lineitem_join.desc;
    lineitem_desc.aggregation.input_channel_count = (int32_t)agg_types.size();
    lineitem_desc.aggregation.input_types = agg_types.data();
    lineitem_desc.aggregation.group_by_count = (int32_t)group_by.size();
    lineitem_desc.aggregation.group_by_channels = group_by.data();
    lineitem_desc.aggregation.hash_channel = -1;
    lineitem_desc.aggregation.step = PA_STEP_SINGLE;
    lineitem_desc.aggregation.aggregate_count = (int32_t)aggregates.size();
    lineitem_desc.aggregation.aggregates = aggregates.data();
    lineitem_desc.aggregation.output_mem = PA_MEM_DEVICE;
    lineitem_desc.aggregation.stream = stream;
    const std::vector<int32_t> result_types = {PA_BIGINT, PA_DATE, PA_INTEGER, PA_DOUBLE}, sort_channels = {3, 1},
                               sort_orders = {PA_DESC_NULLS_LAST, PA_ASC_NULLS_LAST};
    pa_topn_desc top{};
    top.input_channel_count = (int32_t)result_types.size();
    top.input_types = result_types.data();
    top.n = 10;
    top.sort_channel_count = 2;
    top.sort_channels = sort_channels.data();
    top.sort_orders = sort_orders.data();
    top.output_mem = PA_MEM_HOST;
    top.stream = stream;

    double stage_ms[3] = {0, 0, 0};
    std::vector<Page> result;
    auto step = [&](bool timed) {
        auto lap = [&](int i, Clock::time_point& t0) {
            check(pa_stream_synchronize(stream));
            const auto t1 = Clock::now();
            if (timed) stage_ms[i] += std::chrono::duration<double, std::milli>(t1 - t0).count();
            t0 = t1;
        };
        auto t0 = Clock::now();
        LookupSourceFactory b1, b2;
        {
            pa_operator* h = nullptr;
            check(pa_filter_project_create(&customer_fp.desc, &h));
            Operator fp(h);
            auto builder = build1.create(b1);
            run_driver(cps.pages, {fp.handle(), builder->handle()});
        }
        lap(0, t0);
        {
            pa_operator* h = nullptr;
            check(pa_fused_join_create(&orders_desc, b1.handle(), &h));
            Operator join(h);
            auto builder = build2.create(b2);
            run_driver(ops.pages, {join.handle(), builder->handle()});
        }
        lap(1, t0);
        {
            lineitem_desc.aggregation.expected_groups = pa_lookup_source_position_count(b2.handle());  // orderkey is unique on the build side
            pa_operator* h = nullptr;
            check(pa_fused_join_aggregation_create(&lineitem_desc, b2.handle(), &h));
            Operator agg(h);
            // (the planner's note that the aggregation's only consumer is this TopN, as presto_amd/q3.py passes it)
            agg.setOutputTopNHint(top.n, sort_channels, sort_orders);
            pa_operator* t = nullptr;
            check(pa_topn_create(&top, &t));
            Operator topn(t);
            result = run_driver(lps.pages, {agg.handle(), topn.handle()});
        }
        lap(2, t0);
    };
    for (int i = 0; i < warmup; i++) step(false);
    check(pa_stream_synchronize(stream));
    const auto t0 = Clock::now();
    for (int i = 0; i < steps; i++) step(true);
    check(pa_stream_synchronize(stream));
    const double ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count() / steps;
    const double rows = (double)(nc + no + nl);
    printf("{\"metric\": \"input rows/s through the TPC-H Q3 operator pipelines, C++ Driver loop over the C ABI\", \"value\": %.6g, \"unit\": \"rows/s\", "
           "\"ms_per_step\": %.4f, \"steps\": %d, \"scale_factor\": %g, \"input_rows\": %.0f, \"stage_ms\": {\"customer\": %.4f, \"orders\": %.4f, \"lineitem\": %.4f}, "
           "\"result\": [",
           rows / (ms / 1e3), ms, steps, sf, rows, stage_ms[0] / steps, stage_ms[1] / steps, stage_ms[2] / steps);
    bool first = true;
    for (const Page& p : result) {
        for (int32_t r = 0; r < p.getPositionCount(); r++) {
            printf("%s[%lld, %lld, %lld, %.17g]", first ? "" : ", ", (long long)p.getBlock(0).getLong(r), (long long)p.getBlock(1).getLong(r),
                   (long long)p.getBlock(2).getLong(r), p.getBlock(3).getDouble(r));
            first = false;
        }
    }
    printf("]}\n");
    check(pa_stream_destroy(stream));
    return 0;
}
