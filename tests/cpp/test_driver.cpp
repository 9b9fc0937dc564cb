// C++ driver test: runs the reference's operator known-answer cases through include/presto_amd.hpp (the C++ host
// mirror) on the GPU, and checks them against the oracle (liboracle.so) on the same pages.
//   fp-1   TestFilterAndProjectOperator.test   core/trino-main/src/test/java/io/trino/operator/TestFilterAndProjectOperator.java:78-124
//   hagg-1 TestHashAggregationOperator.testHashAggregation   core/trino-main/src/test/java/io/trino/operator/TestHashAggregationOperator.java:160-219
//   join-1 TestHashJoinOperator.testInnerJoin   core/trino-main/src/test/java/io/trino/operator/join/TestHashJoinOperator.java:192-229
// plus a two-operator Driver pipeline FilterAndProject -> HashAggregation (Driver.java:355-457 call order).
// Built by __graft_entry__.build(); executed by tests/test_gpu_cpp_driver.py.  Exit code 0 = all cases pass.
#include <cmath>
#include <cstdio>
#include <map>

#include "presto_amd.hpp"
#include "presto_oracle.h"

using namespace presto_amd;

static int failures = 0;
#define EXPECT(cond, ...)                              \
    do {                                               \
        if (!(cond)) {                                 \
            failures++;                                \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);              \
            fprintf(stderr, "\n");                     \
        }                                              \
    } while (0)

static Page sequencePage(int32_t length, int64_t start)
{
    std::vector<std::string> s;
    std::vector<int64_t> v;
    for (int32_t i = 0; i < length; i++) {
        s.push_back(std::to_string(start + i));
        v.push_back(start + i);
    }
    return Page({Block::varchar(s), Block::bigint(v)});
}

// fp-1, the reference's literal expressions (TestFilterAndProjectOperator.java:86-96): filter LESS_THAN_OR_EQUAL(field1, 9),
// projections (field0, ADD(field1, 5)) over the sequence page (100, 0, 0) -- here handed over twice -> rows ("0", 5) .. ("9", 14)
static void testFilterAndProject()
{
    std::vector<int32_t> types = {PA_VARCHAR, PA_BIGINT};
    Expr filter = call(PA_OP_LESS_THAN_OR_EQUAL, PA_BOOLEAN, {field(1, PA_BIGINT), constantLong(9)});
    std::vector<Expr> projections = {field(0, PA_VARCHAR), call(PA_OP_ADD, PA_BIGINT, {field(1, PA_BIGINT), constantLong(5)})};
    auto op = createFilterAndProjectOperator(types, filter, projections);
    std::vector<Page> input = {sequencePage(100, 0), sequencePage(100, 0)};
    auto out = runDriver(input, {op.get()});
    int64_t rows = 0;
    for (const auto& p : out) {
        for (int32_t i = 0; i < p.getPositionCount(); i++, rows++) {
            int64_t expect = rows % 10;
            EXPECT(p.getBlock(0).getSlice(i) == std::to_string(expect), "fp-1 row %ld varchar %s", (long)rows, p.getBlock(0).getSlice(i).c_str());
            EXPECT(p.getBlock(1).getLong(i) == expect + 5, "fp-1 row %ld bigint %ld", (long)rows, (long)p.getBlock(1).getLong(i));
        }
    }
    EXPECT(rows == 20, "fp-1 expected 20 rows, got %ld", (long)rows);

    // the same pages through the oracle's PageProcessor restatement
    SerializedExpression f(filter);
    std::vector<std::unique_ptr<SerializedExpression>> ps;
    std::vector<pa_expr> pe;
    for (auto& p : projections) {
        ps.push_back(std::make_unique<SerializedExpression>(p));
        pe.push_back(*ps.back()->get());
    }
    pa_page in;
    std::vector<pa_column> cols;
    input[0].toNative(&in, cols);
    pa_page ref{};
    int32_t rc = orc_filter_project(&in, f.get(), (int32_t)pe.size(), pe.data(), &ref);
    EXPECT(rc >= 0, "oracle filter_project: %s", orc_last_error());
    Page expected = Page::fromNative(ref);
    orc_free_page(&ref);
    EXPECT(!out.empty() && expected.getPositionCount() == out[0].getPositionCount(), "fp-1 oracle row count");
    for (int32_t i = 0; !out.empty() && i < expected.getPositionCount() && i < out[0].getPositionCount(); i++) {
        EXPECT(expected.getBlock(0).getSlice(i) == out[0].getBlock(0).getSlice(i), "fp-1 oracle varchar row %d", i);
        EXPECT(expected.getBlock(1).getLong(i) == out[0].getBlock(1).getLong(i), "fp-1 oracle bigint row %d", i);
    }
}

// division by zero must surface as the reference's DIVISION_BY_ZERO, not as a value
static void testDivisionByZero()
{
    std::vector<int32_t> types = {PA_VARCHAR, PA_BIGINT};
    std::vector<Expr> projections = {call(PA_OP_DIVIDE, PA_BIGINT, {constantLong(100), field(1, PA_BIGINT)})};
    auto op = createFilterAndProjectOperator(types, nullptr, projections);
    bool thrown = false;
    try {
        op->addInput(sequencePage(10, 0));
        op->getOutput();
    }
    catch (const TrinoException& e) {
        thrown = e.status == PA_ERR_DIVISION_BY_ZERO;
    }
    EXPECT(thrown, "expected DIVISION_BY_ZERO");
}

struct GroupState {
    int64_t count;
    int64_t sum;
    double avg;
};

static std::map<int64_t, GroupState> collect(const std::vector<Page>& pages)
{
    std::map<int64_t, GroupState> groups;
    for (const auto& p : pages) {
        for (int32_t i = 0; i < p.getPositionCount(); i++) {
            int64_t key = p.getBlock(0).getLong(i);
            EXPECT(!groups.count(key), "duplicate group %ld", (long)key);
            groups[key] = GroupState{p.getBlock(1).getLong(i), p.getBlock(2).getLong(i), p.getBlock(3).getDouble(i)};
        }
    }
    return groups;
}

// FilterAndProject (key = field1 % 7, value = field1) -> HashAggregation(count(*), sum(value), avg(value)) by key
static void testPipeline()
{
    std::vector<int32_t> types = {PA_VARCHAR, PA_BIGINT};
    Expr filter = call(PA_OP_GREATER_THAN_OR_EQUAL, PA_BOOLEAN, {field(1, PA_BIGINT), constantLong(3)});
    std::vector<Expr> projections = {call(PA_OP_MODULUS, PA_BIGINT, {field(1, PA_BIGINT), constantLong(7)}), field(1, PA_BIGINT)};
    auto fp = createFilterAndProjectOperator(types, filter, projections);
    std::vector<int32_t> aggTypes = {PA_BIGINT, PA_BIGINT};
    std::vector<pa_aggregate> aggs = {{PA_AGG_COUNT_STAR, -1, -1, PA_BIGINT}, {PA_AGG_SUM, 1, -1, PA_BIGINT}, {PA_AGG_AVG, 1, -1, PA_BIGINT}};
    auto agg = createHashAggregationOperator(aggTypes, {0}, aggs);
    std::vector<Page> input;
    for (int i = 0; i < 8; i++) input.push_back(sequencePage(1000, 1000 * i));
    auto got = collect(runDriver(input, {fp.get(), agg.get()}));

    // oracle: the same filter/projection and GroupByHash + accumulators page by page
    SerializedExpression f(filter);
    std::vector<std::unique_ptr<SerializedExpression>> ps;
    std::vector<pa_expr> pe;
    for (auto& p : projections) {
        ps.push_back(std::make_unique<SerializedExpression>(p));
        pe.push_back(*ps.back()->get());
    }
    std::vector<int32_t> gb = {0};
    pa_hash_aggregation_desc d{};
    d.input_channel_count = 2;
    d.input_types = aggTypes.data();
    d.group_by_count = 1;
    d.group_by_channels = gb.data();
    d.hash_channel = -1;
    d.aggregate_count = (int32_t)aggs.size();
    d.aggregates = aggs.data();
    d.expected_groups = 10000;
    orc_hash_agg* ref = orc_hash_agg_create(&d);
    for (const auto& page : input) {
        pa_page in;
        std::vector<pa_column> cols;
        page.toNative(&in, cols);
        pa_page projected{};
        EXPECT(orc_filter_project(&in, f.get(), (int32_t)pe.size(), pe.data(), &projected) >= 0, "oracle: %s", orc_last_error());
        std::vector<int32_t> ids((size_t)projected.position_count + 1);
        EXPECT(orc_hash_agg_add_page(ref, &projected, ids.data()) >= 0, "oracle: %s", orc_last_error());
        orc_free_page(&projected);
    }
    pa_page result{};
    EXPECT(orc_hash_agg_build_result(ref, &result) >= 0, "oracle: %s", orc_last_error());
    auto expected = collect({Page::fromNative(result)});
    orc_free_page(&result);
    orc_hash_agg_destroy(ref);

    EXPECT(got.size() == 7 && expected.size() == got.size(), "groups: got %zu expected %zu", got.size(), expected.size());
    for (const auto& [key, e] : expected) {
        auto it = got.find(key);
        EXPECT(it != got.end(), "missing group %ld", (long)key);
        if (it == got.end()) continue;
        EXPECT(it->second.count == e.count, "group %ld count %ld != %ld", (long)key, (long)it->second.count, (long)e.count);
        EXPECT(it->second.sum == e.sum, "group %ld sum %ld != %ld", (long)key, (long)it->second.sum, (long)e.sum);
        EXPECT(std::fabs(it->second.avg - e.avg) <= 1e-9 * std::fabs(e.avg), "group %ld avg %.17g != %.17g", (long)key, it->second.avg, e.avg);
    }
}

// hagg-1: TestHashAggregationOperator.testHashAggregation (:160-219): 3 sequence pages of 40 000 rows (VARCHAR @100, VARCHAR @0 =
// the group key, VARCHAR @100000/200000/300000, BIGINT @0, BOOLEAN @500), the test's six aggregates count(*), sum(bigint),
// avg(bigint), max(varchar), count(varchar), count(boolean) -> per key str(i): 3, 3 i, (double) i, str(300000 + i), 3, 3; the
// result comes out in more than one page only on the reference (1 MB pages); compared ignoring order, as the reference does.
static void testHashAggregationKat()
{
    const int32_t n = 40000;
    std::vector<Page> input;
    for (int p = 0; p < 3; p++) {
        std::vector<std::string> a, key, c;
        std::vector<int64_t> v;
        std::vector<uint8_t> b;
        for (int32_t i = 0; i < n; i++) {
            a.push_back(std::to_string(100 + i));
            key.push_back(std::to_string(i));
            c.push_back(std::to_string(100000 * (p + 1) + i));
            v.push_back(i);
            b.push_back((uint8_t)((500 + i) % 2 == 0));  // SequencePageBuilder: BOOLEAN = (start + i) % 2 == 0
        }
        input.push_back(Page({Block::varchar(a), Block::varchar(key), Block::varchar(c), Block::bigint(v), Block::flat<uint8_t>(PA_BOOLEAN, b)}));
    }
    std::vector<int32_t> types = {PA_VARCHAR, PA_VARCHAR, PA_VARCHAR, PA_BIGINT, PA_BOOLEAN};
    std::vector<pa_aggregate> aggs = {{PA_AGG_COUNT_STAR, -1, -1, PA_BIGINT}, {PA_AGG_SUM, 3, -1, PA_BIGINT}, {PA_AGG_AVG, 3, -1, PA_BIGINT},
                                      {PA_AGG_MAX, 2, -1, PA_VARCHAR}, {PA_AGG_COUNT, 0, -1, PA_VARCHAR}, {PA_AGG_COUNT, 4, -1, PA_BOOLEAN}};
    auto agg = createHashAggregationOperator(types, {1}, aggs);
    auto out = runDriver(input, {agg.get()});
    std::map<std::string, int> seen;
    int64_t rows = 0;
    for (const auto& p : out) {
        for (int32_t i = 0; i < p.getPositionCount(); i++, rows++) {
            const std::string k = p.getBlock(0).getSlice(i);
            const int64_t v = atoll(k.c_str());
            EXPECT(seen[k]++ == 0, "hagg-1 duplicate key %s", k.c_str());
            EXPECT(p.getBlock(1).getLong(i) == 3 && p.getBlock(2).getLong(i) == 3 * v && p.getBlock(3).getDouble(i) == (double)v &&
                       p.getBlock(4).getSlice(i) == std::to_string(300000 + v) && p.getBlock(5).getLong(i) == 3 && p.getBlock(6).getLong(i) == 3,
                   "hagg-1 key %s: %ld %ld %g %s %ld %ld", k.c_str(), (long)p.getBlock(1).getLong(i), (long)p.getBlock(2).getLong(i),
                   p.getBlock(3).getDouble(i), p.getBlock(4).getSlice(i).c_str(), (long)p.getBlock(5).getLong(i), (long)p.getBlock(6).getLong(i));
        }
    }
    EXPECT(rows == n && (int64_t)seen.size() == n, "hagg-1 expected %d groups, got %ld", n, (long)rows);
}

// testHashAggregationWithGlobals (TestHashAggregationOperator.java:221-272): no input page, globalAggregationGroupIds (42, 49),
// groupIdChannel 1 among the keys (VARCHAR, BIGINT), produceDefaultOutput -> rows (NULL, 42, 0, NULL, NULL, NULL, 0, 0) and
// (NULL, 49, ...).  Channel layout as in tests/test_oracle_operators.py (the reference's factory is given the key TYPES apart from
// the channels; here they come from the channels).
static void testHashAggregationWithGlobals()
{
    std::vector<int32_t> types = {PA_VARCHAR, PA_VARCHAR, PA_BIGINT, PA_BIGINT, PA_BIGINT, PA_BOOLEAN, PA_VARCHAR};
    std::vector<pa_aggregate> aggs = {{PA_AGG_COUNT_STAR, -1, -1, PA_BIGINT}, {PA_AGG_MIN, 4, -1, PA_BIGINT}, {PA_AGG_AVG, 4, -1, PA_BIGINT},
                                      {PA_AGG_MAX, 6, -1, PA_VARCHAR}, {PA_AGG_COUNT, 0, -1, PA_VARCHAR}, {PA_AGG_COUNT, 5, -1, PA_BOOLEAN}};
    auto agg = createHashAggregationOperator(types, {1, 2}, aggs, PA_STEP_SINGLE, {42, 49}, 1, true);
    auto out = runDriver({}, {agg.get()});
    EXPECT(out.size() == 1 && out[0].getPositionCount() == 2 && out[0].getChannelCount() == 8, "globals: %zu pages", out.size());
    if (out.size() != 1 || out[0].getPositionCount() != 2) return;
    const Page& p = out[0];
    for (int32_t i = 0; i < 2; i++) {
        EXPECT(p.getBlock(0).isNull(i) && !p.getBlock(1).isNull(i) && p.getBlock(1).getLong(i) == (i == 0 ? 42 : 49), "globals row %d keys", i);
        EXPECT(p.getBlock(2).getLong(i) == 0 && p.getBlock(3).isNull(i) && p.getBlock(4).isNull(i) && p.getBlock(5).isNull(i) &&
                   p.getBlock(6).getLong(i) == 0 && p.getBlock(7).getLong(i) == 0, "globals row %d aggregates", i);
    }
    // the oracle's twin agrees
    pa_hash_aggregation_desc d{};
    std::vector<int32_t> gb = {1, 2}, ids = {42, 49};
    d.input_channel_count = (int32_t)types.size();
    d.input_types = types.data();
    d.group_by_count = 2;
    d.group_by_channels = gb.data();
    d.hash_channel = -1;
    d.aggregate_count = (int32_t)aggs.size();
    d.aggregates = aggs.data();
    d.produce_default_output = 1;
    d.group_id_channel = 1;
    d.global_aggregation_group_id_count = 2;
    d.global_aggregation_group_ids = ids.data();
    pa_page ref{};
    EXPECT(orc_hash_agg_default_output(&d, &ref) >= 0 && ref.position_count == 2 && ref.channel_count == 8, "oracle default output");
    orc_free_page(&ref);
}

// join-1: TestHashJoinOperator.testInnerJoin (core/trino-main/src/test/java/io/trino/operator/join/TestHashJoinOperator.java:192-229):
// build (VARCHAR, BIGINT, BIGINT) sequence 10 rows @20,30,40; probe sequence 1000 rows @0,1000,2000; key channel 0.  Then the
// same build probed with a probe-outer join, and a TopN over the join output (TestTopNOperator-style ordering).
static Page sequencePage3(int32_t length, int64_t a, int64_t b, int64_t c)
{
    std::vector<std::string> s;
    std::vector<int64_t> x, y;
    for (int32_t i = 0; i < length; i++) {
        s.push_back(std::to_string(a + i));
        x.push_back(b + i);
        y.push_back(c + i);
    }
    return Page({Block::varchar(s), Block::bigint(x), Block::bigint(y)});
}

static void testJoinAndTopN()
{
    std::vector<int32_t> types = {PA_VARCHAR, PA_BIGINT, PA_BIGINT};
    LookupSourceFactory bridge;
    auto build = createHashBuilderOperator(bridge, types, {0}, {0, 1, 2});
    runDriver({sequencePage3(10, 20, 30, 40)}, {build.get()});
    auto join = createLookupJoinOperator(bridge, types, {0}, {0, 1, 2});
    std::vector<int32_t> joined = {PA_VARCHAR, PA_BIGINT, PA_BIGINT, PA_VARCHAR, PA_BIGINT, PA_BIGINT};
    auto topn = createTopNOperator(joined, 3, {1}, {PA_DESC_NULLS_LAST});
    auto out = runDriver({sequencePage3(1000, 0, 1000, 2000)}, {join.get(), topn.get()});
    EXPECT(out.size() == 1 && out[0].getPositionCount() == 3, "join-1 + TopN: expected one page of 3 rows");
    for (int32_t i = 0; !out.empty() && i < out[0].getPositionCount(); i++) {
        const int64_t k = 29 - i;  // the 10 matches are probe rows "20".."29"; top 3 by probe column 1 descending
        EXPECT(out[0].getBlock(0).getSlice(i) == std::to_string(k) && out[0].getBlock(3).getSlice(i) == std::to_string(k), "join-1 keys row %d", i);
        EXPECT(out[0].getBlock(1).getLong(i) == 1000 + k && out[0].getBlock(2).getLong(i) == 2000 + k, "join-1 probe columns row %d", i);
        EXPECT(out[0].getBlock(4).getLong(i) == 30 + (k - 20) && out[0].getBlock(5).getLong(i) == 40 + (k - 20), "join-1 build columns row %d", i);
    }
    // the join's dynamic filter in the FilterAndProject upstream of the probe: the same 10 rows reach the join (a BIGINT key:
    // the VARCHAR key of join-1 offers no bitmap), and the join result is unchanged
    {
        LookupSourceFactory byNumber;
        auto b = createHashBuilderOperator(byNumber, types, {1}, {0, 1, 2});
        runDriver({sequencePage3(10, 20, 30, 40)}, {b.get()});
        auto fp = createFilterAndProjectOperator(types, nullptr, {field(0, PA_VARCHAR), field(1, PA_BIGINT), field(2, PA_BIGINT)});
        EXPECT(fp->setDynamicFilter(1, byNumber.handle()), "dynamic filter: a BIGINT key offers a bitmap");
        auto j = createLookupJoinOperator(byNumber, types, {1}, {0, 1, 2});
        auto filtered = runDriver({sequencePage3(1000, 0, 0, 2000)}, {fp.get()});
        EXPECT(filtered.size() == 1 && filtered[0].getPositionCount() == 10, "dynamic filter: 10 of 1000 probe rows have a build partner");
        auto joined2 = runDriver(filtered, {j.get()});
        EXPECT(joined2.size() == 1 && joined2[0].getPositionCount() == 10 && joined2[0].getBlock(1).getLong(0) == 30 && joined2[0].getBlock(4).getLong(9) == 39,
               "dynamic filter: join over the filtered rows");
        auto fpv = createFilterAndProjectOperator(types, nullptr, {field(0, PA_VARCHAR)});
        EXPECT(!fpv->setDynamicFilter(0, bridge.handle()), "dynamic filter: none for a VARCHAR key");
    }
    // testProbeOuterJoin (:850-894): 15 probe rows @20: 10 matches, then 5 rows with NULL build columns
    auto outer = createLookupJoinOperator(bridge, types, {0}, {0, 1, 2}, PA_JOIN_PROBE_OUTER);
    auto rows = runDriver({sequencePage3(15, 20, 1020, 2020)}, {outer.get()});
    EXPECT(rows.size() == 1 && rows[0].getPositionCount() == 15, "probe outer: expected 15 rows");
    for (int32_t i = 0; !rows.empty() && i < rows[0].getPositionCount(); i++) {
        EXPECT(rows[0].getBlock(1).getLong(i) == 1020 + i, "probe outer probe column row %d", i);
        EXPECT(rows[0].getBlock(4).isNull(i) == (i >= 10), "probe outer NULL build side row %d", i);
        if (i < 10) EXPECT(rows[0].getBlock(4).getLong(i) == 30 + i, "probe outer build column row %d", i);
    }
}

// TestDynamicFilterSourceOperator.testCollectMultipleOperators (:181-207) first operator, then
// testSingleColumnCollectMinMaxRangeWhenTooManyPositions (:391-406), through the C++ mirror: the pages pass through, the
// predicate is polled after finish
static void testDynamicFilterSource()
{
    auto op = createDynamicFilterSourceOperator({PA_BIGINT}, {0}, 100, 10 * 1024, 1000000);
    auto out = runDriver({Page({Block::bigint({1, 2})}), Page({Block::bigint({3, 5})})}, {op.get()});
    EXPECT(out.size() == 2 && out[0].getPositionCount() == 2 && out[1].getBlock(0).getLong(1) == 5, "dynamic filter: pages must pass through");
    int32_t isAll = 0;
    pa_domain domain{};
    EXPECT(pa_dynamic_filter_poll(op->handle(), &isAll, &domain, 1) == 1 && !isAll, "dynamic filter: predicate ready after finish");
    EXPECT(domain.kind == PA_DOMAIN_VALUES && domain.value_count == 4, "dynamic filter: 4 distinct values");
    const int64_t* v = static_cast<const int64_t*>(domain.values.values);
    EXPECT(domain.value_count == 4 && v[0] == 1 && v[1] == 2 && v[2] == 3 && v[3] == 5, "dynamic filter: values 1, 2, 3, 5");

    std::vector<int64_t> seq;
    for (int64_t i = 0; i <= 100; i++) seq.push_back(i);
    auto big = createDynamicFilterSourceOperator({PA_BIGINT}, {0}, 100, 10 * 1024, 1000000);
    runDriver({Page({Block::bigint(seq)})}, {big.get()});
    EXPECT(pa_dynamic_filter_poll(big->handle(), &isAll, &domain, 1) == 1 && !isAll && domain.kind == PA_DOMAIN_RANGE, "dynamic filter: range after 101 values");
    v = static_cast<const int64_t*>(domain.values.values);
    EXPECT(domain.value_count == 2 && v[0] == 0 && v[1] == 100, "dynamic filter: range [0, 100]");
}

int main()
{
    try {
        check(pa_init(0));
        testFilterAndProject();
        testDivisionByZero();
        testPipeline();
        testHashAggregationKat();
        testHashAggregationWithGlobals();
        testJoinAndTopN();
        testDynamicFilterSource();
        pa_shutdown();
    }
    catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    if (failures) {
        fprintf(stderr, "%d failure(s)\n", failures);
        return 1;
    }
    printf("cpp driver: all cases pass\n");
    return 0;
}
