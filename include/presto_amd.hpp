// presto_amd.hpp -- header-only C++ host side above the C ABI (include/presto_amd.h), mirroring the reference's
// interfaces with the same names and call protocol:
//   io.trino.spi.Page / Block                  core/trino-spi/src/main/java/io/trino/spi/Page.java:33-398
//   io.trino.sql.relational.RowExpression      core/trino-main/src/main/java/io/trino/sql/relational/{Call,Constant,
//                                              InputReference}Expression.java, SpecialForm.java
//   io.trino.operator.Operator                 core/trino-main/src/main/java/io/trino/operator/Operator.java:21-103
//   io.trino.operator.Driver (processInternal) core/trino-main/src/main/java/io/trino/operator/Driver.java:355-457
// The reference is Java; no JDK exists in the build image, so this is the compiled-language host mirror (the Java
// binding itself is in INTEGRATION.md).  Everything here is plumbing: no arithmetic, no fallback.
#pragma once

#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "presto_amd.h"

namespace presto_amd {

// TrinoException(StandardErrorCode, message)
struct TrinoException : std::runtime_error {
    int32_t status;
    TrinoException(int32_t s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline int32_t check(int32_t rc)
{
    if (rc < 0) throw TrinoException(rc, pa_last_error());
    return rc;
}

// ---- Block / Page (host memory) --------------------------------------------------------------------------------
class Block {
public:
    int32_t type = PA_BIGINT;
    std::vector<uint8_t> values;     // element array or VARCHAR bytes
    std::vector<int32_t> offsets;    // VARCHAR only
    std::vector<uint8_t> nulls;      // empty = no nulls
    int32_t positionCount = 0;

    template <typename T>
    static Block flat(int32_t type, const std::vector<T>& v, const std::vector<uint8_t>& valueIsNull = {})
    {
        Block b;
        b.type = type;
        b.positionCount = (int32_t)v.size();
        b.values.resize(v.size() * sizeof(T));
        if (!v.empty()) memcpy(b.values.data(), v.data(), b.values.size());
        b.nulls = valueIsNull;
        return b;
    }
    static Block bigint(const std::vector<int64_t>& v) { return flat<int64_t>(PA_BIGINT, v); }
    static Block doubles(const std::vector<double>& v) { return flat<double>(PA_DOUBLE, v); }
    static Block varchar(const std::vector<std::string>& v)
    {
        Block b;
        b.type = PA_VARCHAR;
        b.positionCount = (int32_t)v.size();
        b.offsets.push_back(0);
        for (const auto& s : v) {
            b.values.insert(b.values.end(), s.begin(), s.end());
            b.offsets.push_back((int32_t)b.values.size());
        }
        if (b.values.empty()) b.values.push_back(0);
        return b;
    }
    int64_t getLong(int32_t position) const
    {
        if (type == PA_INTEGER || type == PA_DATE) { int32_t v; memcpy(&v, values.data() + 4 * (size_t)position, 4); return v; }
        int64_t v;
        memcpy(&v, values.data() + 8 * (size_t)position, 8);
        return v;
    }
    double getDouble(int32_t position) const { double v; memcpy(&v, values.data() + 8 * (size_t)position, 8); return v; }
    std::string getSlice(int32_t position) const
    {
        return std::string(values.begin() + offsets[position], values.begin() + offsets[position + 1]);
    }
    bool isNull(int32_t position) const { return !nulls.empty() && nulls[position] != 0; }
};

class Page {
public:
    std::vector<Block> blocks;
    int32_t positionCount = 0;
    Page() = default;
    explicit Page(std::vector<Block> b) : blocks(std::move(b)) { positionCount = blocks.empty() ? 0 : blocks[0].positionCount; }
    int32_t getPositionCount() const { return positionCount; }
    int32_t getChannelCount() const { return (int32_t)blocks.size(); }
    const Block& getBlock(int32_t channel) const { return blocks[channel]; }

    // fills a pa_page view of this (host) page; `cols` provides the storage
    void toNative(pa_page* out, std::vector<pa_column>& cols) const
    {
        cols.assign(blocks.size() ? blocks.size() : 1, pa_column{});
        for (size_t i = 0; i < blocks.size(); i++) {
            const Block& b = blocks[i];
            cols[i].type = b.type;
            cols[i].encoding = b.type == PA_VARCHAR ? PA_VARWIDTH : PA_FLAT;
            cols[i].values = b.values.data();
            cols[i].offsets = b.type == PA_VARCHAR ? b.offsets.data() : nullptr;
            cols[i].nulls = b.nulls.empty() ? nullptr : b.nulls.data();
        }
        out->position_count = positionCount;
        out->channel_count = (int32_t)blocks.size();
        out->columns = cols.data();
        out->mem = PA_MEM_HOST;
        out->flags = 0;
    }
    // copies a PA_MEM_HOST pa_page returned by pa_op_get_output
    static Page fromNative(const pa_page& p)
    {
        Page page;
        page.positionCount = p.position_count;
        for (int32_t c = 0; c < p.channel_count; c++) {
            const pa_column& col = p.columns[c];
            Block b;
            b.type = col.type;
            b.positionCount = p.position_count;
            size_t n = (size_t)p.position_count;
            if (col.encoding == PA_VARWIDTH) {
                b.offsets.assign(col.offsets, col.offsets + n + 1);
                const uint8_t* v = static_cast<const uint8_t*>(col.values);
                b.values.assign(v, v + (n ? (size_t)col.offsets[n] : 0));
            }
            else {
                size_t w = (col.type == PA_BIGINT || col.type == PA_DOUBLE) ? 8 : (col.type == PA_BOOLEAN ? 1 : 4);
                const uint8_t* v = static_cast<const uint8_t*>(col.values);
                b.values.assign(v, v + n * w);
            }
            if (col.nulls) b.nulls.assign(col.nulls, col.nulls + n);
            page.blocks.push_back(std::move(b));
        }
        return page;
    }
};

// ---- RowExpression (Expressions.field / constant / call, SpecialForm) ------------------------------------------
class RowExpression {
public:
    pa_expr_node node{};
    std::vector<std::shared_ptr<RowExpression>> arguments;
    std::string text;  // VARCHAR constant bytes
};
using Expr = std::shared_ptr<RowExpression>;

inline Expr field(int32_t channel, int32_t type)
{
    auto e = std::make_shared<RowExpression>();
    e->node.kind = PA_EXPR_INPUT_REF;
    e->node.type = type;
    e->node.channel = channel;
    return e;
}
inline Expr constantLong(int64_t v, int32_t type = PA_BIGINT)
{
    auto e = std::make_shared<RowExpression>();
    e->node.kind = PA_EXPR_CONSTANT;
    e->node.type = type;
    e->node.i64 = v;
    return e;
}
inline Expr constantDouble(double v)
{
    auto e = std::make_shared<RowExpression>();
    e->node.kind = PA_EXPR_CONSTANT;
    e->node.type = PA_DOUBLE;
    e->node.f64 = v;
    return e;
}
inline Expr call(int32_t op, int32_t type, std::vector<Expr> args)
{
    auto e = std::make_shared<RowExpression>();
    e->node.kind = PA_EXPR_CALL;
    e->node.op = op;
    e->node.type = type;
    e->arguments = std::move(args);
    return e;
}
inline Expr specialForm(int32_t form, int32_t type, std::vector<Expr> args)
{
    auto e = std::make_shared<RowExpression>();
    e->node.kind = PA_EXPR_SPECIAL;
    e->node.op = form;
    e->node.type = type;
    e->arguments = std::move(args);
    return e;
}

// children-first flattening into pa_expr (what the Java RowExpressionVisitor of INTEGRATION.md emits)
class SerializedExpression {
public:
    explicit SerializedExpression(const Expr& root)
    {
        int32_t r = walk(root);
        for (size_t i = 0; i < nodes_.size(); i++) {
            if (nodes_[i].kind == PA_EXPR_CONSTANT && nodes_[i].type == PA_VARCHAR) nodes_[i].str = strings_[i].data();
        }
        expr_.node_count = (int32_t)nodes_.size();
        expr_.root = r;
        expr_.nodes = nodes_.data();
        expr_.arg_count = (int32_t)args_.size();
        expr_.args = args_.data();
    }
    const pa_expr* get() const { return &expr_; }

private:
    int32_t walk(const Expr& e)
    {
        std::vector<int32_t> children;
        for (const auto& a : e->arguments) children.push_back(walk(a));
        pa_expr_node n = e->node;
        n.nargs = (int32_t)children.size();
        n.first_arg = (int32_t)args_.size();
        n.str_len = (int32_t)e->text.size();
        args_.insert(args_.end(), children.begin(), children.end());
        nodes_.push_back(n);
        strings_.push_back(e->text);
        return (int32_t)nodes_.size() - 1;
    }
    std::vector<pa_expr_node> nodes_;
    std::vector<int32_t> args_;
    std::vector<std::string> strings_;
    pa_expr expr_{};
};

// ---- Operator ---------------------------------------------------------------------------------------------------
class Operator {
public:
    explicit Operator(pa_operator* h) : h_(h) {}
    Operator(const Operator&) = delete;
    Operator& operator=(const Operator&) = delete;
    ~Operator() { close(); }
    bool needsInput() { return check(pa_op_needs_input(h_)) == 1; }
    void addInput(const Page& page)
    {
        pa_page p;
        std::vector<pa_column> cols;
        page.toNative(&p, cols);
        check(pa_op_add_input(h_, &p));
    }
    std::optional<Page> getOutput()
    {
        pa_page out{};
        if (check(pa_op_get_output(h_, &out)) == 0) return std::nullopt;
        return Page::fromNative(out);
    }
    void finish() { check(pa_op_finish(h_)); }
    bool isFinished() { return check(pa_op_is_finished(h_)) == 1; }
    bool isBlocked() { return check(pa_op_is_blocked(h_)) == 1; }
    void close()
    {
        if (h_) pa_op_close(h_);
        h_ = nullptr;
    }
    pa_operator* handle() { return h_; }
    // FilterAndProject only, before the first page: the dynamic filter of the (built) join this operator feeds -- rows whose
    // `channel` value matches no build key are dropped with the filter; true when the source offers one
    bool setDynamicFilter(int32_t channel, pa_lookup_source* source) { return check(pa_filter_project_set_dynamic_filter(h_, channel, source)) == 1; }
    // aggregation operators whose only consumer is a TopN over their output (pa_aggregation_set_output_topn_hint)
    bool setOutputTopNHint(int64_t n, const std::vector<int32_t>& sortChannels, const std::vector<int32_t>& sortOrders)
    {
        return check(pa_aggregation_set_output_topn_hint(h_, n, (int32_t)sortChannels.size(), sortChannels.data(), sortOrders.data())) == 1;
    }

private:
    pa_operator* h_;
};

// OperatorFactory.createOperator equivalents
inline std::unique_ptr<Operator> createFilterAndProjectOperator(const std::vector<int32_t>& inputTypes, const Expr& filter,
                                                                const std::vector<Expr>& projections)
{
    std::unique_ptr<SerializedExpression> f;
    if (filter) f = std::make_unique<SerializedExpression>(filter);
    std::vector<std::unique_ptr<SerializedExpression>> ps;
    std::vector<pa_expr> pexprs;
    for (const auto& p : projections) {
        ps.push_back(std::make_unique<SerializedExpression>(p));
        pexprs.push_back(*ps.back()->get());
    }
    pa_filter_project_desc d{};
    d.input_channel_count = (int32_t)inputTypes.size();
    d.input_types = inputTypes.data();
    d.filter = f ? f->get() : nullptr;
    d.projection_count = (int32_t)pexprs.size();
    d.projections = pexprs.data();
    d.output_mem = PA_MEM_HOST;
    pa_operator* h = nullptr;
    check(pa_filter_project_create(&d, &h));
    return std::make_unique<Operator>(h);
}

// HashAggregationOperatorFactory (HashAggregationOperator.java:120-202); globalAggregationGroupIds / groupIdChannel (an index among the
// group-by columns) / produceDefaultOutput: the default rows of the global grouping sets when no page arrives (:486-492, 545-587)
inline std::unique_ptr<Operator> createHashAggregationOperator(const std::vector<int32_t>& inputTypes, const std::vector<int32_t>& groupByChannels,
                                                               const std::vector<pa_aggregate>& aggregates, int32_t step = PA_STEP_SINGLE,
                                                               const std::vector<int32_t>& globalAggregationGroupIds = {}, int32_t groupIdChannel = -1,
                                                               bool produceDefaultOutput = false)
{
    pa_hash_aggregation_desc d{};
    d.input_channel_count = (int32_t)inputTypes.size();
    d.input_types = inputTypes.data();
    d.group_by_count = (int32_t)groupByChannels.size();
    d.group_by_channels = groupByChannels.data();
    d.hash_channel = -1;
    d.step = step;
    d.aggregate_count = (int32_t)aggregates.size();
    d.aggregates = aggregates.data();
    d.expected_groups = 10000;
    d.output_mem = PA_MEM_HOST;
    d.produce_default_output = produceDefaultOutput ? 1 : 0;
    d.group_id_channel = groupIdChannel;
    d.global_aggregation_group_id_count = (int32_t)globalAggregationGroupIds.size();
    d.global_aggregation_group_ids = globalAggregationGroupIds.data();
    pa_operator* h = nullptr;
    check(pa_hash_aggregation_create(&d, &h));
    return std::make_unique<Operator>(h);
}

inline std::unique_ptr<Operator> createTopNOperator(const std::vector<int32_t>& inputTypes, int32_t n, const std::vector<int32_t>& sortChannels,
                                                    const std::vector<int32_t>& sortOrders)
{
    pa_topn_desc d{};
    d.input_channel_count = (int32_t)inputTypes.size();
    d.input_types = inputTypes.data();
    d.n = n;
    d.sort_channel_count = (int32_t)sortChannels.size();
    d.sort_channels = sortChannels.data();
    d.sort_orders = sortOrders.data();
    d.output_mem = PA_MEM_HOST;
    pa_operator* h = nullptr;
    check(pa_topn_create(&d, &h));
    return std::make_unique<Operator>(h);
}

// DynamicFilterSourceOperatorFactory: the returned operator passes pages through; poll the collected predicate with
// pa_dynamic_filter_poll(op->handle(), ...) after finish()
inline std::unique_ptr<Operator> createDynamicFilterSourceOperator(const std::vector<int32_t>& inputTypes, const std::vector<int32_t>& filterChannels,
                                                                   int32_t maxDistinctValues, int64_t maxFilterSizeBytes, int32_t minMaxCollectionLimit)
{
    pa_dynamic_filter_source_desc d{};
    d.input_channel_count = (int32_t)inputTypes.size();
    d.input_types = inputTypes.data();
    d.filter_channel_count = (int32_t)filterChannels.size();
    d.filter_channels = filterChannels.data();
    d.max_distinct_values = maxDistinctValues;
    d.max_filter_size_bytes = maxFilterSizeBytes;
    d.min_max_collection_limit = minMaxCollectionLimit;
    pa_operator* h = nullptr;
    check(pa_dynamic_filter_source_create(&d, &h));
    return std::make_unique<Operator>(h);
}

// JoinBridge shared by the build operator and its probe operators (JoinBridgeManager / PartitionedLookupSourceFactory)
class LookupSourceFactory {
public:
    LookupSourceFactory() { check(pa_lookup_source_create(&h_)); }
    LookupSourceFactory(const LookupSourceFactory&) = delete;
    LookupSourceFactory& operator=(const LookupSourceFactory&) = delete;
    ~LookupSourceFactory()
    {
        if (h_) pa_lookup_source_destroy(h_);
    }
    pa_lookup_source* handle() { return h_; }

private:
    pa_lookup_source* h_ = nullptr;
};

inline std::unique_ptr<Operator> createHashBuilderOperator(LookupSourceFactory& bridge, const std::vector<int32_t>& inputTypes,
                                                           const std::vector<int32_t>& joinChannels, const std::vector<int32_t>& outputChannels)
{
    pa_hash_builder_desc d{};
    d.input_channel_count = (int32_t)inputTypes.size();
    d.input_types = inputTypes.data();
    d.join_channel_count = (int32_t)joinChannels.size();
    d.join_channels = joinChannels.data();
    d.hash_channel = -1;
    d.output_channel_count = (int32_t)outputChannels.size();
    d.output_channels = outputChannels.data();
    pa_operator* h = nullptr;
    check(pa_hash_builder_create(&d, bridge.handle(), &h));
    return std::make_unique<Operator>(h);
}

// OperatorFactories.innerJoin / probeOuterJoin / lookupOuterJoin / fullOuterJoin: joinType = pa_join_type; filter = the join's
// JoinFilterFunction over [build page channels, probe page channels], or null
inline std::unique_ptr<Operator> createLookupJoinOperator(LookupSourceFactory& bridge, const std::vector<int32_t>& probeTypes,
                                                          const std::vector<int32_t>& probeJoinChannels, const std::vector<int32_t>& probeOutputChannels,
                                                          int32_t joinType = PA_JOIN_INNER, bool outputSingleMatch = false, const Expr& filter = nullptr)
{
    std::unique_ptr<SerializedExpression> f;
    if (filter) f = std::make_unique<SerializedExpression>(filter);
    pa_lookup_join_desc d{};
    d.filter = f ? f->get() : nullptr;
    d.probe_channel_count = (int32_t)probeTypes.size();
    d.probe_types = probeTypes.data();
    d.join_channel_count = (int32_t)probeJoinChannels.size();
    d.probe_join_channels = probeJoinChannels.data();
    d.probe_hash_channel = -1;
    d.probe_output_channel_count = (int32_t)probeOutputChannels.size();
    d.probe_output_channels = probeOutputChannels.data();
    d.output_mem = PA_MEM_HOST;
    d.join_type = joinType;
    d.output_single_match = outputSingleMatch ? 1 : 0;
    pa_operator* h = nullptr;
    check(pa_lookup_join_create(&d, bridge.handle(), &h));
    return std::make_unique<Operator>(h);
}

// ---- Driver (Driver.processInternal): page source -> operators[0] -> ... -> collected output -----------------------
inline std::vector<Page> runDriver(const std::vector<Page>& source, const std::vector<Operator*>& operators)
{
    std::vector<Page> output;
    size_t next = 0;
    bool sourceFinished = false;
    for (int guard = 0; guard < (1 << 22); guard++) {
        if (next < source.size() && operators[0]->needsInput()) {
            if (source[next].getPositionCount() > 0) operators[0]->addInput(source[next]);
            next++;
        }
        if (next == source.size() && !sourceFinished) {
            sourceFinished = true;
            operators[0]->finish();
        }
        for (size_t i = 0; i + 1 < operators.size(); i++) {
            Operator* current = operators[i];
            Operator* nxt = operators[i + 1];
            if (!current->isFinished() && nxt->needsInput()) {
                auto page = current->getOutput();
                if (page && page->getPositionCount() > 0) nxt->addInput(*page);
            }
            if (current->isFinished()) nxt->finish();
        }
        Operator* last = operators.back();
        if (auto page = last->getOutput()) {
            if (page->getPositionCount() > 0) output.push_back(std::move(*page));
        }
        if (last->isFinished()) return output;
    }
    throw std::runtime_error("pipeline did not finish");
}

}  // namespace presto_amd
