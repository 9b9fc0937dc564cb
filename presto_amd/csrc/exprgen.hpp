// exprgen.hpp -- RowExpression -> HIP source, the counterpart of the reference's bytecode generators
// (core/trino-main/src/main/java/io/trino/sql/gen/PageFunctionCompiler.java:219-365 projections,
// :459-544 filters; AndCodeGenerator.java:44-105, OrCodeGenerator.java, BetweenCodeGenerator.java:58-82,
// IfCodeGenerator.java, InCodeGenerator.java, CoalesceCodeGenerator.java, IsNullCodeGenerator.java for
// the SQL NULL rules).  Values are tracked as (value, isNull) pairs exactly as the generated JVM code
// tracks `wasNull`.
#pragma once

#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "common.hpp"

namespace pa {

// Deep copy of a pa_expr: descriptors only live for the duration of the factory call.
struct OwnedExpr {
    std::vector<pa_expr_node> nodes;
    std::vector<int32_t> args;
    std::vector<std::string> strings;  // backing store of VARCHAR constants, one per node (may be empty)
    int32_t root = -1;

    static OwnedExpr copy(const pa_expr& e);
    const pa_expr_node& node(int32_t id) const { return nodes[id]; }
    const int32_t* node_args(int32_t id) const { return args.data() + nodes[id].first_arg; }
    int32_t root_type() const { return nodes[root].type; }
    bool is_input_ref() const { return nodes[root].kind == PA_EXPR_INPUT_REF; }
    void collect_channels(std::set<int32_t>* out) const;
    // structural identity (used to share accumulators between sum(x) and avg(x))
    std::string fingerprint() const;
};

struct ChannelLayout {
    int32_t type = PA_BIGINT;
    bool nullable = false;
};

// A generated scalar: `v` / `n` are C++ expressions (variable names or literals) valid in the scope
// where emit() wrote its statements.  VARCHAR values are a (pointer, length) pair: v and len.
struct GenValue {
    std::string v;
    std::string len;   // VARCHAR only
    std::string n;     // "false" when statically non-null
    int32_t type = PA_BIGINT;
    int32_t param = 0;  // DECIMAL types: PA_DECIMAL_PARAM(precision, scale) of the value's type
    bool nullable() const { return n != "false"; }
};

class RowCodegen {
public:
    // Inputs of channel c are expected in scope as c<c> (value; i64 for all integer types, double,
    // bool, const u8* for VARCHAR), cl<c> (VARCHAR length) and cn<c> (bool, only when nullable).
    // `err` is an expression of type i32* receiving device-raised pa_status codes.
    RowCodegen(const std::vector<ChannelLayout>& channels, std::string err);
    // Appends statements computing `e` to `out` and returns the result names.
    GenValue emit(const OwnedExpr& e, std::ostringstream& out);
    static std::string ctype(int32_t type);

private:
    GenValue emit_node(const OwnedExpr& e, int32_t id, std::ostringstream& out);
    GenValue emit_compare(int32_t op, const GenValue& a, const GenValue& b, std::ostringstream& out);
    bool can_throw(const OwnedExpr& e, int32_t id) const;
    std::string fresh(const char* prefix);
    static std::string or_nulls(const std::vector<std::string>& ns);

    std::vector<ChannelLayout> channels_;
    std::string err_;
    int counter_ = 0;
};

std::string double_literal(double v);
std::string bytes_literal(const std::string& bytes);

}  // namespace pa
