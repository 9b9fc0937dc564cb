// exprgen.cpp -- see exprgen.hpp.
#include <cstring>

#include "exprgen.hpp"

#include <cinttypes>
#include <cmath>

namespace pa {

OwnedExpr OwnedExpr::copy(const pa_expr& e)
{
    OwnedExpr o;
    PA_REQUIRE(e.node_count > 0 && e.nodes != nullptr, PA_ERR_INVALID_ARGUMENT, "empty expression");
    PA_REQUIRE(e.root >= 0 && e.root < e.node_count, PA_ERR_INVALID_ARGUMENT, "expression root out of range");
    o.nodes.assign(e.nodes, e.nodes + e.node_count);
    if (e.arg_count > 0) o.args.assign(e.args, e.args + e.arg_count);
    o.strings.resize(e.node_count);
    for (int32_t i = 0; i < e.node_count; i++) {
        pa_expr_node& n = o.nodes[i];
        PA_REQUIRE(n.nargs >= 0 && n.first_arg >= 0 && n.first_arg + n.nargs <= e.arg_count, PA_ERR_INVALID_ARGUMENT,
                   "expression argument list out of range");
        for (int32_t k = 0; k < n.nargs; k++) {
            int32_t child = e.args[n.first_arg + k];
            PA_REQUIRE(child >= 0 && child < i, PA_ERR_INVALID_ARGUMENT, "expression children must precede parents");
        }
        if (n.kind == PA_EXPR_CONSTANT && n.type == PA_VARCHAR && !n.is_null && n.str_len > 0) {
            o.strings[i].assign(n.str, n.str + n.str_len);
        }
        n.str = nullptr;
    }
    o.root = e.root;
    return o;
}

void OwnedExpr::collect_channels(std::set<int32_t>* out) const
{
    // only nodes reachable from the root
    std::vector<int32_t> stack{root};
    while (!stack.empty()) {
        int32_t id = stack.back();
        stack.pop_back();
        const pa_expr_node& n = nodes[id];
        if (n.kind == PA_EXPR_INPUT_REF) out->insert(n.channel);
        for (int32_t k = 0; k < n.nargs; k++) stack.push_back(args[n.first_arg + k]);
    }
}

std::string OwnedExpr::fingerprint() const
{
    std::ostringstream s;
    std::vector<int32_t> stack{root};
    while (!stack.empty()) {
        int32_t id = stack.back();
        stack.pop_back();
        const pa_expr_node& n = nodes[id];
        uint64_t f64_bits;   // (the raw bits: a long-decimal constant keeps its high half there, and NaN payloads must not collapse)
        memcpy(&f64_bits, &n.f64, 8);
        s << n.kind << ':' << n.op << ':' << n.type << ':' << n.channel << ':' << n.is_null << ':' << n.nargs << ':'
          << n.i64 << ':' << f64_bits << ':' << strings[id].size() << ':' << strings[id] << ';';
        if (n.type == PA_DECIMAL || n.type == PA_LONG_DECIMAL) s << 'p' << n.type_param << ';';
        for (int32_t k = 0; k < n.nargs; k++) stack.push_back(args[n.first_arg + k]);
    }
    return s.str();
}

std::string double_literal(double v)
{
    if (v != v) return "__longlong_as_double(0x7ff8000000000000LL)";
    if (std::isinf(v)) return v > 0 ? "__longlong_as_double(0x7ff0000000000000LL)" : "__longlong_as_double(0xfff0000000000000LL)";
    char buf[64];
    snprintf(buf, sizeof buf, v < 0 || (v == 0 && std::signbit(v)) ? "(%a)" : "%a", v);  // exact hexadecimal floating literal (C++17)
    return buf;
}

std::string bytes_literal(const std::string& bytes)
{
    std::string s = "(const u8*)\"";
    char buf[8];
    for (unsigned char c : bytes) {
        snprintf(buf, sizeof buf, "\\%03o", c);
        s += buf;
    }
    s += "\"";
    return s;
}

RowCodegen::RowCodegen(const std::vector<ChannelLayout>& channels, std::string err)
    : channels_(channels), err_(std::move(err))
{
}

std::string RowCodegen::ctype(int32_t type)
{
    switch (type) {
        case PA_DOUBLE:
            return "double";
        case PA_REAL:
            return "float";
        case PA_BOOLEAN:
            return "bool";
        case PA_VARCHAR:
            return "const u8*";
        case PA_LONG_DECIMAL:
            return "i128";
        default:
            return "i64";
    }
}

std::string RowCodegen::fresh(const char* prefix)
{
    return std::string(prefix) + std::to_string(counter_++);
}

std::string RowCodegen::or_nulls(const std::vector<std::string>& ns)
{
    std::string r;
    for (const auto& n : ns) {
        if (n == "false") continue;
        if (n == "true") return "true";
        if (!r.empty()) r += " || ";
        r += n;
    }
    return r.empty() ? "false" : "(" + r + ")";
}

static bool is_int_type(int32_t t) { return t == PA_BIGINT || t == PA_INTEGER || t == PA_DATE; }
static bool is_decimal_type(int32_t t) { return t == PA_DECIMAL || t == PA_LONG_DECIMAL; }

// 10^k as an expression of the generated code: an i64 literal up to 10^18, an i128 from its halves beyond
static std::string pow10_literal(int k, bool wide)
{
    unsigned __int128 v = 1;
    for (int i = 0; i < k; i++) v *= 10;
    char buf[96];
    if (!wide && k <= 18) {
        snprintf(buf, sizeof buf, "%" PRId64 "LL", (int64_t)v);
        return buf;
    }
    snprintf(buf, sizeof buf, "pa_i128_of(0x%" PRIx64 "ULL, 0x%" PRIx64 "ULL)", (uint64_t)(v >> 64), (uint64_t)v);
    return buf;
}

bool RowCodegen::can_throw(const OwnedExpr& e, int32_t id) const
{
    const pa_expr_node& n = e.node(id);
    if (n.kind == PA_EXPR_CALL && is_int_type(n.type) && n.op <= PA_OP_NEGATE) return true;
    // long decimal arithmetic raises NUMERIC_VALUE_OUT_OF_RANGE at 10^38, casts to a decimal check their precision
    if (n.kind == PA_EXPR_CALL && ((n.type == PA_LONG_DECIMAL && n.op <= PA_OP_NEGATE) || (is_decimal_type(n.type) && n.op == PA_OP_CAST))) return true;
    const int32_t* a = e.node_args(id);
    for (int32_t k = 0; k < n.nargs; k++) {
        if (can_throw(e, a[k])) return true;
    }
    return false;
}

GenValue RowCodegen::emit(const OwnedExpr& e, std::ostringstream& out)
{
    return emit_node(e, e.root, out);
}

GenValue RowCodegen::emit_compare(int32_t op, const GenValue& a, const GenValue& b, std::ostringstream& out)
{
    GenValue r;
    r.type = PA_BOOLEAN;
    r.n = or_nulls({a.n, b.n});
    r.v = fresh("t");
    static const char* sym[] = {"==", "!=", "<", "<=", ">", ">="};
    const char* s = sym[op - PA_OP_EQUAL];
    if (a.type == PA_VARCHAR || b.type == PA_VARCHAR) {
        PA_REQUIRE(a.type == PA_VARCHAR && b.type == PA_VARCHAR, PA_ERR_NOT_SUPPORTED, "VARCHAR compared with non-VARCHAR");
        // operands of a NULL side are unspecified pointers: guard the byte loop
        std::string guard = r.nullable() ? "(" + r.n + ") ? false : " : "";
        if (op == PA_OP_EQUAL || op == PA_OP_NOT_EQUAL) {
            out << "bool " << r.v << " = " << guard << (op == PA_OP_NOT_EQUAL ? "!" : "") << "pa_str_eq(" << a.v << ", " << a.len
                << ", " << b.v << ", " << b.len << ");\n";
        }
        else {
            out << "bool " << r.v << " = " << guard << "(pa_str_cmp(" << a.v << ", " << a.len << ", " << b.v << ", " << b.len << ") "
                << s << " 0);\n";
        }
        return r;
    }
    if (is_decimal_type(a.type) || is_decimal_type(b.type)) {
        // operands of ONE decimal type (the planner casts them): the unscaled values compare (ShortDecimalType / LongDecimalType operators)
        PA_REQUIRE(a.type == b.type && PA_DECIMAL_SCALE(a.param) == PA_DECIMAL_SCALE(b.param), PA_ERR_NOT_SUPPORTED,
                   "comparison between decimals of different types needs an explicit CAST");
        out << "bool " << r.v << " = (" << a.v << " " << s << " " << b.v << ");\n";
        return r;
    }
    bool ad = a.type == PA_DOUBLE, bd = b.type == PA_DOUBLE;
    PA_REQUIRE(ad == bd, PA_ERR_NOT_SUPPORTED, "comparison between DOUBLE and non-DOUBLE needs an explicit CAST");
    // RealOperators' comparisons are Java float comparisons (RealOperators.java: intBitsToFloat(left) < intBitsToFloat(right) ...)
    PA_REQUIRE((a.type == PA_REAL) == (b.type == PA_REAL), PA_ERR_NOT_SUPPORTED, "comparison between REAL and non-REAL needs an explicit CAST");
    out << "bool " << r.v << " = (" << a.v << " " << s << " " << b.v << ");\n";
    return r;
}

GenValue RowCodegen::emit_node(const OwnedExpr& e, int32_t id, std::ostringstream& out)
{
    const pa_expr_node& n = e.node(id);
    const int32_t* a = e.node_args(id);
    GenValue r;
    r.type = n.type;
    switch (n.kind) {
        case PA_EXPR_INPUT_REF: {
            PA_REQUIRE(n.channel >= 0 && n.channel < (int32_t)channels_.size(), PA_ERR_INVALID_ARGUMENT, "input channel out of range");
            const ChannelLayout& c = channels_[n.channel];
            r.type = c.type;
            r.param = is_decimal_type(c.type) ? n.type_param : 0;
            PA_REQUIRE(!is_decimal_type(c.type) || (n.type == c.type && n.type_param != 0), PA_ERR_INVALID_ARGUMENT,
                       "a DECIMAL input reference carries its type parameter (precision, scale)");
            r.v = "c" + std::to_string(n.channel);
            if (c.type == PA_VARCHAR) r.len = "cl" + std::to_string(n.channel);
            r.n = c.nullable ? "cn" + std::to_string(n.channel) : "false";
            return r;
        }
        case PA_EXPR_CONSTANT: {
            r.n = n.is_null ? "true" : "false";
            switch (n.type) {
                case PA_DOUBLE:
                    r.v = n.is_null ? "0.0" : double_literal(n.f64);
                    break;
                case PA_REAL:  // the constant travels in f64 (every float is exactly a double)
                    r.v = n.is_null ? "0.0f" : "((float)" + double_literal((double)(float)n.f64) + ")";
                    break;
                case PA_BOOLEAN:
                    r.v = (!n.is_null && n.i64) ? "true" : "false";
                    break;
                case PA_VARCHAR:
                    r.v = bytes_literal(e.strings[id]);
                    r.len = std::to_string(e.strings[id].size());
                    break;
                case PA_LONG_DECIMAL: {  // two's complement halves: low in i64, high in the bits of f64
                    uint64_t hi = 0;
                    memcpy(&hi, &n.f64, 8);
                    char buf[96];
                    snprintf(buf, sizeof buf, "pa_i128_of(0x%" PRIx64 "ULL, 0x%" PRIx64 "ULL)", n.is_null ? (uint64_t)0 : hi, n.is_null ? (uint64_t)0 : (uint64_t)n.i64);
                    r.v = buf;
                    r.param = n.type_param;
                    break;
                }
                default: {
                    if (n.type == PA_DECIMAL) r.param = n.type_param;
                    char buf[40];
                    int64_t v = n.is_null ? 0 : n.i64;
                    if (v == INT64_MIN) snprintf(buf, sizeof buf, "(-9223372036854775807LL - 1)");
                    else snprintf(buf, sizeof buf, v < 0 ? "(%" PRId64 "LL)" : "%" PRId64 "LL", v);
                    r.v = buf;
                }
            }
            return r;
        }
        case PA_EXPR_CALL: {
            if (n.op >= PA_OP_EQUAL && n.op <= PA_OP_GREATER_THAN_OR_EQUAL) {
                PA_REQUIRE(n.nargs == 2, PA_ERR_INVALID_ARGUMENT, "comparison needs 2 arguments");
                GenValue x = emit_node(e, a[0], out);
                GenValue y = emit_node(e, a[1], out);
                return emit_compare(n.op, x, y, out);
            }
            if (n.op == PA_OP_NOT) {
                GenValue x = emit_node(e, a[0], out);
                r.v = "(!" + x.v + ")";
                r.n = x.n;
                r.type = PA_BOOLEAN;
                return r;
            }
            if (n.op == PA_OP_CAST) {
                GenValue x = emit_node(e, a[0], out);
                r.n = x.n;
                if (is_decimal_type(n.type) && (is_decimal_type(x.type) || is_int_type(x.type))) {
                    // DecimalCasts (bigint -> decimal: value * 10^scale) / DecimalConversions (decimal -> decimal: rescale, a division
                    // rounds half up); out of range once the magnitude reaches 10^precision
                    const int p = PA_DECIMAL_PRECISION(n.type_param), sc = PA_DECIMAL_SCALE(n.type_param);
                    const int sx = is_decimal_type(x.type) ? PA_DECIMAL_SCALE(x.param) : 0;
                    PA_REQUIRE(p >= 1 && p <= 38 && sc <= p, PA_ERR_INVALID_ARGUMENT, "CAST to a DECIMAL without its type parameter");
                    r.param = n.type_param;
                    r.v = fresh("t");
                    const std::string guard = r.nullable() ? "(" + r.n + ") ? (i128)0 : " : "";
                    std::string wide;
                    if (sc >= sx) wide = "pa_dec_mul((i128)" + x.v + ", " + pow10_literal(sc - sx, true) + ", " + err_ + ")";
                    else {
                        PA_REQUIRE(x.type != PA_LONG_DECIMAL && sx - sc <= 18, PA_ERR_NOT_SUPPORTED, "rescaling a long decimal down is not on the device path");
                        wide = "(i128)pa_dec_div_round(" + x.v + ", " + pow10_literal(sx - sc, false) + ")";
                    }
                    out << "const i128 " << r.v << "w = " << guard << "pa_dec_check_bound(" << wide << ", " << pow10_literal(p, true) << ", " << err_ << ");\n";
                    if (n.type == PA_DECIMAL) out << "const i64 " << r.v << " = (i64)" << r.v << "w;\n";
                    else out << "const i128 " << r.v << " = " << r.v << "w;\n";
                    return r;
                }
                if (n.type == PA_DOUBLE && is_int_type(x.type)) r.v = "((double)" + x.v + ")";
                // RealOperators.castToDouble (:164-169), DoubleOperators.castToReal, BigintOperators / IntegerOperators.castToReal: Java
                // widening / narrowing primitive conversions = the C conversions (round to nearest even)
                else if (n.type == PA_DOUBLE && x.type == PA_REAL) r.v = "((double)" + x.v + ")";
                else if (n.type == PA_REAL && (x.type == PA_DOUBLE || is_int_type(x.type))) r.v = "((float)" + x.v + ")";
                else if (n.type == PA_BIGINT && is_int_type(x.type)) r.v = x.v;
                else if (n.type == x.type) { r.v = x.v; r.len = x.len; }
                else throw Error(PA_ERR_NOT_SUPPORTED, "CAST not supported on device");
                return r;
            }
            PA_REQUIRE(n.op >= PA_OP_ADD && n.op <= PA_OP_NEGATE, PA_ERR_NOT_SUPPORTED, "unknown call operator");
            GenValue x = emit_node(e, a[0], out);
            GenValue y;
            if (n.op != PA_OP_NEGATE) {
                PA_REQUIRE(n.nargs == 2, PA_ERR_INVALID_ARGUMENT, "binary operator needs 2 arguments");
                y = emit_node(e, a[1], out);
                PA_REQUIRE((x.type == PA_DOUBLE) == (y.type == PA_DOUBLE), PA_ERR_NOT_SUPPORTED,
                           "mixed DOUBLE / integer arithmetic needs an explicit CAST");
                PA_REQUIRE((x.type == PA_REAL) == (y.type == PA_REAL), PA_ERR_NOT_SUPPORTED, "mixed REAL / other arithmetic needs an explicit CAST");
            }
            else {
                y.n = "false";
            }
            r.n = or_nulls({x.n, y.n});
            r.v = fresh("t");
            if (is_decimal_type(n.type)) {
                // DecimalOperators: add / subtract rescale both operands to the result scale, multiply multiplies the unscaled values;
                // a result of at most 18 digits is long arithmetic (the derived precision cannot overflow it), a long result raises
                // NUMERIC_VALUE_OUT_OF_RANGE once its magnitude reaches 10^38
                PA_REQUIRE(is_decimal_type(x.type) && (n.op == PA_OP_NEGATE || is_decimal_type(y.type)), PA_ERR_NOT_SUPPORTED,
                           "mixed DECIMAL / other arithmetic needs an explicit CAST");
                PA_REQUIRE(n.op == PA_OP_ADD || n.op == PA_OP_SUBTRACT || n.op == PA_OP_MULTIPLY || n.op == PA_OP_NEGATE, PA_ERR_NOT_SUPPORTED,
                           "DECIMAL division / modulus are not on the device path");
                PA_REQUIRE(n.type_param != 0, PA_ERR_INVALID_ARGUMENT, "a DECIMAL expression node carries its type parameter (precision, scale)");
                r.param = n.type_param;
                const int sr = PA_DECIMAL_SCALE(n.type_param), sx = PA_DECIMAL_SCALE(x.param), sy = PA_DECIMAL_SCALE(y.param);
                const bool wide = n.type == PA_LONG_DECIMAL;
                const std::string T = wide ? "i128" : "i64";
                const std::string guard = r.nullable() ? "(" + r.n + ") ? (" + T + ")0 : " : "";
                std::string val;
                if (n.op == PA_OP_NEGATE) val = "-(" + x.v + ")";
                else if (n.op == PA_OP_MULTIPLY) {
                    PA_REQUIRE(sr == sx + sy, PA_ERR_INVALID_ARGUMENT, "DECIMAL multiply: the result scale is the sum of the operand scales");
                    val = wide ? "pa_dec_check38(pa_dec_mul((i128)" + x.v + ", (i128)" + y.v + ", " + err_ + "), " + err_ + ")" : "(" + x.v + " * " + y.v + ")";
                }
                else {
                    PA_REQUIRE(sr >= sx && sr >= sy, PA_ERR_INVALID_ARGUMENT, "DECIMAL add / subtract: the result scale is the larger operand scale");
                    if (wide) {
                        const std::string xs = sr == sx ? "(i128)" + x.v : "pa_dec_mul((i128)" + x.v + ", " + pow10_literal(sr - sx, true) + ", " + err_ + ")";
                        const std::string ys = sr == sy ? "(i128)" + y.v : "pa_dec_mul((i128)" + y.v + ", " + pow10_literal(sr - sy, true) + ", " + err_ + ")";
                        val = std::string("pa_dec_check38(") + (n.op == PA_OP_ADD ? "pa_dec_add(" : "pa_dec_sub(") + xs + ", " + ys + ", " + err_ + "), " + err_ + ")";
                    }
                    else {
                        PA_REQUIRE(x.type == PA_DECIMAL && y.type == PA_DECIMAL, PA_ERR_INVALID_ARGUMENT, "a short DECIMAL result has short operands");
                        const std::string xs = sr == sx ? x.v : "(" + x.v + " * " + pow10_literal(sr - sx, false) + ")";
                        const std::string ys = sr == sy ? y.v : "(" + y.v + " * " + pow10_literal(sr - sy, false) + ")";
                        val = "(" + xs + (n.op == PA_OP_ADD ? " + " : " - ") + ys + ")";
                    }
                }
                out << "const " << T << " " << r.v << " = " << guard << val << ";\n";
                return r;
            }
            if (n.type == PA_DOUBLE) {
                PA_REQUIRE(x.type == PA_DOUBLE, PA_ERR_NOT_SUPPORTED, "DOUBLE arithmetic on non-DOUBLE operands");
                // DoubleOperators.java:59-110; kept as separate statements, compiled with -ffp-contract=off
                out << "double " << r.v << " = ";
                switch (n.op) {
                    case PA_OP_ADD: out << x.v << " + " << y.v; break;
                    case PA_OP_SUBTRACT: out << x.v << " - " << y.v; break;
                    case PA_OP_MULTIPLY: out << x.v << " * " << y.v; break;
                    case PA_OP_DIVIDE: out << x.v << " / " << y.v; break;
                    case PA_OP_MODULUS: out << "fmod(" << x.v << ", " << y.v << ")"; break;
                    default: out << "-(" << x.v << ")"; break;  // (a negative literal operand would read as `--`)
                }
                out << ";\n";
                return r;
            }
            if (n.type == PA_REAL) {
                PA_REQUIRE(x.type == PA_REAL, PA_ERR_NOT_SUPPORTED, "REAL arithmetic on non-REAL operands");
                // RealOperators.java:55-95: Java float arithmetic (IEEE single, no contraction); % is fmodf
                out << "float " << r.v << " = ";
                switch (n.op) {
                    case PA_OP_ADD: out << x.v << " + " << y.v; break;
                    case PA_OP_SUBTRACT: out << x.v << " - " << y.v; break;
                    case PA_OP_MULTIPLY: out << x.v << " * " << y.v; break;
                    case PA_OP_DIVIDE: out << x.v << " / " << y.v; break;
                    case PA_OP_MODULUS: out << "fmodf(" << x.v << ", " << y.v << ")"; break;
                    default: out << "-(" << x.v << ")"; break;
                }
                out << ";\n";
                return r;
            }
            PA_REQUIRE(is_int_type(n.type) && is_int_type(x.type), PA_ERR_NOT_SUPPORTED, "arithmetic type not supported on device");
            // exact arithmetic (BigintOperators.java:47-121); NULL operands carry unspecified values,
            // so the throwing operation only runs on non-null rows
            std::string call;
            switch (n.op) {
                case PA_OP_ADD: call = "pa_add_exact(" + x.v + ", " + y.v + ", " + err_ + ")"; break;
                case PA_OP_SUBTRACT: call = "pa_sub_exact(" + x.v + ", " + y.v + ", " + err_ + ")"; break;
                case PA_OP_MULTIPLY: call = "pa_mul_exact(" + x.v + ", " + y.v + ", " + err_ + ")"; break;
                case PA_OP_DIVIDE: call = "pa_div_exact(" + x.v + ", " + y.v + ", " + err_ + ")"; break;
                case PA_OP_MODULUS: call = "pa_mod_exact(" + x.v + ", " + y.v + ", " + err_ + ")"; break;
                default: call = "pa_sub_exact(0LL, " + x.v + ", " + err_ + ")"; break;
            }
            if (n.type != PA_BIGINT) call = "pa_int_range(" + call + ", " + err_ + ")";
            out << "i64 " << r.v << " = " << (r.nullable() ? "(" + r.n + ") ? 0LL : " : "") << call << ";\n";
            return r;
        }
        case PA_EXPR_SPECIAL: {
            switch (n.op) {
                case PA_FORM_AND:
                case PA_FORM_OR: {
                    const bool is_and = n.op == PA_FORM_AND;
                    r.type = PA_BOOLEAN;
                    bool lazy = false;
                    for (int32_t k = 1; k < n.nargs; k++) lazy = lazy || can_throw(e, a[k]);
                    std::string v = fresh("t"), nn = fresh("n");
                    if (!lazy) {
                        // decided = some term is definitely false (AND) / true (OR)
                        std::vector<std::string> decided, nulls;
                        for (int32_t k = 0; k < n.nargs; k++) {
                            GenValue x = emit_node(e, a[k], out);
                            std::string hit = is_and ? "!" + x.v : x.v;
                            decided.push_back(x.nullable() ? "(!" + x.n + " && " + hit + ")" : "(" + hit + ")");
                            nulls.push_back(x.n);
                        }
                        std::string d = fresh("d");
                        out << "bool " << d << " = ";
                        for (size_t k = 0; k < decided.size(); k++) out << (k ? " || " : "") << decided[k];
                        out << ";\n";
                        std::string anynull = or_nulls(nulls);
                        out << "bool " << v << " = " << (is_and ? "!" + d : d) << ";\n";
                        if (anynull == "false") {
                            r.n = "false";
                        }
                        else {
                            out << "bool " << nn << " = !" << d << " && " << anynull << ";\n";
                            r.n = nn;
                        }
                        r.v = v;
                        return r;
                    }
                    // left-to-right short circuit (AndCodeGenerator.java:63-71): a decided term hides
                    // errors of the terms to its right
                    std::string done = fresh("d");
                    out << "bool " << v << " = " << (is_and ? "true" : "false") << "; bool " << nn << " = false; bool " << done
                        << " = false;\n";
                    for (int32_t k = 0; k < n.nargs; k++) {
                        out << "if (!" << done << ") {\n";
                        GenValue x = emit_node(e, a[k], out);
                        std::string hit = is_and ? "!" + x.v : x.v;
                        if (x.nullable()) {
                            out << "if (" << x.n << ") { " << nn << " = true; } else if (" << hit << ") { " << v << " = "
                                << (is_and ? "false" : "true") << "; " << nn << " = false; " << done << " = true; }\n";
                        }
                        else {
                            out << "if (" << hit << ") { " << v << " = " << (is_and ? "false" : "true") << "; " << nn << " = false; "
                                << done << " = true; }\n";
                        }
                        out << "}\n";
                    }
                    r.v = v;
                    r.n = nn;
                    return r;
                }
                case PA_FORM_BETWEEN: {
                    PA_REQUIRE(n.nargs == 3, PA_ERR_INVALID_ARGUMENT, "BETWEEN needs 3 arguments");
                    GenValue x = emit_node(e, a[0], out);
                    GenValue lo = emit_node(e, a[1], out);
                    GenValue hi = emit_node(e, a[2], out);
                    GenValue c1 = emit_compare(PA_OP_GREATER_THAN_OR_EQUAL, x, lo, out);
                    GenValue c2 = emit_compare(PA_OP_LESS_THAN_OR_EQUAL, x, hi, out);
                    r.type = PA_BOOLEAN;
                    std::string d = fresh("d"), v = fresh("t");
                    auto decided = [](const GenValue& c) {
                        return c.nullable() ? "(!" + c.n + " && !" + c.v + ")" : "(!" + c.v + ")";
                    };
                    out << "bool " << d << " = " << decided(c1) << " || " << decided(c2) << ";\n";
                    out << "bool " << v << " = !" << d << ";\n";
                    std::string anynull = or_nulls({c1.n, c2.n});
                    if (anynull == "false") {
                        r.n = "false";
                    }
                    else {
                        std::string nn = fresh("n");
                        out << "bool " << nn << " = !" << d << " && " << anynull << ";\n";
                        r.n = nn;
                    }
                    r.v = v;
                    return r;
                }
                case PA_FORM_IS_NULL: {
                    GenValue x = emit_node(e, a[0], out);
                    r.type = PA_BOOLEAN;
                    r.v = x.n;
                    r.n = "false";
                    return r;
                }
                case PA_FORM_IF: {
                    PA_REQUIRE(n.nargs == 3, PA_ERR_INVALID_ARGUMENT, "IF needs 3 arguments");
                    PA_REQUIRE(n.type != PA_VARCHAR, PA_ERR_NOT_SUPPORTED, "VARCHAR-valued IF not supported on device");
                    GenValue c = emit_node(e, a[0], out);
                    std::string cond = fresh("t");
                    out << "bool " << cond << " = " << (c.nullable() ? "!" + c.n + " && " : "") << c.v << ";\n";
                    std::string v = fresh("t"), nn = fresh("n");
                    out << ctype(n.type) << " " << v << "; bool " << nn << ";\n";
                    out << "if (" << cond << ") {\n";
                    GenValue x = emit_node(e, a[1], out);
                    out << v << " = " << x.v << "; " << nn << " = " << x.n << ";\n} else {\n";
                    GenValue y = emit_node(e, a[2], out);
                    out << v << " = " << y.v << "; " << nn << " = " << y.n << ";\n}\n";
                    r.v = v;
                    r.n = nn;
                    r.param = is_decimal_type(n.type) ? n.type_param : 0;
                    return r;
                }
                case PA_FORM_COALESCE: {
                    PA_REQUIRE(n.type != PA_VARCHAR, PA_ERR_NOT_SUPPORTED, "VARCHAR-valued COALESCE not supported on device");
                    std::string v = fresh("t"), nn = fresh("n");
                    out << ctype(n.type) << " " << v << " = " << (n.type == PA_DOUBLE ? "0.0" : (n.type == PA_REAL ? "0.0f" : (n.type == PA_BOOLEAN ? "false" : "0LL")))
                        << "; bool " << nn << " = true;\n";
                    for (int32_t k = 0; k < n.nargs; k++) {
                        out << "if (" << nn << ") {\n";
                        GenValue x = emit_node(e, a[k], out);
                        out << v << " = " << x.v << "; " << nn << " = " << x.n << ";\n}\n";
                    }
                    r.v = v;
                    r.n = nn;
                    r.param = is_decimal_type(n.type) ? n.type_param : 0;
                    return r;
                }
                case PA_FORM_IN: {
                    PA_REQUIRE(n.nargs >= 2, PA_ERR_INVALID_ARGUMENT, "IN needs a value and candidates");
                    GenValue x = emit_node(e, a[0], out);
                    std::vector<std::string> hits, nulls;
                    for (int32_t k = 1; k < n.nargs; k++) {
                        GenValue c = emit_node(e, a[k], out);
                        GenValue eq = emit_compare(PA_OP_EQUAL, x, c, out);
                        hits.push_back(eq.nullable() ? "(!" + eq.n + " && " + eq.v + ")" : eq.v);
                        nulls.push_back(c.n);
                    }
                    r.type = PA_BOOLEAN;
                    std::string v = fresh("t");
                    out << "bool " << v << " = ";
                    for (size_t k = 0; k < hits.size(); k++) out << (k ? " || " : "") << hits[k];
                    out << ";\n";
                    std::string cand_null = or_nulls(nulls);
                    if (x.n == "false" && cand_null == "false") {
                        r.n = "false";
                    }
                    else {
                        std::string nn = fresh("n");
                        out << "bool " << nn << " = " << x.n << " || (!" << v << " && " << cand_null << ");\n";
                        r.n = nn;
                    }
                    r.v = v;
                    return r;
                }
                default:
                    throw Error(PA_ERR_NOT_SUPPORTED, "special form not supported on device");
            }
        }
        default:
            throw Error(PA_ERR_NOT_SUPPORTED, "expression kind not supported on device");
    }
}

}  // namespace pa
