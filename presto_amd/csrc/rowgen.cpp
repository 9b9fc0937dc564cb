// rowgen.cpp -- shared pieces of the generated kernels: how the columns of a page reach the per-row
// function (4 rows per lane per step with 16-byte loads when every buffer is aligned, scalar otherwise).
// Kernel argument blocks expose the columns as a.v[c] / a.o[c] / a.nl[c] and the error word as a.err.
#include "rowgen.hpp"

namespace pa {

std::string row_params(const RowInputs& s, const std::vector<ChannelLayout>& layout)
{
    std::ostringstream p;
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used[c]) continue;
        p << ", " << RowCodegen::ctype(layout[c].type) << " c" << c;
        if (layout[c].type == PA_VARCHAR) p << ", i32 cl" << c;
        if (s.short_bound[c] > 0) p << ", u64 cs" << c;
        if (layout[c].nullable) p << ", bool cn" << c;
    }
    return p.str();
}

// the same parameters by name only (", c0, ..."): forwarding from one generated function to another
std::string row_param_names(const RowInputs& s, const std::vector<ChannelLayout>& layout)
{
    std::ostringstream p;
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used[c]) continue;
        p << ", c" << c;
        if (layout[c].type == PA_VARCHAR) p << ", cl" << c;
        if (s.short_bound[c] > 0) p << ", cs" << c;
        if (layout[c].nullable) p << ", cn" << c;
    }
    return p.str();
}

// Kernel prologue: wave-uniform facts about the page used by the speculative VARCHAR(1) path.
void emit_prologue(const RowInputs& s, const std::vector<ChannelLayout>& layout, std::ostringstream& o, const ColumnNames& nm)
{
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used[c] || layout[c].type != PA_VARCHAR || s.short_bound[c] != 1) continue;
        o << "    const i32 P" << c << " = " << nm.n() << " > 0 ? " << nm.o(c) << "[0] : 0;\n";
        o << "    const i64 PB" << c << " = " << nm.n() << " > 0 ? (i64)" << nm.o(c) << "[" << nm.n() << "] - P" << c << " : 0;\n";
    }
}

// vector loads of row quad q and the 4 argument lists.  All independent loads are issued first (one HBM
// round trip per step); work that depends on loaded offsets follows.  VARCHAR(1) keys read their 4 bytes
// speculatively at the position they have when every earlier string of the page is one byte long, and
// fall back to the offset-dependent path otherwise.
void emit_vector_loads(const RowInputs& s, const std::vector<ChannelLayout>& layout, std::ostringstream& o, std::string args[4], const ColumnNames& nm)
{
    static const char* xyzw[4] = {"x", "y", "z", "w"};
    std::ostringstream post;
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used[c]) continue;
        const std::string C = std::to_string(c), V = nm.v(c), O = nm.o(c), NL = nm.nl(c);
        switch (layout[c].type) {
            case PA_BIGINT:
            case PA_DECIMAL:  // ShortDecimalType: a LongArrayBlock of unscaled values
                o << "        pa_i64x2 A" << C << " = ((const pa_i64x2*)" << V << ")[2 * q], B" << C << " = ((const pa_i64x2*)" << V
                  << ")[2 * q + 1];\n";
                for (int r = 0; r < 4; r++) args[r] += ", " + std::string(r < 2 ? "A" : "B") + C + "." + xyzw[r & 1];
                break;
            case PA_LONG_DECIMAL:  // 16 bytes per position: one 16-byte load per row
                for (int r = 0; r < 4; r++) {
                    o << "        pa_i64x2 L" << C << r << " = ((const pa_i64x2*)" << V << ")[4 * q + " << r << "];\n";
                    args[r] += ", pa_ld_from(L" + C + std::to_string(r) + ".x, L" + C + std::to_string(r) + ".y)";
                }
                break;
            case PA_DOUBLE:
                o << "        pa_f64x2 A" << C << " = ((const pa_f64x2*)" << V << ")[2 * q], B" << C << " = ((const pa_f64x2*)" << V
                  << ")[2 * q + 1];\n";
                for (int r = 0; r < 4; r++) args[r] += ", " + std::string(r < 2 ? "A" : "B") + C + "." + xyzw[r & 1];
                break;
            case PA_INTEGER:
            case PA_DATE:
                o << "        pa_i32x4 A" << C << " = ((const pa_i32x4*)" << V << ")[q];\n";
                for (int r = 0; r < 4; r++) args[r] += ", (i64)A" + C + "." + xyzw[r];
                break;
            case PA_REAL:
                o << "        pa_f32x4 A" << C << " = ((const pa_f32x4*)" << V << ")[q];\n";
                for (int r = 0; r < 4; r++) args[r] += ", A" + C + "." + xyzw[r];
                break;
            case PA_BOOLEAN:
                o << "        u32 A" << C << " = ((const u32*)" << V << ")[q];\n";
                for (int r = 0; r < 4; r++) args[r] += ", ((A" + C + " >> " + std::to_string(8 * r) + ") & 0xffu) != 0u";
                break;
            case PA_VARCHAR: {
                o << "        pa_i32x4 O" << C << " = ((const pa_i32x4*)" << O << ")[q]; i32 E" << C << " = " << O << "[4 * q + 4];\n";
                std::string lo[4], len[4];
                for (int r = 0; r < 4; r++) {
                    lo[r] = "O" + C + "." + xyzw[r];
                    len[r] = (r < 3 ? "O" + C + "." + xyzw[r + 1] : "E" + C) + " - " + lo[r];
                }
                if (s.short_bound[c] > 0) {
                    post << "        u64 S" << C << "0, S" << C << "1, S" << C << "2, S" << C << "3;\n";
                    if (s.short_bound[c] == 1) {
                        o << "        u32 K" << C << " = 0u; if (4 * q + 4 <= PB" << C << ") __builtin_memcpy(&K" << C << ", (const u8*)" << V
                          << " + P" << C << " + 4 * q, 4);\n";
                        post << "        if (E" << C << " - " << lo[0] << " == 4) {\n            u32 pk = K" << C << ";\n            if (" << lo[0]
                             << " != P" << C << " + (i32)(4 * q) || 4 * q + 4 > PB" << C << ") __builtin_memcpy(&pk, (const u8*)" << V << " + "
                             << lo[0] << ", 4);\n";
                        for (int r = 0; r < 4; r++) post << "            S" << C << r << " = (pk >> " << 8 * r << ") & 0xffu;\n";
                        post << "        } else {\n";
                    }
                    else {
                        post << "        {\n";
                    }
                    for (int r = 0; r < 4; r++) {
                        post << "            S" << C << r << " = pa_short_bytes((const u8*)" << V << " + " << lo[r] << ", " << len[r] << ", "
                             << s.short_bound[c] << ", a.err);\n";
                    }
                    post << "        }\n";
                }
                for (int r = 0; r < 4; r++) {
                    args[r] += ", (const u8*)" + V + " + " + lo[r] + ", " + len[r];
                    if (s.short_bound[c] > 0) args[r] += ", S" + C + std::to_string(r);
                }
                break;
            }
            default:
                throw Error(PA_ERR_NOT_SUPPORTED, "column type not supported on device");
        }
        if (layout[c].nullable) {
            // a page without a valueIsNull array on a channel that had one earlier passes a null pointer
            o << "        u32 N" << C << " = " << NL << " ? ((const u32*)" << NL << ")[q] : 0u;\n";
            for (int r = 0; r < 4; r++) args[r] += ", ((N" + C + " >> " + std::to_string(8 * r) + ") & 0xffu) != 0u";
        }
    }
    o << post.str();
}

std::string VectorVar::load(const std::string& q) const
{
    const std::string idx = per_quad == 2 ? "2 * (" + q + ")" + (half ? " + 1" : "") : "(" + q + ")";
    const std::string ld = "((const " + type + "*)" + array + ")[" + idx + "]";
    return maybe_null ? "(" + array + " ? " + ld + " : 0u)" : ld;
}

bool vector_load_vars(const RowInputs& s, const std::vector<ChannelLayout>& layout, std::vector<VectorVar>& vars, const ColumnNames& nm)
{
    vars.clear();
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used[c]) continue;
        const std::string C = std::to_string(c);
        switch (layout[c].type) {
            case PA_BIGINT:
            case PA_DECIMAL:
                vars.push_back(VectorVar{"pa_i64x2", "A" + C, nm.v(c), 2, 0, false});
                vars.push_back(VectorVar{"pa_i64x2", "B" + C, nm.v(c), 2, 1, false});
                break;
            case PA_DOUBLE:
                vars.push_back(VectorVar{"pa_f64x2", "A" + C, nm.v(c), 2, 0, false});
                vars.push_back(VectorVar{"pa_f64x2", "B" + C, nm.v(c), 2, 1, false});
                break;
            case PA_INTEGER:
            case PA_DATE: vars.push_back(VectorVar{"pa_i32x4", "A" + C, nm.v(c), 1, 0, false}); break;
            case PA_REAL: vars.push_back(VectorVar{"pa_f32x4", "A" + C, nm.v(c), 1, 0, false}); break;
            case PA_BOOLEAN: vars.push_back(VectorVar{"u32", "A" + C, nm.v(c), 1, 0, false}); break;
            default: return false;
        }
        if (layout[c].nullable) vars.push_back(VectorVar{"u32", "N" + C, nm.nl(c), 1, 0, true});
    }
    return true;
}

// (the argument list emit_vector_loads builds for row r of the quad, over variables with a prefix)
std::string vector_var_args(const RowInputs& s, const std::vector<ChannelLayout>& layout, const std::string& P, int r)
{
    static const char* xyzw[4] = {"x", "y", "z", "w"};
    std::string a;
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used[c]) continue;
        const std::string C = std::to_string(c);
        switch (layout[c].type) {
            case PA_BIGINT:
            case PA_DECIMAL:
            case PA_DOUBLE: a += ", " + P + std::string(r < 2 ? "A" : "B") + C + "." + xyzw[r & 1]; break;
            case PA_INTEGER:
            case PA_DATE: a += ", (i64)" + P + "A" + C + "." + xyzw[r]; break;
            case PA_REAL: a += ", " + P + "A" + C + "." + xyzw[r]; break;
            case PA_BOOLEAN: a += ", ((" + P + "A" + C + " >> " + std::to_string(8 * r) + ") & 0xffu) != 0u"; break;
            default: throw Error(PA_ERR_NOT_SUPPORTED, "column type not supported on device");
        }
        if (layout[c].nullable) a += ", ((" + P + "N" + C + " >> " + std::to_string(8 * r) + ") & 0xffu) != 0u";
    }
    return a;
}

std::string scalar_loads(const RowInputs& s, const std::vector<ChannelLayout>& layout, const std::string& row, const std::string& suffix, std::ostringstream& decl,
                         const ColumnNames& nm)
{
    std::string a;
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used[c]) continue;
        const std::string C = std::to_string(c), V = nm.v(c), O = nm.o(c), NL = nm.nl(c), L = "l" + C + suffix;
        switch (layout[c].type) {
            case PA_BIGINT:
            case PA_DECIMAL: decl << "const i64 " << L << " = ((const i64*)" << V << ")[" << row << "]; "; a += ", " + L; break;
            case PA_LONG_DECIMAL: decl << "const i128 " << L << " = pa_ld_read((const u64*)" << V << " + 2 * (" << row << ")); "; a += ", " + L; break;
            case PA_DOUBLE: decl << "const double " << L << " = ((const double*)" << V << ")[" << row << "]; "; a += ", " + L; break;
            case PA_INTEGER:
            case PA_DATE: decl << "const i64 " << L << " = (i64)((const i32*)" << V << ")[" << row << "]; "; a += ", " + L; break;
            case PA_REAL: decl << "const float " << L << " = ((const float*)" << V << ")[" << row << "]; "; a += ", " + L; break;
            case PA_BOOLEAN: decl << "const bool " << L << " = ((const u8*)" << V << ")[" << row << "] != 0; "; a += ", " + L; break;
            case PA_VARCHAR:
                decl << "const i32 " << L << "o = " << O << "[" << row << "], " << L << "e = " << O << "[(" << row << ") + 1]; ";
                a += ", (const u8*)" + V + " + " + L + "o, " + L + "e - " + L + "o";
                if (s.short_bound[c] > 0) a += ", pa_short_bytes((const u8*)" + V + " + " + L + "o, " + L + "e - " + L + "o, " + std::to_string(s.short_bound[c]) + ", a.err)";
                break;
            default: throw Error(PA_ERR_NOT_SUPPORTED, "column type not supported on device");
        }
        if (layout[c].nullable) {
            decl << "const bool ln" << C << suffix << " = " << NL << " != nullptr && " << NL << "[" << row << "] != 0; ";
            a += ", ln" + C + suffix;
        }
    }
    return a;
}

std::string scalar_args(const RowInputs& s, const std::vector<ChannelLayout>& layout, const ColumnNames& nm)
{
    std::string a;
    for (int c = 0; c < s.n_in; c++) {
        if (!s.used[c]) continue;
        const std::string C = std::to_string(c), V = nm.v(c), O = nm.o(c), NL = nm.nl(c);
        switch (layout[c].type) {
            case PA_BIGINT:
            case PA_DECIMAL: a += ", ((const i64*)" + V + ")[r]"; break;
            case PA_LONG_DECIMAL: a += ", pa_ld_read((const u64*)" + V + " + 2 * r)"; break;
            case PA_DOUBLE: a += ", ((const double*)" + V + ")[r]"; break;
            case PA_INTEGER:
            case PA_DATE: a += ", (i64)((const i32*)" + V + ")[r]"; break;
            case PA_REAL: a += ", ((const float*)" + V + ")[r]"; break;
            case PA_BOOLEAN: a += ", ((const u8*)" + V + ")[r] != 0"; break;
            case PA_VARCHAR:
                a += ", (const u8*)" + V + " + " + O + "[r], " + O + "[r + 1] - " + O + "[r]";
                if (s.short_bound[c] > 0) {
                    a += ", pa_short_bytes((const u8*)" + V + " + " + O + "[r], " + O + "[r + 1] - " + O + "[r], " +
                         std::to_string(s.short_bound[c]) + ", a.err)";
                }
                break;
            default: throw Error(PA_ERR_NOT_SUPPORTED, "column type not supported on device");
        }
        if (layout[c].nullable) a += ", (" + NL + " != nullptr && " + NL + "[r] != 0)";
    }
    return a;
}

}  // namespace pa
