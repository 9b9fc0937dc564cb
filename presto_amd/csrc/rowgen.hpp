// rowgen.hpp -- see rowgen.cpp.
#pragma once

#include <sstream>
#include <string>
#include <vector>

#include "exprgen.hpp"

namespace pa {

struct RowInputs {
    int n_in = 0;
    std::vector<bool> used;        // channels read by the expressions
    std::vector<int> short_bound;  // > 0: short VARCHAR whose packed bytes are passed as cs<c>
};

// How the generated loops name a page's buffers: the kernel argument block (a.v[c], a.o[c], a.nl[c], a.n), or -- `ranged` --
// locals of the loop over a table of row ranges (RV<c>, RO<c>, RNL<c>, RN; op_fused.hpp, the pa_fused_ranges kernels).
struct ColumnNames {
    bool ranged = false;
    std::string v(int c) const { return ranged ? "RV" + std::to_string(c) : "a.v[" + std::to_string(c) + "]"; }
    std::string o(int c) const { return ranged ? "RO" + std::to_string(c) : "a.o[" + std::to_string(c) + "]"; }
    std::string nl(int c) const { return ranged ? "RNL" + std::to_string(c) : "a.nl[" + std::to_string(c) + "]"; }
    std::string n() const { return ranged ? "RN" : "a.n"; }
};

// ", type c0, ..." parameter list of the per-row function
std::string row_params(const RowInputs& s, const std::vector<ChannelLayout>& layout);
std::string row_param_names(const RowInputs& s, const std::vector<ChannelLayout>& layout);
// wave-uniform per-page values the vector loads rely on (emit once, before the loops)
void emit_prologue(const RowInputs& s, const std::vector<ChannelLayout>& layout, std::ostringstream& o, const ColumnNames& names = ColumnNames());
// vector loads of row quad q (into `o`) and the 4 argument lists of the per-row calls
void emit_vector_loads(const RowInputs& s, const std::vector<ChannelLayout>& layout, std::ostringstream& o, std::string args[4],
                       const ColumnNames& names = ColumnNames());
// The vector loads of a quad as a list of variables (type, name, the load of quad `q` as an expression) and the argument lists over
// variables named <prefix><name>, for loops that keep several quads in flight (the next one being loaded, the current one, the one
// whose rows are still to be accumulated).  false: some used channel does not load that simply (VARCHAR, long DECIMAL).
struct VectorVar {
    std::string type, name, array;   // load of quad q: ((const type*)array)[index(q)], guarded when the array may be null
    int per_quad;                    // 2: the variable holds half a quad (index 2 q + half), 1: the whole quad
    int half;
    bool maybe_null;                 // (a valueIsNull array that a page may leave out)
    std::string load(const std::string& q) const;
};
bool vector_load_vars(const RowInputs& s, const std::vector<ChannelLayout>& layout, std::vector<VectorVar>& vars, const ColumnNames& names = ColumnNames());
std::string vector_var_args(const RowInputs& s, const std::vector<ChannelLayout>& layout, const std::string& prefix, int r);
// the same with the loads of the row's values named: declarations of locals <name><suffix> go to `decl` (so that the loads of several
// rows can be issued before the first row is worked on), the returned argument list names them
std::string scalar_loads(const RowInputs& s, const std::vector<ChannelLayout>& layout, const std::string& row, const std::string& suffix,
                         std::ostringstream& decl, const ColumnNames& names = ColumnNames());
// argument list of the scalar (row r) call
std::string scalar_args(const RowInputs& s, const std::vector<ChannelLayout>& layout, const ColumnNames& names = ColumnNames());

}  // namespace pa
