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
// locals of the loop over a table of row ranges (RV<c>, RO<c>, RNL<c>, RN; op_fused.cpp, the pa_fused_ranges kernels).
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
// the same with the loads of the row's values named: declarations of locals <name><suffix> go to `decl` (so that the loads of several
// rows can be issued before the first row is worked on), the returned argument list names them
std::string scalar_loads(const RowInputs& s, const std::vector<ChannelLayout>& layout, const std::string& row, const std::string& suffix,
                         std::ostringstream& decl, const ColumnNames& names = ColumnNames());
// argument list of the scalar (row r) call
std::string scalar_args(const RowInputs& s, const std::vector<ChannelLayout>& layout, const ColumnNames& names = ColumnNames());

}  // namespace pa
