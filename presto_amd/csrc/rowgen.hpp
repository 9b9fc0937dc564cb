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

// ", type c0, ..." parameter list of the per-row function
std::string row_params(const RowInputs& s, const std::vector<ChannelLayout>& layout);
std::string row_param_names(const RowInputs& s, const std::vector<ChannelLayout>& layout);
// wave-uniform per-page values the vector loads rely on (emit once, before the loops)
void emit_prologue(const RowInputs& s, const std::vector<ChannelLayout>& layout, std::ostringstream& o);
// vector loads of row quad q (into `o`) and the 4 argument lists of the per-row calls
void emit_vector_loads(const RowInputs& s, const std::vector<ChannelLayout>& layout, std::ostringstream& o, std::string args[4]);
// argument list of the scalar (row r) call
std::string scalar_args(const RowInputs& s, const std::vector<ChannelLayout>& layout);

}  // namespace pa
