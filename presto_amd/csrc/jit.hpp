// jit.hpp -- query-time device code: the MI355X counterpart of the reference's bytecode generation
// (core/trino-main/src/main/java/io/trino/sql/gen/PageFunctionCompiler.java:148-216 compiles and caches
// one PageFilter / PageProjection class per RowExpression; here one gfx950 code object per
// (expression set, column-layout signature)).
//
// Lookup order for a generated translation unit: in-process cache -> prebuilt code object shipped next
// to the library (csrc/prebuilt/<key>.hsaco, produced by __graft_entry__.build() with hipcc --genco)
// -> on-disk JIT cache -> hiprtc.  A failure is PA_ERR_COMPILER, never a CPU fallback.
#pragma once

#include <string>

#include "common.hpp"

namespace pa {

struct JitKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    std::string name;  // the kernel's name in the code object: `entry`_<first 8 hex digits of the key> (what a kernel trace shows)
};

// FNV-1a 64 of the generated source (+ device header + options): the cache key.
std::string jit_key(const std::string& source);

// Full translation unit for `source` (device header prepended), as written to prebuilt/<key>.hip.
std::string jit_translation_unit(const std::string& source);

// Returns the kernel `entry` of the code object generated from `source` (the source defines it as PA_K(entry): its name in the
// code object is entry_<key8>).
JitKernel jit_get(const std::string& source, const std::string& entry);

// Compile only (no device needed): returns the code object bytes; used by the CPU-side tests and by
// build() to validate generated code without a GPU.
std::string jit_compile_only(const std::string& source);

// Directory holding libpresto_amd.so
std::string library_dir();

}  // namespace pa
