// temporary: operators still under construction report NOT_SUPPORTED
#include "operator.hpp"
namespace pa {
pa_operator* make_filter_project(const pa_filter_project_desc*) { throw Error(PA_ERR_NOT_SUPPORTED, "filter_project: under construction"); }
pa_operator* make_hash_builder(const pa_hash_builder_desc*, pa_lookup_source*) { throw Error(PA_ERR_NOT_SUPPORTED, "hash_builder: under construction"); }
pa_operator* make_lookup_join(const pa_lookup_join_desc*, pa_lookup_source*) { throw Error(PA_ERR_NOT_SUPPORTED, "lookup_join: under construction"); }
}
extern "C" int32_t pa_partition_positions(const int32_t*, int32_t, int32_t, int32_t*, int64_t*, void*) { return PA_ERR_NOT_SUPPORTED; }
extern "C" int64_t pa_codegen_filter_project(const pa_filter_project_desc*, char*, int64_t, char*) { return PA_ERR_NOT_SUPPORTED; }
extern "C" int64_t pa_codegen_compile_filter_project(const pa_filter_project_desc*) { return PA_ERR_NOT_SUPPORTED; }
