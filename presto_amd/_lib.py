"""Loader of libpresto_amd.so -- the C ABI declared in include/presto_amd.h.

The library is built in-tree by presto_amd/csrc/Makefile (see __graft_entry__.build()).  There is no
CPU fallback: a missing library raises here, and a missing GPU raises PA_ERR_NO_DEVICE from pa_init /
the operator factories.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRESTO_AMD_LIB") or os.path.join(_HERE, "libpresto_amd.so")  # PRESTO_AMD_LIB: A/B runs against another build
_LIB = None


class PrestoAmdError(RuntimeError):
    """Raised for a negative pa_status; mirrors TrinoException(StandardErrorCode, message)."""

    def __init__(self, status, message):
        super().__init__("%s: %s" % (abi.STATUS_NAMES.get(status, status), message))
        self.status = status
        self.message = message


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C presto_amd/csrc` (or __graft_entry__.build()); "
            "presto_amd has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.pa_last_error.restype = C.c_char_p
    L.pa_init.argtypes = [C.c_int32]
    L.pa_device_malloc.argtypes = [C.POINTER(vp), C.c_int64]
    L.pa_device_free.argtypes = [vp]
    L.pa_host_malloc_pinned.argtypes = [C.POINTER(vp), C.c_int64]
    L.pa_host_free_pinned.argtypes = [vp]
    L.pa_memcpy_h2d.argtypes = [vp, vp, C.c_int64, vp]
    L.pa_memcpy_d2h.argtypes = [vp, vp, C.c_int64, vp]
    L.pa_stream_synchronize.argtypes = [vp]
    L.pa_memory_set_limit.argtypes = [C.c_int64]
    L.pa_memory_stats.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.pa_stream_create.argtypes = [C.POINTER(vp)]
    L.pa_stream_destroy.argtypes = [vp]
    L.pa_filter_project_create.argtypes = [C.POINTER(abi.pa_filter_project_desc), C.POINTER(vp)]
    L.pa_scan_filter_project_create.argtypes = [C.POINTER(abi.pa_filter_project_desc), C.POINTER(abi.pa_page_source), C.POINTER(vp)]
    L.pa_scan_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.pa_aggregation_create.argtypes = [C.POINTER(abi.pa_aggregation_desc), C.POINTER(vp)]
    L.pa_hash_aggregation_create.argtypes = [C.POINTER(abi.pa_hash_aggregation_desc), C.POINTER(vp)]
    L.pa_fused_aggregation_create.argtypes = [C.POINTER(abi.pa_fused_aggregation_desc), C.POINTER(vp)]
    L.pa_lookup_source_create.argtypes = [C.POINTER(vp)]
    L.pa_lookup_source_destroy.argtypes = [vp]
    L.pa_hash_builder_create.argtypes = [C.POINTER(abi.pa_hash_builder_desc), vp, C.POINTER(vp)]
    L.pa_lookup_join_create.argtypes = [C.POINTER(abi.pa_lookup_join_desc), vp, C.POINTER(vp)]
    L.pa_fused_join_aggregation_create.argtypes = [C.POINTER(abi.pa_fused_join_aggregation_desc), vp, C.POINTER(vp)]
    L.pa_fused_join_create.argtypes = [C.POINTER(abi.pa_fused_join_desc), vp, C.POINTER(vp)]
    L.pa_codegen_fused_join.argtypes = [C.POINTER(abi.pa_fused_join_aggregation_desc), C.POINTER(abi.pa_hash_builder_desc), C.c_int32, C.c_char_p, C.c_int64]
    L.pa_codegen_fused_join.restype = C.c_int64
    L.pa_codegen_compile_fused_join.argtypes = [C.POINTER(abi.pa_fused_join_aggregation_desc), C.POINTER(abi.pa_hash_builder_desc), C.c_int32]
    L.pa_codegen_compile_fused_join.restype = C.c_int64
    L.pa_codegen_fused_join_probe.argtypes = [C.POINTER(abi.pa_fused_join_desc), C.POINTER(abi.pa_hash_builder_desc), C.c_char_p, C.c_int64]
    L.pa_codegen_fused_join_probe.restype = C.c_int64
    L.pa_codegen_compile_fused_join_probe.argtypes = [C.POINTER(abi.pa_fused_join_desc), C.POINTER(abi.pa_hash_builder_desc)]
    L.pa_codegen_compile_fused_join_probe.restype = C.c_int64
    L.pa_filter_project_set_dynamic_filter.argtypes = [vp, C.c_int32, vp]
    L.pa_aggregation_set_output_topn_hint.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.pa_lookup_source_position_count.argtypes = [vp]
    L.pa_lookup_source_key_range.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.pa_lookup_source_key_bitmap.argtypes = [vp, C.c_int64, C.c_uint64, vp, vp]
    L.pa_filter_project_set_dynamic_filter_bitmap.argtypes = [vp, C.c_int32, vp, C.c_int64, C.c_uint64]
    L.pa_dynamic_filter_source_create.argtypes = [C.POINTER(abi.pa_dynamic_filter_source_desc), C.POINTER(vp)]
    L.pa_dynamic_filter_poll.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(abi.pa_domain), C.c_int32]
    for name in ("pa_op_needs_input", "pa_op_finish", "pa_op_is_finished", "pa_op_is_blocked", "pa_op_close"):
        getattr(L, name).argtypes = [vp]
    L.pa_op_add_input.argtypes = [vp, C.POINTER(abi.pa_page)]
    L.pa_op_get_output.argtypes = [vp, C.POINTER(abi.pa_page)]
    L.pa_op_memory_bytes.argtypes = [vp]
    L.pa_op_memory_bytes.restype = C.c_int64
    L.pa_op_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.pa_op_kernel_name.argtypes = [vp, C.c_char_p, C.c_int32]
    L.pa_hash_page.argtypes = [C.POINTER(abi.pa_page), C.c_int32, C.POINTER(C.c_int32), vp, vp]
    L.pa_partition_ids.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]
    L.pa_partition_positions.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.pa_partition_columns_stable.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32), C.c_int32, vp, vp]
    L.pa_partition_columns.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32), C.c_int32, vp, vp]
    L.pa_gather_flat.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp]
    L.pa_tpch_generate.argtypes = [C.c_int32, C.c_double, C.c_int64, C.c_int64, C.c_uint64, vp, vp, vp]
    L.pa_codegen_fused.argtypes = [C.POINTER(abi.pa_fused_aggregation_desc), C.c_int32, C.c_char_p, C.c_int64, C.c_char_p]
    L.pa_codegen_fused.restype = C.c_int64
    L.pa_codegen_compile_fused.argtypes = [C.POINTER(abi.pa_fused_aggregation_desc), C.c_int32]
    L.pa_codegen_compile_fused.restype = C.c_int64
    L.pa_codegen_filter_project.argtypes = [C.POINTER(abi.pa_filter_project_desc), C.c_char_p, C.c_int64, C.c_char_p]
    L.pa_codegen_filter_project.restype = C.c_int64
    L.pa_codegen_compile_filter_project.argtypes = [C.POINTER(abi.pa_filter_project_desc)]
    L.pa_codegen_compile_filter_project.restype = C.c_int64
    L.pa_filter_project_selected_positions.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.pa_lookup_join_match_pairs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32)]
    L.pa_lookup_source_tables.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(C.c_int32)]
    L.pa_varwidth_gather_offsets.argtypes = [vp, vp, C.c_int32, vp, vp, C.POINTER(C.c_int64), vp]
    L.pa_varwidth_gather_bytes.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp]
    L.pa_offsets_from_lengths.argtypes = [vp, C.c_int32, vp, C.POINTER(C.c_int64), vp]
    L.pa_page_serialize.argtypes = [C.POINTER(abi.pa_page), vp, C.c_int64, vp]
    L.pa_page_serialize.restype = C.c_int64
    L.pa_page_deserialize.argtypes = [vp, C.c_int64, vp, C.POINTER(vp)]
    L.pa_page_serialize_lz4.argtypes = [C.POINTER(abi.pa_page), vp, C.c_int64, vp]
    L.pa_page_serialize_lz4.restype = C.c_int64
    L.pa_page_deserialize_typed.argtypes = [vp, C.c_int64, C.POINTER(C.c_int32), C.c_int32, vp, C.POINTER(vp)]
    L.pa_page_buffer_page.argtypes = [vp, C.POINTER(abi.pa_page)]
    L.pa_page_buffer_free.argtypes = [vp]
    L.pa_comm_unique_id.argtypes = [vp]
    L.pa_comm_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.pa_comm_create_host.argtypes = [C.POINTER(abi.pa_host_transport), C.c_int32, C.c_int32, C.POINTER(vp)]
    L.pa_comm_destroy.argtypes = [vp]
    L.pa_comm_rank.argtypes = [vp]
    L.pa_comm_world.argtypes = [vp]
    L.pa_comm_all_reduce_i64.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32, C.c_int32, vp]
    L.pa_comm_all_gather_i64.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, vp]
    L.pa_comm_preflight.argtypes = [vp, C.c_int64, vp]
    L.pa_exchange_create.argtypes = [C.POINTER(abi.pa_exchange_desc), vp, C.POINTER(vp)]
    L.pa_exchange_destroy.argtypes = [vp]
    L.pa_partitioned_output_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.pa_exchange_source_create.argtypes = [vp, C.c_int32, vp, C.POINTER(vp)]
    L.pa_exchange_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.pa_lookup_source_shared_key_bitmap.argtypes = [vp, vp, C.c_int32, vp, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    if L.pa_abi_version() != abi.ABI_VERSION:
        raise ImportError("libpresto_amd.so ABI version %d != %d" % (L.pa_abi_version(), abi.ABI_VERSION))
    _LIB = L
    return L


def check(rc):
    if rc < 0:
        raise PrestoAmdError(rc, lib().pa_last_error().decode("utf-8", "replace"))
    return rc


def init(device=-1):
    """pa_init: binds the thread to a gfx950 device, fails loudly when there is none."""
    check(lib().pa_init(device))


def device_synchronize():
    check(lib().pa_device_synchronize())


class DeviceAllocation:
    """HBM owned through the C ABI (pa_device_malloc), for hosts that do not bring torch."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().pa_device_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            lib().pa_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceStream:
    """A HIP stream owned through the C ABI: one per Driver, shared by the operators of its pipeline."""

    def __init__(self):
        p = C.c_void_p()
        check(lib().pa_stream_create(C.byref(p)))
        self.handle = p.value

    def synchronize(self):
        check(lib().pa_stream_synchronize(self.handle))

    def destroy(self):
        if self.handle:
            lib().pa_stream_destroy(self.handle)
            self.handle = None
