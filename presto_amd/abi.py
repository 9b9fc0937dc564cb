"""ctypes mirror of include/presto_amd.h (struct layouts and enums only, no logic).

Shared by the product binding (presto_amd._lib) and -- for the struct definitions alone -- by the
test-only oracle binding (oracle/oracle.py), exactly as both C sides share the header.
"""
import ctypes as C

ABI_VERSION = 10

# pa_status
OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_ILLEGAL_STATE = -2
ERR_NOT_SUPPORTED = -3
ERR_NUMERIC_VALUE_OUT_OF_RANGE = -4
ERR_DIVISION_BY_ZERO = -5
ERR_INSUFFICIENT_RESOURCES = -6
ERR_DEVICE = -7
ERR_COMPILER = -8
ERR_NO_DEVICE = -9

STATUS_NAMES = {
    -1: "INVALID_ARGUMENT", -2: "ILLEGAL_STATE", -3: "NOT_SUPPORTED", -4: "NUMERIC_VALUE_OUT_OF_RANGE",
    -5: "DIVISION_BY_ZERO", -6: "GENERIC_INSUFFICIENT_RESOURCES", -7: "DEVICE_ERROR", -8: "COMPILER_ERROR",
    -9: "NO_DEVICE",
}

# pa_type
BIGINT, INTEGER, DATE, DOUBLE, BOOLEAN, VARCHAR, ROW, REAL, DECIMAL, LONG_DECIMAL = range(10)
TYPE_NAMES = ["BIGINT", "INTEGER", "DATE", "DOUBLE", "BOOLEAN", "VARCHAR", "ROW", "REAL", "DECIMAL", "LONG_DECIMAL"]
TYPE_WIDTH = {BIGINT: 8, INTEGER: 4, DATE: 4, DOUBLE: 8, BOOLEAN: 1, VARCHAR: 0, ROW: 0, REAL: 4, DECIMAL: 8, LONG_DECIMAL: 16}


class DecimalType(int):
    """DECIMAL(precision, scale) as a pa_type value that knows its parameters: compares and hashes as PA_DECIMAL (precision <= 18,
    ShortDecimalType) or PA_LONG_DECIMAL (LongDecimalType), `param` is PA_DECIMAL_PARAM(precision, scale)."""

    def __new__(cls, precision, scale):
        if not (1 <= precision <= 38 and 0 <= scale <= precision):
            raise ValueError("DECIMAL(%d, %d)" % (precision, scale))
        self = super().__new__(cls, DECIMAL if precision <= 18 else LONG_DECIMAL)
        self.precision, self.scale = precision, scale
        return self

    @property
    def param(self):
        return (self.precision << 8) | self.scale

    def __repr__(self):
        return "DECIMAL(%d, %d)" % (self.precision, self.scale)


def decimal(precision, scale):
    return DecimalType(precision, scale)


def type_param(t):
    """The type parameter a descriptor carries for type t (0 for the types that have none)."""
    return t.param if isinstance(t, DecimalType) else 0

# pa_encoding
FLAT, VARWIDTH, DICTIONARY, RLE, ROW_FIELDS = range(5)
# pa_mem
MEM_HOST, MEM_DEVICE = 0, 1
# pa_expr_kind
EXPR_INPUT_REF, EXPR_CONSTANT, EXPR_CALL, EXPR_SPECIAL = range(4)
# pa_call_op
(OP_ADD, OP_SUBTRACT, OP_MULTIPLY, OP_DIVIDE, OP_MODULUS, OP_NEGATE, OP_EQUAL, OP_NOT_EQUAL, OP_LESS_THAN,
 OP_LESS_THAN_OR_EQUAL, OP_GREATER_THAN, OP_GREATER_THAN_OR_EQUAL, OP_NOT, OP_CAST) = range(14)
# pa_special_form
FORM_AND, FORM_OR, FORM_BETWEEN, FORM_IS_NULL, FORM_IF, FORM_COALESCE, FORM_IN = range(7)
# pa_agg_fn
AGG_COUNT_STAR, AGG_COUNT, AGG_SUM, AGG_AVG, AGG_MIN, AGG_MAX = range(6)
# pa_agg_step
STEP_SINGLE, STEP_PARTIAL, STEP_FINAL = range(3)
# pa_tpch_column
(L_ORDERKEY, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_TAX, L_SHIPDATE, L_RETURNFLAG, L_LINESTATUS,
 O_ORDERKEY, O_CUSTKEY, O_ORDERDATE, O_SHIPPRIORITY, C_CUSTKEY, C_MKTSEGMENT) = range(14)
TPCH_COLUMN_TYPE = {
    L_ORDERKEY: BIGINT, L_QUANTITY: DOUBLE, L_EXTENDEDPRICE: DOUBLE, L_DISCOUNT: DOUBLE, L_TAX: DOUBLE,
    L_SHIPDATE: DATE, L_RETURNFLAG: VARCHAR, L_LINESTATUS: VARCHAR, O_ORDERKEY: BIGINT, O_CUSTKEY: BIGINT,
    O_ORDERDATE: DATE, O_SHIPPRIORITY: INTEGER, C_CUSTKEY: BIGINT, C_MKTSEGMENT: VARCHAR,
}


class pa_column(C.Structure):
    pass


pa_column._fields_ = [
    ("type", C.c_int32),
    ("encoding", C.c_int32),
    ("values", C.c_void_p),
    ("offsets", C.c_void_p),
    ("nulls", C.c_void_p),
    ("ids", C.c_void_p),
    ("dictionary", C.POINTER(pa_column)),
    ("dictionary_size", C.c_int32),
    ("reserved", C.c_int32),
]


class pa_page(C.Structure):
    _fields_ = [
        ("position_count", C.c_int32),
        ("channel_count", C.c_int32),
        ("columns", C.POINTER(pa_column)),
        ("mem", C.c_int32),
        ("flags", C.c_int32),
        ("release", C.c_void_p),       # PAGE_RELEASE function pointer (PAGE_RETAINED only)
        ("release_ctx", C.c_void_p),
    ]


PAGE_RELEASE = C.CFUNCTYPE(None, C.c_void_p)
PAGE_STABLE = 1
PAGE_PINNED = 2
PAGE_RETAINED = 4


class pa_expr_node(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("op", C.c_int32),
        ("type", C.c_int32),
        ("channel", C.c_int32),
        ("is_null", C.c_int32),
        ("nargs", C.c_int32),
        ("first_arg", C.c_int32),
        ("str_len", C.c_int32),
        ("i64", C.c_int64),
        ("f64", C.c_double),
        ("str", C.c_char_p),
    ]


class pa_expr(C.Structure):
    _fields_ = [
        ("node_count", C.c_int32),
        ("root", C.c_int32),
        ("nodes", C.POINTER(pa_expr_node)),
        ("arg_count", C.c_int32),
        ("reserved", C.c_int32),
        ("args", C.POINTER(C.c_int32)),
    ]


class pa_aggregate(C.Structure):
    _fields_ = [
        ("fn", C.c_int32),
        ("input_channel", C.c_int32),
        ("mask_channel", C.c_int32),
        ("input_type", C.c_int32),
    ]


class pa_filter_project_desc(C.Structure):
    _fields_ = [
        ("input_channel_count", C.c_int32),
        ("input_types", C.POINTER(C.c_int32)),
        ("input_type_params", C.POINTER(C.c_int32)),
        ("filter", C.POINTER(pa_expr)),
        ("projection_count", C.c_int32),
        ("projections", C.POINTER(pa_expr)),
        ("output_mem", C.c_int32),
        ("stream", C.c_void_p),
        ("min_output_page_bytes", C.c_int64),
        ("min_output_page_rows", C.c_int32),
        ("max_output_page_bytes", C.c_int32),
        ("output_handover", C.c_int32),
        ("reserved", C.c_int32),
    ]


ASC_NULLS_FIRST, ASC_NULLS_LAST, DESC_NULLS_FIRST, DESC_NULLS_LAST = 0, 1, 2, 3  # io.trino.spi.connector.SortOrder


class pa_topn_desc(C.Structure):
    _fields_ = [
        ("input_channel_count", C.c_int32),
        ("input_types", C.POINTER(C.c_int32)),
        ("n", C.c_int32),
        ("sort_channel_count", C.c_int32),
        ("sort_channels", C.POINTER(C.c_int32)),
        ("sort_orders", C.POINTER(C.c_int32)),
        ("output_mem", C.c_int32),
        ("stream", C.c_void_p),
    ]


class pa_order_by_desc(C.Structure):
    _fields_ = [
        ("input_channel_count", C.c_int32),
        ("input_types", C.POINTER(C.c_int32)),
        ("output_channel_count", C.c_int32),
        ("output_channels", C.POINTER(C.c_int32)),
        ("sort_channel_count", C.c_int32),
        ("sort_channels", C.POINTER(C.c_int32)),
        ("sort_orders", C.POINTER(C.c_int32)),
        ("output_mem", C.c_int32),
        ("stream", C.c_void_p),
    ]


class pa_dynamic_filter_source_desc(C.Structure):
    _fields_ = [
        ("input_channel_count", C.c_int32),
        ("input_types", C.POINTER(C.c_int32)),
        ("filter_channel_count", C.c_int32),
        ("filter_channels", C.POINTER(C.c_int32)),
        ("max_distinct_values", C.c_int32),
        ("min_max_collection_limit", C.c_int32),
        ("max_filter_size_bytes", C.c_int64),
        ("stream", C.c_void_p),
    ]


DOMAIN_ALL, DOMAIN_NONE, DOMAIN_VALUES, DOMAIN_RANGE = 0, 1, 2, 3


class pa_domain(C.Structure):
    _fields_ = [("kind", C.c_int32), ("value_count", C.c_int32), ("values", pa_column)]


class pa_aggregation_desc(C.Structure):
    _fields_ = [
        ("input_channel_count", C.c_int32),
        ("input_types", C.POINTER(C.c_int32)),
        ("aggregate_count", C.c_int32),
        ("aggregates", C.POINTER(pa_aggregate)),
        ("step", C.c_int32),
        ("output_mem", C.c_int32),
        ("stream", C.c_void_p),
        ("state_format", C.c_int32),
        ("reserved", C.c_int32),
    ]


class pa_hash_aggregation_desc(C.Structure):
    _fields_ = [
        ("input_channel_count", C.c_int32),
        ("input_types", C.POINTER(C.c_int32)),
        ("input_type_params", C.POINTER(C.c_int32)),
        ("group_by_count", C.c_int32),
        ("group_by_channels", C.POINTER(C.c_int32)),
        ("hash_channel", C.c_int32),
        ("step", C.c_int32),
        ("aggregate_count", C.c_int32),
        ("aggregates", C.POINTER(pa_aggregate)),
        ("expected_groups", C.c_int32),
        ("output_mem", C.c_int32),
        ("stream", C.c_void_p),
        ("max_partial_memory", C.c_int64),
        ("state_format", C.c_int32),
        ("produce_default_output", C.c_int32),
        ("group_id_channel", C.c_int32),
        ("global_aggregation_group_id_count", C.c_int32),
        ("global_aggregation_group_ids", C.POINTER(C.c_int32)),
    ]


STATES_FLAT, STATES_REFERENCE = 0, 1


class pa_fused_aggregation_desc(C.Structure):
    _fields_ = [
        ("filter_project", pa_filter_project_desc),
        ("aggregation", pa_hash_aggregation_desc),
    ]


class pa_hash_builder_desc(C.Structure):
    _fields_ = [
        ("input_channel_count", C.c_int32),
        ("input_types", C.POINTER(C.c_int32)),
        ("join_channel_count", C.c_int32),
        ("join_channels", C.POINTER(C.c_int32)),
        ("hash_channel", C.c_int32),
        ("output_channel_count", C.c_int32),
        ("output_channels", C.POINTER(C.c_int32)),
        ("expected_positions", C.c_int32),
        ("stream", C.c_void_p),
    ]


class pa_lookup_join_desc(C.Structure):
    _fields_ = [
        ("probe_channel_count", C.c_int32),
        ("probe_types", C.POINTER(C.c_int32)),
        ("join_channel_count", C.c_int32),
        ("probe_join_channels", C.POINTER(C.c_int32)),
        ("probe_hash_channel", C.c_int32),
        ("probe_output_channel_count", C.c_int32),
        ("probe_output_channels", C.POINTER(C.c_int32)),
        ("output_mem", C.c_int32),
        ("stream", C.c_void_p),
        ("join_type", C.c_int32),
        ("output_single_match", C.c_int32),
        ("filter", C.POINTER(pa_expr)),
    ]


class pa_fused_join_desc(C.Structure):
    _fields_ = [
        ("filter_project", pa_filter_project_desc),
        ("join", pa_lookup_join_desc),
    ]


class pa_fused_join_aggregation_desc(C.Structure):
    _fields_ = [
        ("filter_project", pa_filter_project_desc),
        ("join", pa_lookup_join_desc),
        ("aggregation", pa_hash_aggregation_desc),
    ]


class pa_exchange_desc(C.Structure):
    _fields_ = [
        ("channel_count", C.c_int32),
        ("types", C.POINTER(C.c_int32)),
        ("partition_channel_count", C.c_int32),
        ("partition_channels", C.POINTER(C.c_int32)),
        ("hash_channel", C.c_int32),
        ("partition_rule", C.c_int32),
        ("sink_count", C.c_int32),
        ("reserved", C.c_int32),
    ]


NEXT_PAGE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(pa_page))
LOAD_BLOCK = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_int32, C.POINTER(pa_column))
CLOSE_SOURCE = C.CFUNCTYPE(None, C.c_void_p)


class pa_page_source(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("next_page", NEXT_PAGE), ("load_block", LOAD_BLOCK), ("close", CLOSE_SOURCE)]


COMM_ID_BYTES = 128
ALL_GATHER_I64 = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32)
ALL_TO_ALL_V = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                           C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class pa_host_transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("all_gather_i64", ALL_GATHER_I64), ("all_to_all_v", ALL_TO_ALL_V)]


COMM_SUM, COMM_MIN, COMM_MAX = 0, 1, 2

JOIN_INNER, JOIN_PROBE_OUTER, JOIN_LOOKUP_OUTER, JOIN_FULL_OUTER = 0, 1, 2, 3  # LookupJoinOperators.JoinType


def int32_array(values):
    values = list(values)
    return (C.c_int32 * max(len(values), 1))(*values)
