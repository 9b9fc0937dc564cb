/*
 * presto_amd.h -- C ABI of the MI355X-native page-processing library (libpresto_amd.so).
 *
 * This is the drop-in boundary for Trino's vectorized page-processing hot path
 * (SURVEY.md section 8b).  Every entry point is extern "C", takes plain pointers and
 * sizes, returns an int32 status (0 / positive = OK, negative = pa_status error class)
 * and never lets a C++ exception cross.  A JNI shim binds one `native` method per
 * export; tests bind the same symbols through ctypes.
 *
 * Reference interfaces replaced (all paths under /root/reference/core):
 *   Operator protocol           trino-main/src/main/java/io/trino/operator/Operator.java:21-103
 *   OperatorFactory             trino-main/src/main/java/io/trino/operator/OperatorFactory.java:18-50
 *   Page / Block data model     trino-spi/src/main/java/io/trino/spi/Page.java:33-398,
 *                               trino-spi/src/main/java/io/trino/spi/block/LongArrayBlock.java:32-130,
 *                               .../IntArrayBlock.java, .../ByteArrayBlock.java,
 *                               .../VariableWidthBlock.java:34-110, .../DictionaryBlock.java,
 *                               .../RunLengthEncodedBlock.java
 *   RowExpression tree          trino-main/src/main/java/io/trino/sql/relational/{CallExpression,
 *                               ConstantExpression,InputReferenceExpression,SpecialForm}.java
 */
#ifndef PRESTO_AMD_H
#define PRESTO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_ABI_VERSION 10

/* ---- status codes (negative = error).  Mapped by the JNI shim onto TrinoException
 *      StandardErrorCode (trino-spi/.../StandardErrorCode.java). ---- */
typedef enum pa_status {
    PA_OK = 0,
    PA_ERR_INVALID_ARGUMENT = -1,        /* IllegalArgumentException / checkArgument */
    PA_ERR_ILLEGAL_STATE = -2,           /* checkState: addInput when !needsInput, after finish ... */
    PA_ERR_NOT_SUPPORTED = -3,           /* expression / type outside the device subset: caller falls back to Java */
    PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE = -4, /* BigintOperators.java:47-55 (Math.addExact etc.) */
    PA_ERR_DIVISION_BY_ZERO = -5,        /* BigintOperators.java:88-110 */
    PA_ERR_INSUFFICIENT_RESOURCES = -6,  /* BigintGroupByHash.java:264-267 (table > 2^30) / HBM exhausted */
    PA_ERR_DEVICE = -7,                  /* HIP / RCCL runtime error */
    PA_ERR_COMPILER = -8,                /* COMPILER_ERROR: device code generation failed */
    PA_ERR_NO_DEVICE = -9                /* no gfx950 device / HIP extension unusable: fail loudly, never fall back */
} pa_status;

/* ---- SQL types on the path (physical layout follows SURVEY.md section 8 preamble) ---- */
typedef enum pa_type {
    PA_BIGINT = 0,   /* LongArrayBlock, 8 B */
    PA_INTEGER = 1,  /* IntArrayBlock, 4 B */
    PA_DATE = 2,     /* IntArrayBlock, 4 B, days since 1970-01-01 */
    PA_DOUBLE = 3,   /* LongArrayBlock holding doubleToLongBits, 8 B */
    PA_BOOLEAN = 4,  /* ByteArrayBlock, 1 B */
    PA_VARCHAR = 5,  /* VariableWidthBlock: bytes + int32 offsets[positionCount+1] */
    PA_ROW = 6,      /* RowBlock of the fields' types (only as the intermediate state of an aggregate, PA_STATES_REFERENCE) */
    PA_REAL = 7,     /* IntArrayBlock holding floatToRawIntBits, 4 B (RealType.java).  Expressions (arithmetic, comparisons, casts
                      * from / to DOUBLE and from the integer types), sum / avg / min / max / count inputs, payload channels, and
                      * group / join / sort / partition keys with RealType's operators (hash of floatToIntBits with +0 for both zeros,
                      * == for joins, IS NOT DISTINCT for groups, Float.compare order; RealType.java:101-160).  Not a dynamic-filter
                      * channel and not an aggregate input in PA_STATES_REFERENCE (PA_ERR_NOT_SUPPORTED) */
    PA_DECIMAL = 8,  /* ShortDecimalType (core/trino-spi/.../type/ShortDecimalType.java): DECIMAL(p, s), p <= 18 -- a LongArrayBlock of
                      * unscaled values, 8 B.  Precision and scale travel as the channel's type parameter PA_DECIMAL_PARAM(p, s)
                      * (pa_*_desc.input_type_params) and, for expression nodes, in pa_expr_node.type_param.  Expressions: + - *
                      * (DecimalOperators.java: operands rescaled to the result scale the planner derived; a result of more than 18
                      * digits is a PA_LONG_DECIMAL value), comparisons / BETWEEN / IN between operands of one type, negation, CAST
                      * from INTEGER / BIGINT / a decimal of smaller scale.  Aggregates: sum (-> DECIMAL(38, s), DecimalSumAggregation),
                      * avg (-> the input type, DecimalAverageAggregation: sum / count rounded half up), min / max / count.  Group key
                      * of an aggregation: yes (ShortDecimalType: == of longs; $hashvalue = the value itself).  Join / sort keys and
                      * payload channels of joins, sorts and exchanges: declare the channel PA_BIGINT -- the same LongArrayBlock, the same
                      * == and < -- ; as the PARTITION key of an exchange it is not supported (its hash differs from BIGINT's) */
    PA_LONG_DECIMAL = 9 /* LongDecimalType (.../type/LongDecimalType.java, UnscaledDecimal128Arithmetic.java): DECIMAL(p, s), 18 < p <= 38 --
                      * 16 B per position, little endian: the low 64 bits of the magnitude, then the high 63 bits with the SIGN in the
                      * top bit (sign-magnitude, as the reference's Slice holds it).  On the device path: a value inside expressions
                      * (products and sums of short decimals), the result of sum / avg over decimals, the sum half of their PARTIAL
                      * state, an input of sum / avg / count, an operand of comparisons with its own type.  Not a key, not a min / max
                      * input, not divided or rescaled down (PA_ERR_NOT_SUPPORTED).  Overflow: the reference's
                      * LongDecimalWithOverflow(AndLong)State carries an overflow counter through Step.PARTIAL, so a partial sum may pass
                      * 10^38 and come back (and avg over such sums is legal, DecimalAverageAggregation.average); the device state has no
                      * such counter and raises NUMERIC_VALUE_OUT_OF_RANGE as soon as a partial sum leaves +-10^38: the planner keeps
                      * sum / avg over DECIMAL(38, s) inputs whose partial sums can get there on the reference operator */
} pa_type;
#define PA_DECIMAL_PARAM(precision, scale) (((precision) << 8) | (scale))
#define PA_DECIMAL_PRECISION(param) (((param) >> 8) & 0xff)
#define PA_DECIMAL_SCALE(param) ((param) & 0xff)

typedef enum pa_encoding {
    PA_FLAT = 0,        /* values[] (+array_offset), optional nulls[] */
    PA_VARWIDTH = 1,    /* values = bytes, offsets[positionCount+1] (already shifted by array_offset) */
    PA_DICTIONARY = 2,  /* ids[positionCount] into *dictionary */
    PA_RLE = 3,         /* *dictionary holds exactly one position, repeated positionCount times */
    PA_ROW_FIELDS = 4   /* RowBlock (core/trino-spi/.../block/RowBlock.java): `dictionary` points at an ARRAY of dictionary_size
                         * field blocks of positionCount positions each, `nulls` is rowIsNull (fields of a NULL row hold any
                         * value); type == PA_ROW */
} pa_encoding;

typedef enum pa_mem {
    PA_MEM_HOST = 0,    /* pointers are host addresses (JVM-copied / pinned staging) */
    PA_MEM_DEVICE = 1   /* pointers are HBM addresses on the operator's device */
} pa_mem;

/* One Block.  nulls is Trino's boolean[] valueIsNull: 1 byte per position, non-zero = NULL,
 * may be NULL pointer when the block has no nulls (LongArrayBlock.java:37-41). */
typedef struct pa_column {
    int32_t type;                       /* pa_type */
    int32_t encoding;                   /* pa_encoding */
    const void* values;                 /* FLAT: element array; VARWIDTH: byte array */
    const int32_t* offsets;             /* VARWIDTH only */
    const uint8_t* nulls;               /* optional */
    const int32_t* ids;                 /* DICTIONARY only */
    const struct pa_column* dictionary; /* DICTIONARY / RLE */
    int32_t dictionary_size;            /* positions in *dictionary */
    int32_t reserved;
} pa_column;

/* One Page: Block[] + positionCount (Page.java:33-60). */
typedef enum pa_page_flags {
    /* The page's buffers stay valid and unchanged until the operator it is given to is closed -- the contract of the
     * reference, where a Page is immutable and kept alive by whoever references it (SURVEY 8b "Ownership"): pages over pinned
     * / HBM buffers the host keeps for the query, row ranges of a resident table.  NOT set on the pages a device operator
     * returns from pa_op_get_output (valid until the next call on that operator).  An operator may defer its work on a
     * stable page past the return of pa_op_add_input: consecutive small stable pages that continue each other in memory
     * are processed as one range, without a copy. */
    PA_PAGE_STABLE = 1,
    /* PA_MEM_HOST pages only: every buffer of the page is pinned host memory (pa_host_malloc_pinned, or registered with the
     * HIP runtime), so the device can read it directly: small pages are then gathered by one copy kernel instead of one
     * hipMemcpyAsync per block array -- and, when the page is also PA_PAGE_STABLE, by one launch per few thousand pages. */
    PA_PAGE_PINNED = 2,
    /* The page's owner keeps its buffers valid and unchanged until the operator calls page.release(page.release_ctx) -- the C-ABI form of
     * what a reference does for a Java Page (immutable, alive while referenced): a JNI staging slot of PinnedPagePool, the output
     * buffers of an upstream device operator that hands them over.  The operator calls release exactly once per page, on the
     * thread of a later call on the same handle (add_input / needs_input / finish / get_output) or of its close, when nothing
     * reads the page any more; an operator that does not hold on to pages is given the page as a plain one and the library calls
     * release before pa_op_add_input returns.  Until its release such a page is treated like a PA_PAGE_STABLE one: small pages
     * are merged into ranges, listed in range tables or copied by ONE gather launch per few thousand pages -- no launch and no wait
     * per page.  release must not call back into the library. */
    PA_PAGE_RETAINED = 4
} pa_page_flags;
typedef struct pa_page {
    int32_t position_count;
    int32_t channel_count;
    pa_column* columns;
    int32_t mem;                        /* pa_mem: where every pointer of every column lives */
    int32_t flags;                      /* pa_page_flags */
    void (*release)(void* ctx);         /* PA_PAGE_RETAINED only */
    void* release_ctx;
} pa_page;

/* ---- RowExpression tree, flattened (TM/sql/relational package) ---- */
typedef enum pa_expr_kind {
    PA_EXPR_INPUT_REF = 0,   /* InputReferenceExpression(field) */
    PA_EXPR_CONSTANT = 1,    /* ConstantExpression(value, type) */
    PA_EXPR_CALL = 2,        /* CallExpression(resolvedFunction, args) */
    PA_EXPR_SPECIAL = 3      /* SpecialForm(form, args)  (SpecialForm.java:137-152) */
} pa_expr_kind;

typedef enum pa_call_op {    /* operator / function of a CALL node; operand types from children */
    PA_OP_ADD = 0, PA_OP_SUBTRACT = 1, PA_OP_MULTIPLY = 2, PA_OP_DIVIDE = 3, PA_OP_MODULUS = 4,
    PA_OP_NEGATE = 5,
    PA_OP_EQUAL = 6, PA_OP_NOT_EQUAL = 7, PA_OP_LESS_THAN = 8, PA_OP_LESS_THAN_OR_EQUAL = 9,
    PA_OP_GREATER_THAN = 10, PA_OP_GREATER_THAN_OR_EQUAL = 11,
    PA_OP_NOT = 12,          /* $not */
    PA_OP_CAST = 13          /* to node type; INTEGER/DATE->BIGINT, BIGINT/INTEGER->DOUBLE */
} pa_call_op;

typedef enum pa_special_form {
    PA_FORM_AND = 0, PA_FORM_OR = 1, PA_FORM_BETWEEN = 2, PA_FORM_IS_NULL = 3,
    PA_FORM_IF = 4, PA_FORM_COALESCE = 5, PA_FORM_IN = 6
} pa_special_form;

typedef struct pa_expr_node {
    int32_t kind;       /* pa_expr_kind */
    int32_t op;         /* pa_call_op or pa_special_form */
    int32_t type;       /* result pa_type */
    int32_t channel;    /* INPUT_REF: page channel */
    int32_t is_null;    /* CONSTANT: typed NULL */
    int32_t nargs;
    int32_t first_arg;  /* index of first child id in pa_expr.args */
    union {
        int32_t str_len;    /* CONSTANT VARCHAR */
        int32_t type_param; /* nodes of type PA_DECIMAL / PA_LONG_DECIMAL: PA_DECIMAL_PARAM(precision, scale) of the node's type */
    };
    int64_t i64;        /* CONSTANT BIGINT/INTEGER/DATE/BOOLEAN; CONSTANT PA_DECIMAL: the unscaled value; CONSTANT PA_LONG_DECIMAL: its low
                         * 64 bits (two's complement), the high 64 bits in the bits of f64 */
    double f64;          /* CONSTANT of type DOUBLE, or REAL (the float value, widened) */
    const char* str;    /* CONSTANT VARCHAR bytes (not NUL terminated) */
} pa_expr_node;

typedef struct pa_expr {
    int32_t node_count;
    int32_t root;
    const pa_expr_node* nodes;
    int32_t arg_count;
    int32_t reserved;
    const int32_t* args;
} pa_expr;

/* ---- aggregate functions (TM/operator/aggregation package) ---- */
typedef enum pa_agg_fn {
    PA_AGG_COUNT_STAR = 0,   /* CountAggregation.java:33-55 */
    PA_AGG_COUNT = 1,        /* count(x): non-null inputs */
    PA_AGG_SUM = 2,          /* DoubleSumAggregation / LongSumAggregation by input type */
    PA_AGG_AVG = 3,          /* AverageAggregations.java:34-80 */
    PA_AGG_MIN = 4,
    PA_AGG_MAX = 5
} pa_agg_fn;

typedef struct pa_aggregate {
    int32_t fn;              /* pa_agg_fn */
    int32_t input_channel;   /* -1 for count(*) */
    int32_t mask_channel;    /* BOOLEAN channel or -1 (CompilerOperations.java:65-74 semantics) */
    int32_t input_type;      /* pa_type of the input channel */
} pa_aggregate;

typedef enum pa_agg_step { PA_STEP_SINGLE = 0, PA_STEP_PARTIAL = 1, PA_STEP_FINAL = 2 } pa_agg_step;
/* How intermediate states cross the boundary (see the comment above pa_aggregation_desc). */
typedef enum pa_state_format {
    PA_STATES_FLAT = 0,      /* plain channels: [count] / [count, sum] / [count, value] */
    PA_STATES_REFERENCE = 1  /* one channel per aggregate, typed as the reference's AccumulatorStateSerializer types it */
} pa_state_format;

/* ---- operator descriptors (what the planner hands to an OperatorFactory) ---- */

/* FilterAndProjectOperator.createOperatorFactory (FilterAndProjectOperator.java:73-178):
 * PageProcessor(Optional<PageFilter>, List<PageProjection>).  Projections are expressions over the
 * INPUT page channels; an INPUT_REF root is InputPageProjection (identity/gather). */
typedef struct pa_filter_project_desc {
    int32_t input_channel_count;
    const int32_t* input_types;          /* pa_type per input channel */
    const int32_t* input_type_params;    /* per channel: VarcharType.getLength() bound, 0 = unbounded; NULL = all 0 */
    const pa_expr* filter;               /* NULL = no filter */
    int32_t projection_count;
    const pa_expr* projections;
    int32_t output_mem;                  /* pa_mem of pages returned by get_output */
    void* stream;                        /* hipStream_t to launch on; NULL = library-owned stream */
    /* MergePages behind the PageProcessor (FilterAndProjectOperator.java:60-66, MergePages.java:112-172): an output
     * page with fewer than min_output_page_rows rows AND fewer than min_output_page_bytes bytes (Block.getSizeInBytes
     * accounting: value width + 1 per position, VARCHAR bytes + 5 per position) is buffered and emitted merged once the
     * buffer reaches max_output_page_bytes (0 = PageBuilderStatus.DEFAULT_MAX_PAGE_SIZE_IN_BYTES, 1 MiB), a big page
     * arrives, or the input ends.  Both thresholds 0 = pass every page through (the reference's tests). */
    int64_t min_output_page_bytes;
    int32_t min_output_page_rows;
    int32_t max_output_page_bytes;
    /* Non-zero (PA_MEM_DEVICE output only): an output page whose blocks are all the operator's own buffers (no zero-copy view of the
     * input) is handed over -- flagged PA_PAGE_RETAINED, its release frees the buffers -- instead of being lent until the operator's next
     * call: the consumer may keep reading it (a HashBuilder reads a retained build page in place: the join's build side is not copied);
     * the operator takes fresh buffers for its next page.  Whoever takes such a page owes it exactly one release call. */
    int32_t output_handover;
    int32_t reserved;
} pa_filter_project_desc;

/* Intermediate states of Step.PARTIAL / Step.FINAL (AggregationNode.Step): the reference serialises LongState /
 * LongDoubleState / LongLongState rows; here they are flattened into plain channels, per aggregate
 *   count(*), count(x)  ->  [count BIGINT]
 *   sum(x), avg(x)      ->  [count BIGINT, sum DOUBLE]   (sum BIGINT for sum over BIGINT/INTEGER)
 *   min(x), max(x)      ->  [count BIGINT, value of x's type, NULL while count = 0]   (BIGINT / INTEGER / DATE / DOUBLE / REAL / BOOLEAN /
 *                           short DECIMAL / VARCHAR: the string itself)
 * PARTIAL emits keys, ($hashvalue), then these channels; FINAL takes them as input: pa_aggregate.input_channel names
 * the aggregate's count channel (the sum channel follows it), input_type the type of the sum, and it combines with the
 * @CombineFunction of the aggregate (DoubleSumAggregation.java:47-52 etc.).
 *
 * pa_hash_aggregation_desc.state_format = PA_STATES_REFERENCE switches both ends to the reference's own intermediate types, ONE
 * channel per aggregate, so that a GPU PARTIAL step can feed a Java FINAL step across a real exchange and the reverse.  The
 * types are what the generated AccumulatorStateSerializers produce (StateCompiler.java:127-185: one field -> that field's
 * type; several -> an anonymous ROW of the fields sorted by name, :586-625):
 *   count(*), count(x)     LongState                      BIGINT
 *   sum(DOUBLE)            LongDoubleState (+ TwoNullableValueState)   ROW(first BIGINT = count, firstNull BOOLEAN, second DOUBLE = sum, secondNull BOOLEAN)
 *   sum(BIGINT / INTEGER)  LongLongState   (+ TwoNullableValueState)   ROW(first BIGINT = count, firstNull BOOLEAN, second BIGINT = sum, secondNull BOOLEAN)
 *                          (DoubleSumAggregation / LongSumAggregation never touch the two null flags: they keep their initial
 *                          value true, and are ignored on input)
 *   avg(x)                 LongAndDoubleState             ROW(double DOUBLE = sum, long BIGINT = count)
 *   min(x), max(x)         NullableLongState / NullableDoubleState / NullableBooleanState with their hand-written serializers:
 *                          BIGINT (also for INTEGER and DATE inputs: their Java type is long) / DOUBLE / BOOLEAN, NULL = no value
 * With this format a FINAL step's pa_aggregate.input_channel names the aggregate's single state channel. */

/* AggregationOperator (AggregationOperator.java:40-140): global aggregates. */
typedef struct pa_aggregation_desc {
    int32_t input_channel_count;
    const int32_t* input_types;
    int32_t aggregate_count;
    const pa_aggregate* aggregates;
    int32_t step;                        /* pa_agg_step */
    int32_t output_mem;
    void* stream;
    int32_t state_format;                /* a pa_state_format value: PARTIAL output / FINAL input */
    int32_t reserved;
} pa_aggregation_desc;

/* HashAggregationOperatorFactory (HashAggregationOperator.java:120-202). */
typedef struct pa_hash_aggregation_desc {
    int32_t input_channel_count;
    const int32_t* input_types;
    const int32_t* input_type_params;    /* as in pa_filter_project_desc; may be NULL */
    int32_t group_by_count;
    const int32_t* group_by_channels;
    int32_t hash_channel;                /* precomputed $hashvalue channel or -1 */
    int32_t step;                        /* pa_agg_step */
    int32_t aggregate_count;
    const pa_aggregate* aggregates;
    int32_t expected_groups;
    int32_t output_mem;
    void* stream;
    /* Step.PARTIAL only: maxPartialMemory (HashAggregationOperatorFactory, HashAggregationOperator.java:120-202; 16 MB by
     * default in the reference).  Once the aggregation holds more than this the operator is "full"
     * (InMemoryHashAggregationBuilder.updateIsFull, :208-215): needs_input turns false, get_output emits the partial result
     * and the operator starts over with an empty table (HashAggregationOperator.java:372, 431, 501) -- a partial operator may
     * emit a key more than once.  0 = never flush early.  The size counted is the groups' key and state bytes; the reference
     * counts GroupByHash.getEstimatedSize(), a JVM-layout figure. */
    int64_t max_partial_memory;
    /* pa_state_format of the intermediate states a PARTIAL step emits / a FINAL step takes */
    int32_t state_format;
    /* GROUPING SETS with global grouping sets (HashAggregationOperatorFactory's produceDefaultOutput, globalAggregationGroupIds and
     * groupIdChannel, HashAggregationOperator.java:120-202): when the operator finishes without having been given a page it emits one
     * row per id of global_aggregation_group_ids -- NULL in every group-by column except the one at index group_id_channel AMONG THE
     * GROUP-BY COLUMNS (a BIGINT: the id), that row's $hashvalue when hash_channel >= 0 (calculateDefaultOutputHash, :589-600), and
     * every aggregate's output over no input (count 0, sum / avg / min / max NULL; a PARTIAL step: the empty intermediate states) --
     * HashAggregationOperator.java:486-492, 545-587.  produce_default_output = 0: such an operator emits nothing. */
    int32_t produce_default_output;
    int32_t group_id_channel;                    /* only read when global_aggregation_group_id_count > 0 */
    int32_t global_aggregation_group_id_count;
    const int32_t* global_aggregation_group_ids;
} pa_hash_aggregation_desc;

/* Fused pipeline: [Scan]FilterAndProject -> (Hash)AggregationOperator collapsed into one device
 * pass.  Semantically the composition of the two descriptors: the aggregation's channels index the
 * projection outputs.  group_by_count == 0 selects the global AggregationOperator form (Q6);
 * otherwise HashAggregationOperator (Q1). */
typedef struct pa_fused_aggregation_desc {
    pa_filter_project_desc filter_project;
    pa_hash_aggregation_desc aggregation;   /* input_* fields describe the projected page */
} pa_fused_aggregation_desc;

/* HashBuilderOperator + LookupJoinOperator (join/HashBuilderOperator.java:56-330,
 * join/LookupJoinOperatorFactory.java; OperatorFactories.innerJoin, OperatorFactories.java:27-45). */
typedef struct pa_hash_builder_desc {
    int32_t input_channel_count;
    const int32_t* input_types;
    int32_t join_channel_count;
    const int32_t* join_channels;
    int32_t hash_channel;                /* precomputed hash channel or -1 */
    int32_t output_channel_count;
    const int32_t* output_channels;      /* build columns copied to the join output */
    int32_t expected_positions;
    void* stream;
} pa_hash_builder_desc;

/* The four join methods of OperatorFactories (OperatorFactories.java:27-84; LookupJoinOperators.JoinType). */
typedef enum pa_join_type {
    PA_JOIN_INNER = 0,
    PA_JOIN_PROBE_OUTER = 1,    /* LEFT:  a probe row without a match comes out once, build channels NULL (DefaultPageJoiner.java:296-303) */
    PA_JOIN_LOOKUP_OUTER = 2,   /* RIGHT: build rows never appended to an output are emitted by pa_lookup_outer_create's operator */
    PA_JOIN_FULL_OUTER = 3      /* both */
} pa_join_type;

typedef struct pa_lookup_join_desc {
    int32_t probe_channel_count;
    const int32_t* probe_types;
    int32_t join_channel_count;
    const int32_t* probe_join_channels;
    int32_t probe_hash_channel;          /* or -1 */
    int32_t probe_output_channel_count;
    const int32_t* probe_output_channels;
    int32_t output_mem;
    void* stream;
    int32_t join_type;                   /* pa_join_type */
    int32_t output_single_match;         /* LookupJoinOperatorFactory.outputSingleMatch (DefaultPageJoiner.java:276-278): at
                                          * most one output row per probe row -- the first position of its chain */
    /* JoinFilterFunction (JoinFilterFunctionCompiler.java; JoinHash.isJoinPositionEligible, JoinHash.java:116-120): a BOOLEAN
     * expression over one (build row, probe row) pair, or NULL.  Its channels are numbered as the compiled filter numbers its
     * blocks: [0, B) = the channels of the build page (pa_hash_builder_desc.input_types), [B, B + probe_channel_count) = the probe
     * page's.  A position of a probe row's chain is joined only if the filter is TRUE for it (NULL counts as false); a probe row
     * none of whose positions is eligible is unmatched (probe-outer joins emit it NULL-extended; lookup-outer joins do not mark
     * the build rows visited); output_single_match keeps the first ELIGIBLE position (DefaultPageJoiner.java:266-292). */
    const pa_expr* filter;
} pa_lookup_join_desc;

/* Fused pipeline: [Scan]FilterAndProject -> LookupJoinOperator -> (Hash)AggregationOperator, the probe side of a join whose
 * output is only ever aggregated (TPC-H Q3's lineitem pipeline; LocalExecutionPlanner chains exactly these three operator
 * factories in one Driver).  Semantically the composition of the three descriptors: the join's probe page is the projection
 * output (join.probe_* index the projections), the aggregation's input page is the join's output page = [probe output channels,
 * build output channels] (LookupJoinPageBuilder.java:76-139).  When the lookup source has one BIGINT / INTEGER / DATE key and no
 * duplicate keys (known once the build side has finished), filter, probe (JoinProbe.java:87-117, DefaultPageJoiner.java:236-320)
 * and accumulation run as ONE generated kernel: no compacted probe page, no position lists, no join output page; if in
 * addition every group key is the join key or a build column (and the join key is among them), the group IS the build row and
 * the accumulators are indexed by build position.  In every other case the operator runs the three device operators behind
 * each other internally -- same results either way (summation order of DOUBLE sums aside, as for any device aggregation).
 * join.join_type must be PA_JOIN_INNER; join.output_mem and filter_project.output_mem are ignored. */
typedef struct pa_fused_join_aggregation_desc {
    pa_filter_project_desc filter_project;
    pa_lookup_join_desc join;
    pa_hash_aggregation_desc aggregation;   /* input_* fields describe the join's output page */
} pa_fused_join_aggregation_desc;

/* Fused pipeline: [Scan]FilterAndProject -> LookupJoinOperator, the probe side of a join whose output feeds further operators (TPC-H
 * Q3's orders pipeline: orders JOIN customer -> the next build side).  Semantically the composition of the two descriptors (the
 * join's probe page is the projection output); under the same condition as pa_fused_join_aggregation_desc -- one BIGINT / INTEGER /
 * DATE key, no duplicate keys on the build side -- filter and probe select the rows in one pass over the page, and the second pass
 * writes the join's output page [probe output channels, build output channels] directly: no FilterAndProject page, no position
 * lists, no gathers.  In every other case the handle runs the two device operators behind each other.  Output rows come in probe
 * order either way.  join.join_type must be PA_JOIN_INNER, join.filter NULL; filter_project.output_mem is ignored, join.output_mem
 * is where the output page lives; the MergePages thresholds of filter_project apply to the joined output. */
typedef struct pa_fused_join_desc {
    pa_filter_project_desc filter_project;
    pa_lookup_join_desc join;
} pa_fused_join_desc;

/* TopNOperator.createOperatorFactory (TopNOperator.java:43-90): keep the n best rows under (sort_channels, sort_orders)
 * and emit them, ordered, as one page after finish.  Ties between fully equal sort keys come out in arrival order (the
 * reference leaves their order unspecified). */
typedef enum pa_sort_order {             /* io.trino.spi.connector.SortOrder */
    PA_ASC_NULLS_FIRST = 0, PA_ASC_NULLS_LAST = 1, PA_DESC_NULLS_FIRST = 2, PA_DESC_NULLS_LAST = 3
} pa_sort_order;
typedef struct pa_topn_desc {
    int32_t input_channel_count;
    const int32_t* input_types;
    int32_t n;
    int32_t sort_channel_count;
    const int32_t* sort_channels;
    const int32_t* sort_orders;          /* pa_sort_order per sort channel */
    int32_t output_mem;
    void* stream;
} pa_topn_desc;

/* OrderByOperator.OrderByOperatorFactory (OrderByOperator.java:45-120): collect the input, sort it by (sort_channels,
 * sort_orders) when it ends, emit output_channels in order (one page).  Fully tied rows keep arrival order. */
typedef struct pa_order_by_desc {
    int32_t input_channel_count;
    const int32_t* input_types;
    int32_t output_channel_count;
    const int32_t* output_channels;
    int32_t sort_channel_count;
    const int32_t* sort_channels;
    const int32_t* sort_orders;          /* pa_sort_order per sort channel */
    int32_t output_mem;
    void* stream;
} pa_order_by_desc;

/* DynamicFilterSourceOperator.DynamicFilterSourceOperatorFactory (DynamicFilterSourceOperator.java:74-139): on the build
 * side of a join; passes every page through unchanged (the output page IS the input page: same buffers, which the caller
 * keeps alive until it has consumed the output) and collects per filter channel the distinct values (at most
 * max_distinct_values per channel and max_filter_size_bytes over all channels), else the min / max of the orderable,
 * non-floating-point channels while no more than min_max_collection_limit rows were seen, else nothing.  The size of a
 * set is counted as its values' bytes; the reference counts TypedSet.getRetainedSizeInBytes(), a JVM-layout figure. */
typedef struct pa_dynamic_filter_source_desc {
    int32_t input_channel_count;
    const int32_t* input_types;
    int32_t filter_channel_count;
    const int32_t* filter_channels;      /* DynamicFilterSourceOperator.Channel.index per dynamic filter */
    int32_t max_distinct_values;
    int32_t min_max_collection_limit;
    int64_t max_filter_size_bytes;
    void* stream;
} pa_dynamic_filter_source_desc;

/* One column's Domain of the TupleDomain handed to dynamicPredicateConsumer (nullAllowed is always false). */
typedef enum pa_domain_kind {
    PA_DOMAIN_ALL = 0,                   /* channel absent from the TupleDomain: unconstrained */
    PA_DOMAIN_NONE = 1,                  /* Domain.none(type): no build value (all NULL / NaN, or no rows) */
    PA_DOMAIN_VALUES = 2,                /* discrete set: value_count distinct non-NULL, non-NaN values, ascending */
    PA_DOMAIN_RANGE = 3                  /* value_count = 2: [values[0], values[1]], both inclusive */
} pa_domain_kind;
typedef struct pa_domain {
    int32_t kind;                        /* pa_domain_kind */
    int32_t value_count;
    pa_column values;                    /* host memory owned by the operator (valid until it is destroyed), no NULLs */
} pa_domain;

typedef struct pa_operator pa_operator;             /* opaque operator handle */
typedef struct pa_lookup_source pa_lookup_source;   /* opaque: JoinBridge / LookupSourceFactory */

/* ---- library lifecycle ---- */
int32_t pa_abi_version(void);
/* Binds the calling thread (and operators it creates) to HIP device `device`; -1 keeps the
 * current device.  Fails with PA_ERR_NO_DEVICE when no gfx950 device is usable. */
int32_t pa_init(int32_t device);
int32_t pa_shutdown(void);
/* Message of the last failing call on this thread ("" if none). */
const char* pa_last_error(void);
int32_t pa_device_count(void);

/* ---- device memory plumbing for hosts without their own allocator (JNI shim, tests) ---- */
int32_t pa_device_malloc(void** ptr, int64_t bytes);
int32_t pa_device_free(void* ptr);
int32_t pa_host_malloc_pinned(void** ptr, int64_t bytes);
int32_t pa_host_free_pinned(void* ptr);
int32_t pa_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream);
int32_t pa_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream);
int32_t pa_stream_synchronize(void* stream);
int32_t pa_device_synchronize(void);   /* every stream of the calling thread's device has drained */
/* The process's HBM budget for operator memory (the reference's memory pool, seen from the device): with a limit set, an
 * allocation that would exceed it fails with PA_ERR_INSUFFICIENT_RESOURCES -- except that an aggregation operator given a
 * PA_PAGE_STABLE page puts the page aside and reports pa_op_is_blocked() == 1 (needs_input 0) until other operators have
 * released enough (Operator.isBlocked on the memory future, Operator.java:69-80; HashAggregationOperator.java:435-438).
 * 0 = no limit.  pa_memory_stats: bytes held by operators, bytes cached for reuse, the limit. */
int32_t pa_memory_set_limit(int64_t bytes);
int32_t pa_memory_stats(int64_t* in_use, int64_t* cached, int64_t* limit);
/* One HIP stream per Driver: operators chained through PA_MEM_DEVICE pages must be created with the same
 * desc.stream (stream order is what makes a producer's buffer reuse safe), as a Driver runs its operators on
 * one thread (Driver.java:284,317,357). */
int32_t pa_stream_create(void** stream);
int32_t pa_stream_destroy(void* stream);

/* ConnectorPageSource as the library sees it (ScanFilterAndProjectOperator pulls pages instead of being handed them).
 * next_page: ConnectorPageSource.getNextPage -- fills *page (position count, channel count, pa_mem, columns) and returns 1,
 * or returns 0 when the source is finished (isFinished), or a negative pa_status.  A column whose `values` and `dictionary`
 * are both NULL is a LazyBlock that has not been loaded; load_block (LazyBlock.getLoadedBlock) fills it for one channel of
 * the page returned last, returning >= 0 or a negative pa_status.  The page and every loaded block stay valid until the next
 * next_page / close.  close: ConnectorPageSource.close, called once. */
typedef struct pa_page_source {
    void* ctx;
    int32_t (*next_page)(void* ctx, pa_page* page);
    int64_t (*load_block)(void* ctx, int32_t channel, pa_column* column);
    void (*close)(void* ctx);
} pa_page_source;

/* ---- operator factories ---- */
int32_t pa_filter_project_create(const pa_filter_project_desc* desc, pa_operator** out);
/* ScanFilterAndProjectOperator (ScanFilterAndProjectOperator.java:67-114, 357-400): a source operator (never needs input)
 * over a page source, PageProcessor and MergePages as in pa_filter_project_create.  Lazy blocks are loaded by need
 * (PageProcessor.java:307-347): the filter's channels first; the other channels of the projections only for a page in which
 * the filter selected a position; channels no expression reads never. */
int32_t pa_scan_filter_project_create(const pa_filter_project_desc* desc, const pa_page_source* source, pa_operator** out);
/* OperatorContext accounting of a scan operator: positions pulled from the source, bytes of the blocks that were loaded
 * (Block.getSizeInBytes, ScanFilterAndProjectOperator.recordMaterializedBytes :391), blocks loaded, and blocks of the
 * projections left unloaded because no position survived the filter. */
int32_t pa_scan_stats(pa_operator* op, int64_t* processed_positions, int64_t* materialized_bytes, int64_t* blocks_loaded, int64_t* blocks_skipped);
int32_t pa_aggregation_create(const pa_aggregation_desc* desc, pa_operator** out);
int32_t pa_hash_aggregation_create(const pa_hash_aggregation_desc* desc, pa_operator** out);
int32_t pa_fused_aggregation_create(const pa_fused_aggregation_desc* desc, pa_operator** out);
/* `bridge` must already have its build operator (as for pa_lookup_join_create); pa_op_is_blocked is 1 until the build finished. */
int32_t pa_fused_join_aggregation_create(const pa_fused_join_aggregation_desc* desc, pa_lookup_source* bridge, pa_operator** out);
int32_t pa_fused_join_create(const pa_fused_join_desc* desc, pa_lookup_source* bridge, pa_operator** out);
int32_t pa_topn_create(const pa_topn_desc* desc, pa_operator** out);
int32_t pa_order_by_create(const pa_order_by_desc* desc, pa_operator** out);
/* The dynamic filter of an INNER (or lookup-outer) join applied where Trino applies it -- in the scan / filter upstream of the
 * probe (DynamicFilter.getCurrentPredicate consumed by ScanFilterAndProjectOperator's page source): rows of `op` (a
 * FilterAndProject operator that has not seen a page yet) whose BIGINT / INTEGER / DATE `channel` value equals no build key
 * of `source` are dropped together with the rows the filter expression drops.  The predicate is the build side's existence
 * bitmap (exact over its key range), not the reference's value set / min-max Domain: it only ever removes rows the join would
 * not match, so results are unchanged.  `source` must be built (its HashBuilderOperator finished).  Returns 1 when the filter
 * is active, 0 when the source offers none (several key channels, non-integer key, keys too sparse) -- the operator then
 * works as before.  Not for probe-outer / full-outer joins, whose unmatched probe rows are output. */
int32_t pa_filter_project_set_dynamic_filter(pa_operator* op, int32_t channel, pa_lookup_source* source);
/* A planner's note to a (Hash)AggregationOperator (plain, fused, or behind a fused probe; Step.SINGLE / FINAL) whose ONLY consumer is
 * a TopNOperator(n, sort channels, sort orders) over its output page (LocalExecutionPlanner.visitTopN over visitAggregation -- TPC-H
 * Q3's ORDER BY revenue DESC, o_orderdate LIMIT 10; sort_channels index the aggregation's OUTPUT channels: group keys, ($hashvalue),
 * aggregates): groups that cannot be among the TopN's n best rows MAY then be left out of the output.  What is emitted is every group
 * whose first sort channel is not beyond a bound under which at least n groups lie (ties on that channel included), in no particular
 * order -- a superset of the n best, thousands of rows where the table holds millions; the TopN downstream does the exact work.
 * Returns 1 when the operator takes the hint, 0 when it does not (it then emits everything, as without the call).  Without this call
 * nothing changes.  (The reference has no such fusion: TopNOperator.java:30-160 sees every group.) */
int32_t pa_aggregation_set_output_topn_hint(pa_operator* op, int64_t n, int32_t sort_channel_count, const int32_t* sort_channels,
                                            const int32_t* sort_orders);
/* The same across the ranks of a partitioned join, where every rank built the keys of its own partition and the probe rows
 * are filtered BEFORE they are exchanged: (1) pa_lookup_source_key_range -> [min, max] of this rank's build keys (returns 0:
 * no single integer key / no non-NULL key); the ranks agree on the union range; (2) pa_lookup_source_key_bitmap sets, in a
 * device bitmap of (range >> 6) + 1 words that it clears first, bit (key - min_key) for each of this rank's keys inside
 * [min_key, min_key + range]; the ranks OR their bitmaps (all-gather + OR: RCCL has no bitwise reduction);
 * (3) pa_filter_project_set_dynamic_filter_bitmap installs the combined bitmap, which the caller keeps alive as long as the
 * operator lives.  presto_amd/q3.py does this with torch.distributed. */
int32_t pa_lookup_source_key_range(pa_lookup_source* source, int64_t* min_key, int64_t* max_key);
/* LookupSource.getJoinPositionCount of a built source (>= 0), or a negative pa_status. */
int32_t pa_lookup_source_position_count(pa_lookup_source* source);
int32_t pa_lookup_source_key_bitmap(pa_lookup_source* source, int64_t min_key, uint64_t range, uint64_t* bits, void* stream);
int32_t pa_filter_project_set_dynamic_filter_bitmap(pa_operator* op, int32_t channel, const uint64_t* bits, int64_t min_key, uint64_t range);
int32_t pa_dynamic_filter_source_create(const pa_dynamic_filter_source_desc* desc, pa_operator** out);
/* The dynamicPredicateConsumer call of a DynamicFilterSourceOperator: returns 0 while the consumer has not been called
 * (it is called by finish(), or earlier when the operator gives up collecting), 1 once it has; then *is_all != 0 means
 * TupleDomain.all(), else domains[i] (i < filter_channel_count <= domain_capacity) is filter channel i's Domain. */
int32_t pa_dynamic_filter_poll(pa_operator* op, int32_t* is_all, pa_domain* domains, int32_t domain_capacity);
int32_t pa_lookup_source_create(pa_lookup_source** out);
int32_t pa_lookup_source_destroy(pa_lookup_source* ls);
int32_t pa_hash_builder_create(const pa_hash_builder_desc* desc, pa_lookup_source* bridge, pa_operator** out);
int32_t pa_lookup_join_create(const pa_lookup_join_desc* desc, pa_lookup_source* bridge, pa_operator** out);
/* LookupOuterOperator (join/LookupOuterOperator.java:40-215): a source operator (never needs input) that, once every probe
 * operator of the bridge is finished (the caller creates it / pulls from it after they are), emits the build rows no probe
 * row was joined with -- ascending build position, probe output channels NULL (types from desc), then the build output
 * channels (OuterLookupSource.java:120-160).  desc->join_type must be LOOKUP_OUTER or FULL_OUTER, as for the probe operators
 * of the same bridge (they mark the visited positions). */
int32_t pa_lookup_outer_create(const pa_lookup_join_desc* desc, pa_lookup_source* bridge, pa_operator** out);

/* ---- Operator protocol (Operator.java:21-103; call order Driver.java:355-457) ---- */
int32_t pa_op_needs_input(pa_operator* op);                 /* 1 / 0 */
/* The page's buffers must stay valid until the call returns -- and, for pages flagged PA_PAGE_STABLE, until the operator is closed.
 * A PA_MEM_DEVICE page without that flag (what another operator's pa_op_get_output returned) is read in stream order: an operator
 * on the caller-provided desc.stream may still have work on it enqueued there when the call returns -- the producer, on the same
 * stream, is ordered behind it --, an operator on a stream of the library's own has finished with it. */
int32_t pa_op_add_input(pa_operator* op, const pa_page* page);
/* Returns 1 and fills *out when a page is available, 0 when none.  The page's buffers are owned by
 * the operator (or, for zero-copy identity projections, are the caller's own input buffers) and stay valid
 * until the next add_input / get_output / close on the same handle.  PA_MEM_DEVICE pages: when the operator runs on a
 * caller-provided desc.stream the page is valid in that stream's order; when the library owns the stream (desc.stream ==
 * NULL) the page is complete on return. */
int32_t pa_op_get_output(pa_operator* op, pa_page* out);
int32_t pa_op_finish(pa_operator* op);
int32_t pa_op_is_finished(pa_operator* op);                 /* 1 / 0 */
int32_t pa_op_is_blocked(pa_operator* op);                  /* 1 = device work still in flight */
int64_t pa_op_memory_bytes(pa_operator* op);                /* HBM held (OperatorContext memory accounting) */
int32_t pa_op_close(pa_operator* op);
/* Device time (ms) of the kernels the operator launched since creation, measured with HIP events
 * on the operator's stream; *launches = number of timed launches of its dominant kernel. */
int32_t pa_op_kernel_time(pa_operator* op, double* total_ms, int64_t* launches);
/* Name of that dominant kernel as a kernel trace (rocprofv3 --kernel-trace) shows it: generated kernels are called
 * <entry>_<tier>_<first 8 hex digits of the code object's key>, e.g. pa_fused_lds_1f0c9a3e -- one name per plan and page layout.
 * Empty until the operator has launched it. */
int32_t pa_op_kernel_name(pa_operator* op, char* buf, int32_t buf_size);

/* (probe position, build position) pairs of the probe page a LookupJoin operator joined last, in emission
 * order (DefaultPageJoiner.java:236-320), valid after its get_output; device pointers. */
int32_t pa_lookup_join_match_pairs(pa_operator* op, const int32_t** probe_positions, const int32_t** build_positions, int32_t* count);
/* PagesHash.key[] (hash_size ints, -1 = empty) and ArrayPositionLinks.positionLinks[] (positions ints) of a built
 * lookup source; device pointers. */
int32_t pa_lookup_source_tables(pa_lookup_source* ls, const int32_t** key, int32_t* hash_size, const int32_t** position_links,
                                int32_t* positions);

/* SelectedPositions of the page a FilterAndProject operator processed last (valid after its get_output):
 * *is_list == 0 means positionsRange(0, *count) (PageFilter.java:37-39: no or every row selected); otherwise
 * *dev_positions points at *count ascending positions in HBM (PageFilter.java:41-49). */
int32_t pa_filter_project_selected_positions(pa_operator* op, const int32_t** dev_positions, int32_t* count, int32_t* is_list);

/* ---- stand-alone kernels of the path (used by exchange / tests; all device pointers) ---- */
/* rawHash per position = InterpretedHashGenerator.hashPosition over `channels`
 * (InterpretedHashGenerator.java:62-70, CombineHashFunction.java:26-29). */
int32_t pa_hash_page(const pa_page* page, int32_t channel_count, const int32_t* channels,
                     int64_t* out_raw_hash, void* stream);
/* partition = (int) XxHash64.hash(Long.reverse(rawHash)) & (count-1)   (LocalPartitionGenerator.java:45-65)
 * or (rawHash & MAX_LONG) % count when local == 0                      (HashGenerator.java:24-35). */
int32_t pa_partition_ids(const int64_t* raw_hash, int32_t position_count, int32_t partition_count,
                         int32_t local, int32_t* out_partition, void* stream);
/* Stable partition of positions by partition id: out_positions grouped by partition in ascending
 * position order inside each partition (PartitioningExchanger.java:59-82), out_counts[partition_count]. */
int32_t pa_partition_positions(const int32_t* partition, int32_t position_count, int32_t partition_count,
                               int32_t* out_positions, int64_t* out_counts_host, void* stream);
/* The rows of flat columns physically regrouped by partition id in one pass (an LDS-staged multisplit: coalesced reads AND
 * writes, where pa_partition_positions + pa_gather_flat pay a cache line per row and column): out_columns[c] holds the
 * elements of in_columns[c] (elem_bytes[c] = 1, 4 or 8) with partition 0 first, then partition 1 ...; the same permutation
 * for every column; the order of the rows INSIDE a partition is unspecified -- for consumers that do not care (the
 * hash-partitioned aggregation); an exchange that must keep the reference's row order uses pa_partition_positions.
 * All pointers are device pointers; out_counts_host[partition_count]. */
int32_t pa_partition_columns(const int32_t* partition, int32_t position_count, int32_t partition_count, const void* const* in_columns,
                             void* const* out_columns, const int32_t* elem_bytes, int32_t column_count, int64_t* out_counts_host, void* stream);
/* The same with ascending row order kept inside every partition (what PartitioningExchanger produces); at most 256 partitions.
 * The exchange regroups its flat columns with it: one pass instead of pa_partition_positions + a pa_gather_flat per column. */
int32_t pa_partition_columns_stable(const int32_t* partition, int32_t position_count, int32_t partition_count, const void* const* in_columns,
                                    void* const* out_columns, const int32_t* elem_bytes, int32_t column_count, int64_t* out_counts_host, void* stream);
/* Block.copyPositions for a flat column: dst[i] = src[positions[i]]. */
int32_t pa_gather_flat(const void* src, int32_t elem_bytes, const int32_t* positions, int32_t count,
                       void* dst, void* stream);

/* ---- hash-partitioned exchange between the ranks (one per GPU) of a node: RCCL over xGMI ----
 * Replaces, for GPU-resident pages, the reference's page shuffle between the stages of a distributed join / aggregation:
 *   producer side  PartitionedOutputOperator.PagePartitioner.partitionPage (TM/operator/PartitionedOutputOperator.java:411-431),
 *                  local twin PartitioningExchanger.accept (TM/operator/exchange/PartitioningExchanger.java:59-82);
 *   routing rule   HashGenerator.getPartition (TM/operator/HashGenerator.java:24-35) /
 *                  LocalPartitionGenerator.getPartition (TM/operator/exchange/LocalPartitionGenerator.java:45-65);
 *   consumer side  ExchangeOperator.getOutput (TM/operator/ExchangeOperator.java) fed by the ExchangeClient.
 * The reference serialises pages and pulls them over HTTP; here the rows of every destination are kept as raw column
 * segments in HBM and travel in ONE variable all-to-all (a grouped ncclSend / ncclRecv per peer) when the producing
 * pipeline of every rank has finished -- a bulk exchange sized for 288 GB of HBM: every rank takes part in exactly one
 * count all-gather and one all-to-all per exchange, whatever its page count, so the collective order is the program order
 * on every rank.  NULLs travel (1 B / row per nullable column), VARCHAR travels as per-row lengths + bytes.
 * Row order at the consumer = (source rank, source position), what draining the producers in rank order gives. */
typedef struct pa_comm pa_comm;          /* communicator: rank, world, transport */
typedef struct pa_exchange pa_exchange;  /* the OutputBuffer + ExchangeClient pair of one exchange on this rank */
#define PA_COMM_ID_BYTES 128
/* ncclGetUniqueId: called on one rank; the host ships the 128 bytes to the others (the coordinator's job in Trino). */
int32_t pa_comm_unique_id(void* id_out);
/* ncclCommInitRank on the calling thread's device (collective over the ranks). */
int32_t pa_comm_create(const void* unique_id, int32_t rank, int32_t world, pa_comm** out);
/* The same exchange with the two collectives done by the host: several ranks sharing one GPU (RCCL refuses that; the
 * multi-rank tests on a one-GPU box use it over gloo) or a host that moves the bytes itself.  Callbacks return 0 or a
 * negative pa_status; all buffers are host memory. */
typedef struct pa_host_transport {
    void* ctx;
    /* every rank contributes count int64 values, recv gets world * count in rank order */
    int32_t (*all_gather_i64)(void* ctx, const int64_t* send, int64_t* recv, int32_t count);
    /* rank p is sent send[send_offsets[p] .. + send_bytes[p]) and its bytes arrive at recv + recv_offsets[p] */
    int32_t (*all_to_all_v)(void* ctx, const void* send, const int64_t* send_offsets, const int64_t* send_bytes, void* recv,
                            const int64_t* recv_offsets, const int64_t* recv_bytes);
} pa_host_transport;
int32_t pa_comm_create_host(const pa_host_transport* transport, int32_t rank, int32_t world, pa_comm** out);
int32_t pa_comm_destroy(pa_comm* comm);
int32_t pa_comm_rank(pa_comm* comm);
int32_t pa_comm_world(pa_comm* comm);
/* Small host-side reductions between the ranks (blocking): op 0 = SUM, 1 = MIN, 2 = MAX; values in place. */
int32_t pa_comm_all_reduce_i64(pa_comm* comm, int64_t* values, int32_t count, int32_t op, void* stream);
/* Every rank contributes `count` int64 of host memory and receives world * count in rank order (blocking): the Step.PARTIAL
 * pages of a few-groups aggregation travel this way to the rank of the Step.FINAL operator (a few hundred bytes per rank). */
int32_t pa_comm_all_gather_i64(pa_comm* comm, const int64_t* send, int64_t* recv, int32_t count, void* stream);
/* Pre-flight of a fresh communicator (collective): a bytes_per_peer all-to-all, an all-reduce and an all-gather whose results
 * every rank verifies.  PA_ERR_DEVICE when a byte arrived damaged.  A host runs it under its own watchdog: a transport whose peers
 * never answer blocks inside RCCL. */
int32_t pa_comm_preflight(pa_comm* comm, int64_t bytes_per_peer, void* stream);

typedef struct pa_exchange_desc {
    int32_t channel_count;
    const int32_t* types;                /* pa_type per channel */
    int32_t partition_channel_count;
    const int32_t* partition_channels;   /* rawHash = InterpretedHashGenerator over these (ignored with hash_channel >= 0) */
    int32_t hash_channel;                /* precomputed $hashvalue BIGINT channel or -1 */
    int32_t partition_rule;              /* 0 = (rawHash & MAX_LONG) % world (remote exchange), 1 = LocalPartitionGenerator
                                          * (world must be a power of two), -1 = 1 when world is a power of two, else 0 */
    int32_t sink_count;                  /* PartitionedOutput operators that feed it (Drivers of the producing pipeline); 0 = 1 */
    int32_t reserved;
} pa_exchange_desc;
int32_t pa_exchange_create(const pa_exchange_desc* desc, pa_comm* comm, pa_exchange** out);
int32_t pa_exchange_destroy(pa_exchange* exchange);
/* The sink of the producing pipeline (Operator protocol: needs input until finish, never has output). */
int32_t pa_partitioned_output_create(pa_exchange* exchange, void* stream, pa_operator** out);
/* The source of the consuming pipeline: never needs input; is blocked while a sink of this rank is unfinished; its first
 * get_output afterwards performs the collective (every rank must get there) and returns the received rows as one page. */
int32_t pa_exchange_source_create(pa_exchange* exchange, int32_t output_mem, void* stream, pa_operator** out);
/* rows this rank sent / received, payload bytes sent to OTHER ranks, and the device time of the all-to-all (ms). */
int32_t pa_exchange_stats(pa_exchange* exchange, int64_t* rows_sent, int64_t* rows_received, int64_t* bytes_sent_remote, double* transfer_ms);
/* The dynamic filter of a partitioned join, agreed between the ranks (steps (1)-(2) of the comment above
 * pa_lookup_source_key_range, done natively): the ranks take the union key range (MIN / MAX reductions), each sets the
 * bits of its partition's build keys, and the bitmaps are combined with one SUM all-reduce (every key lives on exactly one
 * rank after a build partitioned on the join key -- partitioned_by_key != 0 -- so the set bits are disjoint and SUM == OR;
 * with partitioned_by_key == 0 the bitmaps are all-gathered and OR-ed by a kernel).  Returns 1 and the device bitmap
 * ((range >> 6) + 1 words, owned by the lookup source) / 0 when no filter is possible (no integer key, keys too sparse). */
int32_t pa_lookup_source_shared_key_bitmap(pa_lookup_source* source, pa_comm* comm, int32_t partitioned_by_key, void* stream,
                                           const uint64_t** bits, int64_t* min_key, uint64_t* range);

/* ---- page wire format (PagesSerde, uncompressed / unencrypted / no checksum) ----
 * pa_page_serialize writes the SerializedPage frame (PagesSerdeUtil.java:66-74: positionCount int, markers byte = 0,
 * uncompressedSize int, size int) and the writeRawPage payload (PagesSerdeUtil.java:45-52; LONG_ARRAY / INT_ARRAY /
 * BYTE_ARRAY / VARIABLE_WIDTH block encodings, nulls as bits) of a host or device page into host memory; returns the
 * bytes written (negative = pa_status; PA_ERR_INSUFFICIENT_RESOURCES when capacity is too small).  DOUBLE travels as
 * LONG_ARRAY and DATE as INT_ARRAY, as in the reference, so pa_page_deserialize types the blocks BIGINT / INTEGER /
 * BOOLEAN / VARCHAR: the consumer's declared column types tell DOUBLE from BIGINT and DATE from INTEGER.
 * pa_page_deserialize builds a PA_MEM_DEVICE page owned by the returned buffer (NULL positions of fixed-width blocks
 * hold 0). */
typedef struct pa_page_buffer pa_page_buffer;
int64_t pa_page_serialize(const pa_page* page, void* out_host, int64_t capacity, void* stream);
int32_t pa_page_deserialize(const void* bytes_host, int64_t size, void* stream, pa_page_buffer** out);
/* The same frame with the payload as one LZ4 block and PageCodecMarker.COMPRESSED set, when that shrinks it to <= 0.8 of its
 * size -- PagesSerde.serialize with a compressor (PagesSerde.java:74-95, MINIMUM_COMPRESSION_RATIO); otherwise the frame of
 * pa_page_serialize.  pa_page_deserialize[_typed] read both (ENCRYPTED frames: PA_ERR_NOT_SUPPORTED). */
int64_t pa_page_serialize_lz4(const pa_page* page, void* out_host, int64_t capacity, void* stream);
/* pa_page_deserialize with the consumer's declared channel types: LONG_ARRAY blocks come back as BIGINT or DOUBLE, INT_ARRAY
 * as INTEGER or DATE, as expected_types says; a block whose encoding cannot carry the declared type is refused. */
int32_t pa_page_deserialize_typed(const void* bytes_host, int64_t size, const int32_t* expected_types, int32_t channel_count, void* stream,
                                  pa_page_buffer** out);
int32_t pa_page_buffer_page(pa_page_buffer* buffer, pa_page* out);
int32_t pa_page_buffer_free(pa_page_buffer* buffer);

/* Block.copyPositions for a VariableWidthBlock, in two calls (the byte total sizes the second one's output):
 * pa_varwidth_gather_offsets: out_lengths[i] = length of row positions[i]; out_offsets[count + 1] = their exclusive scan
 * (out_offsets[count] = total); *total_bytes_host = total.  pa_varwidth_gather_bytes copies the bytes behind those offsets.
 * pa_offsets_from_lengths: exclusive scan of per-row lengths into count + 1 offsets (what a receiver of shuffled rows does). */
int32_t pa_varwidth_gather_offsets(const int32_t* offsets, const int32_t* positions, int32_t count, int32_t* out_lengths,
                                   int32_t* out_offsets, int64_t* total_bytes_host, void* stream);
int32_t pa_varwidth_gather_bytes(const void* values, const int32_t* offsets, const int32_t* positions, int32_t count,
                                 const int32_t* out_offsets, void* out_values, void* stream);
int32_t pa_offsets_from_lengths(const int32_t* lengths, int32_t count, int32_t* out_offsets, int64_t* total_bytes_host, void* stream);

/* ---- synthetic TPC-H-shaped column generator, on device ---- */
typedef enum pa_tpch_column {
    PA_L_ORDERKEY = 0, PA_L_QUANTITY = 1, PA_L_EXTENDEDPRICE = 2, PA_L_DISCOUNT = 3, PA_L_TAX = 4,
    PA_L_SHIPDATE = 5, PA_L_RETURNFLAG = 6, PA_L_LINESTATUS = 7,
    PA_O_ORDERKEY = 8, PA_O_CUSTKEY = 9, PA_O_ORDERDATE = 10, PA_O_SHIPPRIORITY = 11,
    PA_C_CUSTKEY = 12, PA_C_MKTSEGMENT = 13
} pa_tpch_column;
/* Fills rows [first_row, first_row+row_count) of the column into `values` (and `offsets`
 * [row_count+1] for VARCHAR columns).  Row r of a column depends only on (seed, column, r, scale). */
int32_t pa_tpch_generate(int32_t column, double scale_factor, int64_t first_row, int64_t row_count,
                         uint64_t seed, void* values, int32_t* offsets, void* stream);

/* ---- query-time code generation without a device (build step, CPU-side tests) ----
 * pa_codegen_*: writes the generated gfx950 translation unit of a descriptor (all-non-null, aligned column
 * layout) into buf (NUL terminated) and its cache key into key[17]; returns the needed buffer size.
 * pa_codegen_compile_*: compiles it with hiprtc; returns the code-object size.  Negative = pa_status.
 * variant: -1 default, 0 GLOBAL (no keys), 1 LDS (few groups), 2 GT (HBM table). */
int64_t pa_codegen_fused(const pa_fused_aggregation_desc* desc, int32_t variant, char* buf, int64_t buf_size, char* key);
int64_t pa_codegen_compile_fused(const pa_fused_aggregation_desc* desc, int32_t variant);
/* the same for ANY column-layout signature and every tier: nullable_channels = bit c set when input channel c carries a valueIsNull
 * array (kernels are generated per signature); variant 0 GLOBAL, 1 LDS (register key table), 2 GT (HBM table), 3 LDSH (LDS table per
 * workgroup), 4 the hash-partition pass, 5 LDSP (partition-owned LDS tables).  buf may be NULL; compile != 0 also compiles the unit with
 * hiprtc for gfx950 and returns the code-object size instead of the source size.  PA_ERR_NOT_SUPPORTED: the tier does not take the shape */
int64_t pa_codegen_fused_layout(const pa_fused_aggregation_desc* desc, int32_t variant, uint64_t nullable_channels, int32_t compile, char* buf,
                                int64_t buf_size);
/* the one-kernel form of a fused join-aggregation over a lookup source shaped as `build` describes (variant: -1 default,
 * 0 GLOBAL, 1 LDS, 2 GT, 3 LDS table per workgroup, 6 BROW = accumulators indexed by build position) */
int64_t pa_codegen_fused_join(const pa_fused_join_aggregation_desc* desc, const pa_hash_builder_desc* build, int32_t variant, char* buf, int64_t buf_size);
int64_t pa_codegen_compile_fused_join(const pa_fused_join_aggregation_desc* desc, const pa_hash_builder_desc* build, int32_t variant);
/* the FilterAndProject kernels with the probe inside: what pa_fused_join_create runs over a lookup source with unique keys */
int64_t pa_codegen_fused_join_probe(const pa_fused_join_desc* desc, const pa_hash_builder_desc* build, char* buf, int64_t buf_size);
int64_t pa_codegen_compile_fused_join_probe(const pa_fused_join_desc* desc, const pa_hash_builder_desc* build);
int64_t pa_codegen_filter_project(const pa_filter_project_desc* desc, char* buf, int64_t buf_size, char* key);
int64_t pa_codegen_compile_filter_project(const pa_filter_project_desc* desc);

#ifdef __cplusplus
}
#endif
#endif /* PRESTO_AMD_H */
