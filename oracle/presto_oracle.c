/*
 * presto_oracle.c -- CPU restatement (scalar, row-at-a-time C) of Trino 359's page-processing
 * hot path.  TEST INFRASTRUCTURE ONLY: see presto_oracle.h.  Never linked into libpresto_amd.so.
 *
 * Citations use SURVEY.md's abbreviations:
 *   TM/  = /root/reference/core/trino-main/src/main/java/io/trino/
 *   SPI/ = /root/reference/core/trino-spi/src/main/java/io/trino/spi/
 *   BM/  = /root/reference/testing/trino-benchmark/src/main/java/io/trino/benchmark/
 *
 * Compile with -ffp-contract=off: Java evaluates double *,+,- unfused (TM/type/DoubleOperators.java:59-78).
 */
#include "presto_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];
const char* orc_last_error(void) { return g_err; }
static int32_t fail(int32_t code, const char* msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
void orc_free(void* p) { free(p); }

/* =====================================================================================
 * Hash arithmetic
 * ===================================================================================== */
#define P64_1 0x9E3779B185EBCA87ULL
#define P64_2 0xC2B2AE3D27D4EB4FULL
#define P64_3 0x165667B19E3779F9ULL
#define P64_4 0x85EBCA77C2B2AE63ULL
#define P64_5 0x27D4EB2F165667C5ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xxh_round(uint64_t acc, uint64_t in) { return rotl64(acc + in * P64_2, 31) * P64_1; }
static inline uint64_t xxh_merge(uint64_t h, uint64_t v) { return (h ^ xxh_round(0, v)) * P64_1 + P64_4; }

/* io.airlift.slice.XxHash64.hash(Slice) == public XXH64 (seed 0 at every reference call site:
 * SPI/block/AbstractVariableWidthBlock.java:92-96, SPI/type/VarcharType.java:246-256). */
uint64_t orc_xxh64(const void* data, int64_t len, uint64_t seed)
{
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed, v4 = seed - P64_1;
        do {
            v1 = xxh_round(v1, rd64(p));
            v2 = xxh_round(v2, rd64(p + 8));
            v3 = xxh_round(v3, rd64(p + 16));
            v4 = xxh_round(v4, rd64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh_merge(h, v1);
        h = xxh_merge(h, v2);
        h = xxh_merge(h, v3);
        h = xxh_merge(h, v4);
    }
    else {
        h = seed + P64_5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) {
        h ^= xxh_round(0, rd64(p));
        h = rotl64(h, 27) * P64_1 + P64_4;
        p += 8;
    }
    if (p + 4 <= end) {
        h ^= (uint64_t)rd32(p) * P64_1;
        h = rotl64(h, 23) * P64_2 + P64_3;
        p += 4;
    }
    while (p < end) {
        h ^= (uint64_t)(*p) * P64_5;
        h = rotl64(h, 11) * P64_1;
        p++;
    }
    h ^= h >> 33;
    h *= P64_2;
    h ^= h >> 29;
    h *= P64_3;
    h ^= h >> 32;
    return h;
}

/* XxHash64.hash(long) == XXH64 of the value's 8 little-endian bytes
 * (call site TM/operator/exchange/LocalPartitionGenerator.java:61-65). */
int64_t orc_xxh64_long(int64_t value)
{
    uint8_t b[8];
    memcpy(b, &value, 8);
    return (int64_t)orc_xxh64(b, 8, 0);
}

/* SPI/type/AbstractLongType.java:126-130 */
int64_t orc_hash_bigint(int64_t value)
{
    return (int64_t)(rotl64((uint64_t)value * P64_2, 31) * P64_1);
}
/* SPI/type/AbstractIntType.java:141-145: AbstractLongType.hash((int) value), sign-extended */
int64_t orc_hash_integer(int32_t value) { return orc_hash_bigint((int64_t)value); }
/* SPI/type/DoubleType.java:163-170: -0.0 folded onto +0.0, then hash(doubleToLongBits) */
int64_t orc_hash_double(double value)
{
    if (value == 0) {
        value = 0;
    }
    int64_t bits;
    if (value != value) {
        bits = 0x7ff8000000000000LL; /* doubleToLongBits canonicalises NaN */
    }
    else {
        memcpy(&bits, &value, 8);
    }
    return orc_hash_bigint(bits);
}
/* SPI/type/BooleanType.java:39-40,151-155: no HASH_CODE operator, falls back to XX_HASH_64
 * (SPI/type/TypeOperators.java:228-233) = XxHash64.hash(1) / hash(0) */
int64_t orc_hash_boolean(int32_t value) { return orc_xxh64_long(value ? 1 : 0); }

/* TM/operator/join/PagesHash.java:225-241 (== fastutil HashCommon.murmurHash3(long)) */
int64_t orc_murmur3_fmix(int64_t x)
{
    uint64_t h = (uint64_t)x;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return (int64_t)h;
}
/* TM/operator/scalar/CombineHashFunction.java:26-29 */
int64_t orc_combine_hash(int64_t previous, int64_t value)
{
    return (int64_t)(31ULL * (uint64_t)previous + (uint64_t)value);
}
/* it.unimi.dsi:fastutil:8.3.0 HashCommon.arraySize(expected, f) =
 * max(2, nextPowerOfTwo(ceil(expected / f))); throws above 2^30. */
int32_t orc_array_size(int32_t expected, float fill)
{
    int64_t need = (int64_t)ceil((double)expected / (double)fill);
    int64_t s = 1;
    while (s < need) {
        s <<= 1;
    }
    if (s < 2) {
        s = 2;
    }
    if (s > (1LL << 30)) {
        return -1;
    }
    return (int32_t)s;
}

/* =====================================================================================
 * Block access
 * ===================================================================================== */
typedef struct orc_val {
    int32_t is_null;
    int32_t type;
    int64_t i;        /* BIGINT / INTEGER / DATE / BOOLEAN */
    double d;         /* DOUBLE */
    const uint8_t* s; /* VARCHAR */
    int32_t slen;
    __int128 q;       /* DECIMAL (short and long): the unscaled value */
} orc_val;

/* ---- DECIMAL: ShortDecimalType (unscaled long) and LongDecimalType (SPI/type/UnscaledDecimal128Arithmetic.java: a 16-byte Slice,
 *      little endian: the low 64 bits of the magnitude, then the high 63 bits with the sign in bit 63 of the second long) ---- */
typedef unsigned __int128 u128;
static inline int type_is_decimal(int32_t t) { return t == PA_DECIMAL || t == PA_LONG_DECIMAL; }
static inline __int128 ld_read(const void* p)
{
    uint64_t lo, hi;
    memcpy(&lo, p, 8);
    memcpy(&hi, (const uint8_t*)p + 8, 8);
    const u128 mag = ((u128)(hi & 0x7fffffffffffffffULL) << 64) | lo;
    return (hi >> 63) ? -(__int128)mag : (__int128)mag;
}
static inline void ld_write(void* p, __int128 v)
{
    const u128 mag = v < 0 ? (u128)(-v) : (u128)v;
    uint64_t lo = (uint64_t)mag, hi = (uint64_t)(mag >> 64) | (v < 0 ? 0x8000000000000000ULL : 0);
    memcpy(p, &lo, 8);
    memcpy((uint8_t*)p + 8, &hi, 8);
}
static __int128 pow10_128(int k)
{
    __int128 r = 1;
    while (k-- > 0) r *= 10;
    return r;
}
/* UnscaledDecimal128Arithmetic.throwIfOverflows: a magnitude of 10^38 or more is an overflow */
static inline int ld_overflows(__int128 v) { return v >= pow10_128(38) || v <= -pow10_128(38); }

static inline int type_is_int(int32_t t) { return t == PA_BIGINT || t == PA_INTEGER || t == PA_DATE; }
/* DOUBLE, and REAL: a REAL value travels in orc_val.d as the double it converts to exactly (RealOperators.castToDouble) */
static inline int type_is_fp(int32_t t) { return t == PA_DOUBLE || t == PA_REAL; }

/* resolve Dictionary / RLE wrappers (SPI/block/DictionaryBlock.java, RunLengthEncodedBlock.java) */
static inline const pa_column* resolve(const pa_column* c, int32_t* pos)
{
    while (c->encoding == PA_DICTIONARY || c->encoding == PA_RLE) {
        if (c->encoding == PA_DICTIONARY) {
            *pos = c->ids[*pos];
        }
        else {
            *pos = 0;
        }
        c = c->dictionary;
    }
    return c;
}

static inline int col_is_null(const pa_column* c, int32_t pos)
{
    c = resolve(c, &pos);
    return c->nulls != NULL && c->nulls[pos] != 0;
}

static inline orc_val col_get(const pa_column* c, int32_t pos)
{
    orc_val v;
    memset(&v, 0, sizeof v);
    c = resolve(c, &pos);
    v.type = c->type;
    if (c->nulls != NULL && c->nulls[pos] != 0) {
        v.is_null = 1;
        return v;
    }
    switch (c->type) {
        case PA_BIGINT:
            v.i = ((const int64_t*)c->values)[pos];
            break;
        case PA_INTEGER:
        case PA_DATE:
            v.i = ((const int32_t*)c->values)[pos];
            break;
        case PA_DOUBLE:
            v.d = ((const double*)c->values)[pos];
            break;
        case PA_REAL: /* IntArrayBlock of floatToRawIntBits (SPI/type/RealType.java) */
            v.d = (double)((const float*)c->values)[pos];
            break;
        case PA_BOOLEAN:
            v.i = ((const uint8_t*)c->values)[pos] != 0;
            break;
        case PA_VARCHAR:
            v.s = (const uint8_t*)c->values + c->offsets[pos];
            v.slen = c->offsets[pos + 1] - c->offsets[pos];
            break;
        case PA_DECIMAL:
            v.i = ((const int64_t*)c->values)[pos];
            v.q = v.i;
            break;
        case PA_LONG_DECIMAL:
            v.q = ld_read((const uint8_t*)c->values + 16 * (size_t)pos);
            break;
        default:
            break;
    }
    return v;
}

/* type hash operator, NULL -> 0 (TM/type/BlockTypeOperators.java:102-108, TM/type/TypeUtils.java:42) */
static int64_t val_hash(const orc_val* v)
{
    if (v->is_null) {
        return 0;
    }
    switch (v->type) {
        case PA_BIGINT:
            return orc_hash_bigint(v->i);
        case PA_INTEGER:
        case PA_DATE:
            return orc_hash_integer((int32_t)v->i);
        case PA_DOUBLE:
            return orc_hash_double(v->d);
        case PA_REAL: { /* SPI/type/RealType.java:107-115: AbstractLongType.hash(floatToIntBits(v == 0 ? 0 : v)) */
            float f = (float)v->d;
            uint32_t bits;
            if (f == 0.0f) {
                f = 0.0f;
            }
            memcpy(&bits, &f, 4);
            if (f != f) {
                bits = 0x7fc00000u; /* floatToIntBits: one NaN */
            }
            return orc_hash_bigint((int64_t)(int32_t)bits);
        }
        case PA_BOOLEAN:
            return orc_hash_boolean((int32_t)v->i);
        case PA_VARCHAR:
            return (int64_t)orc_xxh64(v->s, v->slen, 0);
        case PA_DECIMAL: /* SPI/type/ShortDecimalType.java:127-131: hashCodeOperator(long value) = value */
            return v->i;
        default:
            return 0;
    }
}

/* TM/operator/InterpretedHashGenerator.java:62-70 */
static int64_t hash_position(const pa_page* page, int32_t nch, const int32_t* ch, int32_t pos)
{
    int64_t result = 0; /* HashGenerationOptimizer.INITIAL_HASH_VALUE */
    for (int32_t i = 0; i < nch; i++) {
        orc_val v = col_get(&page->columns[ch[i]], pos);
        result = orc_combine_hash(result, val_hash(&v));
    }
    return result;
}

int32_t orc_hash_page(const pa_page* page, int32_t nch, const int32_t* ch, int64_t* out)
{
    for (int32_t p = 0; p < page->position_count; p++) {
        out[p] = hash_position(page, nch, ch, p);
    }
    return 0;
}

/* local: TM/operator/exchange/LocalPartitionGenerator.java:45-65
 * remote: TM/operator/HashGenerator.java:24-35 */
int32_t orc_partition_ids(const int64_t* raw_hash, int32_t n, int32_t partition_count, int32_t local, int32_t* out)
{
    if (partition_count <= 0) {
        return fail(PA_ERR_INVALID_ARGUMENT, "partitionCount must be positive");
    }
    if (local && (partition_count & (partition_count - 1)) != 0) {
        return fail(PA_ERR_INVALID_ARGUMENT, "partitionCount must be a power of 2");
    }
    for (int32_t i = 0; i < n; i++) {
        if (local) {
            uint64_t x = (uint64_t)raw_hash[i];
            uint64_t r = 0;
            for (int b = 0; b < 64; b++) { /* Long.reverse */
                r = (r << 1) | ((x >> b) & 1);
            }
            out[i] = (int32_t)orc_xxh64_long((int64_t)r) & (partition_count - 1);
        }
        else {
            int64_t h = raw_hash[i] & 0x7fffffffffffffffLL;
            out[i] = (int32_t)(h % partition_count);
        }
    }
    return 0;
}

/* TM/operator/exchange/PartitioningExchanger.java:59-82: per-partition IntArrayList of positions in
 * ascending position order; here concatenated partition by partition. */
int32_t orc_partition_positions(const int32_t* partition, int32_t n, int32_t partition_count,
                                int32_t* out_positions, int64_t* out_counts)
{
    int64_t* start = (int64_t*)calloc((size_t)partition_count + 1, sizeof(int64_t));
    for (int32_t i = 0; i < n; i++) {
        start[partition[i] + 1]++;
    }
    for (int32_t p = 0; p < partition_count; p++) {
        out_counts[p] = start[p + 1];
        start[p + 1] += start[p];
    }
    for (int32_t i = 0; i < n; i++) {
        out_positions[start[partition[i]]++] = i;
    }
    free(start);
    return 0;
}

/* =====================================================================================
 * RowExpression interpreter with SQL three-valued logic
 * (generated code spec: TM/sql/gen/PageFunctionCompiler.java:459-544, AndCodeGenerator.java:44-105,
 *  OrCodeGenerator.java, BetweenCodeGenerator.java:58-82, IfCodeGenerator, InCodeGenerator,
 *  CoalesceCodeGenerator, IsNullCodeGenerator)
 * ===================================================================================== */
typedef struct eval_ctx {
    const pa_page* page;
    const pa_expr* e;
    int32_t pos;
    int32_t error;
} eval_ctx;

static orc_val eval_node(eval_ctx* cx, int32_t id);

static orc_val null_of(int32_t type)
{
    orc_val v;
    memset(&v, 0, sizeof v);
    v.is_null = 1;
    v.type = type;
    return v;
}
static orc_val bool_of(int b)
{
    orc_val v;
    memset(&v, 0, sizeof v);
    v.type = PA_BOOLEAN;
    v.i = b ? 1 : 0;
    return v;
}

static int val_compare_varchar(const orc_val* a, const orc_val* b)
{
    int32_t n = a->slen < b->slen ? a->slen : b->slen;
    int c = n > 0 ? memcmp(a->s, b->s, (size_t)n) : 0; /* Slice.compareTo: unsigned bytes */
    if (c != 0) {
        return c;
    }
    return (a->slen > b->slen) - (a->slen < b->slen);
}

static orc_val eval_compare(eval_ctx* cx, int32_t op, const orc_val* a, const orc_val* b)
{
    if (a->is_null || b->is_null) {
        return null_of(PA_BOOLEAN);
    }
    int lt, eq;
    if (type_is_fp(a->type)) {
        /* TM/type/DoubleOperators, RealOperators: plain IEEE comparisons (NaN compares false, -0 == +0); float comparisons are the
         * comparisons of the widened values */
        lt = a->d < b->d;
        eq = a->d == b->d;
        switch (op) {
            case PA_OP_EQUAL: return bool_of(eq);
            case PA_OP_NOT_EQUAL: return bool_of(a->d != b->d);
            case PA_OP_LESS_THAN: return bool_of(lt);
            case PA_OP_LESS_THAN_OR_EQUAL: return bool_of(a->d <= b->d);
            case PA_OP_GREATER_THAN: return bool_of(a->d > b->d);
            default: return bool_of(a->d >= b->d);
        }
    }
    if (a->type == PA_VARCHAR) {
        int c = val_compare_varchar(a, b);
        lt = c < 0;
        eq = c == 0;
    }
    else if (type_is_decimal(a->type)) { /* operands of one decimal type (the planner casts): their unscaled values compare */
        lt = a->q < b->q;
        eq = a->q == b->q;
    }
    else {
        lt = a->i < b->i;
        eq = a->i == b->i;
    }
    switch (op) {
        case PA_OP_EQUAL: return bool_of(eq);
        case PA_OP_NOT_EQUAL: return bool_of(!eq);
        case PA_OP_LESS_THAN: return bool_of(lt);
        case PA_OP_LESS_THAN_OR_EQUAL: return bool_of(lt || eq);
        case PA_OP_GREATER_THAN: return bool_of(!lt && !eq);
        default: return bool_of(!lt);
    }
    (void)cx;
}

static orc_val eval_arith(eval_ctx* cx, int32_t op, int32_t type, const orc_val* a, const orc_val* b)
{
    if (a->is_null || (b != NULL && b->is_null)) {
        return null_of(type);
    }
    orc_val r;
    memset(&r, 0, sizeof r);
    r.type = type;
    if (type == PA_REAL) { /* TM/type/RealOperators.java:55-95: Java float arithmetic */
        const float x = (float)a->d, y = b ? (float)b->d : 0.0f;
        float z;
        switch (op) {
            case PA_OP_ADD: z = x + y; break;
            case PA_OP_SUBTRACT: z = x - y; break;
            case PA_OP_MULTIPLY: z = x * y; break;
            case PA_OP_DIVIDE: z = x / y; break;
            case PA_OP_MODULUS: z = fmodf(x, y); break;
            default: z = -x; break;
        }
        r.d = (double)z;
        return r;
    }
    if (type == PA_DOUBLE) { /* TM/type/DoubleOperators.java:59-110 */
        switch (op) {
            case PA_OP_ADD: r.d = a->d + b->d; break;
            case PA_OP_SUBTRACT: r.d = a->d - b->d; break;
            case PA_OP_MULTIPLY: r.d = a->d * b->d; break;
            case PA_OP_DIVIDE: r.d = a->d / b->d; break;
            case PA_OP_MODULUS: r.d = fmod(a->d, b->d); break;
            default: r.d = -a->d; break;
        }
        return r;
    }
    /* TM/type/BigintOperators.java:47-121, IntegerOperators.java: exact arithmetic */
    int64_t x = a->i, y = b ? b->i : 0, z = 0;
    int ovf = 0;
    switch (op) {
        case PA_OP_ADD: ovf = __builtin_add_overflow(x, y, &z); break;
        case PA_OP_SUBTRACT: ovf = __builtin_sub_overflow(x, y, &z); break;
        case PA_OP_MULTIPLY: ovf = __builtin_mul_overflow(x, y, &z); break;
        case PA_OP_DIVIDE:
            if (y == 0) { cx->error = PA_ERR_DIVISION_BY_ZERO; return r; }
            if (x == INT64_MIN && y == -1) { ovf = 1; break; }
            z = x / y;
            break;
        case PA_OP_MODULUS:
            if (y == 0) { cx->error = PA_ERR_DIVISION_BY_ZERO; return r; }
            z = (y == -1) ? 0 : x % y;
            break;
        default: /* NEGATE */
            ovf = __builtin_sub_overflow((int64_t)0, x, &z);
            break;
    }
    if (!ovf && type != PA_BIGINT && (z > INT32_MAX || z < INT32_MIN)) {
        ovf = 1; /* IntegerOperators: Math.addExact(int,int) */
    }
    if (ovf) {
        cx->error = PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE;
        return r;
    }
    r.i = z;
    return r;
}

/* TM/type/DecimalOperators.java: add / subtract (:86-231) bring both operands to the result scale (the larger of the two) and add
 * the unscaled values; multiply (:251-330) multiplies them (the result scale is the sum of the scales).  A result type of at most
 * 18 digits is computed in longs (the derived precision cannot overflow them); a long result throws NUMERIC_VALUE_OUT_OF_RANGE
 * once its magnitude reaches 10^38 (UnscaledDecimal128Arithmetic.throwIfOverflows) -- a product that does not even fit 128 bits
 * included.  sa / sb / sr: the scales of the operands and of the result (pa_expr_node.type_param). */
static orc_val eval_decimal_arith(eval_ctx* cx, int32_t op, int32_t type, const orc_val* a, int sa, const orc_val* b, int sb, int sr)
{
    if (a->is_null || (b != NULL && b->is_null)) {
        return null_of(type);
    }
    orc_val r;
    memset(&r, 0, sizeof r);
    r.type = type;
    __int128 x = a->q, y = b ? b->q : 0, z = 0;
    int ovf = 0;
    switch (op) {
        case PA_OP_ADD:
        case PA_OP_SUBTRACT:
            ovf |= __builtin_mul_overflow(x, pow10_128(sr - sa), &x);
            ovf |= __builtin_mul_overflow(y, pow10_128(sr - sb), &y);
            ovf |= op == PA_OP_ADD ? __builtin_add_overflow(x, y, &z) : __builtin_sub_overflow(x, y, &z);
            break;
        case PA_OP_MULTIPLY:
            ovf = __builtin_mul_overflow(x, y, &z);
            break;
        case PA_OP_NEGATE:
            z = -x;
            break;
        default:
            cx->error = PA_ERR_NOT_SUPPORTED;
            return r;
    }
    if (type == PA_LONG_DECIMAL ? (ovf || ld_overflows(z)) : ovf) {
        cx->error = PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE;
        return r;
    }
    r.q = z;
    r.i = (int64_t)z; /* (a short result: the long the reference computed, wrapping as Java's would if the type lied) */
    if (type == PA_DECIMAL) r.q = r.i;
    return r;
}

/* CAST to DECIMAL(p, s): from BIGINT / INTEGER (TM/type/DecimalCasts.java bigintToShortDecimal / bigintToLongDecimal: value * 10^s,
 * out of range once the magnitude reaches 10^p) and from another DECIMAL (DecimalConversions.java shortToShortCast ... : rescale,
 * dividing rounds half up -- away from zero --, then the same range check) */
static orc_val eval_decimal_cast(eval_ctx* cx, int32_t type, int32_t param, const orc_val* a, int sa)
{
    orc_val r;
    memset(&r, 0, sizeof r);
    r.type = type;
    const int p = PA_DECIMAL_PRECISION(param), sc = PA_DECIMAL_SCALE(param);
    __int128 v = type_is_decimal(a->type) ? a->q : (__int128)a->i;
    int ovf = 0;
    if (sc >= sa) {
        ovf = __builtin_mul_overflow(v, pow10_128(sc - sa), &v);
    }
    else {
        const __int128 d = pow10_128(sa - sc), half = d / 2;
        const __int128 q = v / d, rem = v % d;
        v = q + (rem >= half ? 1 : (rem <= -half ? -1 : 0));
    }
    if (ovf || v >= pow10_128(p) || v <= -pow10_128(p)) {
        cx->error = PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE;
        return r;
    }
    r.q = v;
    r.i = (int64_t)v;
    return r;
}

static orc_val eval_node(eval_ctx* cx, int32_t id)
{
    const pa_expr_node* n = &cx->e->nodes[id];
    const int32_t* args = cx->e->args + n->first_arg;
    if (cx->error) {
        return null_of(n->type);
    }
    switch (n->kind) {
        case PA_EXPR_INPUT_REF:
            return col_get(&cx->page->columns[n->channel], cx->pos);
        case PA_EXPR_CONSTANT: {
            orc_val v;
            memset(&v, 0, sizeof v);
            v.type = n->type;
            v.is_null = n->is_null;
            v.i = n->i64;
            v.d = n->type == PA_REAL ? (double)(float)n->f64 : n->f64;
            v.s = (const uint8_t*)n->str;
            v.slen = n->str_len;
            if (n->type == PA_DECIMAL) {
                v.q = n->i64;
            }
            else if (n->type == PA_LONG_DECIMAL) { /* low 64 bits in i64, high 64 bits in the bits of f64 (two's complement) */
                uint64_t hi;
                memcpy(&hi, &n->f64, 8);
                v.q = (__int128)(((u128)hi << 64) | (uint64_t)n->i64);
            }
            return v;
        }
        case PA_EXPR_CALL: {
            if (n->op >= PA_OP_EQUAL && n->op <= PA_OP_GREATER_THAN_OR_EQUAL) {
                orc_val a = eval_node(cx, args[0]);
                orc_val b = eval_node(cx, args[1]);
                return eval_compare(cx, n->op, &a, &b);
            }
            if (n->op == PA_OP_NOT) {
                orc_val a = eval_node(cx, args[0]);
                if (a.is_null) {
                    return null_of(PA_BOOLEAN);
                }
                return bool_of(!a.i);
            }
            if (n->op == PA_OP_NEGATE) {
                orc_val a = eval_node(cx, args[0]);
                if (type_is_decimal(n->type)) {
                    return eval_decimal_arith(cx, n->op, n->type, &a, 0, NULL, 0, 0);
                }
                return eval_arith(cx, n->op, n->type, &a, NULL);
            }
            if (n->op == PA_OP_CAST) {
                orc_val a = eval_node(cx, args[0]);
                if (a.is_null) {
                    return null_of(n->type);
                }
                if (type_is_decimal(n->type) && (type_is_decimal(a.type) || type_is_int(a.type))) {
                    const pa_expr_node* an = &cx->e->nodes[args[0]];
                    return eval_decimal_cast(cx, n->type, n->type_param, &a, type_is_decimal(an->type) ? PA_DECIMAL_SCALE(an->type_param) : 0);
                }
                orc_val r;
                memset(&r, 0, sizeof r);
                r.type = n->type;
                if (n->type == PA_DOUBLE && type_is_int(a.type)) {
                    r.d = (double)a.i; /* BigintOperators.castToDouble */
                }
                else if (n->type == PA_DOUBLE && a.type == PA_REAL) {
                    r.d = a.d; /* RealOperators.castToDouble (:164-169) */
                }
                else if (n->type == PA_REAL && a.type == PA_DOUBLE) {
                    r.d = (double)(float)a.d; /* DoubleOperators.castToReal: (float) value */
                }
                else if (n->type == PA_REAL && type_is_int(a.type)) {
                    r.d = (double)(float)a.i; /* BigintOperators / IntegerOperators.castToReal: (float) value */
                }
                else if (n->type == PA_BIGINT && type_is_int(a.type)) {
                    r.i = a.i;
                }
                else if (n->type == a.type) {
                    r = a;
                }
                else {
                    cx->error = PA_ERR_NOT_SUPPORTED;
                }
                return r;
            }
            orc_val a = eval_node(cx, args[0]);
            orc_val b = eval_node(cx, args[1]);
            if (type_is_decimal(n->type)) {
                return eval_decimal_arith(cx, n->op, n->type, &a, PA_DECIMAL_SCALE(cx->e->nodes[args[0]].type_param), &b,
                                          PA_DECIMAL_SCALE(cx->e->nodes[args[1]].type_param), PA_DECIMAL_SCALE(n->type_param));
            }
            return eval_arith(cx, n->op, n->type, &a, &b);
        }
        case PA_EXPR_SPECIAL: {
            switch (n->op) {
                case PA_FORM_AND: { /* AndCodeGenerator.java:44-105: left-to-right, false short-circuits */
                    int saw_null = 0;
                    for (int32_t k = 0; k < n->nargs; k++) {
                        orc_val t = eval_node(cx, args[k]);
                        if (cx->error) return null_of(PA_BOOLEAN);
                        if (t.is_null) saw_null = 1;
                        else if (!t.i) return bool_of(0);
                    }
                    return saw_null ? null_of(PA_BOOLEAN) : bool_of(1);
                }
                case PA_FORM_OR: {
                    int saw_null = 0;
                    for (int32_t k = 0; k < n->nargs; k++) {
                        orc_val t = eval_node(cx, args[k]);
                        if (cx->error) return null_of(PA_BOOLEAN);
                        if (t.is_null) saw_null = 1;
                        else if (t.i) return bool_of(1);
                    }
                    return saw_null ? null_of(PA_BOOLEAN) : bool_of(0);
                }
                case PA_FORM_BETWEEN: { /* BetweenCodeGenerator.java:64-68: value >= min AND value <= max */
                    orc_val v = eval_node(cx, args[0]);
                    orc_val lo = eval_node(cx, args[1]);
                    orc_val c1 = eval_compare(cx, PA_OP_GREATER_THAN_OR_EQUAL, &v, &lo);
                    if (!c1.is_null && !c1.i) {
                        return bool_of(0);
                    }
                    orc_val hi = eval_node(cx, args[2]);
                    orc_val c2 = eval_compare(cx, PA_OP_LESS_THAN_OR_EQUAL, &v, &hi);
                    if (!c2.is_null && !c2.i) {
                        return bool_of(0);
                    }
                    if (c1.is_null || c2.is_null) {
                        return null_of(PA_BOOLEAN);
                    }
                    return bool_of(1);
                }
                case PA_FORM_IS_NULL: {
                    orc_val v = eval_node(cx, args[0]);
                    return bool_of(v.is_null);
                }
                case PA_FORM_IF: { /* IfCodeGenerator: NULL condition takes the false branch */
                    orc_val c = eval_node(cx, args[0]);
                    if (cx->error) return null_of(n->type);
                    if (!c.is_null && c.i) {
                        return eval_node(cx, args[1]);
                    }
                    return eval_node(cx, args[2]);
                }
                case PA_FORM_COALESCE: {
                    for (int32_t k = 0; k < n->nargs; k++) {
                        orc_val v = eval_node(cx, args[k]);
                        if (cx->error) return null_of(n->type);
                        if (!v.is_null) return v;
                    }
                    return null_of(n->type);
                }
                case PA_FORM_IN: { /* InCodeGenerator: NULL value -> NULL; miss with a NULL candidate -> NULL */
                    orc_val v = eval_node(cx, args[0]);
                    if (v.is_null) {
                        return null_of(PA_BOOLEAN);
                    }
                    int saw_null = 0;
                    for (int32_t k = 1; k < n->nargs; k++) {
                        orc_val c = eval_node(cx, args[k]);
                        orc_val eq = eval_compare(cx, PA_OP_EQUAL, &v, &c);
                        if (eq.is_null) saw_null = 1;
                        else if (eq.i) return bool_of(1);
                    }
                    return saw_null ? null_of(PA_BOOLEAN) : bool_of(0);
                }
                default:
                    cx->error = PA_ERR_NOT_SUPPORTED;
                    return null_of(n->type);
            }
        }
        default:
            cx->error = PA_ERR_NOT_SUPPORTED;
            return null_of(n->type);
    }
}

/* =====================================================================================
 * PageFilter / PageProcessor
 * ===================================================================================== */
/* generated PageFilter.filter loop (TM/sql/gen/PageFunctionCompiler.java:477-499; row result =
 * !wasNull && value, :539-542) then PageFilter.positionsArrayToSelectedPositions
 * (TM/operator/project/PageFilter.java:27-50). */
int32_t orc_filter(const pa_page* page, const pa_expr* filter, int32_t* positions, int32_t* count, int32_t* is_list)
{
    int32_t n = page->position_count;
    uint8_t* selected = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    eval_ctx cx = {page, filter, 0, 0};
    for (int32_t p = 0; p < n; p++) {
        cx.pos = p;
        orc_val v = eval_node(&cx, filter->root);
        if (cx.error) {
            free(selected);
            return fail(cx.error, "filter evaluation failed");
        }
        selected[p] = (!v.is_null && v.i) ? 1 : 0;
    }
    int32_t selected_count = 0;
    for (int32_t i = 0; i < n; i++) {
        if (selected[i]) {
            selected_count++;
        }
    }
    *count = selected_count;
    if (selected_count == 0 || selected_count == n) {
        *is_list = 0; /* positionsRange(0, selectedCount) */
        free(selected);
        return 0;
    }
    int32_t index = 0;
    for (int32_t p = 0; p < n; p++) {
        if (selected[p]) {
            positions[index++] = p;
        }
    }
    *is_list = 1;
    free(selected);
    return 0;
}

/* growable output column (BlockBuilder) */
typedef struct col_builder {
    int32_t type;
    int32_t count;
    int32_t cap;
    uint8_t* values;   /* element array or varchar bytes */
    int64_t bytes;     /* varchar bytes used */
    int64_t bytes_cap;
    int32_t* offsets;
    uint8_t* nulls;
    int32_t has_null;
} col_builder;

static int32_t type_width(int32_t t)
{
    switch (t) {
        case PA_LONG_DECIMAL:
            return 16;
        case PA_BIGINT:
        case PA_DOUBLE:
        case PA_DECIMAL:
            return 8;
        case PA_INTEGER:
        case PA_DATE:
        case PA_REAL:
            return 4;
        case PA_BOOLEAN:
            return 1;
        default:
            return 0;
    }
}

static void cb_init(col_builder* b, int32_t type, int32_t cap)
{
    memset(b, 0, sizeof *b);
    b->type = type;
    b->cap = cap > 16 ? cap : 16;
    int32_t w = type_width(type);
    if (type == PA_VARCHAR) {
        b->bytes_cap = 64;
        b->values = (uint8_t*)malloc((size_t)b->bytes_cap);
        b->offsets = (int32_t*)malloc(sizeof(int32_t) * ((size_t)b->cap + 1));
        b->offsets[0] = 0;
    }
    else {
        b->values = (uint8_t*)malloc((size_t)w * (size_t)b->cap);
    }
    b->nulls = (uint8_t*)calloc((size_t)b->cap, 1);
}

static void cb_reserve(col_builder* b)
{
    if (b->count < b->cap) {
        return;
    }
    int32_t ncap = b->cap * 2;
    int32_t w = type_width(b->type);
    if (b->type == PA_VARCHAR) {
        b->offsets = (int32_t*)realloc(b->offsets, sizeof(int32_t) * ((size_t)ncap + 1));
    }
    else {
        b->values = (uint8_t*)realloc(b->values, (size_t)w * (size_t)ncap);
    }
    b->nulls = (uint8_t*)realloc(b->nulls, (size_t)ncap);
    memset(b->nulls + b->cap, 0, (size_t)(ncap - b->cap));
    b->cap = ncap;
}

static void cb_append(col_builder* b, const orc_val* v)
{
    cb_reserve(b);
    int32_t i = b->count++;
    if (v->is_null) {
        b->nulls[i] = 1;
        b->has_null = 1;
    }
    switch (b->type) {
        case PA_BIGINT:
            ((int64_t*)b->values)[i] = v->is_null ? 0 : v->i;
            break;
        case PA_INTEGER:
        case PA_DATE:
            ((int32_t*)b->values)[i] = v->is_null ? 0 : (int32_t)v->i;
            break;
        case PA_DOUBLE:
            ((double*)b->values)[i] = v->is_null ? 0.0 : v->d;
            break;
        case PA_REAL:
            ((float*)b->values)[i] = v->is_null ? 0.0f : (float)v->d;
            break;
        case PA_BOOLEAN:
            b->values[i] = v->is_null ? 0 : (uint8_t)(v->i != 0);
            break;
        case PA_DECIMAL:
            ((int64_t*)b->values)[i] = v->is_null ? 0 : (int64_t)v->q;
            break;
        case PA_LONG_DECIMAL:
            ld_write(b->values + 16 * (size_t)i, v->is_null ? 0 : v->q);
            break;
        case PA_VARCHAR: {
            int32_t len = v->is_null ? 0 : v->slen;
            while (b->bytes + len > b->bytes_cap) {
                b->bytes_cap *= 2;
                b->values = (uint8_t*)realloc(b->values, (size_t)b->bytes_cap);
            }
            if (len > 0) {
                memcpy(b->values + b->bytes, v->s, (size_t)len);
            }
            b->bytes += len;
            b->offsets[i + 1] = (int32_t)b->bytes;
            break;
        }
        default:
            break;
    }
}

static void cb_finish(col_builder* b, pa_column* out)
{
    memset(out, 0, sizeof *out);
    out->type = b->type;
    out->encoding = b->type == PA_VARCHAR ? PA_VARWIDTH : PA_FLAT;
    out->values = b->values;
    out->offsets = b->offsets;
    if (b->has_null) {
        out->nulls = b->nulls;
    }
    else {
        free(b->nulls);
        out->nulls = NULL;
    }
}

void orc_free_page(pa_page* page)
{
    if (page == NULL || page->columns == NULL) {
        return;
    }
    for (int32_t c = 0; c < page->channel_count; c++) {
        free((void*)page->columns[c].values);
        free((void*)page->columns[c].offsets);
        free((void*)page->columns[c].nulls);
    }
    free(page->columns);
    page->columns = NULL;
}

/* PageProcessor.createWorkProcessor + ProjectSelectedPositions (TM/operator/project/PageProcessor.java:
 * 111-137, 180-263, 307-347): empty page or no selected position -> nothing; otherwise every projection
 * evaluated over the selected positions in input order.  Batch boundaries are not part of results
 * parity (MergePages re-chunks), so the batches are returned concatenated. */
int32_t orc_filter_project(const pa_page* page, const pa_expr* filter, int32_t projection_count,
                           const pa_expr* projections, pa_page* out)
{
    int32_t n = page->position_count;
    memset(out, 0, sizeof *out);
    if (n == 0) {
        return 0; /* PageProcessor.java:113-115 */
    }
    int32_t* positions = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int32_t count = n, is_list = 0;
    if (filter != NULL) {
        int32_t rc = orc_filter(page, filter, positions, &count, &is_list);
        if (rc < 0) {
            free(positions);
            return rc;
        }
    }
    if (count == 0) {
        free(positions);
        return 0; /* PageProcessor.java:127-129 */
    }
    out->position_count = count;
    out->channel_count = projection_count;
    out->mem = PA_MEM_HOST;
    out->columns = (pa_column*)calloc((size_t)(projection_count > 0 ? projection_count : 1), sizeof(pa_column));
    for (int32_t j = 0; j < projection_count; j++) {
        const pa_expr* e = &projections[j];
        col_builder b;
        cb_init(&b, e->nodes[e->root].type, count);
        eval_ctx cx = {page, e, 0, 0};
        for (int32_t k = 0; k < count; k++) {
            cx.pos = is_list ? positions[k] : k;
            orc_val v = eval_node(&cx, e->root);
            if (cx.error) {
                free(positions);
                cb_finish(&b, &out->columns[j]);
                orc_free_page(out);
                return fail(cx.error, "projection evaluation failed");
            }
            cb_append(&b, &v);
        }
        cb_finish(&b, &out->columns[j]);
    }
    free(positions);
    return 1;
}

/* =====================================================================================
 * Aggregation: GroupByHash + accumulators
 * ===================================================================================== */
#define FILL_RATIO 0.75f

/* BigintGroupByHash.calculateMaxFill (TM/operator/BigintGroupByHash.java:325-334) */
static int32_t calculate_max_fill(int32_t hash_size)
{
    int32_t max_fill = (int32_t)ceil(hash_size * (double)FILL_RATIO);
    if (max_fill == hash_size) {
        max_fill--;
    }
    return max_fill;
}

typedef struct acc_state { /* LongDoubleState / LongLongState / LongState per group */
    int64_t count;
    double dsum;
    int64_t lsum;
    int32_t has_value; /* min/max */
    uint8_t* str;      /* min/max over VARCHAR: the value so far (owned copy) */
    int32_t slen;
    /* LongDecimalWithOverflowState / LongDecimalWithOverflowAndLongState (sum / avg over DECIMAL): the 127-bit sign-magnitude sum as
     * the reference's Slice holds it (dneg: its sign bit, which zero may carry too) and the overflow counter beside it */
    u128 dmag;
    int32_t dneg;
    int64_t overflow;
} acc_state;

/* UnscaledDecimal128Arithmetic.addWithOverflow (SPI/type/UnscaledDecimal128Arithmetic.java:354-379) on (sign, 127-bit magnitude)
 * pairs: equal signs add the magnitudes -- a carry out of bit 126 is the overflow, +1 / -1 by the sign, and the sum keeps its low
 * 127 bits --; different signs subtract the smaller magnitude from the larger and take its sign; equal magnitudes give +0 */
static int64_t ld_add_with_overflow(acc_state* s, __int128 x)
{
    const int xneg = x < 0;
    const u128 xmag = xneg ? (u128)(-x) : (u128)x;
    const u128 limit = (u128)1 << 127;
    if (s->dneg == xneg) {
        u128 sum = s->dmag + xmag; /* both < 2^127: no wrap of the 128-bit word */
        const int64_t ovf = sum >= limit ? 1 : 0;
        s->dmag = sum & (limit - 1);
        return xneg ? -ovf : ovf;
    }
    if (s->dmag > xmag) {
        s->dmag -= xmag;
    }
    else if (s->dmag < xmag) {
        s->dmag = xmag - s->dmag;
        s->dneg = xneg;
    }
    else {
        s->dmag = 0;
        s->dneg = 0;
    }
    return 0;
}
/* sum = overflow * 2^127 + (sign, magnitude), divided by count and rounded half up -- away from zero -- at the sum's scale:
 * DecimalAverageAggregation.average (TM/operator/aggregation/DecimalAverageAggregation.java:214-226; BigDecimal.divide(count, scale,
 * ROUND_HALF_UP)).  Exact in 192-bit arithmetic on three limbs.  Returns 0 when the quotient does not fit 127 bits. */
static int ld_average(const acc_state* s, int64_t count, __int128* out)
{
    /* total = overflow * 2^127 +- mag as a two's complement number of 192 bits in three 64-bit limbs */
    uint64_t t[3] = {0, 0, 0};
    {
        /* overflow * 2^127: overflow is a small signed count */
        const __int128 hi = (__int128)s->overflow; /* bits 127.. */
        /* place hi << 127 */
        const u128 lowpart = ((u128)(uint64_t)hi << 127); /* contributes bit 127 of limb 1 from bit 0 of hi */
        t[0] = 0;
        t[1] = (uint64_t)(lowpart >> 64);
        t[2] = (uint64_t)((hi >> 1) & (__int128)0xffffffffffffffffULL);
        /* add the signed magnitude */
        u128 mag = s->dmag;
        uint64_t m[3] = {(uint64_t)mag, (uint64_t)(mag >> 64), 0};
        if (s->dneg) { /* two's complement negate over 192 bits */
            uint64_t carry = 1;
            for (int i = 0; i < 3; i++) {
                uint64_t v = ~m[i] + carry;
                carry = (carry && v == 0) ? 1 : 0;
                m[i] = v;
            }
        }
        uint64_t carry = 0;
        for (int i = 0; i < 3; i++) {
            u128 sum = (u128)t[i] + m[i] + carry;
            t[i] = (uint64_t)sum;
            carry = (uint64_t)(sum >> 64);
        }
    }
    const int neg = (t[2] >> 63) != 0;
    if (neg) {
        uint64_t carry = 1;
        for (int i = 0; i < 3; i++) {
            uint64_t v = ~t[i] + carry;
            carry = (carry && v == 0) ? 1 : 0;
            t[i] = v;
        }
    }
    /* magnitude / count with remainder, limb by limb from the top */
    const uint64_t d = (uint64_t)count;
    uint64_t q[3];
    u128 rem = 0;
    for (int i = 2; i >= 0; i--) {
        const u128 cur = (rem << 64) | t[i];
        q[i] = (uint64_t)(cur / d);
        rem = cur % d;
    }
    if ((u128)rem * 2 >= (u128)d) { /* half up on the magnitude */
        for (int i = 0; i < 3; i++) {
            if (++q[i] != 0) break;
        }
    }
    if (q[2] != 0 || (q[1] >> 63)) {
        return 0;
    }
    const u128 mag = ((u128)q[1] << 64) | q[0];
    *out = neg ? -(__int128)mag : (__int128)mag;
    return 1;
}

struct orc_hash_agg {
    pa_hash_aggregation_desc desc;
    int32_t* input_types;
    int32_t* group_channels;
    pa_aggregate* aggs;
    int32_t bigint_mode; /* GroupByHash.createGroupByHash: one BIGINT key (TM/operator/GroupByHash.java:55-57) */

    int32_t hash_capacity, max_fill, mask, next_group_id;
    /* BigintGroupByHash */
    int64_t* values;
    int32_t* group_ids;
    int64_t* values_by_group_id;
    int32_t null_group_id;
    /* MultiChannelGroupByHash */
    int64_t* group_address_by_hash;
    int32_t* group_ids_by_hash;
    uint8_t* raw_hash_by_hash_position;
    col_builder* key_builders; /* group_by_count (+1 for the precomputed hash) */
    int32_t key_builder_count;
    int64_t* raw_hash_by_group;

    acc_state* states; /* [aggregate][group] */
    int32_t state_cap;
    int32_t error;
};

static void agg_ensure_states(orc_hash_agg* a, int32_t groups)
{
    if (groups <= a->state_cap) {
        return;
    }
    int32_t ncap = a->state_cap ? a->state_cap : 16;
    while (ncap < groups) {
        ncap *= 2;
    }
    acc_state* ns = (acc_state*)calloc((size_t)ncap * (size_t)(a->desc.aggregate_count > 0 ? a->desc.aggregate_count : 1), sizeof(acc_state));
    for (int32_t g = 0; g < a->desc.aggregate_count; g++) {
        if (a->state_cap) {
            memcpy(ns + (size_t)g * ncap, a->states + (size_t)g * a->state_cap, sizeof(acc_state) * (size_t)a->state_cap);
        }
    }
    free(a->states);
    a->states = ns;
    a->state_cap = ncap;
}

orc_hash_agg* orc_hash_agg_create(const pa_hash_aggregation_desc* d)
{
    orc_hash_agg* a = (orc_hash_agg*)calloc(1, sizeof *a);
    a->desc = *d;
    a->input_types = (int32_t*)malloc(sizeof(int32_t) * (size_t)(d->input_channel_count > 0 ? d->input_channel_count : 1));
    memcpy(a->input_types, d->input_types, sizeof(int32_t) * (size_t)d->input_channel_count);
    a->group_channels = (int32_t*)malloc(sizeof(int32_t) * (size_t)(d->group_by_count > 0 ? d->group_by_count : 1));
    if (d->group_by_count > 0) {
        memcpy(a->group_channels, d->group_by_channels, sizeof(int32_t) * (size_t)d->group_by_count);
    }
    a->aggs = (pa_aggregate*)malloc(sizeof(pa_aggregate) * (size_t)(d->aggregate_count > 0 ? d->aggregate_count : 1));
    if (d->aggregate_count > 0) {
        memcpy(a->aggs, d->aggregates, sizeof(pa_aggregate) * (size_t)d->aggregate_count);
    }
    a->null_group_id = -1;
    if (d->group_by_count == 0) {
        a->next_group_id = 1; /* AggregationOperator: one global state */
        agg_ensure_states(a, 1);
        return a;
    }
    int32_t expected = d->expected_groups > 0 ? d->expected_groups : 1;
    a->hash_capacity = orc_array_size(expected, FILL_RATIO);
    a->max_fill = calculate_max_fill(a->hash_capacity);
    a->mask = a->hash_capacity - 1;
    a->bigint_mode = d->group_by_count == 1 && d->input_types[d->group_by_channels[0]] == PA_BIGINT;
    if (a->bigint_mode) { /* BigintGroupByHash ctor (TM/operator/BigintGroupByHash.java:78-101) */
        a->values = (int64_t*)calloc((size_t)a->hash_capacity, 8);
        a->group_ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)a->hash_capacity);
        for (int32_t i = 0; i < a->hash_capacity; i++) {
            a->group_ids[i] = -1;
        }
        a->values_by_group_id = (int64_t*)calloc((size_t)a->hash_capacity, 8);
    }
    else { /* MultiChannelGroupByHash ctor (TM/operator/MultiChannelGroupByHash.java:93-157) */
        a->group_address_by_hash = (int64_t*)malloc(8 * (size_t)a->hash_capacity);
        for (int32_t i = 0; i < a->hash_capacity; i++) {
            a->group_address_by_hash[i] = -1;
        }
        a->raw_hash_by_hash_position = (uint8_t*)calloc((size_t)a->hash_capacity, 1);
        a->group_ids_by_hash = (int32_t*)calloc((size_t)a->hash_capacity, sizeof(int32_t));
        a->key_builder_count = d->group_by_count;
        a->key_builders = (col_builder*)calloc((size_t)a->key_builder_count, sizeof(col_builder));
        for (int32_t i = 0; i < d->group_by_count; i++) {
            cb_init(&a->key_builders[i], d->input_types[d->group_by_channels[i]], 16);
        }
    }
    return a;
}

void orc_hash_agg_destroy(orc_hash_agg* a)
{
    if (!a) {
        return;
    }
    free(a->input_types);
    free(a->group_channels);
    free(a->aggs);
    free(a->values);
    free(a->group_ids);
    free(a->values_by_group_id);
    free(a->group_address_by_hash);
    free(a->group_ids_by_hash);
    free(a->raw_hash_by_hash_position);
    for (int32_t i = 0; i < a->key_builder_count; i++) {
        free(a->key_builders[i].values);
        free(a->key_builders[i].offsets);
        free(a->key_builders[i].nulls);
    }
    free(a->key_builders);
    free(a->raw_hash_by_group);
    if (a->states) {
        const size_t total = (size_t)a->state_cap * (size_t)(a->desc.aggregate_count > 0 ? a->desc.aggregate_count : 1);
        for (size_t i = 0; i < total; i++) free(a->states[i].str);
    }
    free(a->states);
    free(a);
}

int32_t orc_hash_agg_group_count(const orc_hash_agg* a) { return a->next_group_id; }
int32_t orc_hash_agg_capacity(const orc_hash_agg* a) { return a->hash_capacity; }

/* ---- BigintGroupByHash (TM/operator/BigintGroupByHash.java:213-313) ---- */
static int32_t bigint_try_rehash(orc_hash_agg* a)
{
    int64_t new_capacity_long = a->hash_capacity * 2LL;
    if (new_capacity_long > INT32_MAX) {
        a->error = PA_ERR_INSUFFICIENT_RESOURCES;
        return 0;
    }
    int32_t new_capacity = (int32_t)new_capacity_long;
    int32_t new_mask = new_capacity - 1;
    int64_t* new_values = (int64_t*)calloc((size_t)new_capacity, 8);
    int32_t* new_group_ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)new_capacity);
    for (int32_t i = 0; i < new_capacity; i++) {
        new_group_ids[i] = -1;
    }
    for (int32_t group_id = 0; group_id < a->next_group_id; group_id++) {
        if (group_id == a->null_group_id) {
            continue;
        }
        int64_t value = a->values_by_group_id[group_id];
        int64_t hash_position = orc_murmur3_fmix(value) & new_mask;
        while (new_group_ids[hash_position] != -1) {
            hash_position = (hash_position + 1) & new_mask;
        }
        new_values[hash_position] = value;
        new_group_ids[hash_position] = group_id;
    }
    free(a->values);
    free(a->group_ids);
    a->mask = new_mask;
    a->hash_capacity = new_capacity;
    a->max_fill = calculate_max_fill(new_capacity);
    a->values = new_values;
    a->group_ids = new_group_ids;
    a->values_by_group_id = (int64_t*)realloc(a->values_by_group_id, 8 * (size_t)new_capacity);
    return 1;
}

static int32_t bigint_put_if_absent(orc_hash_agg* a, const pa_column* block, int32_t position)
{
    orc_val v = col_get(block, position);
    if (v.is_null) {
        if (a->null_group_id < 0) {
            a->null_group_id = a->next_group_id++;
        }
        return a->null_group_id;
    }
    int64_t value = v.i;
    int64_t hash_position = orc_murmur3_fmix(value) & a->mask;
    while (1) {
        int32_t group_id = a->group_ids[hash_position];
        if (group_id == -1) {
            break;
        }
        if (value == a->values[hash_position]) {
            return group_id;
        }
        hash_position = (hash_position + 1) & a->mask;
    }
    /* addNewGroup */
    int32_t group_id = a->next_group_id++;
    a->values[hash_position] = value;
    a->values_by_group_id[group_id] = value;
    a->group_ids[hash_position] = group_id;
    if (a->next_group_id >= a->max_fill) {
        bigint_try_rehash(a);
    }
    return group_id;
}

/* ---- MultiChannelGroupByHash (TM/operator/MultiChannelGroupByHash.java:275-452) ---- */
static orc_val builder_get(const col_builder* b, int32_t pos)
{
    orc_val v;
    memset(&v, 0, sizeof v);
    v.type = b->type;
    if (b->nulls[pos]) {
        v.is_null = 1;
        return v;
    }
    switch (b->type) {
        case PA_BIGINT: v.i = ((int64_t*)b->values)[pos]; break;
        case PA_INTEGER:
        case PA_DATE: v.i = ((int32_t*)b->values)[pos]; break;
        case PA_DOUBLE: v.d = ((double*)b->values)[pos]; break;
        case PA_REAL: v.d = (double)((float*)b->values)[pos]; break;
        case PA_BOOLEAN: v.i = b->values[pos]; break;
        case PA_DECIMAL: v.i = ((int64_t*)b->values)[pos]; v.q = v.i; break;
        case PA_VARCHAR:
            v.s = b->values + b->offsets[pos];
            v.slen = b->offsets[pos + 1] - b->offsets[pos];
            break;
        default: break;
    }
    return v;
}

/* PagesHashStrategy.positionNotDistinctFromRow: NULL is not distinct from NULL; DOUBLE NaN is not
 * distinct from NaN (TM/sql/gen/JoinCompiler.java positionNotDistinctFromRow -> IS_DISTINCT_FROM,
 * SPI/type/DoubleType.java:172-184) */
static int not_distinct(const orc_val* a, const orc_val* b)
{
    if (a->is_null || b->is_null) {
        return a->is_null == b->is_null;
    }
    switch (a->type) {
        case PA_DOUBLE:
        case PA_REAL: /* SPI/type/RealType.java:127-140: the same rule on floats (the values travel widened, exactly) */
            if (a->d != a->d && b->d != b->d) {
                return 1;
            }
            return a->d == b->d;
        case PA_VARCHAR:
            return a->slen == b->slen && (a->slen == 0 || memcmp(a->s, b->s, (size_t)a->slen) == 0);
        default:
            return a->i == b->i;
    }
}

static int multi_row_matches(const orc_hash_agg* a, int64_t address, int32_t hash_position, const pa_page* page,
                             int32_t position, uint8_t raw_hash_byte)
{
    if (a->raw_hash_by_hash_position[hash_position] != raw_hash_byte) {
        return 0;
    }
    for (int32_t i = 0; i < a->desc.group_by_count; i++) {
        orc_val l = builder_get(&a->key_builders[i], (int32_t)address);
        orc_val r = col_get(&page->columns[a->group_channels[i]], position);
        if (!not_distinct(&l, &r)) {
            return 0;
        }
    }
    return 1;
}

static int32_t multi_try_rehash(orc_hash_agg* a)
{
    int64_t new_capacity_long = a->hash_capacity * 2LL;
    if (new_capacity_long > INT32_MAX) {
        a->error = PA_ERR_INSUFFICIENT_RESOURCES;
        return 0;
    }
    int32_t new_capacity = (int32_t)new_capacity_long;
    int32_t new_mask = new_capacity - 1;
    int64_t* new_key = (int64_t*)malloc(8 * (size_t)new_capacity);
    uint8_t* raw_hashes = (uint8_t*)calloc((size_t)new_capacity, 1);
    int32_t* new_value = (int32_t*)calloc((size_t)new_capacity, sizeof(int32_t));
    for (int32_t i = 0; i < new_capacity; i++) {
        new_key[i] = -1;
    }
    int32_t old_index = 0;
    for (int32_t group_id = 0; group_id < a->next_group_id; group_id++) {
        while (a->group_address_by_hash[old_index] == -1) {
            old_index++;
        }
        int64_t address = a->group_address_by_hash[old_index];
        int64_t raw_hash = a->raw_hash_by_group[address];
        int32_t pos = (int32_t)(orc_murmur3_fmix(raw_hash) & new_mask);
        while (new_key[pos] != -1) {
            pos = (pos + 1) & new_mask;
        }
        new_key[pos] = address;
        raw_hashes[pos] = (uint8_t)raw_hash;
        new_value[pos] = a->group_ids_by_hash[old_index];
        old_index++;
    }
    free(a->group_address_by_hash);
    free(a->raw_hash_by_hash_position);
    free(a->group_ids_by_hash);
    a->mask = new_mask;
    a->hash_capacity = new_capacity;
    a->max_fill = calculate_max_fill(new_capacity);
    a->group_address_by_hash = new_key;
    a->raw_hash_by_hash_position = raw_hashes;
    a->group_ids_by_hash = new_value;
    return 1;
}

static int32_t multi_put_if_absent(orc_hash_agg* a, const pa_page* page, int32_t position, int64_t raw_hash)
{
    int32_t hash_position = (int32_t)(orc_murmur3_fmix(raw_hash) & a->mask);
    int32_t group_id = -1;
    while (a->group_address_by_hash[hash_position] != -1) {
        if (multi_row_matches(a, a->group_address_by_hash[hash_position], hash_position, page, position, (uint8_t)raw_hash)) {
            group_id = a->group_ids_by_hash[hash_position];
            break;
        }
        hash_position = (hash_position + 1) & a->mask;
    }
    if (group_id >= 0) {
        return group_id;
    }
    /* addNewGroup: append the key row; address == group ordinal here (pageIndex/position split is
     * an allocation detail of the reference with no effect on results) */
    for (int32_t i = 0; i < a->desc.group_by_count; i++) {
        orc_val v = col_get(&page->columns[a->group_channels[i]], position);
        cb_append(&a->key_builders[i], &v);
    }
    group_id = a->next_group_id++;
    a->raw_hash_by_group = (int64_t*)realloc(a->raw_hash_by_group, 8 * (size_t)a->next_group_id);
    a->raw_hash_by_group[group_id] = raw_hash;
    a->group_address_by_hash[hash_position] = group_id;
    a->raw_hash_by_hash_position[hash_position] = (uint8_t)raw_hash;
    a->group_ids_by_hash[hash_position] = group_id;
    if (a->next_group_id >= a->max_fill) {
        multi_try_rehash(a);
    }
    return group_id;
}

static int64_t agg_row_hash(const orc_hash_agg* a, const pa_page* page, int32_t position)
{
    if (a->desc.hash_channel >= 0) { /* PrecomputedHashGenerator */
        orc_val h = col_get(&page->columns[a->desc.hash_channel], position);
        return h.i;
    }
    return hash_position(page, a->desc.group_by_count, a->group_channels, position);
}

int32_t orc_hash_agg_contains(const orc_hash_agg* a, const pa_page* page, int32_t position)
{
    if (a->desc.group_by_count == 0) {
        return 1;
    }
    if (a->bigint_mode) { /* BigintGroupByHash.contains (TM/operator/BigintGroupByHash.java:170-192) */
        orc_val v = col_get(&page->columns[a->group_channels[0]], position);
        if (v.is_null) {
            return a->null_group_id >= 0;
        }
        int64_t hp = orc_murmur3_fmix(v.i) & a->mask;
        while (a->group_ids[hp] != -1) {
            if (a->values[hp] == v.i) {
                return 1;
            }
            hp = (hp + 1) & a->mask;
        }
        return 0;
    }
    int64_t raw_hash = agg_row_hash(a, page, position);
    int32_t hp = (int32_t)(orc_murmur3_fmix(raw_hash) & a->mask);
    while (a->group_address_by_hash[hp] != -1) {
        if (multi_row_matches(a, a->group_address_by_hash[hp], hp, page, position, (uint8_t)raw_hash)) {
            return 1;
        }
        hp = (hp + 1) & a->mask;
    }
    return 0;
}

/* mask channel: NULL -> false, non-zero byte -> true (TM/sql/gen/CompilerOperations.java:65-74) */
static int mask_passes(const pa_page* page, int32_t mask_channel, int32_t pos)
{
    if (mask_channel < 0) {
        return 1;
    }
    const pa_column* c = &page->columns[mask_channel];
    int32_t p = pos;
    c = resolve(c, &p);
    if (c->nulls && c->nulls[p]) {
        return 0;
    }
    return ((const uint8_t*)c->values)[p] != 0;
}

/* DoubleType's COMPARISON operator = Double.compare (SPI/type/DoubleType.java:194-198): numeric order, -0.0 < 0.0,
 * NaN equal to itself and above everything */
static int double_compare(double a, double b)
{
    if (a < b) return -1;
    if (a > b) return 1;
    uint64_t x, y;
    memcpy(&x, &a, 8);
    memcpy(&y, &b, 8);
    if (a != a) x = 0x7ff8000000000000ULL; /* doubleToLongBits: one NaN */
    if (b != b) y = 0x7ff8000000000000ULL;
    int64_t sx = (int64_t)x, sy = (int64_t)y; /* Double.compare falls back to the order of doubleToLongBits */
    return sx == sy ? 0 : (sx < sy ? -1 : 1);
}

/* AbstractMinMaxAggregationFunction.compareAndUpdateState (TM/operator/aggregation/AbstractMinMaxAggregationFunction.java:
 * 274-330): the first value is taken as is, later ones when the type's comparison says smaller (min) / larger (max) */
static void min_max_update(acc_state* s, int is_min, const orc_val* v)
{
    int take;
    if (!s->has_value) {
        take = 1;
    }
    else {
        int c;
        if (v->type == PA_VARCHAR) { /* VarcharType's comparison = Slice.compareTo: unsigned bytes, then length */
            int32_t m = v->slen < s->slen ? v->slen : s->slen;
            c = m > 0 ? memcmp(v->s, s->str, (size_t)m) : 0;
            if (c == 0) c = v->slen < s->slen ? -1 : (v->slen > s->slen ? 1 : 0);
        }
        else {
            c = type_is_fp(v->type) ? double_compare(v->d, s->dsum) : (v->i < s->lsum ? -1 : (v->i > s->lsum ? 1 : 0));
        }
        take = is_min ? c < 0 : c > 0;
    }
    if (take) {
        s->has_value = 1;
        s->dsum = v->d;
        s->lsum = v->i;
        if (v->type == PA_VARCHAR) {
            free(s->str);
            s->str = (uint8_t*)malloc((size_t)(v->slen > 0 ? v->slen : 1));
            if (v->slen > 0) memcpy(s->str, v->s, (size_t)v->slen);
            s->slen = v->slen;
        }
    }
}

/* generated GroupedAccumulator.addInput loop (TM/operator/aggregation/AccumulatorCompiler.java:490-569)
 * calling the input functions of SURVEY a15 */
static void accumulate(orc_hash_agg* a, int32_t k, const pa_page* page, const int32_t* gids)
{
    const pa_aggregate* ag = &a->aggs[k];
    acc_state* st = a->states + (size_t)k * a->state_cap;
    if (a->desc.step == PA_STEP_FINAL) {
        /* Step.FINAL: the input channels are intermediate states [count BIGINT] (+ [sum]); the @CombineFunction of
         * every aggregate adds them (DoubleSumAggregation.java:47-52, AverageAggregations.java:60-65,
         * CountAggregation.java:45-49, LongSumAggregation.java:48-53) */
        for (int32_t pos = 0; pos < page->position_count; pos++) {
            acc_state* s = &st[gids ? gids[pos] : 0];
            orc_val c = col_get(&page->columns[ag->input_channel], pos);
            if (c.is_null) {
                continue;
            }
            s->count += c.i;
            if (ag->fn == PA_AGG_MIN || ag->fn == PA_AGG_MAX) { /* combine = compareAndUpdateState with the other state's value */
                orc_val v = col_get(&page->columns[ag->input_channel + 1], pos);
                if (!v.is_null) {
                    min_max_update(s, ag->fn == PA_AGG_MIN, &v);
                }
                continue;
            }
            if (ag->fn == PA_AGG_SUM || ag->fn == PA_AGG_AVG) {
                orc_val v = col_get(&page->columns[ag->input_channel + 1], pos);
                if (v.is_null) {
                    continue;
                }
                if (v.type == PA_DOUBLE) {
                    s->dsum = s->dsum + v.d;
                }
                else if (type_is_decimal(v.type)) { /* DecimalSumAggregation.combine / DecimalAverageAggregation.combine (the flat
                                                      * state carries no overflow count: a partial sum that had one was refused) */
                    s->overflow += ld_add_with_overflow(s, v.q);
                }
                else {
                    int64_t r;
                    if (__builtin_add_overflow(s->lsum, v.i, &r)) {
                        a->error = PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE;
                        return;
                    }
                    s->lsum = r;
                }
            }
        }
        return;
    }
    for (int32_t pos = 0; pos < page->position_count; pos++) {
        if (!mask_passes(page, ag->mask_channel, pos)) {
            continue;
        }
        acc_state* s = &st[gids ? gids[pos] : 0];
        if (ag->fn == PA_AGG_COUNT_STAR) { /* CountAggregation.java:33-37 */
            s->count++;
            continue;
        }
        orc_val v = col_get(&page->columns[ag->input_channel], pos);
        if (v.is_null) {
            continue;
        }
        switch (ag->fn) {
            case PA_AGG_COUNT: /* CountColumn.java */
                s->count++;
                break;
            case PA_AGG_SUM:
                s->count++;
                if (type_is_decimal(v.type)) { /* DecimalSumAggregation.inputShortDecimal / inputLongDecimal (:139-159) */
                    s->overflow += ld_add_with_overflow(s, v.q);
                }
                else if (type_is_fp(v.type)) { /* DoubleSumAggregation.java:33-38; RealSumAggregation.java:36-41: the REAL sum's state is a
                                           * double, every input widened */
                    s->dsum = s->dsum + v.d;
                }
                else { /* LongSumAggregation.java:37-42 -> BigintOperators.add (Math.addExact) */
                    int64_t r;
                    if (__builtin_add_overflow(s->lsum, v.i, &r)) {
                        a->error = PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE;
                        return;
                    }
                    s->lsum = r;
                }
                break;
            case PA_AGG_AVG: /* AverageAggregations.java:34-46 */
                s->count++;
                if (type_is_decimal(v.type)) { /* DecimalAverageAggregation.inputShortDecimal / inputLongDecimal (:153-177) */
                    s->overflow += ld_add_with_overflow(s, v.q);
                    break;
                }
                s->dsum = s->dsum + (type_is_fp(v.type) ? v.d : (double)v.i); /* (RealAverageAggregation.input: double sum of widened floats) */
                break;
            case PA_AGG_MIN:
            case PA_AGG_MAX:
                min_max_update(s, ag->fn == PA_AGG_MIN, &v);
                s->count++;
                break;
            default:
                break;
        }
    }
}

/* InMemoryHashAggregationBuilder.processPage (TM/operator/aggregation/builder/
 * InMemoryHashAggregationBuilder.java:139-155) / AggregationOperator.addInput
 * (TM/operator/AggregationOperator.java:145-160) */
int32_t orc_hash_agg_add_page(orc_hash_agg* a, const pa_page* page, int32_t* group_ids_out)
{
    int32_t n = page->position_count;
    int32_t* gids = NULL;
    if (a->desc.group_by_count > 0) {
        gids = group_ids_out ? group_ids_out : (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        for (int32_t pos = 0; pos < n; pos++) {
            if (a->bigint_mode) {
                gids[pos] = bigint_put_if_absent(a, &page->columns[a->group_channels[0]], pos);
            }
            else {
                gids[pos] = multi_put_if_absent(a, page, pos, agg_row_hash(a, page, pos));
            }
            if (a->error) {
                if (!group_ids_out) free(gids);
                return fail(a->error, "Size of hash table cannot exceed 1 billion entries");
            }
        }
        agg_ensure_states(a, a->next_group_id);
    }
    for (int32_t k = 0; k < a->desc.aggregate_count; k++) {
        accumulate(a, k, page, gids);
        if (a->error) {
            if (gids && !group_ids_out) free(gids);
            return fail(a->error, "bigint addition overflow");
        }
    }
    if (gids && !group_ids_out) {
        free(gids);
    }
    return 0;
}

/* HashAggregationOperator.getGlobalAggregationOutput (TM/operator/HashAggregationOperator.java:545-587) with
 * calculateDefaultOutputHash (:589-600): the rows an operator with produceDefaultOutput emits when it finishes without having seen
 * a page (:486-492) -- one per globalAggregationGroupIds entry: NULL in every group-by column except groupIdChannel (the id), the
 * row's $hashvalue if a hash channel was supplied, then every accumulator's evaluateFinal / evaluateIntermediate over no input (fresh
 * accumulators: the one row of this file's AggregationOperator restatement over nothing). */
int32_t orc_hash_agg_default_output(const pa_hash_aggregation_desc* d, pa_page* out)
{
    memset(out, 0, sizeof *out);
    out->mem = PA_MEM_HOST;
    int32_t n = d->global_aggregation_group_id_count;
    if (!d->produce_default_output || n <= 0) return 0; /* output.isEmpty() -> null */
    pa_hash_aggregation_desc g = *d;
    g.group_by_count = 0;
    g.group_by_channels = NULL;
    g.hash_channel = -1;
    g.produce_default_output = 0;
    g.global_aggregation_group_id_count = 0;
    orc_hash_agg* fresh = orc_hash_agg_create(&g);
    pa_page row;
    int32_t rc = orc_hash_agg_build_result(fresh, &row);
    if (rc < 0 || row.position_count != 1) {
        orc_hash_agg_destroy(fresh);
        return rc < 0 ? rc : PA_ERR_INVALID_ARGUMENT;
    }
    int32_t nkeys = d->group_by_count, has_hash = d->hash_channel >= 0;
    int32_t ncols = nkeys + has_hash + row.channel_count;
    out->position_count = n;
    out->channel_count = ncols;
    out->columns = (pa_column*)calloc((size_t)ncols, sizeof(pa_column));
    int32_t c = 0;
    for (int32_t k = 0; k < nkeys; k++) {
        col_builder b;
        cb_init(&b, d->input_types[d->group_by_channels[k]], n);
        for (int32_t i = 0; i < n; i++) {
            orc_val v = null_of(b.type);
            if (k == d->group_id_channel) {
                v.is_null = 0;
                v.i = d->global_aggregation_group_ids[i];
            }
            cb_append(&b, &v);
        }
        cb_finish(&b, &out->columns[c++]);
    }
    if (has_hash) {
        col_builder b;
        cb_init(&b, PA_BIGINT, n);
        for (int32_t i = 0; i < n; i++) {
            int64_t result = 0; /* INITIAL_HASH_VALUE */
            for (int32_t k = 0; k < nkeys; k++) {
                result = orc_combine_hash(result, k != d->group_id_channel ? 0 /* NULL_HASH_CODE */ : orc_hash_bigint(d->global_aggregation_group_ids[i]));
            }
            orc_val v;
            memset(&v, 0, sizeof v);
            v.type = PA_BIGINT;
            v.i = result;
            cb_append(&b, &v);
        }
        cb_finish(&b, &out->columns[c++]);
    }
    for (int32_t a = 0; a < row.channel_count; a++) {
        col_builder b;
        cb_init(&b, row.columns[a].type, n);
        orc_val v = col_get(&row.columns[a], 0);
        for (int32_t i = 0; i < n; i++) cb_append(&b, &v);
        cb_finish(&b, &out->columns[c++]);
    }
    orc_free_page(&row);
    orc_hash_agg_destroy(fresh);
    return 0;
}

/* InMemoryHashAggregationBuilder.buildResult (…/InMemoryHashAggregationBuilder.java:244-298): group ids
 * 0..n-1 in order; output columns = keys, ($hashvalue), aggregates (toTypes :419-431).  Output
 * functions: DoubleSumAggregation.java:54-63, LongSumAggregation.java:55-64, AverageAggregations.java:
 * 68-80, CountAggregation.java:51-55.  AggregationOperator.getOutput (TM/operator/AggregationOperator.java:
 * 164-186) is the one-row, no-key form. */
int32_t orc_hash_agg_build_result(orc_hash_agg* a, pa_page* out)
{
    int32_t groups = a->next_group_id;
    int32_t nkeys = a->desc.group_by_count;
    int32_t has_hash = nkeys > 0 && a->desc.hash_channel >= 0;
    int32_t partial = a->desc.step == PA_STEP_PARTIAL;
    int32_t agg_cols = 0;
    for (int32_t k = 0; k < a->desc.aggregate_count; k++) {
        agg_cols += (partial && a->aggs[k].fn != PA_AGG_COUNT && a->aggs[k].fn != PA_AGG_COUNT_STAR) ? 2 : 1;
    }
    int32_t ncols = nkeys + has_hash + agg_cols;
    memset(out, 0, sizeof *out);
    out->position_count = groups;
    out->channel_count = ncols;
    out->mem = PA_MEM_HOST;
    out->columns = (pa_column*)calloc((size_t)(ncols > 0 ? ncols : 1), sizeof(pa_column));
    int32_t c = 0;
    if (nkeys > 0) {
        if (a->bigint_mode) {
            col_builder b;
            cb_init(&b, PA_BIGINT, groups);
            for (int32_t g = 0; g < groups; g++) {
                orc_val v;
                memset(&v, 0, sizeof v);
                v.type = PA_BIGINT;
                if (g == a->null_group_id) {
                    v.is_null = 1;
                }
                else {
                    v.i = a->values_by_group_id[g];
                }
                cb_append(&b, &v);
            }
            cb_finish(&b, &out->columns[c++]);
            if (has_hash) { /* BigintGroupByHash.appendValuesTo: BIGINT.hash(value), NULL -> 0 */
                col_builder hb;
                cb_init(&hb, PA_BIGINT, groups);
                for (int32_t g = 0; g < groups; g++) {
                    orc_val v;
                    memset(&v, 0, sizeof v);
                    v.type = PA_BIGINT;
                    v.i = g == a->null_group_id ? 0 : orc_hash_bigint(a->values_by_group_id[g]);
                    cb_append(&hb, &v);
                }
                cb_finish(&hb, &out->columns[c++]);
            }
        }
        else {
            for (int32_t i = 0; i < nkeys; i++) {
                col_builder b;
                cb_init(&b, a->key_builders[i].type, groups);
                for (int32_t g = 0; g < groups; g++) {
                    orc_val v = builder_get(&a->key_builders[i], g);
                    cb_append(&b, &v);
                }
                cb_finish(&b, &out->columns[c++]);
            }
            if (has_hash) {
                col_builder hb;
                cb_init(&hb, PA_BIGINT, groups);
                for (int32_t g = 0; g < groups; g++) {
                    orc_val v;
                    memset(&v, 0, sizeof v);
                    v.type = PA_BIGINT;
                    v.i = a->raw_hash_by_group[g];
                    cb_append(&hb, &v);
                }
                cb_finish(&hb, &out->columns[c++]);
            }
        }
    }
    for (int32_t k = 0; k < a->desc.aggregate_count; k++) {
        const pa_aggregate* ag = &a->aggs[k];
        const acc_state* st = a->states + (size_t)k * a->state_cap;
        if (partial) {
            /* Step.PARTIAL: the states themselves, flattened: [count BIGINT] (+ [sum DOUBLE | BIGINT]) */
            int value_double = ag->fn == PA_AGG_AVG || type_is_fp(ag->input_type); /* REAL sums / averages: NullableDoubleState / DoubleState */
            int min_max = ag->fn == PA_AGG_MIN || ag->fn == PA_AGG_MAX;
            /* sum / avg over DECIMAL: [count BIGINT, sum DECIMAL(38, s)] -- the flat form of LongDecimalWithOverflow(AndLong)State; a sum
             * whose overflow counter is not zero, or that reached 10^38, does not fit it (NUMERIC_VALUE_OUT_OF_RANGE) */
            int decimal_sum = !min_max && type_is_decimal(ag->input_type) && (ag->fn == PA_AGG_SUM || ag->fn == PA_AGG_AVG);
            for (int part = 0; part < ((ag->fn == PA_AGG_COUNT || ag->fn == PA_AGG_COUNT_STAR) ? 1 : 2); part++) {
                col_builder b;
                /* min / max: [count BIGINT, value of the input type (NULL while no value was seen)] */
                cb_init(&b, part == 0 ? PA_BIGINT : (decimal_sum ? PA_LONG_DECIMAL : (min_max ? ag->input_type : (value_double ? PA_DOUBLE : PA_BIGINT))), groups);
                for (int32_t g = 0; g < groups; g++) {
                    orc_val v;
                    memset(&v, 0, sizeof v);
                    v.type = b.type;
                    if (part == 0) v.i = st[g].count;
                    else if (min_max && !st[g].has_value) v.is_null = 1;
                    else if (decimal_sum) {
                        v.q = st[g].dneg ? -(__int128)st[g].dmag : (__int128)st[g].dmag;
                        if (st[g].overflow != 0 || ld_overflows(v.q)) {
                            a->error = PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE;
                        }
                    }
                    else { v.d = st[g].dsum; v.i = st[g].lsum; v.q = st[g].lsum; v.s = st[g].str; v.slen = st[g].slen; }
                    cb_append(&b, &v);
                }
                cb_finish(&b, &out->columns[c++]);
            }
            continue;
        }
        int32_t out_type;
        switch (ag->fn) {
            case PA_AGG_COUNT_STAR:
            case PA_AGG_COUNT: out_type = PA_BIGINT; break;
            case PA_AGG_AVG: out_type = ag->input_type == PA_REAL ? PA_REAL : (type_is_decimal(ag->input_type) ? ag->input_type : PA_DOUBLE); break; /* RealAverageAggregation.output writes a REAL; avg(DECIMAL(p, s)) is a DECIMAL(p, s) */
            case PA_AGG_SUM: out_type = type_is_decimal(ag->input_type) ? PA_LONG_DECIMAL : ag->input_type; break; /* sum(DECIMAL(p, s)) is a DECIMAL(38, s) */
            default: out_type = ag->input_type; break;
        }
        col_builder b;
        cb_init(&b, out_type, groups);
        for (int32_t g = 0; g < groups; g++) {
            const acc_state* s = &st[g];
            orc_val v;
            memset(&v, 0, sizeof v);
            v.type = out_type;
            switch (ag->fn) {
                case PA_AGG_COUNT_STAR:
                case PA_AGG_COUNT:
                    v.i = s->count;
                    break;
                case PA_AGG_SUM:
                    if (s->count == 0) v.is_null = 1;
                    else if (type_is_decimal(ag->input_type)) { /* DecimalSumAggregation.outputLongDecimal (:177-190) */
                        v.q = s->dneg ? -(__int128)s->dmag : (__int128)s->dmag;
                        if (s->overflow != 0 || ld_overflows(v.q)) a->error = PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE;
                    }
                    else if (type_is_fp(out_type)) v.d = out_type == PA_REAL ? (double)(float)s->dsum : s->dsum; /* RealSumAggregation.output: (float) sum */
                    else v.i = s->lsum;
                    break;
                case PA_AGG_AVG:
                    if (s->count == 0) v.is_null = 1;
                    else if (type_is_decimal(ag->input_type)) { /* DecimalAverageAggregation.outputShortDecimal / outputLongDecimal */
                        if (!ld_average(s, s->count, &v.q) || (out_type == PA_DECIMAL && (v.q > INT64_MAX || v.q < INT64_MIN)) ||
                            (out_type == PA_LONG_DECIMAL && ld_overflows(v.q))) {
                            a->error = PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE;
                        }
                        v.i = (int64_t)v.q;
                    }
                    else if (out_type == PA_REAL) v.d = (double)(float)(s->dsum / (double)s->count); /* RealAverageAggregation.java:150-158 */
                    else v.d = s->dsum / (double)s->count;
                    break;
                default:
                    if (!s->has_value) v.is_null = 1;
                    else { v.d = s->dsum; v.i = s->lsum; v.q = s->lsum; v.s = s->str; v.slen = s->slen; }
                    break;
            }
            cb_append(&b, &v);
        }
        cb_finish(&b, &out->columns[c++]);
    }
    if (a->error) {
        return fail(a->error, "Decimal overflow");
    }
    return 1;
}

/* =====================================================================================
 * Hash join: PagesIndex + PagesHash + ArrayPositionLinks + DefaultPageJoiner
 * ===================================================================================== */
struct orc_join {
    pa_hash_builder_desc desc;
    int32_t* input_types;
    int32_t* join_channels;
    int32_t* output_channels;
    col_builder* cols; /* all build channels concatenated: address == position */
    int32_t positions;
    int32_t hash_size, mask;
    int32_t* key;
    uint8_t* position_to_hashes;
    int32_t* position_links;
    uint8_t* visited; /* OuterLookupSource.OuterPositionTracker.visitedPositions (OuterLookupSource.java:162-215) */
};

orc_join* orc_join_create(const pa_hash_builder_desc* d)
{
    orc_join* j = (orc_join*)calloc(1, sizeof *j);
    j->desc = *d;
    j->input_types = (int32_t*)malloc(sizeof(int32_t) * (size_t)d->input_channel_count);
    memcpy(j->input_types, d->input_types, sizeof(int32_t) * (size_t)d->input_channel_count);
    j->join_channels = (int32_t*)malloc(sizeof(int32_t) * (size_t)d->join_channel_count);
    memcpy(j->join_channels, d->join_channels, sizeof(int32_t) * (size_t)d->join_channel_count);
    j->output_channels = (int32_t*)malloc(sizeof(int32_t) * (size_t)(d->output_channel_count > 0 ? d->output_channel_count : 1));
    if (d->output_channel_count > 0) {
        memcpy(j->output_channels, d->output_channels, sizeof(int32_t) * (size_t)d->output_channel_count);
    }
    j->cols = (col_builder*)calloc((size_t)d->input_channel_count, sizeof(col_builder));
    for (int32_t c = 0; c < d->input_channel_count; c++) {
        cb_init(&j->cols[c], d->input_types[c], 16);
    }
    return j;
}

void orc_join_destroy(orc_join* j)
{
    if (!j) {
        return;
    }
    for (int32_t c = 0; c < j->desc.input_channel_count; c++) {
        free(j->cols[c].values);
        free(j->cols[c].offsets);
        free(j->cols[c].nulls);
    }
    free(j->cols);
    free(j->input_types);
    free(j->join_channels);
    free(j->output_channels);
    free(j->key);
    free(j->position_to_hashes);
    free(j->position_links);
    free(j->visited);
    free(j);
}

/* PagesIndex.addPage (TM/operator/PagesIndex.java:212-241) */
int32_t orc_join_add_build_page(orc_join* j, const pa_page* page)
{
    for (int32_t c = 0; c < j->desc.input_channel_count; c++) {
        for (int32_t p = 0; p < page->position_count; p++) {
            orc_val v = col_get(&page->columns[c], p);
            cb_append(&j->cols[c], &v);
        }
    }
    j->positions += page->position_count;
    return 0;
}

int32_t orc_join_build_positions(const orc_join* j) { return j->positions; }

static int64_t join_build_hash(const orc_join* j, int32_t pos)
{
    if (j->desc.hash_channel >= 0) {
        return builder_get(&j->cols[j->desc.hash_channel], pos).i;
    }
    int64_t result = 0;
    for (int32_t i = 0; i < j->desc.join_channel_count; i++) {
        orc_val v = builder_get(&j->cols[j->join_channels[i]], pos);
        result = orc_combine_hash(result, val_hash(&v));
    }
    return result;
}

/* PagesHashStrategy.positionEqualsPositionIgnoreNulls / positionEqualsRowIgnoreNulls: plain EQUAL on
 * non-null values (SPI/type/DoubleType.java:157-161: left == right) */
static int equals_ignore_nulls(const orc_val* a, const orc_val* b)
{
    switch (a->type) {
        case PA_DOUBLE:
        case PA_REAL: /* SPI/type/RealType.java:101-105 */
            return a->d == b->d;
        case PA_VARCHAR: return a->slen == b->slen && (a->slen == 0 || memcmp(a->s, b->s, (size_t)a->slen) == 0);
        default: return a->i == b->i;
    }
}

/* PagesHash ctor (TM/operator/join/PagesHash.java:54-126) with ArrayPositionLinks.FactoryBuilder.link
 * (TM/operator/join/ArrayPositionLinks.java:45-50).  The 128 KB step batching of the reference only
 * bounds a temporary array; insertion order is ascending position either way. */
int32_t orc_join_build(orc_join* j)
{
    int32_t n = j->positions;
    j->hash_size = orc_array_size(n, 0.75f);
    j->mask = j->hash_size - 1;
    j->key = (int32_t*)malloc(sizeof(int32_t) * (size_t)j->hash_size);
    for (int32_t i = 0; i < j->hash_size; i++) {
        j->key[i] = -1;
    }
    j->position_to_hashes = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    j->position_links = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    j->visited = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int32_t i = 0; i < n; i++) {
        j->position_links[i] = -1;
    }
    for (int32_t real = 0; real < n; real++) {
        int64_t hash = join_build_hash(j, real);
        j->position_to_hashes[real] = (uint8_t)hash;
        int is_null = 0; /* isPositionNull: any join channel NULL (PagesHash.java:95-97) */
        for (int32_t i = 0; i < j->desc.join_channel_count; i++) {
            if (j->cols[j->join_channels[i]].nulls[real]) {
                is_null = 1;
            }
        }
        if (is_null) {
            continue;
        }
        int32_t pos = (int32_t)(orc_murmur3_fmix(hash) & j->mask);
        int32_t real_position = real;
        while (j->key[pos] != -1) {
            int32_t current_key = j->key[pos];
            int eq = ((uint8_t)hash) == j->position_to_hashes[current_key];
            for (int32_t i = 0; eq && i < j->desc.join_channel_count; i++) {
                orc_val l = builder_get(&j->cols[j->join_channels[i]], current_key);
                orc_val r = builder_get(&j->cols[j->join_channels[i]], real_position);
                eq = equals_ignore_nulls(&l, &r);
            }
            if (eq) {
                j->position_links[real_position] = current_key; /* link(left=new, right=head) */
                break;
            }
            pos = (pos + 1) & j->mask;
        }
        j->key[pos] = real_position;
    }
    return 0;
}

int32_t orc_join_tables(const orc_join* j, int32_t* hash_size, int32_t* key, int32_t* position_links)
{
    if (hash_size) {
        *hash_size = j->hash_size;
    }
    if (key) {
        memcpy(key, j->key, sizeof(int32_t) * (size_t)j->hash_size);
    }
    if (position_links) {
        memcpy(position_links, j->position_links, sizeof(int32_t) * (size_t)j->positions);
    }
    return 0;
}

/* PagesHash.getAddressIndex (TM/operator/join/PagesHash.java:158-170) */
static int32_t join_address_index(const orc_join* j, const pa_page* probe, const int32_t* probe_join_channels,
                                  int32_t position, int64_t raw_hash)
{
    int32_t pos = (int32_t)(orc_murmur3_fmix(raw_hash) & j->mask);
    while (j->key[pos] != -1) {
        int32_t left = j->key[pos];
        int eq = j->position_to_hashes[left] == (uint8_t)raw_hash;
        for (int32_t i = 0; eq && i < j->desc.join_channel_count; i++) {
            orc_val l = builder_get(&j->cols[j->join_channels[i]], left);
            orc_val r = col_get(&probe->columns[probe_join_channels[i]], position);
            eq = equals_ignore_nulls(&l, &r);
        }
        if (eq) {
            return left;
        }
        pos = (pos + 1) & j->mask;
    }
    return -1;
}

/* DefaultPageJoiner.processProbe / joinCurrentPosition / advanceProbePosition for an inner join without
 * filter function (TM/operator/join/DefaultPageJoiner.java:236-320), JoinProbe.getCurrentJoinPosition
 * (TM/operator/join/JoinProbe.java:87-117: NULL key never matches), LookupJoinPageBuilder.appendRow/build
 * (TM/operator/join/LookupJoinPageBuilder.java:76-139). */
int32_t orc_join_probe(const orc_join* j, const pa_lookup_join_desc* d, const pa_page* probe,
                       pa_page* out, int32_t** probe_indices, int32_t** build_positions, int32_t* match_count)
{
    int32_t cap = 1024, cnt = 0;
    int32_t* pi = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    for (int32_t position = 0; position < probe->position_count; position++) {
        int has_null = 0;
        for (int32_t i = 0; i < d->join_channel_count; i++) {
            if (col_is_null(&probe->columns[d->probe_join_channels[i]], position)) {
                has_null = 1;
            }
        }
        const int probe_outer = d->join_type == PA_JOIN_PROBE_OUTER || d->join_type == PA_JOIN_FULL_OUTER;
        const int track = d->join_type == PA_JOIN_LOOKUP_OUTER || d->join_type == PA_JOIN_FULL_OUTER;
        if (has_null) {
            if (probe_outer) { /* DefaultPageJoiner.outerJoinCurrentPosition -> appendNullForBuild (:296-303) */
                if (cnt == cap) {
                    cap *= 2;
                    pi = (int32_t*)realloc(pi, sizeof(int32_t) * (size_t)cap);
                    bi = (int32_t*)realloc(bi, sizeof(int32_t) * (size_t)cap);
                }
                pi[cnt] = position;
                bi[cnt] = -1;
                cnt++;
            }
            continue;
        }
        int64_t raw_hash;
        if (d->probe_hash_channel >= 0) {
            raw_hash = col_get(&probe->columns[d->probe_hash_channel], position).i;
        }
        else {
            raw_hash = hash_position(probe, d->join_channel_count, d->probe_join_channels, position);
        }
        int32_t join_position = join_address_index(j, probe, d->probe_join_channels, position, raw_hash);
        int produced = 0;
        while (join_position >= 0 || (probe_outer && !produced)) {
            if (cnt == cap) {
                cap *= 2;
                pi = (int32_t*)realloc(pi, sizeof(int32_t) * (size_t)cap);
                bi = (int32_t*)realloc(bi, sizeof(int32_t) * (size_t)cap);
            }
            pi[cnt] = position;
            bi[cnt] = join_position >= 0 ? join_position : -1;
            cnt++;
            produced = 1;
            if (join_position < 0) {
                break;
            }
            if (track) {
                ((orc_join*)j)->visited[join_position] = 1; /* OuterLookupSource.appendTo -> positionVisited */
            }
            if (d->output_single_match) {
                break; /* DefaultPageJoiner.java:276-278: joinPosition = -1 once the probe row produced a row */
            }
            join_position = j->position_links[join_position]; /* ArrayPositionLinks.next */
        }
    }
    *match_count = cnt;
    if (out != NULL) {
        int32_t ncols = d->probe_output_channel_count + j->desc.output_channel_count;
        memset(out, 0, sizeof *out);
        out->position_count = cnt;
        out->channel_count = ncols;
        out->mem = PA_MEM_HOST;
        out->columns = (pa_column*)calloc((size_t)(ncols > 0 ? ncols : 1), sizeof(pa_column));
        int32_t c = 0;
        for (int32_t i = 0; i < d->probe_output_channel_count; i++) {
            const pa_column* src = &probe->columns[d->probe_output_channels[i]];
            int32_t t = src->type;
            if (src->encoding == PA_DICTIONARY || src->encoding == PA_RLE) {
                int32_t zero = 0;
                t = resolve(src, &zero)->type;
            }
            col_builder b;
            cb_init(&b, t, cnt);
            for (int32_t k = 0; k < cnt; k++) {
                orc_val v = col_get(src, pi[k]);
                cb_append(&b, &v);
            }
            cb_finish(&b, &out->columns[c++]);
        }
        for (int32_t i = 0; i < j->desc.output_channel_count; i++) {
            const col_builder* src = &j->cols[j->output_channels[i]];
            col_builder b;
            cb_init(&b, src->type, cnt);
            for (int32_t k = 0; k < cnt; k++) {
                orc_val v;
                if (bi[k] >= 0) {
                    v = builder_get(src, bi[k]);
                }
                else { /* LookupJoinPageBuilder.appendNullForBuild (:84-99) */
                    memset(&v, 0, sizeof v);
                    v.is_null = 1;
                    v.type = src->type;
                }
                cb_append(&b, &v);
            }
            cb_finish(&b, &out->columns[c++]);
        }
    }
    if (probe_indices) {
        *probe_indices = pi;
    }
    else {
        free(pi);
    }
    if (build_positions) {
        *build_positions = bi;
    }
    else {
        free(bi);
    }
    return 0;
}

/* LookupOuterOperator (TM/operator/join/LookupOuterOperator.java:150-215) over the shared OuterPositionIterator
 * (TM/operator/join/OuterLookupSource.java:120-160): the build positions no probe row was joined with, ascending; probe
 * output channels NULL, then the build output channels. */
int32_t orc_join_outer(const orc_join* j, const pa_lookup_join_desc* d, pa_page* out)
{
    int32_t cnt = 0;
    for (int32_t p = 0; p < j->positions; p++) {
        cnt += j->visited[p] ? 0 : 1;
    }
    int32_t ncols = d->probe_output_channel_count + j->desc.output_channel_count;
    memset(out, 0, sizeof *out);
    out->position_count = cnt;
    out->channel_count = ncols;
    out->mem = PA_MEM_HOST;
    out->columns = (pa_column*)calloc((size_t)(ncols > 0 ? ncols : 1), sizeof(pa_column));
    int32_t c = 0;
    for (int32_t i = 0; i < d->probe_output_channel_count; i++) {
        col_builder b;
        cb_init(&b, d->probe_types[d->probe_output_channels[i]], cnt);
        for (int32_t k = 0; k < cnt; k++) {
            orc_val v;
            memset(&v, 0, sizeof v);
            v.is_null = 1;
            v.type = b.type;
            cb_append(&b, &v);
        }
        cb_finish(&b, &out->columns[c++]);
    }
    for (int32_t i = 0; i < j->desc.output_channel_count; i++) {
        const col_builder* src = &j->cols[j->output_channels[i]];
        col_builder b;
        cb_init(&b, src->type, cnt);
        for (int32_t p = 0; p < j->positions; p++) {
            if (!j->visited[p]) {
                orc_val v = builder_get(src, p);
                cb_append(&b, &v);
            }
        }
        cb_finish(&b, &out->columns[c++]);
    }
    return 0;
}

/* =====================================================================================
 * Synthetic TPC-H-shaped columns.  Not a restatement of io.trino.tpch:tpch:1.1 (un-vendored,
 * SURVEY 8c item 3): a counter-based generator whose value ranges and group proportions follow
 * plugin/trino-tpch/src/main/resources/tpch/statistics/sf1.0/{lineitem,orders,customer}.json.
 * Row r of a column depends only on (seed, column, r, scale).  The device generator in
 * presto_amd/csrc must produce bit-identical columns (tests/test_tpch_gen.py).
 * ===================================================================================== */
static inline uint64_t mix64(uint64_t z) /* splitmix64 output function */
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t rnd(uint64_t seed, uint32_t stream, uint64_t row)
{
    /* row-th output of splitmix64 seeded with seed + stream */
    return mix64(seed + (uint64_t)stream * 0xD1342543DE82EF95ULL + (row + 1) * 0x9E3779B97F4A7C15ULL);
}

enum { S_QTY = 1, S_PRICE = 2, S_DISC = 3, S_TAX = 4, S_SHIP = 5, S_RECEIPT = 6, S_FLAG = 7,
       S_CUST = 8, S_ODATE = 9, S_SEG = 10 };

static inline int64_t tpch_order_count(double sf) { return (int64_t)(1500000.0 * sf); }
static inline int64_t sparse_orderkey(int64_t o) { return (o >> 3) * 32 + (o & 7) + 1; }
static inline int64_t lineitem_order_index(int64_t r, int64_t orders)
{
    static const int8_t order_in_block[28] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4,
                                              5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 6};
    int64_t o = (r / 28) * 7 + order_in_block[r % 28];
    return o < orders ? o : orders - 1;
}
static inline int32_t l_shipdate(uint64_t seed, int64_t r) { return 8036 + (int32_t)(rnd(seed, S_SHIP, (uint64_t)r) % 2526); }
static const char* const SEGMENTS[5] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"};
static const int32_t SEGMENT_LEN[5] = {10, 8, 9, 9, 9};
/* each block of 5 consecutive customers holds the 5 segments in a seeded rotation+stride order, so a
 * block is always 45 bytes and offsets need no prefix scan */
static inline int32_t c_segment(uint64_t seed, int64_t r)
{
    uint64_t u = rnd(seed, S_SEG, (uint64_t)(r / 5));
    int32_t start = (int32_t)(u % 5);
    int32_t stride = 1 + (int32_t)((u >> 8) % 4);
    return (start + stride * (int32_t)(r % 5)) % 5;
}

int32_t orc_tpch_generate(int32_t column, double sf, int64_t first_row, int64_t row_count,
                          uint64_t seed, void* values, int32_t* offsets)
{
    int64_t orders = tpch_order_count(sf);
    for (int64_t k = 0; k < row_count; k++) {
        int64_t r = first_row + k;
        switch (column) {
            case PA_L_ORDERKEY:
                ((int64_t*)values)[k] = sparse_orderkey(lineitem_order_index(r, orders));
                break;
            case PA_L_QUANTITY:
                ((double*)values)[k] = (double)(1 + rnd(seed, S_QTY, (uint64_t)r) % 50);
                break;
            case PA_L_EXTENDEDPRICE:
                ((double*)values)[k] = (double)(90100 + rnd(seed, S_PRICE, (uint64_t)r) % 10404851ULL) / 100.0;
                break;
            case PA_L_DISCOUNT:
                ((double*)values)[k] = (double)(rnd(seed, S_DISC, (uint64_t)r) % 11) / 100.0;
                break;
            case PA_L_TAX:
                ((double*)values)[k] = (double)(rnd(seed, S_TAX, (uint64_t)r) % 9) / 100.0;
                break;
            case PA_L_SHIPDATE:
                ((int32_t*)values)[k] = l_shipdate(seed, r);
                break;
            case PA_L_RETURNFLAG: {
                int32_t receipt = l_shipdate(seed, r) + 1 + (int32_t)(rnd(seed, S_RECEIPT, (uint64_t)r) % 30);
                char f = 'N';
                if (receipt <= 9298) {
                    f = (rnd(seed, S_FLAG, (uint64_t)r) & 1) ? 'R' : 'A';
                }
                ((uint8_t*)values)[k] = (uint8_t)f;
                offsets[k] = (int32_t)k;
                offsets[k + 1] = (int32_t)(k + 1);
                break;
            }
            case PA_L_LINESTATUS:
                ((uint8_t*)values)[k] = (uint8_t)(l_shipdate(seed, r) > 9298 ? 'O' : 'F');
                offsets[k] = (int32_t)k;
                offsets[k + 1] = (int32_t)(k + 1);
                break;
            case PA_O_ORDERKEY:
                ((int64_t*)values)[k] = sparse_orderkey(r);
                break;
            case PA_O_CUSTKEY: {
                uint64_t customers = (uint64_t)(150000.0 * sf);
                uint64_t usable = customers - customers / 3; /* keys that are not multiples of 3 */
                uint64_t u = rnd(seed, S_CUST, (uint64_t)r) % usable;
                ((int64_t*)values)[k] = (int64_t)((u / 2) * 3 + (u % 2) + 1);
                break;
            }
            case PA_O_ORDERDATE:
                ((int32_t*)values)[k] = 8035 + (int32_t)(rnd(seed, S_ODATE, (uint64_t)r) % 2406);
                break;
            case PA_O_SHIPPRIORITY:
                ((int32_t*)values)[k] = 0;
                break;
            case PA_C_CUSTKEY:
                ((int64_t*)values)[k] = r + 1;
                break;
            case PA_C_MKTSEGMENT: {
                /* byte offset of row r relative to first_row's block-aligned base */
                int64_t base_block = first_row / 5;
                int64_t off = (r / 5 - base_block) * 45;
                for (int64_t q = (r / 5) * 5; q < r; q++) {
                    off += SEGMENT_LEN[c_segment(seed, q)];
                }
                if (first_row % 5 != 0) {
                    return fail(PA_ERR_INVALID_ARGUMENT, "mktsegment first_row must be a multiple of 5");
                }
                int32_t s = c_segment(seed, r);
                memcpy((uint8_t*)values + off, SEGMENTS[s], (size_t)SEGMENT_LEN[s]);
                offsets[k] = (int32_t)off;
                offsets[k + 1] = (int32_t)(off + SEGMENT_LEN[s]);
                break;
            }
            default:
                return fail(PA_ERR_INVALID_ARGUMENT, "unknown tpch column");
        }
    }
    return 0;
}

/* =====================================================================================
 * Hand-written Q6 twin for the CPU baseline: TpchQuery6Filter.filter + generated field*field
 * projection + DoubleSumAggregation over selected rows (BM/HandTpchQuery6.java:95-141; SQL form
 * BM/SqlTpchQuery6.java:26-32 multiplies extendedprice * discount).
 * ===================================================================================== */
int32_t orc_q6(const int32_t* shipdate, const double* discount, const double* quantity,
               const double* extendedprice, int64_t n, double* sum, int64_t* count)
{
    double s = 0.0;
    int64_t c = 0;
    for (int64_t i = 0; i < n; i++) {
        if (shipdate[i] >= 8766 && shipdate[i] < 9131 && discount[i] >= 0.05 && discount[i] <= 0.07 && quantity[i] < 24.0) {
            s = s + extendedprice[i] * discount[i];
            c++;
        }
    }
    *sum = s;
    *count = c;
    return 0;
}

/* =====================================================================================
 * Q1 twin for the CPU baseline: TpchQuery1Operator.filterAndProjectRowOriented
 * (BM/HandTpchQuery1.java:241-330, with the SQL form's 8 aggregates of BM/SqlTpchQuery1.java:29-36)
 * feeding HashAggregationOperator -> InMemoryHashAggregationBuilder -> MultiChannelGroupByHash in
 * pages of up to 8192 projected rows, exactly as one Driver thread does.  `agg` must have been created
 * for the projected layout (VARCHAR, VARCHAR, DOUBLE x5) with group-by channels {0,1}.
 * ===================================================================================== */
int32_t orc_q1_add(orc_hash_agg* agg, const uint8_t* returnflag, const int32_t* rf_offsets, const uint8_t* linestatus,
                   const int32_t* ls_offsets, const double* quantity, const double* extendedprice, const double* discount,
                   const double* tax, const int32_t* shipdate, int64_t n)
{
    enum { BATCH = 8192 };
    uint8_t rf[BATCH * 4], ls[BATCH * 4];
    int32_t rfo[BATCH + 1], lso[BATCH + 1];
    static __thread double qty[BATCH], price[BATCH], disc_price[BATCH], charge[BATCH], disc[BATCH];
    int64_t i = 0;
    while (i < n) {
        int32_t m = 0;
        int32_t rfb = 0, lsb = 0;
        rfo[0] = 0;
        lso[0] = 0;
        for (; i < n && m < BATCH; i++) {
            if (shipdate[i] <= 10471) {
                int32_t l1 = rf_offsets[i + 1] - rf_offsets[i], l2 = ls_offsets[i + 1] - ls_offsets[i];
                if (l1 > 4 || l2 > 4) {
                    return fail(PA_ERR_NOT_SUPPORTED, "q1 twin expects VARCHAR(1..4) flags");
                }
                memcpy(rf + rfb, returnflag + rf_offsets[i], (size_t)l1);
                rfb += l1;
                rfo[m + 1] = rfb;
                memcpy(ls + lsb, linestatus + ls_offsets[i], (size_t)l2);
                lsb += l2;
                lso[m + 1] = lsb;
                qty[m] = quantity[i];
                price[m] = extendedprice[i];
                disc_price[m] = extendedprice[i] * (1 - discount[i]);
                charge[m] = extendedprice[i] * (1 - discount[i]) * (1 + tax[i]);
                disc[m] = discount[i];
                m++;
            }
        }
        if (m == 0) {
            continue;
        }
        pa_column cols[7];
        memset(cols, 0, sizeof cols);
        cols[0].type = PA_VARCHAR; cols[0].encoding = PA_VARWIDTH; cols[0].values = rf; cols[0].offsets = rfo;
        cols[1].type = PA_VARCHAR; cols[1].encoding = PA_VARWIDTH; cols[1].values = ls; cols[1].offsets = lso;
        const double* d[5] = {qty, price, disc_price, charge, disc};
        for (int c = 0; c < 5; c++) {
            cols[2 + c].type = PA_DOUBLE;
            cols[2 + c].encoding = PA_FLAT;
            cols[2 + c].values = d[c];
        }
        pa_page page = {m, 7, cols, PA_MEM_HOST, 0, NULL, NULL};
        int32_t rc = orc_hash_agg_add_page(agg, &page, NULL);
        if (rc < 0) {
            return rc;
        }
    }
    return 0;
}

/* =====================================================================================
 * OrderByOperator / TopNOperator over flat columns: the CPU twins bench.py times beside the device operators (the
 * Python restatements of oracle.py -- oracle.order_by, oracle.topn -- are the parity checkers for small pages; these
 * follow the same reference code at a speed worth timing).
 *
 * orc_sort_positions_bigint: PagesIndex.sort (core/trino-main/src/main/java/io/trino/operator/PagesIndex.java:386-396) ->
 *   PagesIndexOrdering.quickSort (…/operator/PagesIndexOrdering.java:52-142; forked from fastutil: insertion sort below
 *   SMALL = 7, median of 3 above, pseudo-median of 9 above MEDIUM = 40, the "moving target" fat partition) over the row
 *   addresses, compared through one BIGINT sort channel (SimplePagesIndexComparator -> SortOrder.compareBlockValue,
 *   ASC_NULLS_LAST, no NULLs).  positions[] must hold 0..n-1 on entry (the arrival order); sorted on return.
 * ===================================================================================== */
typedef struct {
    const int64_t* keys;
    int32_t* pos;
} sort_ctx;

static inline int sort_cmp(const sort_ctx* c, int32_t i, int32_t j)
{
    const int64_t a = c->keys[c->pos[i]], b = c->keys[c->pos[j]];
    return a < b ? -1 : (a > b ? 1 : 0);
}
static inline void sort_swap(const sort_ctx* c, int32_t i, int32_t j)
{
    const int32_t t = c->pos[i];
    c->pos[i] = c->pos[j];
    c->pos[j] = t;
}
static int32_t sort_median3(const sort_ctx* c, int32_t a, int32_t b, int32_t d)
{
    const int ab = sort_cmp(c, a, b), ac = sort_cmp(c, a, d), bc = sort_cmp(c, b, d);
    return ab < 0 ? (bc < 0 ? b : ac < 0 ? d : a) : (bc > 0 ? b : ac > 0 ? d : a);
}
static void sort_vector_swap(const sort_ctx* c, int32_t from, int32_t l, int32_t s)
{
    for (int32_t i = 0; i < s; i++, from++, l++) sort_swap(c, from, l);
}
static void sort_quick(const sort_ctx* x, int32_t from, int32_t to)
{
    enum { SMALL = 7, MEDIUM = 40 };
    const int32_t len = to - from;
    if (len < SMALL) {
        for (int32_t i = from; i < to; i++) {
            for (int32_t j = i; j > from && sort_cmp(x, j - 1, j) > 0; j--) sort_swap(x, j, j - 1);
        }
        return;
    }
    int32_t m = from + len / 2;
    if (len > SMALL) {
        int32_t l = from, n = to - 1;
        if (len > MEDIUM) {
            const int32_t s = len / 8;
            l = sort_median3(x, l, l + s, l + 2 * s);
            m = sort_median3(x, m - s, m, m + s);
            n = sort_median3(x, n - 2 * s, n - s, n);
        }
        m = sort_median3(x, l, m, n);
    }
    int32_t a = from, b = a, c = to - 1, d = c;
    for (;;) {
        int comparison;
        while (b <= c && (comparison = sort_cmp(x, b, m)) <= 0) {
            if (comparison == 0) {
                if (a == m) m = b;
                else if (b == m) m = a;
                sort_swap(x, a++, b);
            }
            b++;
        }
        while (c >= b && (comparison = sort_cmp(x, c, m)) >= 0) {
            if (comparison == 0) {
                if (c == m) m = d;
                else if (d == m) m = c;
                sort_swap(x, c, d--);
            }
            c--;
        }
        if (b > c) break;
        if (b == m) m = d;
        else if (c == m) m = c;
        sort_swap(x, b++, c--);
    }
    int32_t s, n = to;
    s = a - from < b - a ? a - from : b - a;
    sort_vector_swap(x, from, b - s, s);
    s = d - c < n - d - 1 ? d - c : n - d - 1;
    sort_vector_swap(x, b, n - s, s);
    if ((s = b - a) > 1) sort_quick(x, from, from + s);
    if ((s = d - c) > 1) sort_quick(x, n - s, n);
}
int32_t orc_sort_positions_bigint(const int64_t* keys, int32_t n, int32_t* positions)
{
    if (n < 0 || (n > 0 && (!keys || !positions))) return fail(PA_ERR_INVALID_ARGUMENT, "orc_sort_positions_bigint: null argument");
    sort_ctx c = {keys, positions};
    sort_quick(&c, 0, n);
    return 0;
}

/* orc_topn_double_desc_bigint_asc: TopNProcessor (…/operator/TopNProcessor.java:35-110) with one group: the `limit` first rows
 * under (channel 0 DOUBLE DESC_NULLS_LAST, channel 1 BIGINT ASC_NULLS_LAST) -- Q3's ORDER BY revenue DESC, o_orderdate --, kept
 * in a binary heap whose root is the WORST kept row: a new row enters only when it precedes the root (GroupedTopNBuilder's
 * RowHeap discipline).  DOUBLE order is Double.compare.  out_positions: the kept rows, best first. */
static inline int dbl_compare(double a, double b)   /* Double.compare: -0.0 < 0.0, NaN greatest and equal to itself */
{
    if (a < b) return -1;
    if (a > b) return 1;
    int64_t x, y;
    memcpy(&x, &a, 8);
    memcpy(&y, &b, 8);
    if (a != a) x = 0x7ff8000000000000LL;
    if (b != b) y = 0x7ff8000000000000LL;
    return x == y ? 0 : (x < y ? -1 : 1);
}
typedef struct {
    const double* v;
    const int64_t* k;
} topn_ctx;
static inline int topn_cmp(const topn_ctx* c, int32_t i, int32_t j)   /* < 0: row i precedes row j in the output order */
{
    const int d = dbl_compare(c->v[i], c->v[j]);
    if (d != 0) return -d;
    return c->k[i] < c->k[j] ? -1 : (c->k[i] > c->k[j] ? 1 : 0);
}
static int topn_qsort_cmp_ctx_set;
static const topn_ctx* topn_qsort_ctx;
static int topn_qsort_cmp(const void* a, const void* b)
{
    const int c = topn_cmp(topn_qsort_ctx, *(const int32_t*)a, *(const int32_t*)b);
    return c != 0 ? c : (*(const int32_t*)a < *(const int32_t*)b ? -1 : 1);
}
int32_t orc_topn_double_desc_bigint_asc(const double* values, const int64_t* keys, int64_t n, int32_t limit, int32_t* out_positions)
{
    if (limit < 0 || n < 0 || n > 0x7fffffffLL) return fail(PA_ERR_INVALID_ARGUMENT, "orc_topn: bad sizes");
    if (limit == 0 || n == 0) return 0;
    topn_ctx c = {values, keys};
    int32_t* heap = (int32_t*)malloc(sizeof(int32_t) * (size_t)limit);
    int32_t size = 0;
    for (int32_t r = 0; r < (int32_t)n; r++) {
        if (size < limit) {
            int32_t i = size++;
            heap[i] = r;
            while (i > 0) {   /* sift up: a parent must not precede its children (max-heap on the output order) */
                const int32_t p = (i - 1) / 2;
                if (topn_cmp(&c, heap[p], heap[i]) >= 0) break;
                const int32_t t = heap[p]; heap[p] = heap[i]; heap[i] = t;
                i = p;
            }
        }
        else if (topn_cmp(&c, r, heap[0]) < 0) {
            heap[0] = r;
            int32_t i = 0;
            for (;;) {
                int32_t l = 2 * i + 1, rr = l + 1, w = i;
                if (l < size && topn_cmp(&c, heap[l], heap[w]) > 0) w = l;
                if (rr < size && topn_cmp(&c, heap[rr], heap[w]) > 0) w = rr;
                if (w == i) break;
                const int32_t t = heap[w]; heap[w] = heap[i]; heap[i] = t;
                i = w;
            }
        }
    }
    (void)topn_qsort_cmp_ctx_set;
    topn_qsort_ctx = &c;
    qsort(heap, (size_t)size, sizeof(int32_t), topn_qsort_cmp);
    memcpy(out_positions, heap, sizeof(int32_t) * (size_t)size);
    free(heap);
    return size;
}

/* =====================================================================================
 * Known-answer hooks of the DECIMAL accumulator states: TestDecimalSumAggregation / TestDecimalAverageAggregation look INSIDE the
 * state (LongDecimalWithOverflowState.getOverflow / getLongDecimal) after every input, which the operator surface does not show.
 * values: n long decimals in the reference's 16-byte layout, added in order to an empty state (inputLongDecimal); *overflow, state
 * (16 bytes) = the state afterwards; *average (16 bytes, when non-NULL) = DecimalAverageAggregation.average at scale 0.
 * ===================================================================================== */
int32_t orc_decimal_state_after(const uint8_t* values, int32_t n, int64_t* overflow, uint8_t* state, uint8_t* average)
{
    acc_state s;
    memset(&s, 0, sizeof s);
    for (int32_t i = 0; i < n; i++) {
        s.overflow += ld_add_with_overflow(&s, ld_read(values + 16 * (size_t)i));
        s.count++;
    }
    *overflow = s.overflow;
    /* the Slice itself: magnitude + sign bit (zero may be negative: -2^126 + -2^126 leaves "-0" with one underflow) */
    uint64_t lo = (uint64_t)s.dmag, hi = (uint64_t)(s.dmag >> 64) | (s.dneg ? 0x8000000000000000ULL : 0);
    memcpy(state, &lo, 8);
    memcpy(state + 8, &hi, 8);
    if (average) {
        __int128 avg = 0;
        if (n == 0 || !ld_average(&s, s.count, &avg)) {
            return fail(PA_ERR_NUMERIC_VALUE_OUT_OF_RANGE, "average does not fit");
        }
        ld_write(average, avg);
    }
    return 0;
}
