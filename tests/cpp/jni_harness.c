/* jni_harness.c -- runs jni/presto_amd_jni.c WITHOUT a JVM.
 *
 * The image this repository is written in has no JDK, so the JNI shim (464 lines of offset unpacking, descriptor filling and
 * status -> exception mapping) had never executed.  This harness is a JVM stand-in for exactly what the shim touches: it
 * implements the members of JNINativeInterface_ that jni/stub/jni.h declares over plain malloc'ed arrays, direct buffers,
 * strings and exception objects, links the shim + libpresto_amd.so (+ liboracle.so as the checker) and drives the reference's
 * operator known-answer cases through the Java_io_trino_gpu_GpuNative_* symbols the way java/io/trino/gpu does it:
 *   PinnedPagePool.stage      block arrays copied to offsets inside ONE pinned direct buffer (hostMallocPinned)
 *   RowExpressionSerializer   RowExpression trees flattened children-first into parallel arrays (newExpression)
 *   GpuOperator               addInput / getOutput (long[2 + 6 c] + wrapAddress) / finish / isFinished / close
 *   GpuNativeException        pending after a failing native, with the pa_status
 * Cases (expected values from the reference's tests, and the oracle on the same pages):
 *   fp-1    TestFilterAndProjectOperator.java:78-124       hagg-1  TestHashAggregationOperator.java:160-219 (device subset)
 *   join-1  join/TestHashJoinOperator.java:192-229         q6      one fused scan-filter-project-aggregate page (HandTpchQuery6.java:95-141)
 *   errors  DIVISION_BY_ZERO -> GpuNativeException(status), a device-output operator refused by getOutput
 * Array elements are handed to the shim as COPIES (as a JVM may) and every Get must meet its Release: the count is checked.
 * Built by __graft_entry__.build(); executed by tests/test_gpu_jni_harness.py.  Exit code 0 = all cases pass. */
#include <jni.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "presto_amd.h"
#include "presto_oracle.h"

/* ---- the natives under test (jni/presto_amd_jni.c) ---- */
jint Java_io_trino_gpu_GpuNative_abiVersion(JNIEnv*, jclass);
void Java_io_trino_gpu_GpuNative_init(JNIEnv*, jclass, jint);
jobject Java_io_trino_gpu_GpuNative_hostMallocPinned(JNIEnv*, jclass, jlong);
void Java_io_trino_gpu_GpuNative_hostFreePinned(JNIEnv*, jclass, jobject);
jlong Java_io_trino_gpu_GpuNative_newExpression(JNIEnv*, jclass, jint, jintArray, jintArray, jintArray, jintArray, jintArray, jintArray, jintArray, jintArray,
                                                jlongArray, jdoubleArray, jobjectArray, jintArray);
void Java_io_trino_gpu_GpuNative_freeExpression(JNIEnv*, jclass, jlong);
jlong Java_io_trino_gpu_GpuNative_createFilterProject(JNIEnv*, jclass, jintArray, jintArray, jlong, jlongArray, jlong, jint, jint);
jlong Java_io_trino_gpu_GpuNative_createHashAggregation(JNIEnv*, jclass, jintArray, jintArray, jintArray, jint, jint, jintArray, jintArray, jintArray, jintArray,
                                                        jint, jint);
jlong Java_io_trino_gpu_GpuNative_createFusedAggregation(JNIEnv*, jclass, jintArray, jintArray, jlong, jlongArray, jintArray, jintArray, jint, jintArray,
                                                         jintArray, jintArray, jintArray, jint, jint);
jlong Java_io_trino_gpu_GpuNative_createLookupSource(JNIEnv*, jclass);
void Java_io_trino_gpu_GpuNative_destroyLookupSource(JNIEnv*, jclass, jlong);
jlong Java_io_trino_gpu_GpuNative_createHashBuilder(JNIEnv*, jclass, jlong, jintArray, jintArray, jint, jintArray, jint);
jlong Java_io_trino_gpu_GpuNative_createLookupJoin(JNIEnv*, jclass, jlong, jintArray, jintArray, jint, jintArray, jint, jboolean, jboolean, jint, jlong);
jboolean Java_io_trino_gpu_GpuNative_needsInput(JNIEnv*, jclass, jlong);
jboolean Java_io_trino_gpu_GpuNative_isBlocked(JNIEnv*, jclass, jlong);
jboolean Java_io_trino_gpu_GpuNative_isFinished(JNIEnv*, jclass, jlong);
void Java_io_trino_gpu_GpuNative_finish(JNIEnv*, jclass, jlong);
void Java_io_trino_gpu_GpuNative_close(JNIEnv*, jclass, jlong);
void Java_io_trino_gpu_GpuNative_addInput(JNIEnv*, jclass, jlong, jint, jint, jintArray, jintArray, jlongArray, jlongArray, jlongArray, jlongArray, jintArray,
                                          jintArray, jobject, jboolean);
jlongArray Java_io_trino_gpu_GpuNative_getOutput(JNIEnv*, jclass, jlong);
jobject Java_io_trino_gpu_GpuNative_wrapAddress(JNIEnv*, jclass, jlong, jlong);

static int failures = 0;
#define EXPECT(cond, ...)                                           \
    do {                                                            \
        if (!(cond)) {                                              \
            failures++;                                             \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);    \
            fprintf(stderr, __VA_ARGS__);                           \
            fprintf(stderr, "\n");                                  \
        }                                                           \
    } while (0)

/* =========================================== the JVM stand-in =========================================== */
enum { K_INTS = 1, K_LONGS, K_DOUBLES, K_BYTES, K_OBJECTS, K_BUFFER, K_STRING, K_CLASS, K_EXCEPTION };
struct _jobject {
    int kind;
    jsize len;          /* arrays: elements */
    void* data;         /* arrays: elements; buffer: address; string / class: chars */
    jlong capacity;     /* buffer */
    jint status;        /* exception */
    struct _jobject* message;
};
static struct _jobject* pending_exception = 0;
static int outstanding_elements = 0;     /* Get*ArrayElements not yet released */
static struct _jmethodID { int dummy; } exception_ctor;

static jobject new_object(int kind, jsize len, size_t bytes)
{
    jobject o = (jobject)calloc(1, sizeof(struct _jobject));
    o->kind = kind;
    o->len = len;
    o->data = calloc(bytes ? bytes : 1, 1);
    return o;
}
static size_t element_size(int kind) { return kind == K_INTS ? 4 : kind == K_BYTES ? 1 : 8; }
static jarray new_array(int kind, jsize n, const void* init)
{
    jarray a = new_object(kind, n, (size_t)n * element_size(kind));
    if (init) memcpy(a->data, init, (size_t)n * element_size(kind));
    return a;
}
static jintArray ints(jsize n, const jint* v) { return new_array(K_INTS, n, v); }
static jlongArray longs(jsize n, const jlong* v) { return new_array(K_LONGS, n, v); }
static jdoubleArray doubles(jsize n, const jdouble* v) { return new_array(K_DOUBLES, n, v); }

static jclass jvm_FindClass(JNIEnv* env, const char* name)
{
    EXPECT(strcmp(name, "io/trino/gpu/GpuNativeException") == 0, "FindClass(%s): only the exception class is looked up", name);
    jclass c = new_object(K_CLASS, 0, strlen(name) + 1);
    strcpy((char*)c->data, name);
    return c;
}
static jmethodID jvm_GetMethodID(JNIEnv* env, jclass cls, const char* name, const char* sig)
{
    EXPECT(cls && cls->kind == K_CLASS && strcmp(name, "<init>") == 0 && strcmp(sig, "(ILjava/lang/String;)V") == 0,
           "GetMethodID(%s, %s): GpuNativeException(int, String) expected", name, sig);
    return &exception_ctor;
}
static jobject jvm_NewObject(JNIEnv* env, jclass cls, jmethodID ctor, ...)
{
    va_list ap;
    va_start(ap, ctor);
    jobject e = new_object(K_EXCEPTION, 0, 0);
    e->status = va_arg(ap, jint);
    e->message = va_arg(ap, jobject);
    va_end(ap);
    EXPECT(ctor == &exception_ctor && e->message && e->message->kind == K_STRING, "NewObject: (status, message) expected");
    return e;
}
static jint jvm_Throw(JNIEnv* env, jthrowable t)
{
    EXPECT(t && t->kind == K_EXCEPTION, "Throw of something that is no exception");
    pending_exception = t;
    return 0;
}
static jstring jvm_NewStringUTF(JNIEnv* env, const char* s)
{
    jstring o = new_object(K_STRING, (jsize)strlen(s), strlen(s) + 1);
    strcpy((char*)o->data, s);
    return o;
}
static jsize jvm_GetArrayLength(JNIEnv* env, jarray a) { return a->len; }
static jobject jvm_GetObjectArrayElement(JNIEnv* env, jobjectArray a, jsize i)
{
    EXPECT(a->kind == K_OBJECTS && i >= 0 && i < a->len, "GetObjectArrayElement out of bounds");
    return ((jobject*)a->data)[i];
}
/* elements are handed out as copies, as a JVM that cannot pin the array does */
static void* get_elements(jarray a, int kind)
{
    EXPECT(a && a->kind == kind, "Get<Type>ArrayElements on an array of another type");
    void* copy = malloc((size_t)a->len * element_size(kind) + 1);
    memcpy(copy, a->data, (size_t)a->len * element_size(kind));
    outstanding_elements++;
    return copy;
}
static void release_elements(jarray a, int kind, void* elems, jint mode)
{
    EXPECT(a && a->kind == kind, "Release<Type>ArrayElements on an array of another type");
    if (mode != JNI_ABORT) memcpy(a->data, elems, (size_t)a->len * element_size(kind));   /* 0: copy back and free */
    free(elems);
    outstanding_elements--;
}
static jint* jvm_GetIntArrayElements(JNIEnv* env, jintArray a, jboolean* c) { return (jint*)get_elements(a, K_INTS); }
static jlong* jvm_GetLongArrayElements(JNIEnv* env, jlongArray a, jboolean* c) { return (jlong*)get_elements(a, K_LONGS); }
static jdouble* jvm_GetDoubleArrayElements(JNIEnv* env, jdoubleArray a, jboolean* c) { return (jdouble*)get_elements(a, K_DOUBLES); }
static jbyte* jvm_GetByteArrayElements(JNIEnv* env, jbyteArray a, jboolean* c) { return (jbyte*)get_elements(a, K_BYTES); }
static void jvm_ReleaseIntArrayElements(JNIEnv* env, jintArray a, jint* e, jint m) { release_elements(a, K_INTS, e, m); }
static void jvm_ReleaseLongArrayElements(JNIEnv* env, jlongArray a, jlong* e, jint m) { release_elements(a, K_LONGS, e, m); }
static void jvm_ReleaseDoubleArrayElements(JNIEnv* env, jdoubleArray a, jdouble* e, jint m) { release_elements(a, K_DOUBLES, e, m); }
static void jvm_ReleaseByteArrayElements(JNIEnv* env, jbyteArray a, jbyte* e, jint m) { release_elements(a, K_BYTES, e, m); }
static void jvm_SetLongArrayRegion(JNIEnv* env, jlongArray a, jsize start, jsize n, const jlong* v)
{
    EXPECT(a->kind == K_LONGS && start >= 0 && start + n <= a->len, "SetLongArrayRegion out of bounds");
    memcpy((jlong*)a->data + start, v, (size_t)n * 8);
}
static void jvm_SetIntArrayRegion(JNIEnv* env, jintArray a, jsize start, jsize n, const jint* v)
{
    EXPECT(a->kind == K_INTS && start >= 0 && start + n <= a->len, "SetIntArrayRegion out of bounds");
    memcpy((jint*)a->data + start, v, (size_t)n * 4);
}
static jlongArray jvm_NewLongArray(JNIEnv* env, jsize n) { return new_array(K_LONGS, n, 0); }
static jobject jvm_NewDirectByteBuffer(JNIEnv* env, void* address, jlong capacity)
{
    jobject b = (jobject)calloc(1, sizeof(struct _jobject));
    b->kind = K_BUFFER;
    b->data = address;
    b->capacity = capacity;
    return b;
}
static void* jvm_GetDirectBufferAddress(JNIEnv* env, jobject b)
{
    EXPECT(b && b->kind == K_BUFFER, "GetDirectBufferAddress of something that is no direct buffer");
    return b->data;
}
static jlong jvm_GetDirectBufferCapacity(JNIEnv* env, jobject b) { return b->capacity; }

static const struct JNINativeInterface_ jvm_functions = {
    jvm_FindClass, jvm_GetMethodID, jvm_NewObject, jvm_Throw, jvm_NewStringUTF, jvm_GetArrayLength, jvm_GetObjectArrayElement,
    jvm_GetIntArrayElements, jvm_GetLongArrayElements, jvm_GetDoubleArrayElements, jvm_GetByteArrayElements,
    jvm_ReleaseIntArrayElements, jvm_ReleaseLongArrayElements, jvm_ReleaseDoubleArrayElements, jvm_ReleaseByteArrayElements,
    jvm_SetLongArrayRegion, jvm_SetIntArrayRegion, jvm_NewLongArray, jvm_NewDirectByteBuffer, jvm_GetDirectBufferAddress, jvm_GetDirectBufferCapacity,
};
static JNIEnv jvm_env = &jvm_functions;
static JNIEnv* env = &jvm_env;

/* the status of a pending GpuNativeException (0 = none), cleared -- what a `catch` does */
static jint take_exception(char* message, size_t cap)
{
    if (!pending_exception) return 0;
    jint status = pending_exception->status;
    if (message) snprintf(message, cap, "%s", (const char*)pending_exception->message->data);
    pending_exception = 0;
    return status;
}
#define NO_EXCEPTION(what)                                                            \
    do {                                                                              \
        char m_[512];                                                                 \
        jint s_ = take_exception(m_, sizeof m_);                                      \
        EXPECT(s_ == 0, "%s threw GpuNativeException(%d, %s)", what, (int)s_, m_);    \
    } while (0)

/* =========================================== the Java side, in C =========================================== */
/* RowExpressionSerializer: nodes appended children-first, a node's children named by indices into `args` */
typedef struct {
    jint kind[32], op[32], type[32], type_param[32], channel[32], is_null[32], nargs[32], first_arg[32], args[64];
    jlong i64[32];
    jdouble f64[32];
    jint n, na;
} expr_builder;
static jint e_field(expr_builder* b, jint channel, jint type)
{
    jint i = b->n++;
    b->kind[i] = PA_EXPR_INPUT_REF; b->type[i] = type; b->channel[i] = channel;
    return i;
}
static jint e_long(expr_builder* b, jlong v, jint type)
{
    jint i = b->n++;
    b->kind[i] = PA_EXPR_CONSTANT; b->type[i] = type; b->i64[i] = v;
    return i;
}
static jint e_double(expr_builder* b, jdouble v)
{
    jint i = b->n++;
    b->kind[i] = PA_EXPR_CONSTANT; b->type[i] = PA_DOUBLE; b->f64[i] = v;
    return i;
}
static jint e_node(expr_builder* b, jint kind, jint op, jint type, jint nargs, const jint* children)
{
    jint i = b->n++;
    b->kind[i] = kind; b->op[i] = op; b->type[i] = type; b->nargs[i] = nargs; b->first_arg[i] = b->na;
    for (jint k = 0; k < nargs; k++) b->args[b->na++] = children[k];
    return i;
}
static jint e_call2(expr_builder* b, jint op, jint type, jint l, jint r)
{
    jint c[2] = {l, r};
    return e_node(b, PA_EXPR_CALL, op, type, 2, c);
}
static jlong e_finish(expr_builder* b, jint root)
{
    jobjectArray strings = new_object(K_OBJECTS, b->n, (size_t)b->n * sizeof(jobject));   /* no VARCHAR constants here: all null */
    jlong h = Java_io_trino_gpu_GpuNative_newExpression(env, 0, root, ints(b->n, b->kind), ints(b->n, b->op), ints(b->n, b->type), ints(b->n, b->type_param),
                                                        ints(b->n, b->channel),
                                                        ints(b->n, b->is_null), ints(b->n, b->nargs), ints(b->n, b->first_arg), longs(b->n, b->i64),
                                                        doubles(b->n, b->f64), strings, ints(b->na, b->args));
    NO_EXCEPTION("newExpression");
    return h;
}

/* a Block as the JVM holds it: long[] / int[] / byte[] values (or Slice bytes + int[] offsets), boolean[] valueIsNull */
typedef struct {
    int32_t type;
    int32_t positions;
    const void* values;
    int64_t value_bytes;
    const int32_t* offsets;       /* VARCHAR */
    const uint8_t* nulls;         /* may be 0 */
} jblock;

static int64_t width_of(int32_t type) { return type == PA_BOOLEAN ? 1 : (type == PA_INTEGER || type == PA_DATE || type == PA_REAL) ? 4 : 8; }
static int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }

/* PinnedPagePool.stage + GpuOperator.addInput: every array of every block at an aligned offset of one pinned direct buffer */
static jobject add_input(jlong op, const jblock* blocks, jint channels, jint positions)
{
    jint types[16], encodings[16], dict_channel[16], dict_size[16];
    jlong voff[16], ooff[16], noff[16], ioff[16];
    int64_t at = 0;
    for (jint c = 0; c < channels; c++) {
        const jblock* b = &blocks[c];
        types[c] = b->type;
        encodings[c] = b->type == PA_VARCHAR ? PA_VARWIDTH : PA_FLAT;
        dict_channel[c] = -1; dict_size[c] = 0; ioff[c] = -1;
        voff[c] = at; at = align16(at + (b->type == PA_VARCHAR ? b->value_bytes : width_of(b->type) * positions));
        if (b->type == PA_VARCHAR) { ooff[c] = at; at = align16(at + 4 * ((int64_t)positions + 1)); } else ooff[c] = -1;
        if (b->nulls) { noff[c] = at; at = align16(at + positions); } else noff[c] = -1;
    }
    jobject buffer = Java_io_trino_gpu_GpuNative_hostMallocPinned(env, 0, at > 0 ? at : 16);
    NO_EXCEPTION("hostMallocPinned");
    char* base = (char*)buffer->data;
    for (jint c = 0; c < channels; c++) {
        const jblock* b = &blocks[c];
        memcpy(base + voff[c], b->values, (size_t)(b->type == PA_VARCHAR ? b->value_bytes : width_of(b->type) * positions));
        if (b->type == PA_VARCHAR) memcpy(base + ooff[c], b->offsets, 4 * ((size_t)positions + 1));
        if (b->nulls) memcpy(base + noff[c], b->nulls, (size_t)positions);
    }
    EXPECT(Java_io_trino_gpu_GpuNative_needsInput(env, 0, op), "needsInput before addInput");
    Java_io_trino_gpu_GpuNative_addInput(env, 0, op, positions, channels, ints(channels, types), ints(channels, encodings), longs(channels, voff),
                                         longs(channels, ooff), longs(channels, noff), longs(channels, ioff), ints(channels, dict_channel),
                                         ints(channels, dict_size), buffer, 1 /* the pool keeps the slab until the operator is closed */);
    return buffer;
}

/* GpuOperator.getOutput: long[2 + 6 c] -> arrays copied out of the wrapped addresses */
typedef struct {
    int32_t type;
    int64_t* longs;       /* BIGINT / DOUBLE bits */
    int32_t* ints;        /* INTEGER / DATE / REAL / VARCHAR offsets */
    uint8_t* bytes;       /* BOOLEAN values / VARCHAR bytes */
    uint8_t* nulls;
} oblock;
typedef struct { int32_t positions, channels; oblock blocks[16]; } opage;

static int get_output(jlong op, opage* page)
{
    memset(page, 0, sizeof *page);
    jlongArray out = Java_io_trino_gpu_GpuNative_getOutput(env, 0, op);
    if (pending_exception || !out) return 0;
    const jlong* v = (const jlong*)out->data;
    page->positions = (int32_t)v[0];
    page->channels = (int32_t)v[1];
    EXPECT(out->len == 2 + 6 * page->channels && page->channels <= 16, "getOutput: long[2 + 6 channels]");
    for (int32_t c = 0; c < page->channels; c++) {
        const jlong* r = v + 2 + 6 * c;
        oblock* b = &page->blocks[c];
        b->type = (int32_t)r[0];
        jobject values = Java_io_trino_gpu_GpuNative_wrapAddress(env, 0, r[1], r[2]);
        EXPECT(values->capacity == r[2], "wrapAddress capacity");
        if (b->type == PA_VARCHAR) {
            jobject offsets = Java_io_trino_gpu_GpuNative_wrapAddress(env, 0, r[3], 4 * ((jlong)page->positions + 1));
            b->ints = (int32_t*)malloc(4 * ((size_t)page->positions + 1));
            memcpy(b->ints, offsets->data, 4 * ((size_t)page->positions + 1));
            EXPECT(r[2] == b->ints[page->positions] - 0 || page->positions == 0, "VARCHAR value bytes = last offset");
            b->bytes = (uint8_t*)malloc((size_t)r[2] + 1);
            memcpy(b->bytes, values->data, (size_t)r[2]);
        }
        else {
            EXPECT(r[2] == width_of(b->type) * page->positions, "channel %d: %ld value bytes for %d positions of type %d", c, (long)r[2], page->positions, b->type);
            void* copy = malloc((size_t)r[2] + 1);
            memcpy(copy, values->data, (size_t)r[2]);
            if (width_of(b->type) == 8) b->longs = (int64_t*)copy;
            else if (width_of(b->type) == 4) b->ints = (int32_t*)copy;
            else b->bytes = (uint8_t*)copy;
        }
        if (r[4]) {
            jobject nulls = Java_io_trino_gpu_GpuNative_wrapAddress(env, 0, r[4], page->positions);
            b->nulls = (uint8_t*)malloc((size_t)page->positions + 1);
            memcpy(b->nulls, nulls->data, (size_t)page->positions);
        }
    }
    return 1;
}
static int slice_equals(const oblock* b, int32_t i, const char* s)
{
    const int32_t len = b->ints[i + 1] - b->ints[i];
    return len == (int32_t)strlen(s) && memcmp(b->bytes + b->ints[i], s, (size_t)len) == 0;
}

/* SequencePageBuilder columns (TT/SequencePageBuilder.java:44-84): VARCHAR = decimal string of start + i, BIGINT = start + i */
typedef struct { char* bytes; int32_t* offsets; int64_t n_bytes; } varchar_column;
static varchar_column sequence_varchar(int32_t n, int64_t start)
{
    varchar_column c;
    c.bytes = (char*)malloc((size_t)n * 12 + 16);
    c.offsets = (int32_t*)malloc(4 * ((size_t)n + 1));
    int32_t at = 0;
    for (int32_t i = 0; i < n; i++) {
        c.offsets[i] = at;
        at += sprintf(c.bytes + at, "%lld", (long long)(start + i));
    }
    c.offsets[n] = at;
    c.n_bytes = at;
    return c;
}
static int64_t* sequence_bigint(int32_t n, int64_t start)
{
    int64_t* v = (int64_t*)malloc(8 * (size_t)n + 8);
    for (int32_t i = 0; i < n; i++) v[i] = start + i;
    return v;
}
static jblock varchar_block(const varchar_column* c, int32_t n) { jblock b = {PA_VARCHAR, n, c->bytes, c->n_bytes, c->offsets, 0}; return b; }
static jblock flat_block(int32_t type, const void* v, int32_t n, const uint8_t* nulls) { jblock b = {type, n, v, 0, 0, nulls}; return b; }

/* ---- fp-1: filter field1 <= 9, projections (field0, field1 + 5) over the sequence page (VARCHAR @0, BIGINT @0), 100 rows ---- */
static void test_filter_and_project(void)
{
    expr_builder f = {0}, p0 = {0}, p1 = {0};
    jlong filter = e_finish(&f, e_call2(&f, PA_OP_LESS_THAN_OR_EQUAL, PA_BOOLEAN, e_field(&f, 1, PA_BIGINT), e_long(&f, 9, PA_BIGINT)));
    jlong proj[2];
    proj[0] = e_finish(&p0, e_field(&p0, 0, PA_VARCHAR));
    proj[1] = e_finish(&p1, e_call2(&p1, PA_OP_ADD, PA_BIGINT, e_field(&p1, 1, PA_BIGINT), e_long(&p1, 5, PA_BIGINT)));
    jint types[2] = {PA_VARCHAR, PA_BIGINT};
    jlong op = Java_io_trino_gpu_GpuNative_createFilterProject(env, 0, ints(2, types), ints(0, 0), filter, longs(2, proj), 0, 0, PA_MEM_HOST);
    NO_EXCEPTION("createFilterProject");
    varchar_column s = sequence_varchar(100, 0);
    int64_t* v = sequence_bigint(100, 0);
    jblock blocks[2] = {varchar_block(&s, 100), flat_block(PA_BIGINT, v, 100, 0)};
    jobject slab = add_input(op, blocks, 2, 100);
    NO_EXCEPTION("addInput");
    opage out;
    EXPECT(get_output(op, &out) && out.positions == 10 && out.channels == 2, "fp-1: one page of 10 rows, got %d", out.positions);
    NO_EXCEPTION("getOutput");
    for (int32_t i = 0; i < out.positions && i < 10; i++) {
        char expect[8];
        sprintf(expect, "%d", i);
        EXPECT(slice_equals(&out.blocks[0], i, expect) && out.blocks[1].longs[i] == i + 5, "fp-1 row %d", i);
    }
    /* and against the oracle's PageProcessor restatement of the same expressions on the same page */
    pa_expr_node fn[3] = {{PA_EXPR_INPUT_REF, 0, PA_BIGINT, 1}, {PA_EXPR_CONSTANT, 0, PA_BIGINT}, {PA_EXPR_CALL, PA_OP_LESS_THAN_OR_EQUAL, PA_BOOLEAN, 0, 0, 2, 0}};
    fn[1].i64 = 9;
    int32_t fargs[2] = {0, 1};
    pa_expr fe = {3, 2, fn, 2, 0, fargs};
    pa_expr_node p0n[1] = {{PA_EXPR_INPUT_REF, 0, PA_VARCHAR, 0}};
    pa_expr_node p1n[3] = {{PA_EXPR_INPUT_REF, 0, PA_BIGINT, 1}, {PA_EXPR_CONSTANT, 0, PA_BIGINT}, {PA_EXPR_CALL, PA_OP_ADD, PA_BIGINT, 0, 0, 2, 0}};
    p1n[1].i64 = 5;
    pa_expr pe[2] = {{1, 0, p0n, 0, 0, fargs}, {3, 2, p1n, 2, 0, fargs}};
    pa_column cols[2];
    memset(cols, 0, sizeof cols);
    cols[0].type = PA_VARCHAR; cols[0].encoding = PA_VARWIDTH; cols[0].values = s.bytes; cols[0].offsets = s.offsets;
    cols[1].type = PA_BIGINT; cols[1].encoding = PA_FLAT; cols[1].values = v;
    pa_page in = {100, 2, cols, PA_MEM_HOST, 0}, ref;
    memset(&ref, 0, sizeof ref);
    EXPECT(orc_filter_project(&in, &fe, 2, pe, &ref) == 1 && ref.position_count == out.positions, "fp-1: oracle row count");
    for (int32_t i = 0; i < ref.position_count && i < out.positions; i++) {
        EXPECT(((const int64_t*)ref.columns[1].values)[i] == out.blocks[1].longs[i], "fp-1 oracle bigint row %d", i);
        EXPECT(ref.columns[0].offsets[i + 1] - ref.columns[0].offsets[i] == out.blocks[0].ints[i + 1] - out.blocks[0].ints[i], "fp-1 oracle varchar row %d", i);
    }
    orc_free_page(&ref);
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    EXPECT(get_output(op, &out) == 0 && Java_io_trino_gpu_GpuNative_isFinished(env, 0, op), "fp-1: finished after finish");
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    NO_EXCEPTION("finish / close");
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);
    NO_EXCEPTION("hostFreePinned");
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, filter);
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj[0]);
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj[1]);
}

/* ---- status -> exception: 100 / field1 over a page holding 0 is the reference's DIVISION_BY_ZERO (BigintOperators.java:88-110) ---- */
static void test_exceptions(void)
{
    expr_builder p = {0};
    jlong proj = e_finish(&p, e_call2(&p, PA_OP_DIVIDE, PA_BIGINT, e_long(&p, 100, PA_BIGINT), e_field(&p, 1, PA_BIGINT)));
    jint types[2] = {PA_VARCHAR, PA_BIGINT};
    jlong op = Java_io_trino_gpu_GpuNative_createFilterProject(env, 0, ints(2, types), ints(0, 0), 0, longs(1, &proj), 0, 0, PA_MEM_HOST);
    NO_EXCEPTION("createFilterProject");
    varchar_column s = sequence_varchar(10, 0);
    int64_t* v = sequence_bigint(10, 0);
    jblock blocks[2] = {varchar_block(&s, 10), flat_block(PA_BIGINT, v, 10, 0)};
    jobject slab = add_input(op, blocks, 2, 10);
    opage out;
    get_output(op, &out);
    char message[512] = "";
    jint status = take_exception(message, sizeof message);
    EXPECT(status == PA_ERR_DIVISION_BY_ZERO, "DIVISION_BY_ZERO expected, got %d (%s)", (int)status, message);
    EXPECT(strlen(message) > 0, "the exception carries pa_last_error()");
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    take_exception(0, 0);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);

    /* a descriptor the device path refuses: NOT_SUPPORTED at creation, handle 0 (the planner then keeps the Java operator) */
    expr_builder q = {0};
    jlong bad = e_finish(&q, e_call2(&q, PA_OP_ADD, PA_VARCHAR, e_field(&q, 0, PA_VARCHAR), e_field(&q, 0, PA_VARCHAR)));
    jlong none = Java_io_trino_gpu_GpuNative_createFilterProject(env, 0, ints(2, types), ints(0, 0), 0, longs(1, &bad), 0, 0, PA_MEM_HOST);
    status = take_exception(message, sizeof message);
    EXPECT(none == 0 && status < 0, "VARCHAR + VARCHAR must be refused at creation (status %d)", (int)status);

    /* getOutput of an operator created with device output: refused, nothing dereferenced on the host */
    expr_builder r = {0};
    jlong ident = e_finish(&r, e_field(&r, 1, PA_BIGINT));
    jlong dev = Java_io_trino_gpu_GpuNative_createFilterProject(env, 0, ints(2, types), ints(0, 0), 0, longs(1, &ident), 0, 0, PA_MEM_DEVICE);
    NO_EXCEPTION("createFilterProject (device output)");
    slab = add_input(dev, blocks, 2, 10);
    NO_EXCEPTION("addInput");
    EXPECT(get_output(dev, &out) == 0, "getOutput of a PA_MEM_DEVICE operator returns nothing");
    status = take_exception(message, sizeof message);
    EXPECT(status == PA_ERR_ILLEGAL_STATE, "getOutput of a PA_MEM_DEVICE operator: ILLEGAL_STATE expected, got %d", (int)status);
    Java_io_trino_gpu_GpuNative_close(env, 0, dev);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);
    take_exception(0, 0);
}

/* ---- hagg-1 (the columns the device path covers): 3 sequence pages of 40 000 rows, key = VARCHAR @0; count(*), sum(bigint),
 *      avg(bigint), count(varchar), count(boolean) -> per key str(i): 3, 3 i, (double) i, 3, 3 ---- */
static void test_hash_aggregation(void)
{
    const int32_t n = 40000;
    jint types[5] = {PA_VARCHAR, PA_VARCHAR, PA_VARCHAR, PA_BIGINT, PA_BOOLEAN};
    jint group_by[1] = {1};
    jint fns[5] = {PA_AGG_COUNT_STAR, PA_AGG_SUM, PA_AGG_AVG, PA_AGG_COUNT, PA_AGG_COUNT};
    jint inputs[5] = {-1, 3, 3, 0, 4}, masks[5] = {-1, -1, -1, -1, -1}, in_types[5] = {PA_BIGINT, PA_BIGINT, PA_BIGINT, PA_VARCHAR, PA_BOOLEAN};
    jlong op = Java_io_trino_gpu_GpuNative_createHashAggregation(env, 0, ints(5, types), ints(0, 0), ints(1, group_by), -1, PA_STEP_SINGLE, ints(5, fns),
                                                                 ints(5, inputs), ints(5, masks), ints(5, in_types), 100000, PA_MEM_HOST);
    NO_EXCEPTION("createHashAggregation");
    jobject slabs[3];
    for (int p = 0; p < 3; p++) {
        varchar_column a = sequence_varchar(n, 100), key = sequence_varchar(n, 0), c = sequence_varchar(n, 100000 * (p + 1));
        int64_t* v = sequence_bigint(n, 0);
        uint8_t* b = (uint8_t*)malloc((size_t)n);
        for (int32_t i = 0; i < n; i++) b[i] = (uint8_t)((500 + i) % 2 == 0);
        jblock blocks[5] = {varchar_block(&a, n), varchar_block(&key, n), varchar_block(&c, n), flat_block(PA_BIGINT, v, n, 0), flat_block(PA_BOOLEAN, b, n, 0)};
        while (Java_io_trino_gpu_GpuNative_isBlocked(env, 0, op) && !Java_io_trino_gpu_GpuNative_needsInput(env, 0, op)) {}
        slabs[p] = add_input(op, blocks, 5, n);
        NO_EXCEPTION("addInput");
    }
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    NO_EXCEPTION("finish");
    uint8_t* seen = (uint8_t*)calloc((size_t)n, 1);
    int64_t rows = 0;
    opage out;
    for (int guard = 0; guard < 1000 && !Java_io_trino_gpu_GpuNative_isFinished(env, 0, op); guard++) {
        if (!get_output(op, &out)) continue;
        EXPECT(out.channels == 6, "hagg-1: key + 5 aggregates");
        for (int32_t i = 0; i < out.positions; i++, rows++) {
            char key[16] = "";
            const int32_t len = out.blocks[0].ints[i + 1] - out.blocks[0].ints[i];
            memcpy(key, out.blocks[0].bytes + out.blocks[0].ints[i], (size_t)(len < 15 ? len : 15));
            const int64_t k = atoll(key);
            EXPECT(k >= 0 && k < n && !seen[k], "hagg-1: key %s twice or out of range", key);
            if (k >= 0 && k < n) seen[k] = 1;
            double avg;
            memcpy(&avg, &out.blocks[3].longs[i], 8);
            EXPECT(out.blocks[1].longs[i] == 3 && out.blocks[2].longs[i] == 3 * k && avg == (double)k && out.blocks[4].longs[i] == 3 && out.blocks[5].longs[i] == 3,
                   "hagg-1 key %s: %ld %ld %g %ld %ld", key, (long)out.blocks[1].longs[i], (long)out.blocks[2].longs[i], avg, (long)out.blocks[4].longs[i],
                   (long)out.blocks[5].longs[i]);
        }
    }
    NO_EXCEPTION("getOutput");
    EXPECT(rows == n, "hagg-1: %d groups expected, got %ld", n, (long)rows);
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    for (int p = 0; p < 3; p++) Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slabs[p]);
    NO_EXCEPTION("close");
}

/* ---- join-1: build (VARCHAR, BIGINT, BIGINT) 10 rows @20, 30, 40; probe 1000 rows @0, 1000, 2000; key channel 0 ---- */
static void test_join(void)
{
    jint types[3] = {PA_VARCHAR, PA_BIGINT, PA_BIGINT}, key[1] = {0}, all[3] = {0, 1, 2};
    jlong bridge = Java_io_trino_gpu_GpuNative_createLookupSource(env, 0);
    jlong build = Java_io_trino_gpu_GpuNative_createHashBuilder(env, 0, bridge, ints(3, types), ints(1, key), -1, ints(3, all), 10);
    jlong join = Java_io_trino_gpu_GpuNative_createLookupJoin(env, 0, bridge, ints(3, types), ints(1, key), -1, ints(3, all), PA_JOIN_INNER, 0, 0, PA_MEM_HOST, 0);
    NO_EXCEPTION("join factories");
    EXPECT(Java_io_trino_gpu_GpuNative_isBlocked(env, 0, join) && !Java_io_trino_gpu_GpuNative_needsInput(env, 0, join), "the probe waits for the lookup source");
    varchar_column bs = sequence_varchar(10, 20);
    int64_t *b1 = sequence_bigint(10, 30), *b2 = sequence_bigint(10, 40);
    jblock bb[3] = {varchar_block(&bs, 10), flat_block(PA_BIGINT, b1, 10, 0), flat_block(PA_BIGINT, b2, 10, 0)};
    jobject s1 = add_input(build, bb, 3, 10);
    Java_io_trino_gpu_GpuNative_finish(env, 0, build);
    NO_EXCEPTION("build");
    EXPECT(!Java_io_trino_gpu_GpuNative_isBlocked(env, 0, join), "the lookup source was lent");
    varchar_column ps = sequence_varchar(1000, 0);
    int64_t *p1 = sequence_bigint(1000, 1000), *p2 = sequence_bigint(1000, 2000);
    jblock pb[3] = {varchar_block(&ps, 1000), flat_block(PA_BIGINT, p1, 1000, 0), flat_block(PA_BIGINT, p2, 1000, 0)};
    jobject s2 = add_input(join, pb, 3, 1000);
    NO_EXCEPTION("probe addInput");
    opage out;
    EXPECT(get_output(join, &out) && out.positions == 10 && out.channels == 6, "join-1: 10 rows of 6 channels, got %d x %d", out.positions, out.channels);
    NO_EXCEPTION("probe getOutput");
    for (int32_t i = 0; i < out.positions && i < 10; i++) {
        char k[8];
        sprintf(k, "%d", 20 + i);
        EXPECT(slice_equals(&out.blocks[0], i, k) && slice_equals(&out.blocks[3], i, k), "join-1 keys row %d", i);
        EXPECT(out.blocks[1].longs[i] == 1020 + i && out.blocks[2].longs[i] == 2020 + i && out.blocks[4].longs[i] == 30 + i && out.blocks[5].longs[i] == 40 + i,
               "join-1 row %d", i);
    }
    Java_io_trino_gpu_GpuNative_finish(env, 0, join);
    Java_io_trino_gpu_GpuNative_close(env, 0, join);
    Java_io_trino_gpu_GpuNative_close(env, 0, build);
    Java_io_trino_gpu_GpuNative_destroyLookupSource(env, 0, bridge);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, s1);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, s2);
    NO_EXCEPTION("join teardown");
}

/* ---- one fused Q6 page: 8766 <= shipdate < 9131 AND 0.05 <= discount <= 0.07 AND quantity < 24 -> sum(extendedprice * discount), count(*) ---- */
static void test_fused_q6(void)
{
    const int32_t n = 60175;   /* the lineitem rows of TPC-H tiny */
    int32_t* shipdate = (int32_t*)malloc(4 * (size_t)n);
    double *discount = (double*)malloc(8 * (size_t)n), *quantity = (double*)malloc(8 * (size_t)n), *price = (double*)malloc(8 * (size_t)n);
    const uint64_t seed = 0x5EED0000;
    /* column ids: include/presto_amd.h pa_tpch_column */
    orc_tpch_generate(PA_L_SHIPDATE, 0.01, 0, n, seed, shipdate, 0);
    orc_tpch_generate(PA_L_DISCOUNT, 0.01, 0, n, seed, discount, 0);
    orc_tpch_generate(PA_L_QUANTITY, 0.01, 0, n, seed, quantity, 0);
    orc_tpch_generate(PA_L_EXTENDEDPRICE, 0.01, 0, n, seed, price, 0);
    double ref_sum = 0;
    int64_t ref_count = 0;
    orc_q6(shipdate, discount, quantity, price, n, &ref_sum, &ref_count);

    expr_builder f = {0}, p = {0};
    jint conj[5];
    conj[0] = e_call2(&f, PA_OP_GREATER_THAN_OR_EQUAL, PA_BOOLEAN, e_field(&f, 0, PA_DATE), e_long(&f, 8766, PA_DATE));
    conj[1] = e_call2(&f, PA_OP_LESS_THAN, PA_BOOLEAN, e_field(&f, 0, PA_DATE), e_long(&f, 9131, PA_DATE));
    conj[2] = e_call2(&f, PA_OP_GREATER_THAN_OR_EQUAL, PA_BOOLEAN, e_field(&f, 1, PA_DOUBLE), e_double(&f, 0.05));
    conj[3] = e_call2(&f, PA_OP_LESS_THAN_OR_EQUAL, PA_BOOLEAN, e_field(&f, 1, PA_DOUBLE), e_double(&f, 0.07));
    conj[4] = e_call2(&f, PA_OP_LESS_THAN, PA_BOOLEAN, e_field(&f, 2, PA_DOUBLE), e_double(&f, 24.0));
    jlong filter = e_finish(&f, e_node(&f, PA_EXPR_SPECIAL, PA_FORM_AND, PA_BOOLEAN, 5, conj));
    jlong proj = e_finish(&p, e_call2(&p, PA_OP_MULTIPLY, PA_DOUBLE, e_field(&p, 3, PA_DOUBLE), e_field(&p, 1, PA_DOUBLE)));
    jint types[4] = {PA_DATE, PA_DOUBLE, PA_DOUBLE, PA_DOUBLE}, ptypes[1] = {PA_DOUBLE};
    jint fns[2] = {PA_AGG_SUM, PA_AGG_COUNT_STAR}, inputs[2] = {0, -1}, masks[2] = {-1, -1}, in_types[2] = {PA_DOUBLE, PA_BIGINT};
    jlong op = Java_io_trino_gpu_GpuNative_createFusedAggregation(env, 0, ints(4, types), ints(0, 0), filter, longs(1, &proj), ints(1, ptypes), ints(0, 0),
                                                                  PA_STEP_SINGLE, ints(2, fns), ints(2, inputs), ints(2, masks), ints(2, in_types), 1, PA_MEM_HOST);
    NO_EXCEPTION("createFusedAggregation");
    /* 8192-row pages, as a Driver delivers them */
    jobject slabs[16];
    int pages = 0;
    for (int32_t at = 0; at < n; at += 8192, pages++) {
        const int32_t m = n - at < 8192 ? n - at : 8192;
        jblock blocks[4] = {flat_block(PA_DATE, shipdate + at, m, 0), flat_block(PA_DOUBLE, discount + at, m, 0), flat_block(PA_DOUBLE, quantity + at, m, 0),
                            flat_block(PA_DOUBLE, price + at, m, 0)};
        while (Java_io_trino_gpu_GpuNative_isBlocked(env, 0, op) && !Java_io_trino_gpu_GpuNative_needsInput(env, 0, op)) {}
        slabs[pages] = add_input(op, blocks, 4, m);
        NO_EXCEPTION("addInput");
    }
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    opage out;
    int got = 0;
    for (int guard = 0; guard < 1000 && !got; guard++) got = get_output(op, &out);
    NO_EXCEPTION("getOutput");
    EXPECT(got && out.positions == 1 && out.channels == 2, "q6: one row of (sum, count)");
    if (got && out.positions == 1) {
        double sum;
        memcpy(&sum, &out.blocks[0].longs[0], 8);
        EXPECT(out.blocks[1].longs[0] == ref_count && ref_count > 500, "q6 count %ld != oracle %ld", (long)out.blocks[1].longs[0], (long)ref_count);
        EXPECT(fabs(sum - ref_sum) <= 1e-9 * fabs(ref_sum), "q6 sum %.17g != oracle %.17g", sum, ref_sum);
    }
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    for (int i = 0; i < pages; i++) Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slabs[i]);
    NO_EXCEPTION("close");
}

int main(void)
{
    EXPECT(Java_io_trino_gpu_GpuNative_abiVersion(env, 0) == PA_ABI_VERSION, "abiVersion");
    Java_io_trino_gpu_GpuNative_init(env, 0, 0);
    char message[512];
    jint status = take_exception(message, sizeof message);
    if (status) {
        fprintf(stderr, "init: GpuNativeException(%d, %s)\n", (int)status, message);
        return 2;
    }
    test_filter_and_project();
    test_exceptions();
    test_hash_aggregation();
    test_join();
    test_fused_q6();
    EXPECT(outstanding_elements == 0, "%d Get<Type>ArrayElements without their Release", outstanding_elements);
    EXPECT(pending_exception == 0, "an exception was left pending");
    if (failures) {
        fprintf(stderr, "%d failure(s)\n", failures);
        return 1;
    }
    printf("jni harness: all cases pass\n");
    return 0;
}
