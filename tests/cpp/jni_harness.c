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
 *   fp-1    TestFilterAndProjectOperator.java:78-124       hagg-1  TestHashAggregationOperator.java:160-219 (all six aggregates)
 *   join-1  join/TestHashJoinOperator.java:192-229         q6      one fused scan-filter-project-aggregate page (HandTpchQuery6.java:95-141)
 *   globals TestHashAggregationOperator.java:221-272       scan    TestScanFilterAndProjectOperator.java:98-224 over a fake GpuPageSource whose
 *           nextPage / loadBlock / close the shim calls back (CallObjectMethod through GetJavaVM / GetEnv), lazy loads counted
 *   errors  DIVISION_BY_ZERO -> GpuNativeException(status), a device-output operator refused by getOutput
 *   and every other native at least once: createAggregation / createOrderBy / createTopN + setOutputTopNHint, createDynamicFilterSource +
 *   dynamicFilterPoll, memorySetLimit / memoryStats / memoryBytes / deviceCount, getOutputSerialized -> addInputSerialized (PagesSerde frames,
 *   LZ4), addInput with retention 2 + drainReleased, createFusedJoin, setDynamicFilter, commUniqueId / commCreate / exchangeCreate /
 *   createPartitionedOutput / createExchangeSource / exchangeDestroy / commDestroy on a world of one rank
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
jlong Java_io_trino_gpu_GpuNative_createHashAggregation(JNIEnv*, jclass, jintArray, jintArray, jintArray, jintArray, jint, jint, jint, jboolean, jintArray, jintArray,
                                                        jintArray, jintArray, jint, jlong, jint, jint);
jlong Java_io_trino_gpu_GpuNative_createAggregation(JNIEnv*, jclass, jintArray, jint, jintArray, jintArray, jintArray, jintArray, jint, jint);
jlong Java_io_trino_gpu_GpuNative_createOrderBy(JNIEnv*, jclass, jintArray, jintArray, jintArray, jintArray, jint);
jlong Java_io_trino_gpu_GpuNative_createDynamicFilterSource(JNIEnv*, jclass, jintArray, jintArray, jint, jint, jlong);
jlongArray Java_io_trino_gpu_GpuNative_dynamicFilterPoll(JNIEnv*, jclass, jlong, jint);
jlong Java_io_trino_gpu_GpuNative_createScanFilterProject(JNIEnv*, jclass, jobject, jintArray, jintArray, jlong, jlongArray, jlong, jint, jint);
jlongArray Java_io_trino_gpu_GpuNative_scanStats(JNIEnv*, jclass, jlong);
void Java_io_trino_gpu_GpuNative_memorySetLimit(JNIEnv*, jclass, jlong);
jlongArray Java_io_trino_gpu_GpuNative_memoryStats(JNIEnv*, jclass);
jint Java_io_trino_gpu_GpuNative_drainReleased(JNIEnv*, jclass, jlongArray);
jlong Java_io_trino_gpu_GpuNative_getOutputSerialized(JNIEnv*, jclass, jlong, jobject, jboolean);
void Java_io_trino_gpu_GpuNative_addInputSerialized(JNIEnv*, jclass, jlong, jobject, jlong, jintArray);
jlong Java_io_trino_gpu_GpuNative_bufferAddress(JNIEnv*, jclass, jobject);
jint Java_io_trino_gpu_GpuNative_deviceCount(JNIEnv*, jclass);
jlong Java_io_trino_gpu_GpuNative_memoryBytes(JNIEnv*, jclass, jlong);
jlong Java_io_trino_gpu_GpuNative_createTopN(JNIEnv*, jclass, jintArray, jint, jintArray, jintArray, jint);
jboolean Java_io_trino_gpu_GpuNative_setDynamicFilter(JNIEnv*, jclass, jlong, jint, jlong);
jboolean Java_io_trino_gpu_GpuNative_setOutputTopNHint(JNIEnv*, jclass, jlong, jlong, jintArray, jintArray);
jlong Java_io_trino_gpu_GpuNative_createFusedJoin(JNIEnv*, jclass, jlong, jintArray, jintArray, jlong, jlongArray, jintArray, jintArray, jintArray, jintArray, jintArray,
                                                  jint, jintArray, jintArray, jintArray, jintArray, jint, jint);
void Java_io_trino_gpu_GpuNative_commUniqueId(JNIEnv*, jclass, jbyteArray);
jlong Java_io_trino_gpu_GpuNative_commCreate(JNIEnv*, jclass, jbyteArray, jint, jint);
void Java_io_trino_gpu_GpuNative_commDestroy(JNIEnv*, jclass, jlong);
jlong Java_io_trino_gpu_GpuNative_exchangeCreate(JNIEnv*, jclass, jlong, jintArray, jintArray, jint, jint);
void Java_io_trino_gpu_GpuNative_exchangeDestroy(JNIEnv*, jclass, jlong);
jlong Java_io_trino_gpu_GpuNative_createPartitionedOutput(JNIEnv*, jclass, jlong);
jlong Java_io_trino_gpu_GpuNative_createExchangeSource(JNIEnv*, jclass, jlong, jint);
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
                                          jintArray, jobject, jint, jlong);
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
enum { K_INTS = 1, K_LONGS, K_DOUBLES, K_BYTES, K_OBJECTS, K_BUFFER, K_STRING, K_CLASS, K_EXCEPTION, K_PAGE_SOURCE };
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
static struct _jmethodID { int dummy; } exception_ctor, source_next_page, source_load_block, source_close;
static int global_refs = 0;              /* NewGlobalRef not yet deleted */

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
    EXPECT(cls && cls->kind == K_CLASS, "GetMethodID on something that is no class");
    if (strcmp((const char*)cls->data, "io/trino/gpu/GpuPageSource") == 0) {   /* the three methods the shim calls back */
        if (strcmp(name, "nextPage") == 0 && strcmp(sig, "()[J") == 0) return &source_next_page;
        if (strcmp(name, "loadBlock") == 0 && strcmp(sig, "(I)[J") == 0) return &source_load_block;
        if (strcmp(name, "close") == 0 && strcmp(sig, "()V") == 0) return &source_close;
        EXPECT(0, "GetMethodID(GpuPageSource.%s%s): no such method", name, sig);
        return 0;
    }
    EXPECT(strcmp(name, "<init>") == 0 && strcmp(sig, "(ILjava/lang/String;)V") == 0, "GetMethodID(%s, %s): GpuNativeException(int, String) expected", name, sig);
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
static jboolean jvm_ExceptionCheck(JNIEnv* env) { return pending_exception != 0; }
static jclass jvm_GetObjectClass(JNIEnv* env, jobject o)
{
    EXPECT(o && o->kind == K_PAGE_SOURCE, "GetObjectClass: only the page source's class is asked for");
    jclass c = new_object(K_CLASS, 0, 64);
    strcpy((char*)c->data, "io/trino/gpu/GpuPageSource");
    return c;
}
static jobject jvm_NewGlobalRef(JNIEnv* env, jobject o) { global_refs++; return o; }
static void jvm_DeleteGlobalRef(JNIEnv* env, jobject o) { global_refs--; }
/* the Java methods of the fake GpuPageSource (defined with the scan cases below) */
static jobject page_source_next_page(jobject source);
static jobject page_source_load_block(jobject source, jint channel);
static void page_source_close(jobject source);
static jobject jvm_CallObjectMethod(JNIEnv* env, jobject o, jmethodID m, ...)
{
    EXPECT(o && o->kind == K_PAGE_SOURCE, "CallObjectMethod on something that is no GpuPageSource");
    if (m == &source_next_page) return page_source_next_page(o);
    if (m == &source_load_block) {
        va_list ap;
        va_start(ap, m);
        const jint channel = va_arg(ap, jint);
        va_end(ap);
        return page_source_load_block(o, channel);
    }
    EXPECT(0, "CallObjectMethod: unknown method");
    return 0;
}
static void jvm_CallVoidMethod(JNIEnv* env, jobject o, jmethodID m, ...)
{
    EXPECT(o && o->kind == K_PAGE_SOURCE && m == &source_close, "CallVoidMethod: GpuPageSource.close expected");
    page_source_close(o);
}
static jint vm_GetEnv(JavaVM* vm, void** out, jint version);
static const struct JNIInvokeInterface_ vm_functions = {vm_GetEnv};
static JavaVM the_vm = &vm_functions;
static jint jvm_GetJavaVM(JNIEnv* env, JavaVM** vm) { *vm = &the_vm; return JNI_OK; }

static const struct JNINativeInterface_ jvm_functions = {
    jvm_FindClass, jvm_GetMethodID, jvm_NewObject, jvm_Throw, jvm_NewStringUTF, jvm_GetArrayLength, jvm_GetObjectArrayElement,
    jvm_GetIntArrayElements, jvm_GetLongArrayElements, jvm_GetDoubleArrayElements, jvm_GetByteArrayElements,
    jvm_ReleaseIntArrayElements, jvm_ReleaseLongArrayElements, jvm_ReleaseDoubleArrayElements, jvm_ReleaseByteArrayElements,
    jvm_SetLongArrayRegion, jvm_SetIntArrayRegion, jvm_NewLongArray, jvm_NewDirectByteBuffer, jvm_GetDirectBufferAddress, jvm_GetDirectBufferCapacity,
    jvm_ExceptionCheck, jvm_GetJavaVM, jvm_GetObjectClass, jvm_NewGlobalRef, jvm_DeleteGlobalRef, jvm_CallObjectMethod, jvm_CallVoidMethod,
};
static JNIEnv jvm_env = &jvm_functions;
static JNIEnv* env = &jvm_env;
static jint vm_GetEnv(JavaVM* vm, void** out, jint version)
{
    EXPECT(vm == &the_vm && version == JNI_VERSION_1_8, "GetEnv(JNI_VERSION_1_8) of the VM GetJavaVM returned");
    *out = env;
    return JNI_OK;
}

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
static jobject add_input_retention(jlong op, const jblock* blocks, jint channels, jint positions, jint retention, jlong token);
static jobject add_input(jlong op, const jblock* blocks, jint channels, jint positions)
{
    return add_input_retention(op, blocks, channels, positions, 1 /* the pool keeps the slab until the operator is closed */, 0);
}
static jobject add_input_retention(jlong op, const jblock* blocks, jint channels, jint positions, jint retention, jlong token)
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
                                         ints(channels, dict_size), buffer, retention, token);
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

/* ---- hagg-1 (TestHashAggregationOperator.java:160-219): 3 sequence pages of 40 000 rows, key = VARCHAR @0; the test's six aggregates
 *      count(*), sum(bigint), avg(bigint), max(varchar), count(varchar), count(boolean) -> per key str(i): 3, 3 i, (double) i, str(300000 + i), 3, 3 ---- */
static void test_hash_aggregation(void)
{
    const int32_t n = 40000;
    jint types[5] = {PA_VARCHAR, PA_VARCHAR, PA_VARCHAR, PA_BIGINT, PA_BOOLEAN};
    jint group_by[1] = {1};
    jint fns[6] = {PA_AGG_COUNT_STAR, PA_AGG_SUM, PA_AGG_AVG, PA_AGG_MAX, PA_AGG_COUNT, PA_AGG_COUNT};
    jint inputs[6] = {-1, 3, 3, 2, 0, 4}, masks[6] = {-1, -1, -1, -1, -1, -1}, in_types[6] = {PA_BIGINT, PA_BIGINT, PA_BIGINT, PA_VARCHAR, PA_VARCHAR, PA_BOOLEAN};
    jlong op = Java_io_trino_gpu_GpuNative_createHashAggregation(env, 0, ints(5, types), ints(0, 0), ints(1, group_by), 0, -1, -1, PA_STEP_SINGLE, 0, ints(6, fns),
                                                                 ints(6, inputs), ints(6, masks), ints(6, in_types), 100000, 0, PA_STATES_FLAT, PA_MEM_HOST);
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
    EXPECT(Java_io_trino_gpu_GpuNative_memoryBytes(env, 0, op) > 0, "memoryBytes: the aggregation holds HBM");
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    NO_EXCEPTION("finish");
    uint8_t* seen = (uint8_t*)calloc((size_t)n, 1);
    int64_t rows = 0;
    opage out;
    for (int guard = 0; guard < 1000 && !Java_io_trino_gpu_GpuNative_isFinished(env, 0, op); guard++) {
        if (!get_output(op, &out)) continue;
        EXPECT(out.channels == 7, "hagg-1: key + 6 aggregates");
        for (int32_t i = 0; i < out.positions; i++, rows++) {
            char key[16] = "", expect_max[32];
            const int32_t len = out.blocks[0].ints[i + 1] - out.blocks[0].ints[i];
            memcpy(key, out.blocks[0].bytes + out.blocks[0].ints[i], (size_t)(len < 15 ? len : 15));
            const int64_t k = atoll(key);
            EXPECT(k >= 0 && k < n && !seen[k], "hagg-1: key %s twice or out of range", key);
            if (k >= 0 && k < n) seen[k] = 1;
            double avg;
            memcpy(&avg, &out.blocks[3].longs[i], 8);
            sprintf(expect_max, "%lld", (long long)(300000 + k));
            EXPECT(out.blocks[1].longs[i] == 3 && out.blocks[2].longs[i] == 3 * k && avg == (double)k && slice_equals(&out.blocks[4], i, expect_max) &&
                       out.blocks[5].longs[i] == 3 && out.blocks[6].longs[i] == 3,
                   "hagg-1 key %s: %ld %ld %g %ld %ld", key, (long)out.blocks[1].longs[i], (long)out.blocks[2].longs[i], avg, (long)out.blocks[5].longs[i],
                   (long)out.blocks[6].longs[i]);
        }
    }
    NO_EXCEPTION("getOutput");
    EXPECT(rows == n, "hagg-1: %d groups expected, got %ld", n, (long)rows);
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    for (int p = 0; p < 3; p++) Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slabs[p]);
    NO_EXCEPTION("close");
}

/* ---- testHashAggregationWithGlobals (TestHashAggregationOperator.java:221-272): no input, globalAggregationGroupIds (42, 49), groupIdChannel 1
 *      among the keys (VARCHAR, BIGINT), produceDefaultOutput -> (NULL, 42, 0, NULL, NULL, NULL, 0, 0), (NULL, 49, ...).  Channel layout as in
 *      tests/test_oracle_operators.py. ---- */
static void test_hash_aggregation_with_globals(void)
{
    jint types[7] = {PA_VARCHAR, PA_VARCHAR, PA_BIGINT, PA_BIGINT, PA_BIGINT, PA_BOOLEAN, PA_VARCHAR};
    jint group_by[2] = {1, 2}, ids[2] = {42, 49};
    jint fns[6] = {PA_AGG_COUNT_STAR, PA_AGG_MIN, PA_AGG_AVG, PA_AGG_MAX, PA_AGG_COUNT, PA_AGG_COUNT};
    jint inputs[6] = {-1, 4, 4, 6, 0, 5}, masks[6] = {-1, -1, -1, -1, -1, -1}, in_types[6] = {PA_BIGINT, PA_BIGINT, PA_BIGINT, PA_VARCHAR, PA_VARCHAR, PA_BOOLEAN};
    jlong op = Java_io_trino_gpu_GpuNative_createHashAggregation(env, 0, ints(7, types), ints(0, 0), ints(2, group_by), ints(2, ids), -1, 1, PA_STEP_SINGLE, 1,
                                                                 ints(6, fns), ints(6, inputs), ints(6, masks), ints(6, in_types), 100000, 0, PA_STATES_FLAT,
                                                                 PA_MEM_HOST);
    NO_EXCEPTION("createHashAggregation (globals)");
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    opage out;
    EXPECT(get_output(op, &out) && out.positions == 2 && out.channels == 8, "globals: 2 rows of 8 channels, got %d x %d", out.positions, out.channels);
    NO_EXCEPTION("getOutput (globals)");
    for (int32_t i = 0; i < out.positions && i < 2; i++) {
        EXPECT(out.blocks[0].nulls && out.blocks[0].nulls[i] && !(out.blocks[1].nulls && out.blocks[1].nulls[i]) && out.blocks[1].longs[i] == ids[i], "globals row %d keys", i);
        EXPECT(out.blocks[2].longs[i] == 0 && out.blocks[3].nulls && out.blocks[3].nulls[i] && out.blocks[4].nulls && out.blocks[4].nulls[i] &&
                   out.blocks[5].nulls && out.blocks[5].nulls[i] && out.blocks[6].longs[i] == 0 && out.blocks[7].longs[i] == 0, "globals row %d aggregates", i);
    }
    EXPECT(Java_io_trino_gpu_GpuNative_isFinished(env, 0, op), "globals: finished after the default rows");
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    NO_EXCEPTION("close (globals)");
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

/* =========================================== the fake GpuPageSource =========================================== */
/* What java/io/trino/gpu/GpuPageSource.java does over a ConnectorPageSource, over a fixed list of pages here (FixedPageSource): nextPage
 * stages the loaded blocks into a pinned slab, an unloaded LazyBlock is announced with loaded = 0 and staged by loadBlock. */
typedef struct {
    int32_t positions, channels;
    jblock blocks[4];
    uint8_t lazy[4];             /* the block is a LazyBlock that was not loaded when the page was returned */
} source_page;
typedef struct {
    source_page pages[8];
    int32_t page_count, next;
    jobject slab, lazy_slab[4];
    int32_t loads[4];            /* LazyBlock.getLoadedBlock calls per channel */
    int32_t closed;
} page_source_state;
static jobject new_page_source(page_source_state* st)
{
    jobject o = (jobject)calloc(1, sizeof(struct _jobject));
    o->kind = K_PAGE_SOURCE;
    o->data = st;
    return o;
}
static int64_t block_bytes(const jblock* b, int32_t n) { return b->type == PA_VARCHAR ? b->value_bytes : width_of(b->type) * n; }
static jobject page_source_next_page(jobject source)
{
    page_source_state* st = (page_source_state*)source->data;
    if (st->next >= st->page_count) return 0;   /* ConnectorPageSource.isFinished */
    const source_page* pg = &st->pages[st->next++];
    jlong v[3 + 6 * 4];
    int64_t at = 0, off[4][3];
    for (int32_t c = 0; c < pg->channels; c++) {
        const jblock* b = &pg->blocks[c];
        off[c][0] = off[c][1] = off[c][2] = -1;
        if (pg->lazy[c]) continue;
        off[c][0] = at; at = align16(at + block_bytes(b, pg->positions));
        if (b->type == PA_VARCHAR) { off[c][1] = at; at = align16(at + 4 * ((int64_t)pg->positions + 1)); }
        if (b->nulls) { off[c][2] = at; at = align16(at + pg->positions); }
    }
    st->slab = Java_io_trino_gpu_GpuNative_hostMallocPinned(env, 0, at > 0 ? at : 16);   /* (a pool would reuse the previous page's slab) */
    char* base = (char*)st->slab->data;
    v[0] = pg->positions; v[1] = pg->channels; v[2] = Java_io_trino_gpu_GpuNative_bufferAddress(env, 0, st->slab);
    for (int32_t c = 0; c < pg->channels; c++) {
        const jblock* b = &pg->blocks[c];
        jlong* r = v + 3 + 6 * c;
        r[0] = b->type; r[1] = b->type == PA_VARCHAR ? PA_VARWIDTH : PA_FLAT; r[2] = off[c][0]; r[3] = off[c][1]; r[4] = off[c][2]; r[5] = pg->lazy[c] ? 0 : 1;
        if (pg->lazy[c]) continue;
        memcpy(base + off[c][0], b->values, (size_t)block_bytes(b, pg->positions));
        if (b->type == PA_VARCHAR) memcpy(base + off[c][1], b->offsets, 4 * ((size_t)pg->positions + 1));
        if (b->nulls) memcpy(base + off[c][2], b->nulls, (size_t)pg->positions);
    }
    return longs(3 + 6 * pg->channels, v);
}
static jobject page_source_load_block(jobject source, jint channel)
{
    page_source_state* st = (page_source_state*)source->data;
    const source_page* pg = &st->pages[st->next - 1];
    EXPECT(channel >= 0 && channel < pg->channels && pg->lazy[channel], "loadBlock(%d): not a lazy channel of the current page", (int)channel);
    st->loads[channel]++;
    const jblock* b = &pg->blocks[channel];
    const int64_t vb = align16(block_bytes(b, pg->positions)), ob = b->type == PA_VARCHAR ? align16(4 * ((int64_t)pg->positions + 1)) : 0;
    jobject slab = Java_io_trino_gpu_GpuNative_hostMallocPinned(env, 0, vb + ob + pg->positions + 16);
    st->lazy_slab[channel] = slab;
    char* base = (char*)slab->data;
    memcpy(base, b->values, (size_t)block_bytes(b, pg->positions));
    if (ob) memcpy(base + vb, b->offsets, 4 * ((size_t)pg->positions + 1));
    if (b->nulls) memcpy(base + vb + ob, b->nulls, (size_t)pg->positions);
    jlong v[3] = {(jlong)(intptr_t)base, ob ? (jlong)(intptr_t)(base + vb) : 0, b->nulls ? (jlong)(intptr_t)(base + vb + ob) : 0};
    return longs(3, v);
}
static void page_source_close(jobject source) { ((page_source_state*)source->data)->closed++; }

static int drain(jlong op, opage* pages, int cap)
{
    int got = 0;
    for (int guard = 0; guard < 10000 && !Java_io_trino_gpu_GpuNative_isFinished(env, 0, op); guard++) {
        opage out;
        if (get_output(op, &out) && out.positions > 0 && got < cap) pages[got++] = out;
        if (pending_exception) break;
    }
    return got;
}

/* ---- TestScanFilterAndProjectOperator.java:98-255 through createScanFilterProject ---- */
static void test_scan(void)
{
    opage out[8];
    /* testPageSource (:98-131): one page of 10 000 VARCHAR rows, projection field0 -> the page itself */
    {
        static page_source_state st;
        memset(&st, 0, sizeof st);
        varchar_column sv = sequence_varchar(10000, 0);
        st.page_count = 1;
        st.pages[0].positions = 10000; st.pages[0].channels = 1; st.pages[0].blocks[0] = varchar_block(&sv, 10000);
        expr_builder p = {0};
        jlong proj = e_finish(&p, e_field(&p, 0, PA_VARCHAR));
        jint types[1] = {PA_VARCHAR};
        jlong op = Java_io_trino_gpu_GpuNative_createScanFilterProject(env, 0, new_page_source(&st), ints(1, types), ints(0, 0), 0, longs(1, &proj), 0, 0, PA_MEM_HOST);
        NO_EXCEPTION("createScanFilterProject");
        EXPECT(!Java_io_trino_gpu_GpuNative_needsInput(env, 0, op), "a source operator never needs input");
        int n = drain(op, out, 8);
        NO_EXCEPTION("scan-1 getOutput");
        int64_t rows = 0;
        for (int i = 0; i < n; i++) {
            for (int32_t r = 0; r < out[i].positions; r++, rows++) {
                char expect[12];
                sprintf(expect, "%lld", (long long)rows);
                EXPECT(slice_equals(&out[i].blocks[0], r, expect), "scan-1 row %ld", (long)rows);
            }
        }
        EXPECT(rows == 10000, "scan-1: 10 000 rows, got %ld", (long)rows);
        jlongArray stats = Java_io_trino_gpu_GpuNative_scanStats(env, 0, op);
        EXPECT(stats && ((jlong*)stats->data)[0] == 10000, "scanStats: processed positions");
        Java_io_trino_gpu_GpuNative_close(env, 0, op);
        EXPECT(st.closed == 1, "the page source was closed once (%d)", st.closed);
        Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj);
        NO_EXCEPTION("scan-1 close");
    }
    /* testPageSourceMergeOutput (:133-181): 4 pages of BIGINT 0..99, filter field0 = 10, min page 64 KB / 2 rows -> ONE page [10, 10, 10, 10] */
    {
        static page_source_state st;
        memset(&st, 0, sizeof st);
        int64_t* v = sequence_bigint(100, 0);
        st.page_count = 4;
        for (int i = 0; i < 4; i++) { st.pages[i].positions = 100; st.pages[i].channels = 1; st.pages[i].blocks[0] = flat_block(PA_BIGINT, v, 100, 0); }
        expr_builder f = {0}, p = {0};
        jlong filter = e_finish(&f, e_call2(&f, PA_OP_EQUAL, PA_BOOLEAN, e_field(&f, 0, PA_BIGINT), e_long(&f, 10, PA_BIGINT)));
        jlong proj = e_finish(&p, e_field(&p, 0, PA_BIGINT));
        jint types[1] = {PA_BIGINT};
        jlong op = Java_io_trino_gpu_GpuNative_createScanFilterProject(env, 0, new_page_source(&st), ints(1, types), ints(0, 0), filter, longs(1, &proj), 64 * 1024, 2,
                                                                       PA_MEM_HOST);
        NO_EXCEPTION("createScanFilterProject (merge)");
        int n = drain(op, out, 8);
        NO_EXCEPTION("scan-2 getOutput");
        EXPECT(n == 1 && out[0].positions == 4, "scan-2: one merged page of 4 rows, got %d page(s)", n);
        for (int32_t r = 0; n == 1 && r < out[0].positions; r++) EXPECT(out[0].blocks[0].longs[r] == 10, "scan-2 row %d", r);
        Java_io_trino_gpu_GpuNative_close(env, 0, op);
        Java_io_trino_gpu_GpuNative_freeExpression(env, 0, filter);
        Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj);
        NO_EXCEPTION("scan-2 close");
    }
    /* testPageSourceLazyLoad (:183-224) and the load order of PageProcessor.java:307-347: channel 1 is a LazyBlock.  Projection field0 alone:
     * never loaded.  Filter on channel 0 + projection of channel 1: loaded only for the page in which the filter selects a row. */
    {
        static page_source_state st;
        memset(&st, 0, sizeof st);
        int64_t *v = sequence_bigint(100, 0), *w = sequence_bigint(100, 1000);
        st.page_count = 1;
        st.pages[0].positions = 100; st.pages[0].channels = 2;
        st.pages[0].blocks[0] = flat_block(PA_BIGINT, v, 100, 0); st.pages[0].blocks[1] = flat_block(PA_BIGINT, w, 100, 0); st.pages[0].lazy[1] = 1;
        expr_builder p = {0};
        jlong proj = e_finish(&p, e_field(&p, 0, PA_BIGINT));
        jint types[2] = {PA_BIGINT, PA_BIGINT};
        jlong op = Java_io_trino_gpu_GpuNative_createScanFilterProject(env, 0, new_page_source(&st), ints(2, types), ints(0, 0), 0, longs(1, &proj), 0, 0, PA_MEM_HOST);
        int n = drain(op, out, 8);
        NO_EXCEPTION("scan-3 getOutput");
        EXPECT(n == 1 && out[0].positions == 100 && out[0].blocks[0].longs[99] == 99, "scan-3: the loaded channel comes back");
        EXPECT(st.loads[1] == 0, "scan-3: the lazy block must not be loaded (%d loads)", st.loads[1]);
        jlongArray stats = Java_io_trino_gpu_GpuNative_scanStats(env, 0, op);
        EXPECT(stats && ((jlong*)stats->data)[2] == 1 && ((jlong*)stats->data)[3] == 0, "scanStats: one block loaded (channel 0), the channel no expression reads is not even counted");
        Java_io_trino_gpu_GpuNative_close(env, 0, op);
        Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj);

        static page_source_state st2;
        memset(&st2, 0, sizeof st2);
        int64_t* far = sequence_bigint(100, 5000);
        st2.page_count = 2;
        for (int i = 0; i < 2; i++) {
            st2.pages[i].positions = 100; st2.pages[i].channels = 2;
            st2.pages[i].blocks[0] = flat_block(PA_BIGINT, i == 0 ? far : v, 100, 0);   /* page 0: no row below 10; page 1: rows 0..9 */
            st2.pages[i].blocks[1] = flat_block(PA_BIGINT, w, 100, 0); st2.pages[i].lazy[1] = 1;
        }
        expr_builder f = {0}, q = {0};
        jlong filter = e_finish(&f, e_call2(&f, PA_OP_LESS_THAN, PA_BOOLEAN, e_field(&f, 0, PA_BIGINT), e_long(&f, 10, PA_BIGINT)));
        jlong proj1 = e_finish(&q, e_field(&q, 1, PA_BIGINT));
        op = Java_io_trino_gpu_GpuNative_createScanFilterProject(env, 0, new_page_source(&st2), ints(2, types), ints(0, 0), filter, longs(1, &proj1), 0, 0, PA_MEM_HOST);
        n = drain(op, out, 8);
        NO_EXCEPTION("scan-4 getOutput");
        EXPECT(n == 1 && out[0].positions == 10 && out[0].blocks[0].longs[0] == 1000 && out[0].blocks[0].longs[9] == 1009, "scan-4: channel 1 of the 10 selected rows");
        EXPECT(st2.loads[1] == 1, "scan-4: the lazy block is loaded for the one page the filter selects rows of (%d loads)", st2.loads[1]);
        stats = Java_io_trino_gpu_GpuNative_scanStats(env, 0, op);
        EXPECT(stats && ((jlong*)stats->data)[0] == 200 && ((jlong*)stats->data)[3] == 1, "scanStats: 200 positions, one projection block left unloaded (no row survived the filter)");
        Java_io_trino_gpu_GpuNative_close(env, 0, op);
        Java_io_trino_gpu_GpuNative_freeExpression(env, 0, filter);
        Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj1);
        NO_EXCEPTION("scan-3/4 close");
    }
    EXPECT(global_refs == 0, "%d global references to page sources left", global_refs);
}

/* ---- AggregationOperator (TestAggregationOperator.java:119-156 shape: count / sum / avg over a BIGINT sequence) and OrderBy / TopN ---- */
static void test_aggregation_order_by_topn(void)
{
    jint types[2] = {PA_BIGINT, PA_DOUBLE};
    int64_t* v = sequence_bigint(100, 0);
    double d[100];
    for (int i = 0; i < 100; i++) d[i] = (double)((i * 37) % 100);
    jblock blocks[2] = {flat_block(PA_BIGINT, v, 100, 0), flat_block(PA_DOUBLE, d, 100, 0)};
    jint fns[3] = {PA_AGG_COUNT_STAR, PA_AGG_SUM, PA_AGG_AVG}, inputs[3] = {-1, 0, 0}, masks[3] = {-1, -1, -1}, in_types[3] = {PA_BIGINT, PA_BIGINT, PA_BIGINT};
    jlong op = Java_io_trino_gpu_GpuNative_createAggregation(env, 0, ints(2, types), PA_STEP_SINGLE, ints(3, fns), ints(3, inputs), ints(3, masks), ints(3, in_types),
                                                             PA_STATES_FLAT, PA_MEM_HOST);
    NO_EXCEPTION("createAggregation");
    jobject slab = add_input(op, blocks, 2, 100);
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    opage out;
    int got = 0;
    for (int guard = 0; guard < 1000 && !got; guard++) got = get_output(op, &out);
    NO_EXCEPTION("aggregation getOutput");
    double avg = 0;
    if (got) memcpy(&avg, &out.blocks[2].longs[0], 8);
    EXPECT(got && out.positions == 1 && out.blocks[0].longs[0] == 100 && out.blocks[1].longs[0] == 4950 && avg == 49.5, "agg: count 100, sum 4950, avg 49.5");
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);

    /* OrderBy: rows by the DOUBLE channel descending; output (DOUBLE, BIGINT) */
    jint outc[2] = {1, 0}, sortc[1] = {1}, sorto[1] = {PA_DESC_NULLS_LAST};
    op = Java_io_trino_gpu_GpuNative_createOrderBy(env, 0, ints(2, types), ints(2, outc), ints(1, sortc), ints(1, sorto), PA_MEM_HOST);
    NO_EXCEPTION("createOrderBy");
    slab = add_input(op, blocks, 2, 100);
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    got = 0;
    for (int guard = 0; guard < 1000 && !got; guard++) got = get_output(op, &out);
    NO_EXCEPTION("orderBy getOutput");
    EXPECT(got && out.positions == 100 && out.channels == 2, "orderBy: 100 rows");
    for (int32_t i = 0; got && i < out.positions; i++) {
        double x;
        memcpy(&x, &out.blocks[0].longs[i], 8);
        EXPECT(x == (double)(99 - i) && d[out.blocks[1].longs[i]] == x, "orderBy row %d: %g", i, x);
    }
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);

    /* TopN: the 5 largest DOUBLEs */
    op = Java_io_trino_gpu_GpuNative_createTopN(env, 0, ints(2, types), 5, ints(1, sortc), ints(1, sorto), PA_MEM_HOST);
    NO_EXCEPTION("createTopN");
    slab = add_input(op, blocks, 2, 100);
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    got = 0;
    for (int guard = 0; guard < 1000 && !got; guard++) got = get_output(op, &out);
    NO_EXCEPTION("topN getOutput");
    EXPECT(got && out.positions == 5, "topN: 5 rows");
    for (int32_t i = 0; got && i < out.positions; i++) {
        double x;
        memcpy(&x, &out.blocks[1].longs[i], 8);
        EXPECT(x == (double)(99 - i), "topN row %d: %g", i, x);
    }
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);

    /* an aggregation whose only consumer is a TopN: the planner's hint is taken by a grouped SINGLE step */
    jint gb[1] = {0}, hfns[1] = {PA_AGG_SUM}, hin[1] = {1}, hm[1] = {-1}, ht[1] = {PA_DOUBLE}, hsc[1] = {1}, hso[1] = {PA_DESC_NULLS_LAST};
    op = Java_io_trino_gpu_GpuNative_createHashAggregation(env, 0, ints(2, types), ints(0, 0), ints(1, gb), 0, -1, -1, PA_STEP_SINGLE, 0, ints(1, hfns), ints(1, hin),
                                                           ints(1, hm), ints(1, ht), 1000, 0, PA_STATES_FLAT, PA_MEM_HOST);
    EXPECT(Java_io_trino_gpu_GpuNative_setOutputTopNHint(env, 0, op, 3, ints(1, hsc), ints(1, hso)), "setOutputTopNHint is taken");
    NO_EXCEPTION("setOutputTopNHint");
    slab = add_input(op, blocks, 2, 100);
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    opage pages[4];
    int n = drain(op, pages, 4);
    int64_t best = 0;
    for (int i = 0; i < n; i++) {
        for (int32_t r = 0; r < pages[i].positions; r++) {
            double x;
            memcpy(&x, &pages[i].blocks[1].longs[r], 8);
            if (x >= 97.0) best++;
        }
    }
    EXPECT(best == 3, "the hinted aggregation still emits the TopN's 3 rows (%ld)", (long)best);
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);
    NO_EXCEPTION("aggregation / orderBy / topN");
}

/* ---- DynamicFilterSourceOperator (TestDynamicFilterSourceOperator shapes): values collected, polled as a TupleDomain; memory natives ---- */
static void test_dynamic_filter_and_memory(void)
{
    jint types[1] = {PA_BIGINT}, ch[1] = {0};
    jlong op = Java_io_trino_gpu_GpuNative_createDynamicFilterSource(env, 0, ints(1, types), ints(1, ch), 100, 1000, 1 << 20);
    NO_EXCEPTION("createDynamicFilterSource");
    EXPECT(Java_io_trino_gpu_GpuNative_dynamicFilterPoll(env, 0, op, 1) == 0, "no TupleDomain before the operator has finished");
    NO_EXCEPTION("dynamicFilterPoll (early)");
    int64_t v[6] = {5, 3, 5, 9, 3, 7};
    jblock blocks[1] = {flat_block(PA_BIGINT, v, 6, 0)};
    jobject slab = add_input(op, blocks, 1, 6);
    opage out;
    EXPECT(get_output(op, &out) && out.positions == 6, "the page passes through");
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    jlongArray dom = 0;
    for (int guard = 0; guard < 1000 && !dom; guard++) dom = Java_io_trino_gpu_GpuNative_dynamicFilterPoll(env, 0, op, 1);
    NO_EXCEPTION("dynamicFilterPoll");
    EXPECT(dom && dom->len == 7, "TupleDomain of one channel");
    if (dom && dom->len == 7) {
        const jlong* r = (const jlong*)dom->data;
        EXPECT(r[0] == 0 && r[1] == PA_DOMAIN_VALUES && r[2] == 4 && r[3] == PA_BIGINT && r[5] == 32, "Domain: 4 distinct BIGINT values (%ld %ld %ld)", (long)r[1], (long)r[2], (long)r[5]);
        jobject values = Java_io_trino_gpu_GpuNative_wrapAddress(env, 0, r[4], r[5]);
        const int64_t* x = (const int64_t*)values->data;
        EXPECT(x[0] == 3 && x[1] == 5 && x[2] == 7 && x[3] == 9, "Domain values ascending");
    }
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);

    EXPECT(Java_io_trino_gpu_GpuNative_deviceCount(env, 0) >= 1, "deviceCount");
    Java_io_trino_gpu_GpuNative_memorySetLimit(env, 0, (jlong)8 << 30);
    jlongArray ms = Java_io_trino_gpu_GpuNative_memoryStats(env, 0);
    EXPECT(ms && ms->len == 3 && ((jlong*)ms->data)[2] == (jlong)8 << 30, "memoryStats reports the limit that was set");
    Java_io_trino_gpu_GpuNative_memorySetLimit(env, 0, 0);
    NO_EXCEPTION("memory natives");
}

/* ---- PagesSerde through the shim: a device-output operator's page -> SerializedPage frame -> another operator's input; retained slabs ---- */
static void test_serde_and_retained_pages(void)
{
    expr_builder p0 = {0}, p1 = {0};
    jlong proj[2];
    proj[0] = e_finish(&p0, e_field(&p0, 0, PA_VARCHAR));
    proj[1] = e_finish(&p1, e_call2(&p1, PA_OP_ADD, PA_BIGINT, e_field(&p1, 1, PA_BIGINT), e_long(&p1, 5, PA_BIGINT)));
    jint types[2] = {PA_VARCHAR, PA_BIGINT};
    jlong producer = Java_io_trino_gpu_GpuNative_createFilterProject(env, 0, ints(2, types), ints(0, 0), 0, longs(2, proj), 0, 0, PA_MEM_DEVICE);
    varchar_column sv = sequence_varchar(1000, 0);
    int64_t* v = sequence_bigint(1000, 0);
    jblock blocks[2] = {varchar_block(&sv, 1000), flat_block(PA_BIGINT, v, 1000, 0)};
    jobject slab = add_input(producer, blocks, 2, 1000);
    jobject frame = Java_io_trino_gpu_GpuNative_hostMallocPinned(env, 0, 1 << 20);
    for (int compress = 0; compress < 2; compress++) {
        if (compress) {
            Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);
            slab = add_input(producer, blocks, 2, 1000);
        }
        jlong size = 0;
        for (int guard = 0; guard < 1000 && !size; guard++) size = Java_io_trino_gpu_GpuNative_getOutputSerialized(env, 0, producer, frame, (jboolean)compress);
        NO_EXCEPTION("getOutputSerialized");
        EXPECT(size > 13 && *(int32_t*)frame->data == 1000, "a SerializedPage frame of 1000 positions (%ld bytes)", (long)size);
        EXPECT(((uint8_t*)frame->data)[4] == (compress ? 1 : 0) || !compress, "the COMPRESSED marker follows the request when compression pays");
        /* the consumer: count + sum over the decoded page */
        jint fns[2] = {PA_AGG_COUNT_STAR, PA_AGG_SUM}, inputs[2] = {-1, 1}, masks[2] = {-1, -1}, in_types[2] = {PA_BIGINT, PA_BIGINT};
        jlong consumer = Java_io_trino_gpu_GpuNative_createAggregation(env, 0, ints(2, types), PA_STEP_SINGLE, ints(2, fns), ints(2, inputs), ints(2, masks), ints(2, in_types),
                                                                       PA_STATES_FLAT, PA_MEM_HOST);
        Java_io_trino_gpu_GpuNative_addInputSerialized(env, 0, consumer, frame, size, ints(2, types));
        NO_EXCEPTION("addInputSerialized");
        Java_io_trino_gpu_GpuNative_finish(env, 0, consumer);
        opage out;
        int got = 0;
        for (int guard = 0; guard < 1000 && !got; guard++) got = get_output(consumer, &out);
        EXPECT(got && out.blocks[0].longs[0] == 1000 && out.blocks[1].longs[0] == 499500 + 5000, "serde: count 1000, sum of (i + 5)");
        Java_io_trino_gpu_GpuNative_close(env, 0, consumer);
    }
    /* a truncated frame is refused, not read past its end */
    jlong consumer = Java_io_trino_gpu_GpuNative_createFilterProject(env, 0, ints(2, types), ints(0, 0), 0, longs(2, proj), 0, 0, PA_MEM_HOST);
    Java_io_trino_gpu_GpuNative_addInputSerialized(env, 0, consumer, frame, 9, ints(2, types));
    EXPECT(take_exception(0, 0) == PA_ERR_INVALID_ARGUMENT, "a truncated frame: INVALID_ARGUMENT");
    Java_io_trino_gpu_GpuNative_close(env, 0, consumer);
    Java_io_trino_gpu_GpuNative_close(env, 0, producer);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slab);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, frame);

    /* retention 2: every slab comes back through drainReleased -- once, at the latest when the operator is closed */
    jint q6t[1] = {PA_BIGINT}, fns[2] = {PA_AGG_COUNT_STAR, PA_AGG_SUM}, inputs[2] = {-1, 0}, masks[2] = {-1, -1}, in_types[2] = {PA_BIGINT, PA_BIGINT};
    expr_builder id = {0};
    jlong ident = e_finish(&id, e_field(&id, 0, PA_BIGINT));
    jlong op = Java_io_trino_gpu_GpuNative_createFusedAggregation(env, 0, ints(1, q6t), ints(0, 0), 0, longs(1, &ident), ints(1, q6t), ints(0, 0), PA_STEP_SINGLE,
                                                                  ints(2, fns), ints(2, inputs), ints(2, masks), ints(2, in_types), 1, PA_MEM_HOST);
    NO_EXCEPTION("createFusedAggregation (retained)");
    enum { PAGES = 40 };
    jobject slabs[PAGES];
    int seen[PAGES] = {0};
    jlongArray tokens = longs(PAGES, 0);
    int64_t* big = sequence_bigint(8192, 0);
    for (int i = 0; i < PAGES; i++) {
        jblock b[1] = {flat_block(PA_BIGINT, big, 8192, 0)};
        while (Java_io_trino_gpu_GpuNative_isBlocked(env, 0, op) && !Java_io_trino_gpu_GpuNative_needsInput(env, 0, op)) {}
        slabs[i] = add_input_retention(op, b, 1, 8192, 2, 1000 + i);
        NO_EXCEPTION("addInput (retained)");
        const jint k = Java_io_trino_gpu_GpuNative_drainReleased(env, 0, tokens);
        for (jint j = 0; j < k; j++) seen[((jlong*)tokens->data)[j] - 1000]++;
    }
    Java_io_trino_gpu_GpuNative_finish(env, 0, op);
    opage out;
    int got = 0;
    for (int guard = 0; guard < 1000 && !got; guard++) got = get_output(op, &out);
    EXPECT(got && out.blocks[0].longs[0] == (int64_t)PAGES * 8192 && out.blocks[1].longs[0] == (int64_t)PAGES * (8191LL * 8192 / 2), "retained pages: count and sum");
    Java_io_trino_gpu_GpuNative_close(env, 0, op);
    for (;;) {
        const jint k = Java_io_trino_gpu_GpuNative_drainReleased(env, 0, tokens);
        if (k == 0) break;
        for (jint j = 0; j < k; j++) seen[((jlong*)tokens->data)[j] - 1000]++;
    }
    for (int i = 0; i < PAGES; i++) {
        EXPECT(seen[i] == 1, "retained page %d released %d times", i, seen[i]);
        Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, slabs[i]);
    }
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, ident);
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj[0]);
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj[1]);
    NO_EXCEPTION("serde / retained");
}

/* ---- FilterAndProject -> LookupJoin behind one handle, the join-side dynamic filter, and the exchange natives on a world of one rank ---- */
static void test_fused_join_dynamic_filter_exchange(void)
{
    jint btypes[2] = {PA_BIGINT, PA_BIGINT}, key[1] = {0}, bout[1] = {1};
    jlong bridge = Java_io_trino_gpu_GpuNative_createLookupSource(env, 0);
    jlong build = Java_io_trino_gpu_GpuNative_createHashBuilder(env, 0, bridge, ints(2, btypes), ints(1, key), -1, ints(1, bout), 100);
    int64_t *bk = sequence_bigint(100, 1000), *bp = sequence_bigint(100, 7000);
    jblock bb[2] = {flat_block(PA_BIGINT, bk, 100, 0), flat_block(PA_BIGINT, bp, 100, 0)};
    jobject s1 = add_input(build, bb, 2, 100);
    Java_io_trino_gpu_GpuNative_finish(env, 0, build);
    NO_EXCEPTION("build");
    /* probe pipeline: filter value < 5000, projections (key, value); join on projection 0; output (probe key, probe value, build payload) */
    expr_builder f = {0}, p0 = {0}, p1 = {0};
    jlong filter = e_finish(&f, e_call2(&f, PA_OP_LESS_THAN, PA_BOOLEAN, e_field(&f, 1, PA_BIGINT), e_long(&f, 5000, PA_BIGINT)));
    jlong proj[2] = {e_finish(&p0, e_field(&p0, 0, PA_BIGINT)), e_finish(&p1, e_field(&p1, 1, PA_BIGINT))};
    jint ptypes[2] = {PA_BIGINT, PA_BIGINT}, pout[2] = {0, 1}, joined[3] = {PA_BIGINT, PA_BIGINT, PA_BIGINT};
    jlong join = Java_io_trino_gpu_GpuNative_createFusedJoin(env, 0, bridge, ints(2, ptypes), ints(0, 0), filter, longs(2, proj), ints(2, ptypes), ints(1, key), ints(2, pout),
                                                             ints(3, joined), ints(0, 0), PA_STEP_SINGLE, 0, 0, 0, 0, 0, PA_MEM_HOST);
    NO_EXCEPTION("createFusedJoin");
    int64_t *pk = sequence_bigint(4000, 0), *pv = sequence_bigint(4000, 3000);   /* keys 0..3999, values 3000..6999: value < 5000 keeps keys < 2000 */
    jblock pb[2] = {flat_block(PA_BIGINT, pk, 4000, 0), flat_block(PA_BIGINT, pv, 4000, 0)};
    jobject s2 = add_input(join, pb, 2, 4000);
    Java_io_trino_gpu_GpuNative_finish(env, 0, join);
    opage pages[4];
    int n = drain(join, pages, 4);
    NO_EXCEPTION("fused join getOutput");
    int64_t rows = 0;
    for (int i = 0; i < n; i++) {
        for (int32_t r = 0; r < pages[i].positions; r++, rows++) {
            const int64_t k = pages[i].blocks[0].longs[r];
            EXPECT(k >= 1000 && k < 1100 && pages[i].blocks[1].longs[r] == k + 3000 && pages[i].blocks[2].longs[r] == k + 6000, "fused join row: key %ld", (long)k);
        }
    }
    EXPECT(rows == 100, "fused join: the 100 build keys all match a filtered probe row (%ld)", (long)rows);
    Java_io_trino_gpu_GpuNative_close(env, 0, join);

    /* the same probe side as a plain FilterAndProject with the build side's dynamic filter installed on its key channel */
    jlong fp = Java_io_trino_gpu_GpuNative_createFilterProject(env, 0, ints(2, ptypes), ints(0, 0), filter, longs(2, proj), 0, 0, PA_MEM_HOST);
    EXPECT(Java_io_trino_gpu_GpuNative_setDynamicFilter(env, 0, fp, 0, bridge), "setDynamicFilter: dense integer keys give a filter");
    NO_EXCEPTION("setDynamicFilter");
    jobject s3 = add_input(fp, pb, 2, 4000);
    opage out;
    EXPECT(get_output(fp, &out) && out.positions == 100, "the dynamic filter keeps only probe rows whose key exists on the build side (%d)", out.positions);
    Java_io_trino_gpu_GpuNative_close(env, 0, fp);
    Java_io_trino_gpu_GpuNative_close(env, 0, build);
    Java_io_trino_gpu_GpuNative_destroyLookupSource(env, 0, bridge);

    /* exchange natives, world of one rank (RCCL with itself): PartitionedOutput -> ExchangeSource hands every row back */
    jbyteArray id = new_array(K_BYTES, PA_COMM_ID_BYTES, 0);
    Java_io_trino_gpu_GpuNative_commUniqueId(env, 0, id);
    jlong comm = Java_io_trino_gpu_GpuNative_commCreate(env, 0, id, 0, 1);
    NO_EXCEPTION("commCreate");
    jlong ex = Java_io_trino_gpu_GpuNative_exchangeCreate(env, 0, comm, ints(2, ptypes), ints(1, key), -1, 1);
    jlong sink = Java_io_trino_gpu_GpuNative_createPartitionedOutput(env, 0, ex);
    jlong source = Java_io_trino_gpu_GpuNative_createExchangeSource(env, 0, ex, PA_MEM_HOST);
    NO_EXCEPTION("exchange factories");
    jobject s4 = add_input(sink, pb, 2, 4000);
    Java_io_trino_gpu_GpuNative_finish(env, 0, sink);
    n = drain(source, pages, 4);
    NO_EXCEPTION("exchange source getOutput");
    rows = 0;
    int64_t sum = 0;
    for (int i = 0; i < n; i++) {
        for (int32_t r = 0; r < pages[i].positions; r++, rows++) sum += pages[i].blocks[1].longs[r] - pages[i].blocks[0].longs[r];
    }
    EXPECT(rows == 4000 && sum == 4000LL * 3000, "exchange over one rank: every row comes back (%ld rows)", (long)rows);
    Java_io_trino_gpu_GpuNative_close(env, 0, sink);
    Java_io_trino_gpu_GpuNative_close(env, 0, source);
    Java_io_trino_gpu_GpuNative_exchangeDestroy(env, 0, ex);
    Java_io_trino_gpu_GpuNative_commDestroy(env, 0, comm);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, s1);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, s2);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, s3);
    Java_io_trino_gpu_GpuNative_hostFreePinned(env, 0, s4);
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, filter);
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj[0]);
    Java_io_trino_gpu_GpuNative_freeExpression(env, 0, proj[1]);
    NO_EXCEPTION("fused join / dynamic filter / exchange teardown");
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
    test_hash_aggregation_with_globals();
    test_join();
    test_fused_q6();
    test_scan();
    test_aggregation_order_by_topn();
    test_dynamic_filter_and_memory();
    test_serde_and_retained_pages();
    test_fused_join_dynamic_filter_exchange();
    EXPECT(outstanding_elements == 0, "%d Get<Type>ArrayElements without their Release", outstanding_elements);
    EXPECT(pending_exception == 0, "an exception was left pending");
    if (failures) {
        fprintf(stderr, "%d failure(s)\n", failures);
        return 1;
    }
    printf("jni harness: all cases pass\n");
    return 0;
}
