/* presto_amd_jni.c -- the JNI shim between io.trino.gpu.* (java/io/trino/gpu) and libpresto_amd.so (include/presto_amd.h).
 *
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/presto_amd_jni.c \
 *       -Lpresto_amd -lpresto_amd -o libpresto_amd_jni.so
 *
 * One native method per C-ABI entry; nothing is computed here.  The image this repository is written in has no JDK, so
 * the file is not built there: tests/test_lib_cpu.py compiles it with -fsyntax-only against jni/stub/jni.h, which keeps it
 * in step with include/presto_amd.h (a renamed entry point or a changed descriptor field fails that test).
 *
 * Data crosses as the reference's blocks lay it out (SURVEY 8b "Ownership"): the Java side copies long[] / int[] / byte[] /
 * Slice bytes / boolean[] valueIsNull of every Block into ONE pinned direct ByteBuffer (allocated by hostMallocPinned
 * below, pooled by PinnedPagePool) and passes per-channel offsets into it; JVM heap arrays are movable and never reach the
 * device side.  A buffer stays untouched until the operator has read it: pages are flagged PA_PAGE_STABLE only when the
 * pool keeps the buffer until the operator is closed. */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "presto_amd.h"

#define MAX_CHANNELS 64

static void throw_native(JNIEnv* env, int32_t status)
{
    if ((*env)->ExceptionCheck(env)) return;   /* a Java exception raised inside a page-source callback is the one that propagates */
    /* GpuNativeException(int status, String message) maps pa_status onto io.trino.spi.StandardErrorCode:
     * NUMERIC_VALUE_OUT_OF_RANGE, DIVISION_BY_ZERO, GENERIC_INSUFFICIENT_RESOURCES, NOT_SUPPORTED, GENERIC_INTERNAL_ERROR */
    jclass cls = (*env)->FindClass(env, "io/trino/gpu/GpuNativeException");
    jmethodID ctor = (*env)->GetMethodID(env, cls, "<init>", "(ILjava/lang/String;)V");
    jstring msg = (*env)->NewStringUTF(env, pa_last_error());
    (*env)->Throw(env, (jthrowable)(*env)->NewObject(env, cls, ctor, (jint)status, msg));
}
static void throw_message(JNIEnv* env, int32_t status, const char* message)
{
    jclass cls = (*env)->FindClass(env, "io/trino/gpu/GpuNativeException");
    jmethodID ctor = (*env)->GetMethodID(env, cls, "<init>", "(ILjava/lang/String;)V");
    (*env)->Throw(env, (jthrowable)(*env)->NewObject(env, cls, ctor, (jint)status, (*env)->NewStringUTF(env, message)));
}
#define CHECK(rc) do { if ((rc) < 0) throw_native(env, (rc)); } while (0)

/* ---- library ---- */
JNIEXPORT jint JNICALL Java_io_trino_gpu_GpuNative_abiVersion(JNIEnv* env, jclass c) { return pa_abi_version(); }
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_init(JNIEnv* env, jclass c, jint device) { CHECK(pa_init(device)); }
JNIEXPORT jint JNICALL Java_io_trino_gpu_GpuNative_deviceCount(JNIEnv* env, jclass c) { return pa_device_count(); }
JNIEXPORT jobject JNICALL Java_io_trino_gpu_GpuNative_hostMallocPinned(JNIEnv* env, jclass c, jlong bytes)
{
    void* p = 0;
    int32_t rc = pa_host_malloc_pinned(&p, bytes);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (*env)->NewDirectByteBuffer(env, p, bytes);
}
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_hostFreePinned(JNIEnv* env, jclass c, jobject buffer)
{
    CHECK(pa_host_free_pinned((*env)->GetDirectBufferAddress(env, buffer)));
}

/* ---- RowExpression trees: RowExpressionSerializer walks the tree children-first and hands over the flat arrays ---- */
typedef struct native_expr {
    pa_expr expr;
    pa_expr_node* nodes;
    int32_t* args;
    char** strings;
    int32_t count;
} native_expr;

/* typeParams: per node PA_DECIMAL_PARAM(precision, scale) for nodes of a DECIMAL type, else 0 (a long decimal constant carries its low
 * 64 bits in longs[i] and its high 64 bits as the bit pattern of doubles[i]) */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_newExpression(JNIEnv* env, jclass c, jint root, jintArray kinds, jintArray ops, jintArray types,
        jintArray typeParams, jintArray channels, jintArray isNull, jintArray nargs, jintArray firstArg, jlongArray longs, jdoubleArray doubles,
        jobjectArray strings, jintArray args)
{
    const jsize n = (*env)->GetArrayLength(env, kinds), na = (*env)->GetArrayLength(env, args);
    native_expr* e = (native_expr*)calloc(1, sizeof(native_expr));
    e->nodes = (pa_expr_node*)calloc((size_t)(n > 0 ? n : 1), sizeof(pa_expr_node));
    e->args = (int32_t*)calloc((size_t)(na > 0 ? na : 1), sizeof(int32_t));
    e->strings = (char**)calloc((size_t)(n > 0 ? n : 1), sizeof(char*));
    e->count = n;
    jint* k = (*env)->GetIntArrayElements(env, kinds, 0);
    jint* o = (*env)->GetIntArrayElements(env, ops, 0);
    jint* t = (*env)->GetIntArrayElements(env, types, 0);
    jint* tp = (*env)->GetIntArrayElements(env, typeParams, 0);
    jint* ch = (*env)->GetIntArrayElements(env, channels, 0);
    jint* nl = (*env)->GetIntArrayElements(env, isNull, 0);
    jint* na_ = (*env)->GetIntArrayElements(env, nargs, 0);
    jint* fa = (*env)->GetIntArrayElements(env, firstArg, 0);
    jlong* lv = (*env)->GetLongArrayElements(env, longs, 0);
    jdouble* dv = (*env)->GetDoubleArrayElements(env, doubles, 0);
    jint* av = (*env)->GetIntArrayElements(env, args, 0);
    for (jsize i = 0; i < n; i++) {
        pa_expr_node* nd = &e->nodes[i];
        nd->kind = k[i]; nd->op = o[i]; nd->type = t[i]; nd->channel = ch[i]; nd->is_null = nl[i];
        nd->nargs = na_[i]; nd->first_arg = fa[i]; nd->i64 = lv[i]; nd->f64 = dv[i];
        if (t[i] == PA_DECIMAL || t[i] == PA_LONG_DECIMAL) nd->type_param = tp[i];
        jbyteArray s = (jbyteArray)(*env)->GetObjectArrayElement(env, strings, i);   /* VARCHAR constants: UTF-8 bytes, else null */
        if (s) {
            const jsize len = (*env)->GetArrayLength(env, s);
            jbyte* b = (*env)->GetByteArrayElements(env, s, 0);
            e->strings[i] = (char*)malloc((size_t)(len > 0 ? len : 1));
            memcpy(e->strings[i], b, (size_t)len);
            (*env)->ReleaseByteArrayElements(env, s, b, JNI_ABORT);
            nd->str = e->strings[i];
            nd->str_len = len;
        }
    }
    for (jsize i = 0; i < na; i++) e->args[i] = av[i];
    (*env)->ReleaseIntArrayElements(env, kinds, k, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, ops, o, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, types, t, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, typeParams, tp, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, channels, ch, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, isNull, nl, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, nargs, na_, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, firstArg, fa, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, longs, lv, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, doubles, dv, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, args, av, JNI_ABORT);
    e->expr.node_count = n;
    e->expr.root = root;
    e->expr.nodes = e->nodes;
    e->expr.arg_count = na;
    e->expr.args = e->args;
    return (jlong)(intptr_t)e;
}
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_freeExpression(JNIEnv* env, jclass c, jlong h)
{
    native_expr* e = (native_expr*)(intptr_t)h;
    if (!e) return;
    for (int32_t i = 0; i < e->count; i++) free(e->strings[i]);
    free(e->strings); free(e->args); free(e->nodes); free(e);
}

/* copies a Java int[] into a malloc'ed int32_t[] (descriptor arrays only have to live for the create call) */
static int32_t* ints_of(JNIEnv* env, jintArray a, jsize* n)
{
    *n = a ? (*env)->GetArrayLength(env, a) : 0;
    int32_t* out = (int32_t*)calloc((size_t)(*n > 0 ? *n : 1), sizeof(int32_t));
    if (a) {
        jint* v = (*env)->GetIntArrayElements(env, a, 0);
        memcpy(out, v, (size_t)*n * sizeof(int32_t));
        (*env)->ReleaseIntArrayElements(env, a, v, JNI_ABORT);
    }
    return out;
}

/* ---- operator factories (OperatorFactory.createOperator) ---- */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createFilterProject(JNIEnv* env, jclass c, jintArray inputTypes, jintArray typeParams, jlong filter,
        jlongArray projections, jlong minOutputPageBytes, jint minOutputPageRows, jint outputMem)
{
    jsize n, np_, nproj = (*env)->GetArrayLength(env, projections);
    pa_filter_project_desc d;
    memset(&d, 0, sizeof d);
    int32_t* types = ints_of(env, inputTypes, &n);
    int32_t* params = ints_of(env, typeParams, &np_);
    pa_expr* pe = (pa_expr*)calloc((size_t)(nproj > 0 ? nproj : 1), sizeof(pa_expr));
    jlong* ph = (*env)->GetLongArrayElements(env, projections, 0);
    for (jsize i = 0; i < nproj; i++) pe[i] = ((native_expr*)(intptr_t)ph[i])->expr;
    (*env)->ReleaseLongArrayElements(env, projections, ph, JNI_ABORT);
    d.input_channel_count = n;
    d.input_types = types;
    d.input_type_params = np_ == n ? params : 0;
    d.filter = filter ? &((native_expr*)(intptr_t)filter)->expr : 0;
    d.projection_count = nproj;
    d.projections = pe;
    d.output_mem = outputMem;
    d.min_output_page_bytes = minOutputPageBytes;      /* FilterAndProjectOperator's MergePages thresholds */
    d.min_output_page_rows = minOutputPageRows;
    pa_operator* op = 0;
    int32_t rc = pa_filter_project_create(&d, &op);
    free(pe); free(params); free(types);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

static void fill_aggregates(JNIEnv* env, jintArray fns, jintArray inputs, jintArray masks, jintArray inputTypes, pa_aggregate** out, jsize* count)
{
    jsize n1, n2, n3, n4;
    int32_t *f = ints_of(env, fns, &n1), *in = ints_of(env, inputs, &n2), *m = ints_of(env, masks, &n3), *t = ints_of(env, inputTypes, &n4);
    *out = (pa_aggregate*)calloc((size_t)(n1 > 0 ? n1 : 1), sizeof(pa_aggregate));
    for (jsize i = 0; i < n1; i++) { (*out)[i].fn = f[i]; (*out)[i].input_channel = in[i]; (*out)[i].mask_channel = m[i]; (*out)[i].input_type = t[i]; }
    *count = n1;
    free(f); free(in); free(m); free(t);
}

/* HashAggregationOperatorFactory's arguments (HashAggregationOperator.java:120-202): globalAggregationGroupIds (may be null), groupIdChannel
 * (an index among the group-by columns, -1 = Optional.empty()), produceDefaultOutput, maxPartialMemory in bytes (0 = no early flush),
 * stateFormat = pa_state_format of a PARTIAL step's output / a FINAL step's input. */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createHashAggregation(JNIEnv* env, jclass c, jintArray inputTypes, jintArray typeParams,
        jintArray groupByChannels, jintArray globalAggregationGroupIds, jint hashChannel, jint groupIdChannel, jint step, jboolean produceDefaultOutput,
        jintArray aggFns, jintArray aggInputs, jintArray aggMasks, jintArray aggInputTypes, jint expectedGroups, jlong maxPartialMemory, jint stateFormat,
        jint outputMem)
{
    jsize n, np_, ng, na, nids;
    pa_hash_aggregation_desc d;
    memset(&d, 0, sizeof d);
    int32_t *types = ints_of(env, inputTypes, &n), *params = ints_of(env, typeParams, &np_), *gb = ints_of(env, groupByChannels, &ng);
    int32_t* ids = ints_of(env, globalAggregationGroupIds, &nids);
    pa_aggregate* aggs;
    fill_aggregates(env, aggFns, aggInputs, aggMasks, aggInputTypes, &aggs, &na);
    d.input_channel_count = n; d.input_types = types; d.input_type_params = np_ == n ? params : 0;
    d.group_by_count = ng; d.group_by_channels = gb; d.hash_channel = hashChannel; d.step = step;
    d.aggregate_count = na; d.aggregates = aggs; d.expected_groups = expectedGroups; d.output_mem = outputMem;
    d.max_partial_memory = maxPartialMemory; d.state_format = stateFormat;
    d.produce_default_output = produceDefaultOutput ? 1 : 0; d.group_id_channel = groupIdChannel;
    d.global_aggregation_group_id_count = nids; d.global_aggregation_group_ids = ids;
    pa_operator* op = 0;
    int32_t rc = pa_hash_aggregation_create(&d, &op);   /* group_by_count == 0: AggregationOperator */
    free(aggs); free(ids); free(gb); free(params); free(types);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

/* AggregationOperator.AggregationOperatorFactory (AggregationOperator.java:40-140; LocalExecutionPlanner.java:3389): ungrouped aggregates */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createAggregation(JNIEnv* env, jclass c, jintArray inputTypes, jint step, jintArray aggFns, jintArray aggInputs,
        jintArray aggMasks, jintArray aggInputTypes, jint stateFormat, jint outputMem)
{
    jsize n, na;
    pa_aggregation_desc d;
    memset(&d, 0, sizeof d);
    int32_t* types = ints_of(env, inputTypes, &n);
    pa_aggregate* aggs;
    fill_aggregates(env, aggFns, aggInputs, aggMasks, aggInputTypes, &aggs, &na);
    d.input_channel_count = n; d.input_types = types; d.aggregate_count = na; d.aggregates = aggs; d.step = step; d.output_mem = outputMem;
    d.state_format = stateFormat;
    pa_operator* op = 0;
    int32_t rc = pa_aggregation_create(&d, &op);
    free(aggs); free(types);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

/* OrderByOperator.OrderByOperatorFactory (OrderByOperator.java:45-120) */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createOrderBy(JNIEnv* env, jclass c, jintArray inputTypes, jintArray outputChannels, jintArray sortChannels,
        jintArray sortOrders, jint outputMem)
{
    jsize n, no, ns, nso;
    pa_order_by_desc d;
    memset(&d, 0, sizeof d);
    int32_t *types = ints_of(env, inputTypes, &n), *oc = ints_of(env, outputChannels, &no), *sc = ints_of(env, sortChannels, &ns), *so = ints_of(env, sortOrders, &nso);
    d.input_channel_count = n; d.input_types = types; d.output_channel_count = no; d.output_channels = oc; d.sort_channel_count = ns; d.sort_channels = sc;
    d.sort_orders = so; d.output_mem = outputMem;
    pa_operator* op = 0;
    int32_t rc = ns == nso ? pa_order_by_create(&d, &op) : PA_ERR_INVALID_ARGUMENT;
    free(so); free(sc); free(oc); free(types);
    if (rc == PA_ERR_INVALID_ARGUMENT && ns != nso) { throw_message(env, rc, "one sort order per sort channel"); return 0; }
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

/* DynamicFilterSourceOperator.DynamicFilterSourceOperatorFactory (DynamicFilterSourceOperator.java:74-139; LocalExecutionPlanner.java:2536-2539) */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createDynamicFilterSource(JNIEnv* env, jclass c, jintArray inputTypes, jintArray filterChannels,
        jint maxDistinctValues, jint minMaxCollectionLimit, jlong maxFilterSizeBytes)
{
    jsize n, nf;
    pa_dynamic_filter_source_desc d;
    memset(&d, 0, sizeof d);
    int32_t *types = ints_of(env, inputTypes, &n), *fc = ints_of(env, filterChannels, &nf);
    d.input_channel_count = n; d.input_types = types; d.filter_channel_count = nf; d.filter_channels = fc; d.max_distinct_values = maxDistinctValues;
    d.min_max_collection_limit = minMaxCollectionLimit; d.max_filter_size_bytes = maxFilterSizeBytes;
    pa_operator* op = 0;
    int32_t rc = pa_dynamic_filter_source_create(&d, &op);
    free(fc); free(types);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}
/* The TupleDomain handed to dynamicPredicateConsumer once the operator has finished: null while it is not ready, else
 * long[1 + 6 * channels]: [0] isAll (TupleDomain.all(): nothing was collected), then per filter channel: kind (pa_domain_kind), valueCount,
 * type, values address, values bytes, offsets address (VARCHAR) -- host memory of the operator, valid until it is closed. */
static int64_t value_width(int32_t type);
JNIEXPORT jlongArray JNICALL Java_io_trino_gpu_GpuNative_dynamicFilterPoll(JNIEnv* env, jclass c, jlong h, jint channels)
{
    if (channels < 0 || channels > MAX_CHANNELS) { throw_message(env, PA_ERR_INVALID_ARGUMENT, "bad filter channel count"); return 0; }
    pa_domain domains[MAX_CHANNELS];
    memset(domains, 0, sizeof domains);
    int32_t is_all = 0;
    int32_t rc = pa_dynamic_filter_poll((pa_operator*)(intptr_t)h, &is_all, domains, channels);
    if (rc < 0) { throw_native(env, rc); return 0; }
    if (rc == 0) return 0;
    const jsize len = 1 + 6 * channels;
    jlong* v = (jlong*)calloc((size_t)len, sizeof(jlong));
    v[0] = is_all;
    for (jint i = 0; i < channels; i++) {
        jlong* r = v + 1 + 6 * i;
        const pa_column* col = &domains[i].values;
        r[0] = domains[i].kind; r[1] = domains[i].value_count; r[2] = col->type; r[3] = (jlong)(intptr_t)col->values;
        r[4] = col->encoding == PA_VARWIDTH ? (domains[i].value_count > 0 && col->offsets ? col->offsets[domains[i].value_count] : 0)
                                             : (jlong)domains[i].value_count * value_width(col->type);
        r[5] = (jlong)(intptr_t)col->offsets;
    }
    jlongArray result = (*env)->NewLongArray(env, len);
    (*env)->SetLongArrayRegion(env, result, 0, len, v);
    free(v);
    return result;
}

/* The worker's HBM budget for operator memory (the memory pool as the device sees it) and its use: long[3] = in use, cached, limit */
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_memorySetLimit(JNIEnv* env, jclass c, jlong bytes) { CHECK(pa_memory_set_limit(bytes)); }
JNIEXPORT jlongArray JNICALL Java_io_trino_gpu_GpuNative_memoryStats(JNIEnv* env, jclass c)
{
    int64_t v[3] = {0, 0, 0};
    int32_t rc = pa_memory_stats(&v[0], &v[1], &v[2]);
    if (rc < 0) { throw_native(env, rc); return 0; }
    jlong j[3] = {v[0], v[1], v[2]};
    jlongArray result = (*env)->NewLongArray(env, 3);
    (*env)->SetLongArrayRegion(env, result, 0, 3, j);
    return result;
}

/* ---- ScanFilterAndProjectOperator: the ConnectorPageSource stays on the Java side, the native operator pulls pages through it ----
 * pageSource: an io.trino.gpu.GpuPageSource (java/io/trino/gpu/GpuPageSource.java), whose three methods are called back from inside
 * getOutput, on the Driver's thread:
 *   long[] nextPage()        ConnectorPageSource.getNextPage staged into the source's pinned slab; null = finished, else
 *                            [positions, channels, slab address, then per channel: type, encoding, values offset, offsets offset,
 *                             nulls offset, loaded] -- offsets inside the slab, -1 = absent; loaded = 0: a LazyBlock not loaded yet
 *   long[] loadBlock(int c)  LazyBlock.getLoadedBlock of channel c of that page, staged (possibly into another slab): the ADDRESSES of
 *                            [values, offsets, nulls], 0 = absent
 *   void close()             ConnectorPageSource.close
 * A Java exception thrown by one of them stays pending and surfaces when the native returns (the operator reports PA_ERR_DEVICE). */
typedef struct scan_source {
    JavaVM* vm;
    jobject source;              /* global reference */
    jmethodID next_page, load_block, close;
    pa_column cols[MAX_CHANNELS];
    char* base;
} scan_source;
static JNIEnv* scan_env(scan_source* s)
{
    JNIEnv* env = 0;
    return (*s->vm)->GetEnv(s->vm, (void**)&env, JNI_VERSION_1_8) == JNI_OK ? env : 0;
}
static int32_t scan_next_page(void* ctx, pa_page* page)
{
    scan_source* s = (scan_source*)ctx;
    JNIEnv* env = scan_env(s);
    if (!env) return PA_ERR_DEVICE;
    jlongArray a = (jlongArray)(*env)->CallObjectMethod(env, s->source, s->next_page);
    if ((*env)->ExceptionCheck(env)) return PA_ERR_DEVICE;
    if (!a) return 0;
    const jsize len = (*env)->GetArrayLength(env, a);
    jlong* v = (*env)->GetLongArrayElements(env, a, 0);
    int32_t rc = 1;
    const jlong channels = len >= 3 ? v[1] : -1;
    if (channels < 0 || channels > MAX_CHANNELS || len != 3 + 6 * channels) rc = PA_ERR_INVALID_ARGUMENT;
    else {
        s->base = (char*)(intptr_t)v[2];
        memset(s->cols, 0, sizeof s->cols);
        for (jlong i = 0; i < channels; i++) {
            const jlong* r = v + 3 + 6 * i;
            pa_column* col = &s->cols[i];
            col->type = (int32_t)r[0];
            col->encoding = (int32_t)r[1];
            if (r[5]) {   /* loaded; else values and dictionary stay NULL: a LazyBlock (include/presto_amd.h, pa_page_source) */
                col->values = r[2] >= 0 ? s->base + r[2] : 0;
                col->offsets = r[3] >= 0 ? (const int32_t*)(s->base + r[3]) : 0;
                col->nulls = r[4] >= 0 ? (const uint8_t*)(s->base + r[4]) : 0;
            }
        }
        memset(page, 0, sizeof *page);
        page->position_count = (int32_t)v[0];
        page->channel_count = (int32_t)channels;
        page->columns = s->cols;
        page->mem = PA_MEM_HOST;
    }
    (*env)->ReleaseLongArrayElements(env, a, v, JNI_ABORT);
    return rc;
}
static int64_t scan_load_block(void* ctx, int32_t channel, pa_column* column)
{
    scan_source* s = (scan_source*)ctx;
    JNIEnv* env = scan_env(s);
    if (!env) return PA_ERR_DEVICE;
    jlongArray a = (jlongArray)(*env)->CallObjectMethod(env, s->source, s->load_block, (jint)channel);
    if ((*env)->ExceptionCheck(env) || !a) return PA_ERR_DEVICE;
    jlong* v = (*env)->GetLongArrayElements(env, a, 0);
    const int ok = (*env)->GetArrayLength(env, a) == 3 && channel >= 0 && channel < MAX_CHANNELS;
    if (ok) {
        *column = s->cols[channel];   /* type and encoding as nextPage announced them */
        column->values = (const void*)(intptr_t)v[0];
        column->offsets = (const int32_t*)(intptr_t)v[1];
        column->nulls = (const uint8_t*)(intptr_t)v[2];
    }
    (*env)->ReleaseLongArrayElements(env, a, v, JNI_ABORT);
    return ok && column->values ? 0 : PA_ERR_INVALID_ARGUMENT;
}
static void scan_close(void* ctx)
{
    scan_source* s = (scan_source*)ctx;
    JNIEnv* env = scan_env(s);
    if (env) {
        (*env)->CallVoidMethod(env, s->source, s->close);
        (*env)->DeleteGlobalRef(env, s->source);
    }
    free(s);
}
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createScanFilterProject(JNIEnv* env, jclass c, jobject pageSource, jintArray inputTypes, jintArray typeParams,
        jlong filter, jlongArray projections, jlong minOutputPageBytes, jint minOutputPageRows, jint outputMem)
{
    jsize n, np_, nproj = (*env)->GetArrayLength(env, projections);
    pa_filter_project_desc d;
    memset(&d, 0, sizeof d);
    scan_source* s = (scan_source*)calloc(1, sizeof(scan_source));
    (*env)->GetJavaVM(env, &s->vm);
    jclass cls = (*env)->GetObjectClass(env, pageSource);
    s->next_page = (*env)->GetMethodID(env, cls, "nextPage", "()[J");
    s->load_block = (*env)->GetMethodID(env, cls, "loadBlock", "(I)[J");
    s->close = (*env)->GetMethodID(env, cls, "close", "()V");
    if (!s->next_page || !s->load_block || !s->close) {
        free(s);
        throw_message(env, PA_ERR_INVALID_ARGUMENT, "pageSource must be an io.trino.gpu.GpuPageSource");
        return 0;
    }
    s->source = (*env)->NewGlobalRef(env, pageSource);
    int32_t* types = ints_of(env, inputTypes, &n);
    int32_t* params = ints_of(env, typeParams, &np_);
    pa_expr* pe = (pa_expr*)calloc((size_t)(nproj > 0 ? nproj : 1), sizeof(pa_expr));
    jlong* ph = (*env)->GetLongArrayElements(env, projections, 0);
    for (jsize i = 0; i < nproj; i++) pe[i] = ((native_expr*)(intptr_t)ph[i])->expr;
    (*env)->ReleaseLongArrayElements(env, projections, ph, JNI_ABORT);
    d.input_channel_count = n; d.input_types = types; d.input_type_params = np_ == n ? params : 0;
    d.filter = filter ? &((native_expr*)(intptr_t)filter)->expr : 0;
    d.projection_count = nproj; d.projections = pe; d.output_mem = outputMem;
    d.min_output_page_bytes = minOutputPageBytes; d.min_output_page_rows = minOutputPageRows;
    pa_page_source src = {s, scan_next_page, scan_load_block, scan_close};
    pa_operator* op = 0;
    int32_t rc = pa_scan_filter_project_create(&d, &src, &op);
    free(pe); free(params); free(types);
    if (rc < 0) {   /* the operator was not made: the source is still ours */
        (*env)->DeleteGlobalRef(env, s->source);
        free(s);
        throw_native(env, rc);
        return 0;
    }
    return (jlong)(intptr_t)op;
}
/* ScanFilterAndProjectOperator's statistics: long[4] = processed positions, materialised bytes, lazy blocks loaded, lazy blocks never loaded */
JNIEXPORT jlongArray JNICALL Java_io_trino_gpu_GpuNative_scanStats(JNIEnv* env, jclass c, jlong h)
{
    int64_t v[4] = {0, 0, 0, 0};
    int32_t rc = pa_scan_stats((pa_operator*)(intptr_t)h, &v[0], &v[1], &v[2], &v[3]);
    if (rc < 0) { throw_native(env, rc); return 0; }
    jlong j[4] = {v[0], v[1], v[2], v[3]};
    jlongArray result = (*env)->NewLongArray(env, 4);
    (*env)->SetLongArrayRegion(env, result, 0, 4, j);
    return result;
}

/* [Scan]FilterAndProject -> (Hash)Aggregation of one pipeline as ONE device pass (pa_fused_aggregation_create): the planner hook
 * collapses the two neighbouring operator factories when both are on the device.  Aggregate channels index the projections. */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createFusedAggregation(JNIEnv* env, jclass c, jintArray inputTypes, jintArray typeParams, jlong filter,
        jlongArray projections, jintArray projectionTypes, jintArray groupByChannels, jint step, jintArray aggFns, jintArray aggInputs, jintArray aggMasks,
        jintArray aggInputTypes, jint expectedGroups, jint outputMem)
{
    jsize n, np_, npt, ng, na, nproj = (*env)->GetArrayLength(env, projections);
    pa_fused_aggregation_desc d;
    memset(&d, 0, sizeof d);
    int32_t *types = ints_of(env, inputTypes, &n), *params = ints_of(env, typeParams, &np_), *ptypes = ints_of(env, projectionTypes, &npt);
    int32_t* gb = ints_of(env, groupByChannels, &ng);
    pa_aggregate* aggs;
    fill_aggregates(env, aggFns, aggInputs, aggMasks, aggInputTypes, &aggs, &na);
    pa_expr* pe = (pa_expr*)calloc((size_t)(nproj > 0 ? nproj : 1), sizeof(pa_expr));
    jlong* ph = (*env)->GetLongArrayElements(env, projections, 0);
    for (jsize i = 0; i < nproj; i++) pe[i] = ((native_expr*)(intptr_t)ph[i])->expr;
    (*env)->ReleaseLongArrayElements(env, projections, ph, JNI_ABORT);
    d.filter_project.input_channel_count = n; d.filter_project.input_types = types; d.filter_project.input_type_params = np_ == n ? params : 0;
    d.filter_project.filter = filter ? &((native_expr*)(intptr_t)filter)->expr : 0;
    d.filter_project.projection_count = nproj; d.filter_project.projections = pe; d.filter_project.output_mem = outputMem;
    d.aggregation.input_channel_count = npt; d.aggregation.input_types = ptypes; d.aggregation.group_by_count = ng; d.aggregation.group_by_channels = gb;
    d.aggregation.hash_channel = -1; d.aggregation.step = step; d.aggregation.aggregate_count = na; d.aggregation.aggregates = aggs;
    d.aggregation.expected_groups = expectedGroups; d.aggregation.output_mem = outputMem;
    pa_operator* op = 0;
    int32_t rc = npt == nproj ? pa_fused_aggregation_create(&d, &op) : PA_ERR_INVALID_ARGUMENT;
    free(pe); free(aggs); free(gb); free(ptypes); free(params); free(types);
    if (rc == PA_ERR_INVALID_ARGUMENT && npt != nproj) { throw_message(env, rc, "one projection type per projection"); return 0; }
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createLookupSource(JNIEnv* env, jclass c)
{
    pa_lookup_source* ls = 0;
    int32_t rc = pa_lookup_source_create(&ls);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)ls;
}
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_destroyLookupSource(JNIEnv* env, jclass c, jlong h) { CHECK(pa_lookup_source_destroy((pa_lookup_source*)(intptr_t)h)); }

JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createHashBuilder(JNIEnv* env, jclass c, jlong bridge, jintArray inputTypes, jintArray joinChannels,
        jint hashChannel, jintArray outputChannels, jint expectedPositions)
{
    jsize n, nj, no;
    pa_hash_builder_desc d;
    memset(&d, 0, sizeof d);
    int32_t *types = ints_of(env, inputTypes, &n), *jc = ints_of(env, joinChannels, &nj), *oc = ints_of(env, outputChannels, &no);
    d.input_channel_count = n; d.input_types = types; d.join_channel_count = nj; d.join_channels = jc; d.hash_channel = hashChannel;
    d.output_channel_count = no; d.output_channels = oc; d.expected_positions = expectedPositions;
    pa_operator* op = 0;
    int32_t rc = pa_hash_builder_create(&d, (pa_lookup_source*)(intptr_t)bridge, &op);
    free(oc); free(jc); free(types);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

/* OperatorFactories.innerJoin / probeOuterJoin / lookupOuterJoin / fullOuterJoin: joinType = pa_join_type; outer = 1 creates the
 * LookupOuterOperator of the same bridge; filter = a newExpression handle over [build channels, probe channels] (the
 * JoinFilterFunction the planner compiled for this join, JoinFilterFunctionCompiler.java) or 0 */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createLookupJoin(JNIEnv* env, jclass c, jlong bridge, jintArray probeTypes, jintArray probeJoinChannels,
        jint probeHashChannel, jintArray probeOutputChannels, jint joinType, jboolean outputSingleMatch, jboolean outer, jint outputMem, jlong filter)
{
    jsize n, nj, no;
    pa_lookup_join_desc d;
    memset(&d, 0, sizeof d);
    int32_t *types = ints_of(env, probeTypes, &n), *jc = ints_of(env, probeJoinChannels, &nj), *oc = ints_of(env, probeOutputChannels, &no);
    d.probe_channel_count = n; d.probe_types = types; d.join_channel_count = nj; d.probe_join_channels = jc; d.probe_hash_channel = probeHashChannel;
    d.probe_output_channel_count = no; d.probe_output_channels = oc; d.output_mem = outputMem; d.join_type = joinType;
    d.output_single_match = outputSingleMatch ? 1 : 0;
    d.filter = filter ? &((native_expr*)(intptr_t)filter)->expr : 0;
    pa_operator* op = 0;
    int32_t rc = outer ? pa_lookup_outer_create(&d, (pa_lookup_source*)(intptr_t)bridge, &op) : pa_lookup_join_create(&d, (pa_lookup_source*)(intptr_t)bridge, &op);
    free(oc); free(jc); free(types);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

/* FilterAndProject -> LookupJoin (INNER, no filter function) [-> (Hash)Aggregation] of one pipeline behind one handle:
 * pa_fused_join_create (aggFns == null) / pa_fused_join_aggregation_create.  The probe page is the projection output; joinedTypes =
 * types of the join's output page [probe output channels, build output channels] (the aggregation's input). */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createFusedJoin(JNIEnv* env, jclass c, jlong bridge, jintArray inputTypes, jintArray typeParams, jlong filter,
        jlongArray projections, jintArray projectionTypes, jintArray probeJoinChannels, jintArray probeOutputChannels, jintArray joinedTypes,
        jintArray groupByChannels, jint step, jintArray aggFns, jintArray aggInputs, jintArray aggMasks, jintArray aggInputTypes, jint expectedGroups, jint outputMem)
{
    jsize n, np_, npt, nj, no, njt, ng, na = 0, nproj = (*env)->GetArrayLength(env, projections);
    pa_fused_join_aggregation_desc d;
    memset(&d, 0, sizeof d);
    int32_t *types = ints_of(env, inputTypes, &n), *params = ints_of(env, typeParams, &np_), *ptypes = ints_of(env, projectionTypes, &npt);
    int32_t *jc = ints_of(env, probeJoinChannels, &nj), *oc = ints_of(env, probeOutputChannels, &no), *jt = ints_of(env, joinedTypes, &njt);
    int32_t* gb = ints_of(env, groupByChannels, &ng);
    pa_aggregate* aggs = 0;
    if (aggFns) fill_aggregates(env, aggFns, aggInputs, aggMasks, aggInputTypes, &aggs, &na);
    pa_expr* pe = (pa_expr*)calloc((size_t)(nproj > 0 ? nproj : 1), sizeof(pa_expr));
    jlong* ph = (*env)->GetLongArrayElements(env, projections, 0);
    for (jsize i = 0; i < nproj; i++) pe[i] = ((native_expr*)(intptr_t)ph[i])->expr;
    (*env)->ReleaseLongArrayElements(env, projections, ph, JNI_ABORT);
    d.filter_project.input_channel_count = n; d.filter_project.input_types = types; d.filter_project.input_type_params = np_ == n ? params : 0;
    d.filter_project.filter = filter ? &((native_expr*)(intptr_t)filter)->expr : 0;
    d.filter_project.projection_count = nproj; d.filter_project.projections = pe; d.filter_project.output_mem = PA_MEM_DEVICE;
    d.join.probe_channel_count = npt; d.join.probe_types = ptypes; d.join.join_channel_count = nj; d.join.probe_join_channels = jc;
    d.join.probe_hash_channel = -1; d.join.probe_output_channel_count = no; d.join.probe_output_channels = oc; d.join.join_type = PA_JOIN_INNER;
    d.join.output_mem = aggFns ? PA_MEM_DEVICE : outputMem;
    d.aggregation.input_channel_count = njt; d.aggregation.input_types = jt; d.aggregation.group_by_count = ng; d.aggregation.group_by_channels = gb;
    d.aggregation.hash_channel = -1; d.aggregation.step = step; d.aggregation.aggregate_count = na; d.aggregation.aggregates = aggs;
    d.aggregation.expected_groups = expectedGroups; d.aggregation.output_mem = outputMem;
    pa_operator* op = 0;
    int32_t rc;
    if (aggFns) rc = pa_fused_join_aggregation_create(&d, (pa_lookup_source*)(intptr_t)bridge, &op);
    else {
        pa_fused_join_desc j;
        memset(&j, 0, sizeof j);
        j.filter_project = d.filter_project;
        j.join = d.join;
        rc = pa_fused_join_create(&j, (pa_lookup_source*)(intptr_t)bridge, &op);
    }
    free(pe); free(aggs); free(gb); free(jt); free(oc); free(jc); free(ptypes); free(params); free(types);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createTopN(JNIEnv* env, jclass c, jintArray inputTypes, jint count, jintArray sortChannels, jintArray sortOrders, jint outputMem)
{
    jsize n, ns, no;
    pa_topn_desc d;
    memset(&d, 0, sizeof d);
    int32_t *types = ints_of(env, inputTypes, &n), *sc = ints_of(env, sortChannels, &ns), *so = ints_of(env, sortOrders, &no);
    d.input_channel_count = n; d.input_types = types; d.n = count; d.sort_channel_count = ns; d.sort_channels = sc; d.sort_orders = so; d.output_mem = outputMem;
    pa_operator* op = 0;
    int32_t rc = pa_topn_create(&d, &op);
    free(so); free(sc); free(types);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

/* The dynamic filter of an inner join, installed by the probe-side ScanFilterAndProject's factory once the build is done. */
JNIEXPORT jboolean JNICALL Java_io_trino_gpu_GpuNative_setDynamicFilter(JNIEnv* env, jclass c, jlong fpOperator, jint channel, jlong lookupSource)
{
    int32_t rc = pa_filter_project_set_dynamic_filter((pa_operator*)(intptr_t)fpOperator, channel, (pa_lookup_source*)(intptr_t)lookupSource);
    CHECK(rc);
    return rc == 1;
}

/* The planner's note that the only consumer of an aggregation's output is a TopN over it (LocalExecutionPlanner.visitTopN over
 * visitAggregation): groups that cannot be among its n best rows may be left out. */
JNIEXPORT jboolean JNICALL Java_io_trino_gpu_GpuNative_setOutputTopNHint(JNIEnv* env, jclass c, jlong aggregationOperator, jlong n, jintArray sortChannels,
                                                                          jintArray sortOrders)
{
    jsize count = (*env)->GetArrayLength(env, sortChannels);
    jint* ch = (*env)->GetIntArrayElements(env, sortChannels, 0);
    jint* od = (*env)->GetIntArrayElements(env, sortOrders, 0);
    int32_t rc = pa_aggregation_set_output_topn_hint((pa_operator*)(intptr_t)aggregationOperator, n, (int32_t)count, (const int32_t*)ch, (const int32_t*)od);
    (*env)->ReleaseIntArrayElements(env, sortChannels, ch, 0);
    (*env)->ReleaseIntArrayElements(env, sortOrders, od, 0);
    CHECK(rc);
    return rc == 1;
}

/* ---- exchange between the GPUs of a node (one JVM worker per GPU): the coordinator ships the 128-byte id ---- */
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_commUniqueId(JNIEnv* env, jclass c, jbyteArray out)
{
    jbyte* b = (*env)->GetByteArrayElements(env, out, 0);
    int32_t rc = pa_comm_unique_id(b);
    (*env)->ReleaseByteArrayElements(env, out, b, 0);
    CHECK(rc);
}
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_commCreate(JNIEnv* env, jclass c, jbyteArray id, jint rank, jint world)
{
    pa_comm* comm = 0;
    jbyte* b = (*env)->GetByteArrayElements(env, id, 0);
    int32_t rc = pa_comm_create(b, rank, world, &comm);
    (*env)->ReleaseByteArrayElements(env, id, b, JNI_ABORT);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)comm;
}
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_commDestroy(JNIEnv* env, jclass c, jlong comm) { CHECK(pa_comm_destroy((pa_comm*)(intptr_t)comm)); }
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_exchangeCreate(JNIEnv* env, jclass c, jlong comm, jintArray types, jintArray partitionChannels, jint hashChannel, jint sinkCount)
{
    jsize n, np_;
    pa_exchange_desc d;
    memset(&d, 0, sizeof d);
    int32_t *t = ints_of(env, types, &n), *pc = ints_of(env, partitionChannels, &np_);
    d.channel_count = n; d.types = t; d.partition_channel_count = np_; d.partition_channels = pc; d.hash_channel = hashChannel;
    d.partition_rule = -1; d.sink_count = sinkCount;
    pa_exchange* ex = 0;
    int32_t rc = pa_exchange_create(&d, (pa_comm*)(intptr_t)comm, &ex);
    free(pc); free(t);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)ex;
}
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_exchangeDestroy(JNIEnv* env, jclass c, jlong ex) { CHECK(pa_exchange_destroy((pa_exchange*)(intptr_t)ex)); }
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createPartitionedOutput(JNIEnv* env, jclass c, jlong ex)
{
    pa_operator* op = 0;
    int32_t rc = pa_partitioned_output_create((pa_exchange*)(intptr_t)ex, 0, &op);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_createExchangeSource(JNIEnv* env, jclass c, jlong ex, jint outputMem)
{
    pa_operator* op = 0;
    int32_t rc = pa_exchange_source_create((pa_exchange*)(intptr_t)ex, outputMem, 0, &op);
    if (rc < 0) { throw_native(env, rc); return 0; }
    return (jlong)(intptr_t)op;
}

/* ---- Operator protocol (Operator.java:21-103) ---- */
JNIEXPORT jboolean JNICALL Java_io_trino_gpu_GpuNative_needsInput(JNIEnv* env, jclass c, jlong h)
{
    int32_t rc = pa_op_needs_input((pa_operator*)(intptr_t)h);
    CHECK(rc);
    return rc == 1;
}
JNIEXPORT jboolean JNICALL Java_io_trino_gpu_GpuNative_isBlocked(JNIEnv* env, jclass c, jlong h)
{
    int32_t rc = pa_op_is_blocked((pa_operator*)(intptr_t)h);
    CHECK(rc);
    return rc == 1;
}
JNIEXPORT jboolean JNICALL Java_io_trino_gpu_GpuNative_isFinished(JNIEnv* env, jclass c, jlong h)
{
    int32_t rc = pa_op_is_finished((pa_operator*)(intptr_t)h);
    CHECK(rc);
    return rc == 1;
}
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_finish(JNIEnv* env, jclass c, jlong h) { CHECK(pa_op_finish((pa_operator*)(intptr_t)h)); }
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_close(JNIEnv* env, jclass c, jlong h) { CHECK(pa_op_close((pa_operator*)(intptr_t)h)); }
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_memoryBytes(JNIEnv* env, jclass c, jlong h) { return pa_op_memory_bytes((pa_operator*)(intptr_t)h); }

/* Released PA_PAGE_RETAINED pages: the operator's release callback may run on any Driver thread, inside any later native call on the
 * operator's handle; it must not call back into the library (or the JVM): the token is queued and PinnedPagePool collects it. */
static struct { int64_t* tokens; int32_t count, capacity; volatile int lock; } released = {0, 0, 0, 0};
static void released_lock(void) { while (__sync_lock_test_and_set(&released.lock, 1)) {} }
static void released_unlock(void) { __sync_lock_release(&released.lock); }
static void released_token(void* ctx)
{
    released_lock();
    if (released.count == released.capacity) {
        released.capacity = released.capacity ? 2 * released.capacity : 256;
        released.tokens = (int64_t*)realloc(released.tokens, (size_t)released.capacity * sizeof(int64_t));
    }
    released.tokens[released.count++] = (int64_t)(intptr_t)ctx;
    released_unlock();
}
/* moves up to out.length released tokens into `out`; returns how many */
JNIEXPORT jint JNICALL Java_io_trino_gpu_GpuNative_drainReleased(JNIEnv* env, jclass c, jlongArray out)
{
    const jsize cap = (*env)->GetArrayLength(env, out);
    jlong* v = (*env)->GetLongArrayElements(env, out, 0);
    released_lock();
    const int32_t n = released.count < cap ? released.count : cap;
    for (int32_t i = 0; i < n; i++) v[i] = released.tokens[released.count - n + i];
    released.count -= n;
    released_unlock();
    (*env)->ReleaseLongArrayElements(env, out, v, 0);
    return n;
}

/* addInput: block arrays at offsets inside one pinned direct ByteBuffer.  Per channel: type, encoding (FLAT / VARWIDTH /
 * DICTIONARY / RLE), and byte offsets of values / offsets / nulls / ids (-1 = absent); a DICTIONARY / RLE channel names its
 * dictionary as one more "channel" behind the page's own (dictionaryChannel[i], -1 = none) with dictionarySize positions. */
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_addInput(JNIEnv* env, jclass c, jlong h, jint positions, jint channels, jintArray types,
        jintArray encodings, jlongArray valueOffsets, jlongArray offsetOffsets, jlongArray nullOffsets, jlongArray idOffsets, jintArray dictionaryChannel,
        jintArray dictionarySize, jobject pinned, jint retention, jlong releaseToken)
{
    char* base = (char*)(*env)->GetDirectBufferAddress(env, pinned);
    const jsize total = (*env)->GetArrayLength(env, types);   /* page channels + dictionaries */
    if (total > MAX_CHANNELS || channels > total) { throw_message(env, PA_ERR_INVALID_ARGUMENT, "page with more than 64 blocks (dictionaries included)"); return; }
    pa_column cols[MAX_CHANNELS];
    jint* t = (*env)->GetIntArrayElements(env, types, 0);
    jint* e = (*env)->GetIntArrayElements(env, encodings, 0);
    jlong* v = (*env)->GetLongArrayElements(env, valueOffsets, 0);
    jlong* o = (*env)->GetLongArrayElements(env, offsetOffsets, 0);
    jlong* n = (*env)->GetLongArrayElements(env, nullOffsets, 0);
    jlong* ids = (*env)->GetLongArrayElements(env, idOffsets, 0);
    jint* dc = (*env)->GetIntArrayElements(env, dictionaryChannel, 0);
    jint* ds = (*env)->GetIntArrayElements(env, dictionarySize, 0);
    for (jsize i = 0; i < total; i++) {
        pa_column col;
        memset(&col, 0, sizeof col);
        col.type = t[i];
        col.encoding = e[i];
        col.values = v[i] >= 0 ? base + v[i] : 0;
        col.offsets = o[i] >= 0 ? (const int32_t*)(base + o[i]) : 0;
        col.nulls = n[i] >= 0 ? (const uint8_t*)(base + n[i]) : 0;      /* boolean[] valueIsNull, 1 B / position */
        col.ids = ids[i] >= 0 ? (const int32_t*)(base + ids[i]) : 0;
        col.dictionary = dc[i] >= 0 ? &cols[dc[i]] : 0;
        col.dictionary_size = ds[i];
        cols[i] = col;
    }
    pa_page page;
    memset(&page, 0, sizeof page);
    page.position_count = positions;
    page.channel_count = channels;
    page.columns = cols;
    page.mem = PA_MEM_HOST;
    /* retention (PinnedPagePool): 0 = the slab may be reused when this call returns; 1 = it is kept until the operator is closed
     * (PA_PAGE_STABLE); 2 = it is kept until the operator releases the page (PA_PAGE_RETAINED): releaseToken then shows up in
     * drainReleased and the pool puts the slab back on its free list.  The slab is pa_host_malloc_pinned memory: the device reads it in place. */
    page.flags = retention == 1 ? (PA_PAGE_STABLE | PA_PAGE_PINNED) : retention == 2 ? (PA_PAGE_RETAINED | PA_PAGE_PINNED) : 0;
    if (retention == 2) {
        page.release = &released_token;
        page.release_ctx = (void*)(intptr_t)releaseToken;
    }
    int32_t rc = pa_op_add_input((pa_operator*)(intptr_t)h, &page);
    (*env)->ReleaseIntArrayElements(env, types, t, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, encodings, e, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, valueOffsets, v, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, offsetOffsets, o, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, nullOffsets, n, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, idOffsets, ids, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, dictionaryChannel, dc, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, dictionarySize, ds, JNI_ABORT);
    CHECK(rc);
}

/* getOutput: returns null when the operator has no page, else long[2 + 6 * channels]:
 *   [0] positionCount  [1] channelCount  then per channel: type, values address, values bytes, offsets address, nulls address, 0
 * The addresses are the operator's pinned output buffers (PA_MEM_HOST operators), valid until the next call on the handle;
 * GpuOperator wraps them (NewDirectByteBuffer on the Java side through wrapAddress) and copies them into fresh long[] /
 * int[] / byte[] for new LongArrayBlock(n, Optional.ofNullable(valueIsNull), values) etc. */
static int64_t value_width(int32_t type)   /* bytes per position of a flat block (common.hpp type_width) */
{
    switch (type) {
        case PA_BIGINT: case PA_DOUBLE: case PA_DECIMAL: return 8;
        case PA_LONG_DECIMAL: return 16;
        case PA_INTEGER: case PA_DATE: case PA_REAL: return 4;
        case PA_BOOLEAN: return 1;
        default: return 0;
    }
}
JNIEXPORT jlongArray JNICALL Java_io_trino_gpu_GpuNative_getOutput(JNIEnv* env, jclass c, jlong h)
{
    pa_page out;
    memset(&out, 0, sizeof out);
    int32_t rc = pa_op_get_output((pa_operator*)(intptr_t)h, &out);
    if (rc < 0) { throw_native(env, rc); return 0; }
    if (rc == 0) return 0;
    if (out.mem != PA_MEM_HOST) {   /* the addresses below are read by the JVM: a device page has nothing to wrap */
        throw_message(env, PA_ERR_ILLEGAL_STATE, "getOutput: the operator was created with PA_MEM_DEVICE output");
        return 0;
    }
    const jsize len = 2 + 6 * out.channel_count;
    jlong* v = (jlong*)calloc((size_t)len, sizeof(jlong));
    v[0] = out.position_count;
    v[1] = out.channel_count;
    for (int32_t i = 0; i < out.channel_count; i++) {
        const pa_column* col = &out.columns[i];
        jlong* r = v + 2 + 6 * i;
        r[0] = col->type;
        r[1] = (jlong)(intptr_t)col->values;
        if (col->encoding == PA_VARWIDTH) r[2] = out.position_count > 0 ? col->offsets[out.position_count] : 0;
        else r[2] = (jlong)out.position_count * value_width(col->type);
        r[3] = (jlong)(intptr_t)col->offsets;
        r[4] = (jlong)(intptr_t)col->nulls;
    }
    jlongArray result = (*env)->NewLongArray(env, len);
    (*env)->SetLongArrayRegion(env, result, 0, len, v);
    free(v);
    return result;
}
JNIEXPORT jobject JNICALL Java_io_trino_gpu_GpuNative_wrapAddress(JNIEnv* env, jclass c, jlong address, jlong bytes)
{
    return (*env)->NewDirectByteBuffer(env, (void*)(intptr_t)address, bytes);
}
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_bufferAddress(JNIEnv* env, jclass c, jobject buffer)
{
    return (jlong)(intptr_t)(*env)->GetDirectBufferAddress(env, buffer);
}

/* ---- page wire format (PagesSerde): exchange pages of CPU workers feed device operators and the reverse, without a Java-side decode ---- */
/* The page an operator would return next, serialised into `out` (a direct buffer): the SerializedPage frame of PagesSerde.serialize
 * (LZ4 when compress and it pays).  Returns the frame's size, 0 when the operator has no page; the operator may have been created with
 * PA_MEM_DEVICE output -- the page never has to exist on the Java heap. */
JNIEXPORT jlong JNICALL Java_io_trino_gpu_GpuNative_getOutputSerialized(JNIEnv* env, jclass c, jlong h, jobject out, jboolean compress)
{
    pa_page page;
    memset(&page, 0, sizeof page);
    int32_t rc = pa_op_get_output((pa_operator*)(intptr_t)h, &page);
    if (rc < 0) { throw_native(env, rc); return 0; }
    if (rc == 0) return 0;
    void* dst = (*env)->GetDirectBufferAddress(env, out);
    const jlong cap = (*env)->GetDirectBufferCapacity(env, out);
    int64_t size = compress ? pa_page_serialize_lz4(&page, dst, cap, 0) : pa_page_serialize(&page, dst, cap, 0);
    if (size < 0) { throw_native(env, (int32_t)size); return 0; }
    return size;
}
/* A SerializedPage frame (direct buffer, `size` bytes) decoded into HBM and handed to the operator as its next input page; expectedTypes:
 * the consumer's declared channel types (LONG_ARRAY blocks become BIGINT or DOUBLE, INT_ARRAY blocks INTEGER or DATE ...). */
JNIEXPORT void JNICALL Java_io_trino_gpu_GpuNative_addInputSerialized(JNIEnv* env, jclass c, jlong h, jobject bytes, jlong size, jintArray expectedTypes)
{
    jsize n;
    int32_t* types = ints_of(env, expectedTypes, &n);
    pa_page_buffer* buffer = 0;
    int32_t rc = pa_page_deserialize_typed((*env)->GetDirectBufferAddress(env, bytes), size, types, n, 0, &buffer);
    free(types);
    if (rc < 0) { throw_native(env, rc); return; }
    pa_page page;
    memset(&page, 0, sizeof page);
    rc = pa_page_buffer_page(buffer, &page);
    if (rc >= 0 && page.position_count > 0) rc = pa_op_add_input((pa_operator*)(intptr_t)h, &page);   /* Driver.java:391: never an empty page */
    int32_t rc2 = pa_page_buffer_free(buffer);   /* (the operator has read or copied the page: it is not flagged stable) */
    if (rc < 0) { throw_native(env, rc); return; }
    CHECK(rc2);
}
