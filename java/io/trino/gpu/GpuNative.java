/*
 * The native methods of jni/presto_amd_jni.c, one per entry of include/presto_amd.h.  Not compiled in the authoring image
 * (no JDK there); the C side is kept in step with the header by tests/test_lib_cpu.py.
 */
package io.trino.gpu;

import java.nio.ByteBuffer;

final class GpuNative
{
    static {
        System.loadLibrary("presto_amd_jni"); // links libpresto_amd.so
        if (abiVersion() != 10) {
            throw new IllegalStateException("libpresto_amd.so ABI version " + abiVersion() + " != 10");
        }
    }

    private GpuNative() {}

    static native int abiVersion();
    static native void init(int device);
    static native int deviceCount();
    static native ByteBuffer hostMallocPinned(long bytes);
    static native void hostFreePinned(ByteBuffer buffer);

    static native long newExpression(int root, int[] kinds, int[] ops, int[] types, int[] typeParams, int[] channels, int[] isNull, int[] nargs,
            int[] firstArg, long[] longs, double[] doubles, byte[][] strings, int[] args);
    static native void freeExpression(long expression);

    static native long createFilterProject(int[] inputTypes, int[] typeParams, long filter, long[] projections, long minOutputPageBytes,
            int minOutputPageRows, int outputMem);
    /**
     * HashAggregationOperatorFactory's arguments: globalAggregationGroupIds (may be null), groupIdChannel (an index among the group-by
     * columns, -1 = Optional.empty()), produceDefaultOutput, maxPartialMemory in bytes (0 = no early flush of a PARTIAL step),
     * stateFormat 0 = flat state channels, 1 = the reference's own intermediate types (ROW(...) / BIGINT per aggregate).
     */
    static native long createHashAggregation(int[] inputTypes, int[] typeParams, int[] groupByChannels, int[] globalAggregationGroupIds, int hashChannel,
            int groupIdChannel, int step, boolean produceDefaultOutput, int[] aggFns, int[] aggInputs, int[] aggMasks, int[] aggInputTypes,
            int expectedGroups, long maxPartialMemory, int stateFormat, int outputMem);
    /** AggregationOperator: ungrouped aggregates. */
    static native long createAggregation(int[] inputTypes, int step, int[] aggFns, int[] aggInputs, int[] aggMasks, int[] aggInputTypes, int stateFormat,
            int outputMem);
    static native long createOrderBy(int[] inputTypes, int[] outputChannels, int[] sortChannels, int[] sortOrders, int outputMem);
    static native long createDynamicFilterSource(int[] inputTypes, int[] filterChannels, int maxDistinctValues, int minMaxCollectionLimit,
            long maxFilterSizeBytes);
    /** null while the operator has not finished; else long[1 + 6 * channels]: isAll, then per channel kind, valueCount, type, values address, values bytes, offsets address. */
    static native long[] dynamicFilterPoll(long dynamicFilterSourceOperator, int channels);
    /** ScanFilterAndProjectOperator over a page source that stays on the Java side (GpuPageSource: nextPage / loadBlock / close are called back). */
    static native long createScanFilterProject(GpuPageSource pageSource, int[] inputTypes, int[] typeParams, long filter, long[] projections,
            long minOutputPageBytes, int minOutputPageRows, int outputMem);
    /** long[4]: processed positions, materialised bytes, lazy blocks loaded, lazy blocks never loaded. */
    static native long[] scanStats(long scanOperator);
    static native void memorySetLimit(long bytes);
    /** long[3]: bytes held by operators, bytes cached for reuse, the limit. */
    static native long[] memoryStats();
    /** [Scan]FilterAndProject -> (Hash)Aggregation of one pipeline as one device pass; aggregate channels index the projections. */
    static native long createFusedAggregation(int[] inputTypes, int[] typeParams, long filter, long[] projections, int[] projectionTypes,
            int[] groupByChannels, int step, int[] aggFns, int[] aggInputs, int[] aggMasks, int[] aggInputTypes, int expectedGroups, int outputMem);
    static native long createLookupSource();
    static native void destroyLookupSource(long lookupSource);
    static native long createHashBuilder(long bridge, int[] inputTypes, int[] joinChannels, int hashChannel, int[] outputChannels, int expectedPositions);
    static native long createLookupJoin(long bridge, int[] probeTypes, int[] probeJoinChannels, int probeHashChannel, int[] probeOutputChannels,
            int joinType, boolean outputSingleMatch, boolean outer, int outputMem, long filterExpression);
    /** FilterAndProject -> LookupJoin [-> aggregation] of one pipeline behind one handle (aggFns == null: no aggregation). */
    static native long createFusedJoin(long bridge, int[] inputTypes, int[] typeParams, long filter, long[] projections, int[] projectionTypes,
            int[] probeJoinChannels, int[] probeOutputChannels, int[] joinedTypes, int[] groupByChannels, int step, int[] aggFns, int[] aggInputs,
            int[] aggMasks, int[] aggInputTypes, int expectedGroups, int outputMem);
    static native long createTopN(int[] inputTypes, int count, int[] sortChannels, int[] sortOrders, int outputMem);
    static native boolean setDynamicFilter(long filterProjectOperator, int channel, long lookupSource);

    /** The only consumer of the aggregation's output is TopNOperator(n, sortChannels, sortOrders): groups that cannot be among its n best rows may be left out. */
    static native boolean setOutputTopNHint(long aggregationOperator, long n, int[] sortChannels, int[] sortOrders);

    static native void commUniqueId(byte[] out128);
    static native long commCreate(byte[] id128, int rank, int world);
    static native void commDestroy(long comm);
    static native long exchangeCreate(long comm, int[] types, int[] partitionChannels, int hashChannel, int sinkCount);
    static native void exchangeDestroy(long exchange);
    static native long createPartitionedOutput(long exchange);
    static native long createExchangeSource(long exchange, int outputMem);

    static native boolean needsInput(long operator);
    static native boolean isBlocked(long operator);
    static native boolean isFinished(long operator);
    static native void finish(long operator);
    static native void close(long operator);
    static native long memoryBytes(long operator);
    /**
     * retention: 0 = the slab may be reused when the call returns; 1 = it is kept until the operator is closed; 2 = it is kept until the
     * operator releases the page -- releaseToken then shows up in drainReleased.
     */
    static native void addInput(long operator, int positions, int channels, int[] types, int[] encodings, long[] valueOffsets, long[] offsetOffsets,
            long[] nullOffsets, long[] idOffsets, int[] dictionaryChannel, int[] dictionarySize, ByteBuffer pinned, int retention, long releaseToken);
    /** Moves up to out.length tokens of released pages into out; returns how many. */
    static native int drainReleased(long[] out);
    /** The operator's next output page as a SerializedPage frame in `out` (0 = no page); works for operators with device output. */
    static native long getOutputSerialized(long operator, ByteBuffer out, boolean compress);
    /** A SerializedPage frame decoded on the device and handed to the operator as its next input page. */
    static native void addInputSerialized(long operator, ByteBuffer frame, long size, int[] expectedTypes);
    static native long[] getOutput(long operator);
    static native ByteBuffer wrapAddress(long address, long bytes);
    static native long bufferAddress(ByteBuffer directBuffer);
}
