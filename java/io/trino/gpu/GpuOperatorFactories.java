package io.trino.gpu;

import io.trino.operator.OperatorFactories;
import io.trino.operator.OperatorFactory;
import io.trino.operator.TrinoOperatorFactories;
import io.trino.operator.join.JoinBridgeManager;
import io.trino.operator.join.LookupSourceFactory;
import io.trino.spi.type.Type;
import io.trino.spiller.PartitioningSpillerFactory;
import io.trino.sql.planner.plan.PlanNodeId;
import io.trino.type.BlockTypeOperators;

import java.util.ArrayList;
import java.util.List;
import java.util.Map;
import java.util.Optional;
import java.util.OptionalInt;
import java.util.concurrent.ConcurrentHashMap;
import java.util.stream.IntStream;

/**
 * OperatorFactories (core/trino-main/src/main/java/io/trino/operator/OperatorFactories.java:27-84) on the device.  Bound with
 *   newOptionalBinder(binder, OperatorFactories.class).setBinding().to(GpuOperatorFactories.class)
 * (core/trino-main/src/main/java/io/trino/server/ServerMainModule.java:303 declares the default, TrinoOperatorFactories).
 * The JoinBridgeManager identifies the join: the build side's GpuHashBuilder factory registers its pa_lookup_source under the
 * same manager (GpuJoinBridges), the probe factories created here look it up.  The reference hands the join's filter function to
 * the BUILD side (LocalExecutionPlanner.createLookupSourceFactory -> HashBuilderOperatorFactory's JoinFilterFunctionFactory); the
 * device HashBuilder factory serialises that RowExpression (build channels first, then probe channels: JoinFilterFunctionCompiler's
 * numbering) and registers it next to the bridge (FILTERS), pa_lookup_join_desc.filter takes it.  A join the device path does not
 * cover (a filter expression RowExpressionSerializer cannot express, an unsupported key type -> PA_ERR_NOT_SUPPORTED) falls back
 * to the reference factories.
 */
public final class GpuOperatorFactories
        implements OperatorFactories
{
    // pa_join_type
    private static final int INNER = 0, PROBE_OUTER = 1, LOOKUP_OUTER = 2, FULL_OUTER = 3;
    private final OperatorFactories fallback = new TrinoOperatorFactories();
    /** pa_lookup_source* per join bridge; filled by the device HashBuilder factory of the same join. */
    static final Map<JoinBridgeManager<?>, Long> BRIDGES = new ConcurrentHashMap<>();
    /** newExpression handle of the join's filter function, for joins that have one and whose filter the serialiser covers. */
    static final Map<JoinBridgeManager<?>, Long> FILTERS = new ConcurrentHashMap<>();

    @Override
    public OperatorFactory innerJoin(int operatorId, PlanNodeId planNodeId, JoinBridgeManager<? extends LookupSourceFactory> lookupSourceFactory,
            boolean outputSingleMatch, boolean waitForBuild, boolean hasFilter, List<Type> probeTypes, List<Integer> probeJoinChannel,
            OptionalInt probeHashChannel, Optional<List<Integer>> probeOutputChannels, OptionalInt totalOperatorsCount,
            PartitioningSpillerFactory partitioningSpillerFactory, BlockTypeOperators blockTypeOperators)
    {
        Long bridge = BRIDGES.get(lookupSourceFactory);
        if (bridge == null || (hasFilter && !FILTERS.containsKey(lookupSourceFactory))) {
            return fallback.innerJoin(operatorId, planNodeId, lookupSourceFactory, outputSingleMatch, waitForBuild, hasFilter, probeTypes, probeJoinChannel,
                    probeHashChannel, probeOutputChannels, totalOperatorsCount, partitioningSpillerFactory, blockTypeOperators);
        }
        return lookupJoin(operatorId, planNodeId, bridge, INNER, outputSingleMatch, probeTypes, probeJoinChannel, probeHashChannel, probeOutputChannels,
                hasFilter ? FILTERS.get(lookupSourceFactory) : 0L);
    }

    @Override
    public OperatorFactory probeOuterJoin(int operatorId, PlanNodeId planNodeId, JoinBridgeManager<? extends LookupSourceFactory> lookupSourceFactory,
            boolean outputSingleMatch, boolean hasFilter, List<Type> probeTypes, List<Integer> probeJoinChannel, OptionalInt probeHashChannel,
            Optional<List<Integer>> probeOutputChannels, OptionalInt totalOperatorsCount, PartitioningSpillerFactory partitioningSpillerFactory,
            BlockTypeOperators blockTypeOperators)
    {
        Long bridge = BRIDGES.get(lookupSourceFactory);
        if (bridge == null || (hasFilter && !FILTERS.containsKey(lookupSourceFactory))) {
            return fallback.probeOuterJoin(operatorId, planNodeId, lookupSourceFactory, outputSingleMatch, hasFilter, probeTypes, probeJoinChannel,
                    probeHashChannel, probeOutputChannels, totalOperatorsCount, partitioningSpillerFactory, blockTypeOperators);
        }
        return lookupJoin(operatorId, planNodeId, bridge, PROBE_OUTER, outputSingleMatch, probeTypes, probeJoinChannel, probeHashChannel, probeOutputChannels,
                hasFilter ? FILTERS.get(lookupSourceFactory) : 0L);
    }

    @Override
    public OperatorFactory lookupOuterJoin(int operatorId, PlanNodeId planNodeId, JoinBridgeManager<? extends LookupSourceFactory> lookupSourceFactory,
            boolean waitForBuild, boolean hasFilter, List<Type> probeTypes, List<Integer> probeJoinChannel, OptionalInt probeHashChannel,
            Optional<List<Integer>> probeOutputChannels, OptionalInt totalOperatorsCount, PartitioningSpillerFactory partitioningSpillerFactory,
            BlockTypeOperators blockTypeOperators)
    {
        Long bridge = BRIDGES.get(lookupSourceFactory);
        if (bridge == null || (hasFilter && !FILTERS.containsKey(lookupSourceFactory))) {
            return fallback.lookupOuterJoin(operatorId, planNodeId, lookupSourceFactory, waitForBuild, hasFilter, probeTypes, probeJoinChannel,
                    probeHashChannel, probeOutputChannels, totalOperatorsCount, partitioningSpillerFactory, blockTypeOperators);
        }
        // the outer operator of the same bridge (LookupJoinOperatorFactory.createOuterOperatorFactory) is createLookupJoin(..., outer = true)
        return lookupJoin(operatorId, planNodeId, bridge, LOOKUP_OUTER, false, probeTypes, probeJoinChannel, probeHashChannel, probeOutputChannels,
                hasFilter ? FILTERS.get(lookupSourceFactory) : 0L);
    }

    @Override
    public OperatorFactory fullOuterJoin(int operatorId, PlanNodeId planNodeId, JoinBridgeManager<? extends LookupSourceFactory> lookupSourceFactory,
            boolean hasFilter, List<Type> probeTypes, List<Integer> probeJoinChannel, OptionalInt probeHashChannel, Optional<List<Integer>> probeOutputChannels,
            OptionalInt totalOperatorsCount, PartitioningSpillerFactory partitioningSpillerFactory, BlockTypeOperators blockTypeOperators)
    {
        Long bridge = BRIDGES.get(lookupSourceFactory);
        if (bridge == null || (hasFilter && !FILTERS.containsKey(lookupSourceFactory))) {
            return fallback.fullOuterJoin(operatorId, planNodeId, lookupSourceFactory, hasFilter, probeTypes, probeJoinChannel, probeHashChannel,
                    probeOutputChannels, totalOperatorsCount, partitioningSpillerFactory, blockTypeOperators);
        }
        return lookupJoin(operatorId, planNodeId, bridge, FULL_OUTER, false, probeTypes, probeJoinChannel, probeHashChannel, probeOutputChannels,
                hasFilter ? FILTERS.get(lookupSourceFactory) : 0L);
    }

    private static OperatorFactory lookupJoin(int operatorId, PlanNodeId planNodeId, long bridge, int joinType, boolean outputSingleMatch, List<Type> probeTypes,
            List<Integer> probeJoinChannel, OptionalInt probeHashChannel, Optional<List<Integer>> probeOutputChannels, long filterExpression)
    {
        int[] types = probeTypes.stream().mapToInt(RowExpressionSerializer::typeOf).toArray();
        int[] joinChannels = probeJoinChannel.stream().mapToInt(Integer::intValue).toArray();
        // default probe output = every probe channel (TrinoOperatorFactories.java:63)
        int[] outputChannels = probeOutputChannels.map(c -> c.stream().mapToInt(Integer::intValue).toArray())
                .orElseGet(() -> IntStream.range(0, probeTypes.size()).toArray());
        List<Type> outputTypes = new ArrayList<>();
        for (int c : outputChannels) {
            outputTypes.add(probeTypes.get(c));
        }
        return new GpuOperatorFactory(operatorId, planNodeId, "GpuLookupJoinOperator", probeTypes, outputTypes,
                () -> GpuNative.createLookupJoin(bridge, types, joinChannels, probeHashChannel.orElse(-1), outputChannels, joinType, outputSingleMatch, false, 0, filterExpression));
    }
}
