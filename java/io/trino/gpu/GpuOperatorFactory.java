package io.trino.gpu;

import io.trino.operator.DriverContext;
import io.trino.operator.Operator;
import io.trino.operator.OperatorContext;
import io.trino.operator.OperatorFactory;
import io.trino.spi.type.Type;
import io.trino.sql.planner.plan.PlanNodeId;

import java.util.List;
import java.util.function.LongSupplier;

/**
 * OperatorFactory (core/trino-main/src/main/java/io/trino/operator/OperatorFactory.java:18-50): built once per plan node by
 * the planner with the serialised descriptor captured in `create`; createOperator() per Driver only calls the native factory.
 */
public final class GpuOperatorFactory
        implements OperatorFactory
{
    private final int operatorId;
    private final PlanNodeId planNodeId;
    private final String operatorType;
    private final List<Type> inputTypes;
    private final List<Type> outputTypes;
    private final LongSupplier create;   // -> pa_operator* (throws GpuNativeException)
    private boolean closed;

    GpuOperatorFactory(int operatorId, PlanNodeId planNodeId, String operatorType, List<Type> inputTypes, List<Type> outputTypes, LongSupplier create)
    {
        this.operatorId = operatorId;
        this.planNodeId = planNodeId;
        this.operatorType = operatorType;
        this.inputTypes = inputTypes;
        this.outputTypes = outputTypes;
        this.create = create;
    }

    @Override
    public Operator createOperator(DriverContext driverContext)
    {
        if (closed) {
            throw new IllegalStateException("Factory is already closed");
        }
        OperatorContext context = driverContext.addOperatorContext(operatorId, planNodeId, operatorType);
        return new GpuOperator(context, create.getAsLong(), outputTypes, new PinnedPagePool(inputTypes), driverContext.getYieldExecutor());
    }

    @Override
    public void noMoreOperators()
    {
        closed = true;
    }

    @Override
    public OperatorFactory duplicate()
    {
        return new GpuOperatorFactory(operatorId, planNodeId, operatorType, inputTypes, outputTypes, create);
    }
}
