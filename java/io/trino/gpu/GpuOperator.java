package io.trino.gpu;

import com.google.common.util.concurrent.ListenableFuture;
import com.google.common.util.concurrent.SettableFuture;
import io.trino.operator.Operator;
import io.trino.operator.OperatorContext;
import io.trino.spi.Page;
import io.trino.spi.block.Block;
import io.trino.spi.block.ByteArrayBlock;
import io.trino.spi.block.IntArrayBlock;
import io.trino.spi.block.LongArrayBlock;
import io.trino.spi.block.VariableWidthBlock;
import io.trino.spi.type.Type;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.List;
import java.util.Optional;
import java.util.concurrent.ScheduledExecutorService;
import java.util.concurrent.TimeUnit;

import static io.airlift.slice.Slices.wrappedBuffer;

/**
 * io.trino.operator.Operator (core/trino-main/src/main/java/io/trino/operator/Operator.java:21-103) over one pa_operator
 * handle: one native call per method.  The Python twin, which the parity tests drive, is presto_amd/operators.py; the C++
 * twin include/presto_amd.hpp.
 */
public final class GpuOperator
        implements Operator
{
    private final OperatorContext operatorContext;
    private final long handle;                     // pa_operator*
    private final List<Type> outputTypes;
    private final PinnedPagePool staging;          // pinned direct ByteBuffers, reused across pages
    private final ScheduledExecutorService poller; // completes isBlocked futures (the task's yield executor)
    private boolean closed;

    GpuOperator(OperatorContext operatorContext, long handle, List<Type> outputTypes, PinnedPagePool staging, ScheduledExecutorService poller)
    {
        this.operatorContext = operatorContext;
        this.handle = handle;
        this.outputTypes = outputTypes;
        this.staging = staging;
        this.poller = poller;
    }

    @Override
    public OperatorContext getOperatorContext()
    {
        return operatorContext;
    }

    /**
     * pa_op_is_blocked: device work in flight with the operator's launch queue full, a probe waiting for its build, an exchange
     * source waiting for its sinks.  The Driver yields on the future (Driver.java:355-457); it is completed by polling --
     * the native side never parks a Driver thread.
     */
    @Override
    public ListenableFuture<Void> isBlocked()
    {
        if (!GpuNative.isBlocked(handle)) {
            return NOT_BLOCKED;
        }
        SettableFuture<Void> future = SettableFuture.create();
        poll(future);
        return future;
    }

    private void poll(SettableFuture<Void> future)
    {
        poller.schedule(() -> {
            if (closed || !GpuNative.isBlocked(handle)) {
                future.set(null);
            }
            else {
                poll(future);
            }
        }, 50, TimeUnit.MICROSECONDS);
    }

    @Override
    public boolean needsInput()
    {
        return GpuNative.needsInput(handle);
    }

    @Override
    public void addInput(Page page)
    {
        Page loaded = page.getLoadedPage();                  // LazyBlock -> loaded (PageProcessor.java:341-343)
        PinnedPagePool.StagedPage s = staging.stage(loaded); // long[] / int[] / byte[] / Slice bytes + offsets + valueIsNull
        GpuNative.addInput(handle, s.positions, s.channels, s.types, s.encodings, s.valueOffsets, s.offsetOffsets, s.nullOffsets, s.idOffsets,
                s.dictionaryChannel, s.dictionarySize, s.buffer, s.stable ? 1 : 0, 0L);
        operatorContext.recordProcessedInput(page.getSizeInBytes(), page.getPositionCount());
    }

    @Override
    public Page getOutput()
    {
        long[] out = GpuNative.getOutput(handle);
        if (out == null) {
            return null;
        }
        int positions = (int) out[0];
        int channels = (int) out[1];
        Block[] blocks = new Block[channels];
        for (int c = 0; c < channels; c++) {
            int at = 2 + 6 * c;
            int type = (int) out[at];
            ByteBuffer values = GpuNative.wrapAddress(out[at + 1], out[at + 2]).order(ByteOrder.LITTLE_ENDIAN);
            boolean[] valueIsNull = out[at + 4] == 0 ? null : nulls(GpuNative.wrapAddress(out[at + 4], positions), positions);
            switch (type) {
                case RowExpressionSerializer.PA_BIGINT:
                case RowExpressionSerializer.PA_DOUBLE: {     // DoubleType stores doubleToLongBits in a LongArrayBlock (DoubleType.java:98-108)
                    long[] v = new long[positions];
                    values.asLongBuffer().get(v);
                    blocks[c] = new LongArrayBlock(positions, Optional.ofNullable(valueIsNull), v);
                    break;
                }
                case RowExpressionSerializer.PA_INTEGER:
                case RowExpressionSerializer.PA_DATE:
                case RowExpressionSerializer.PA_REAL: {        // RealType stores floatToRawIntBits in an IntArrayBlock
                    int[] v = new int[positions];
                    values.asIntBuffer().get(v);
                    blocks[c] = new IntArrayBlock(positions, Optional.ofNullable(valueIsNull), v);
                    break;
                }
                case RowExpressionSerializer.PA_BOOLEAN: {
                    byte[] v = new byte[positions];
                    values.get(v);
                    blocks[c] = new ByteArrayBlock(positions, Optional.ofNullable(valueIsNull), v);
                    break;
                }
                default: {                                    // VARCHAR: bytes + int offsets[positions + 1]
                    int[] offsets = new int[positions + 1];
                    GpuNative.wrapAddress(out[at + 3], 4L * (positions + 1)).order(ByteOrder.LITTLE_ENDIAN).asIntBuffer().get(offsets);
                    byte[] bytes = new byte[(int) out[at + 2]];
                    values.get(bytes);
                    blocks[c] = new VariableWidthBlock(positions, wrappedBuffer(bytes), offsets, Optional.ofNullable(valueIsNull));
                }
            }
        }
        Page page = new Page(positions, blocks);
        operatorContext.recordOutput(page.getSizeInBytes(), page.getPositionCount());
        return page;
    }

    private static boolean[] nulls(ByteBuffer flags, int positions)
    {
        boolean[] out = new boolean[positions];
        for (int i = 0; i < positions; i++) {
            out[i] = flags.get(i) != 0;
        }
        return out;
    }

    @Override
    public void finish()
    {
        GpuNative.finish(handle);
    }

    @Override
    public boolean isFinished()
    {
        return GpuNative.isFinished(handle);
    }

    @Override
    public void close()
    {
        if (!closed) {
            closed = true;
            GpuNative.close(handle);
            staging.releaseAll(); // PA_PAGE_STABLE pages stay untouched until here
        }
    }

    long handle()
    {
        return handle;
    }
}
