package io.trino.gpu;

import io.trino.spi.ErrorCodeSupplier;
import io.trino.spi.TrinoException;

import static io.trino.spi.StandardErrorCode.DIVISION_BY_ZERO;
import static io.trino.spi.StandardErrorCode.GENERIC_INSUFFICIENT_RESOURCES;
import static io.trino.spi.StandardErrorCode.GENERIC_INTERNAL_ERROR;
import static io.trino.spi.StandardErrorCode.NOT_SUPPORTED;
import static io.trino.spi.StandardErrorCode.NUMERIC_VALUE_OUT_OF_RANGE;

/** Thrown by the JNI shim for a negative pa_status (include/presto_amd.h): the reference's error codes for the same conditions. */
public final class GpuNativeException
        extends TrinoException
{
    private final int status;

    public GpuNativeException(int status, String message)
    {
        super(errorCode(status), message);
        this.status = status;
    }

    /** PA_ERR_NOT_SUPPORTED from a factory: the planner hook keeps the reference operator for that plan node. */
    public boolean isNotSupported()
    {
        return status == -3;
    }

    private static ErrorCodeSupplier errorCode(int status)
    {
        switch (status) {
            case -3: return NOT_SUPPORTED;
            case -4: return NUMERIC_VALUE_OUT_OF_RANGE;      // BigintOperators.java:47-55
            case -5: return DIVISION_BY_ZERO;                 // BigintOperators.java:88-110
            case -6: return GENERIC_INSUFFICIENT_RESOURCES;   // BigintGroupByHash.java:264-267
            default: return GENERIC_INTERNAL_ERROR;
        }
    }
}
