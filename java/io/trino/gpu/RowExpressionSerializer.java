package io.trino.gpu;

import io.airlift.slice.Slice;
import io.trino.spi.type.Type;
import io.trino.sql.relational.CallExpression;
import io.trino.sql.relational.ConstantExpression;
import io.trino.sql.relational.InputReferenceExpression;
import io.trino.sql.relational.LambdaDefinitionExpression;
import io.trino.sql.relational.RowExpression;
import io.trino.sql.relational.RowExpressionVisitor;
import io.trino.sql.relational.SpecialForm;
import io.trino.sql.relational.VariableReferenceExpression;

import java.util.ArrayList;
import java.util.List;

import static io.trino.spi.type.BigintType.BIGINT;
import static io.trino.spi.type.BooleanType.BOOLEAN;
import static io.trino.spi.type.DateType.DATE;
import static io.trino.spi.type.DoubleType.DOUBLE;
import static io.trino.spi.type.IntegerType.INTEGER;

/**
 * Walks a RowExpression (core/trino-main/src/main/java/io/trino/sql/relational/) children-first into the flat arrays of
 * pa_expr (include/presto_amd.h): nodes in post-order, every node naming its children through (first_arg, nargs) into one
 * args array.  The C++ twin is SerializedExpression in include/presto_amd.hpp.  Anything outside the device subset throws
 * UnsupportedOnDevice, and the planner hook keeps the reference operator.
 */
final class RowExpressionSerializer
        implements RowExpressionVisitor<Integer, Void>
{
    static final class UnsupportedOnDevice
            extends RuntimeException
    {
        UnsupportedOnDevice(String what)
        {
            super(what);
        }
    }

    // pa_type / pa_expr_kind / pa_call_op / pa_special_form ordinals
    static final int PA_BIGINT = 0, PA_INTEGER = 1, PA_DATE = 2, PA_DOUBLE = 3, PA_BOOLEAN = 4, PA_VARCHAR = 5, PA_REAL = 7, PA_DECIMAL = 8, PA_LONG_DECIMAL = 9;
    private static final int INPUT_REF = 0, CONSTANT = 1, CALL = 2, SPECIAL = 3;

    private final List<int[]> nodes = new ArrayList<>();      // kind, op, type, typeParam, channel, isNull, nargs, firstArg
    private final List<Long> longs = new ArrayList<>();
    private final List<Double> doubles = new ArrayList<>();
    private final List<byte[]> strings = new ArrayList<>();
    private final List<Integer> args = new ArrayList<>();

    /** Returns the native handle of the flattened tree (GpuNative.freeExpression releases it). */
    static long serialize(RowExpression expression)
    {
        RowExpressionSerializer s = new RowExpressionSerializer();
        int root = expression.accept(s, null);
        int n = s.nodes.size();
        int[][] columns = new int[8][n];
        long[] longs = new long[n];
        double[] doubles = new double[n];
        for (int i = 0; i < n; i++) {
            for (int f = 0; f < 8; f++) {
                columns[f][i] = s.nodes.get(i)[f];
            }
            longs[i] = s.longs.get(i);
            doubles[i] = s.doubles.get(i);
        }
        return GpuNative.newExpression(root, columns[0], columns[1], columns[2], columns[3], columns[4], columns[5], columns[6], columns[7], longs, doubles,
                s.strings.toArray(new byte[0][]), s.args.stream().mapToInt(Integer::intValue).toArray());
    }

    static int typeOf(Type type)
    {
        if (type.equals(BIGINT)) {
            return PA_BIGINT;
        }
        if (type.equals(INTEGER)) {
            return PA_INTEGER;
        }
        if (type.equals(DATE)) {
            return PA_DATE;
        }
        if (type.equals(DOUBLE)) {
            return PA_DOUBLE;
        }
        if (type.equals(BOOLEAN)) {
            return PA_BOOLEAN;
        }
        if (type.equals(io.trino.spi.type.RealType.REAL)) {
            return PA_REAL;
        }
        if (type instanceof io.trino.spi.type.VarcharType) {
            return PA_VARCHAR;
        }
        if (type instanceof io.trino.spi.type.DecimalType) {
            return ((io.trino.spi.type.DecimalType) type).isShort() ? PA_DECIMAL : PA_LONG_DECIMAL;
        }
        throw new UnsupportedOnDevice("type " + type);
    }

    /** PA_DECIMAL_PARAM(precision, scale) for DECIMAL types, else 0. */
    static int typeParamOf(Type type)
    {
        if (type instanceof io.trino.spi.type.DecimalType) {
            io.trino.spi.type.DecimalType decimal = (io.trino.spi.type.DecimalType) type;
            return (decimal.getPrecision() << 8) | decimal.getScale();
        }
        return 0;
    }

    private int add(int kind, int op, Type type, int channel, boolean isNull, List<Integer> children, long longValue, double doubleValue, byte[] string)
    {
        nodes.add(new int[] {kind, op, typeOf(type), typeParamOf(type), channel, isNull ? 1 : 0, children.size(), args.size()});
        args.addAll(children);
        longs.add(longValue);
        doubles.add(doubleValue);
        strings.add(string);
        return nodes.size() - 1;
    }

    private List<Integer> children(List<RowExpression> arguments)
    {
        List<Integer> ids = new ArrayList<>();
        for (RowExpression argument : arguments) {
            ids.add(argument.accept(this, null));
        }
        return ids;
    }

    @Override
    public Integer visitCall(CallExpression call, Void context)
    {
        // ResolvedFunction names of the operators the device code generator knows (pa_call_op)
        String name = call.getResolvedFunction().getSignature().getName();
        int op;
        switch (name) {
            case "$operator$add": op = 0; break;
            case "$operator$subtract": op = 1; break;
            case "$operator$multiply": op = 2; break;
            case "$operator$divide": op = 3; break;
            case "$operator$modulus": op = 4; break;
            case "$operator$negation": op = 5; break;
            case "$operator$equal": op = 6; break;
            case "$operator$less_than": op = 8; break;
            case "$operator$less_than_or_equal": op = 9; break;
            case "$not": op = 12; break;
            case "$operator$cast": op = 13; break;
            default: throw new UnsupportedOnDevice("function " + name);
        }
        // (>, >=, <> reach the RowExpression level as swapped / negated forms of the three comparison operators above)
        return add(CALL, op, call.getType(), 0, false, children(call.getArguments()), 0, 0, null);
    }

    @Override
    public Integer visitSpecialForm(SpecialForm form, Void context)
    {
        int op;
        switch (form.getForm()) {
            case AND: op = 0; break;
            case OR: op = 1; break;
            case BETWEEN: op = 2; break;
            case IS_NULL: op = 3; break;
            case IF: op = 4; break;
            case COALESCE: op = 5; break;
            case IN: op = 6; break;
            default: throw new UnsupportedOnDevice("special form " + form.getForm());
        }
        return add(SPECIAL, op, form.getType(), 0, false, children(form.getArguments()), 0, 0, null);
    }

    @Override
    public Integer visitInputReference(InputReferenceExpression reference, Void context)
    {
        return add(INPUT_REF, 0, reference.getType(), reference.getField(), false, List.of(), 0, 0, null);
    }

    @Override
    public Integer visitConstant(ConstantExpression literal, Void context)
    {
        Object value = literal.getValue();
        Type type = literal.getType();
        if (value == null) {
            return add(CONSTANT, 0, type, 0, true, List.of(), 0, 0, null);
        }
        switch (typeOf(type)) {
            case PA_DOUBLE: return add(CONSTANT, 0, type, 0, false, List.of(), 0, (Double) value, null);
            // a REAL literal is a Long holding floatToRawIntBits (RealType's Java type); the native side takes the value widened
            case PA_REAL: return add(CONSTANT, 0, type, 0, false, List.of(), 0, Float.intBitsToFloat(((Number) value).intValue()), null);
            case PA_BOOLEAN: return add(CONSTANT, 0, type, 0, false, List.of(), (Boolean) value ? 1 : 0, 0, null);
            case PA_VARCHAR: return add(CONSTANT, 0, type, 0, false, List.of(), 0, 0, ((Slice) value).getBytes());
            case PA_LONG_DECIMAL: {
                // a long decimal literal is a Slice in UnscaledDecimal128Arithmetic's layout: it travels as its two's complement halves
                java.math.BigInteger unscaled = io.trino.spi.type.Decimals.decodeUnscaledValue((Slice) value);
                return add(CONSTANT, 0, type, 0, false, List.of(), unscaled.longValue(), Double.longBitsToDouble(unscaled.shiftRight(64).longValue()), null);
            }
            default: return add(CONSTANT, 0, type, 0, false, List.of(), ((Number) value).longValue(), 0, null);
        }
    }

    @Override
    public Integer visitLambda(LambdaDefinitionExpression lambda, Void context)
    {
        throw new UnsupportedOnDevice("lambda");
    }

    @Override
    public Integer visitVariableReference(VariableReferenceExpression reference, Void context)
    {
        throw new UnsupportedOnDevice("variable reference");
    }
}
