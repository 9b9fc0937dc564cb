"""Mirror of the reference's RowExpression IR and its serialisation into the C-ABI descriptor.

Mirrors core/trino-main/src/main/java/io/trino/sql/relational/{CallExpression, ConstantExpression,
InputReferenceExpression, SpecialForm}.java (forms at SpecialForm.java:137-152) and the helper
constructors of core/trino-main/src/main/java/io/trino/sql/relational/Expressions.java
(`field`, `constant`, `call`).  `serialize` flattens a tree into pa_expr (include/presto_amd.h),
which is what the JNI shim's RowExpression walker would emit.
"""
import ctypes as C

from . import abi

_INT_TYPES = (abi.BIGINT, abi.INTEGER, abi.DATE)


class RowExpression:
    def __init__(self, kind, type_, op=0, args=(), channel=-1, value=None, is_null=False):
        self.kind = kind
        self.type = type_
        self.op = op
        self.args = list(args)
        self.channel = channel
        self.value = value
        self.is_null = is_null

    # sugar so tests read like SQL
    def _bin(self, op, other, type_=None):
        other = _lift(other, self.type)
        if type_ is None and isinstance(self.type, abi.DecimalType) and isinstance(other.type, abi.DecimalType):
            type_ = decimal_result_type(op, self.type, other.type)
        return call(op, type_ if type_ is not None else self.type, self, other)

    def __add__(self, o): return self._bin(abi.OP_ADD, o)
    def __sub__(self, o): return self._bin(abi.OP_SUBTRACT, o)
    def __mul__(self, o): return self._bin(abi.OP_MULTIPLY, o)
    def __truediv__(self, o): return self._bin(abi.OP_DIVIDE, o)
    def __mod__(self, o): return self._bin(abi.OP_MODULUS, o)
    def __neg__(self): return call(abi.OP_NEGATE, self.type, self)
    def __radd__(self, o): return _lift(o, self.type)._bin(abi.OP_ADD, self)
    def __rsub__(self, o): return _lift(o, self.type)._bin(abi.OP_SUBTRACT, self)
    def __rmul__(self, o): return _lift(o, self.type)._bin(abi.OP_MULTIPLY, self)
    def eq(self, o): return self._bin(abi.OP_EQUAL, o, abi.BOOLEAN)
    def ne(self, o): return self._bin(abi.OP_NOT_EQUAL, o, abi.BOOLEAN)
    def __lt__(self, o): return self._bin(abi.OP_LESS_THAN, o, abi.BOOLEAN)
    def __le__(self, o): return self._bin(abi.OP_LESS_THAN_OR_EQUAL, o, abi.BOOLEAN)
    def __gt__(self, o): return self._bin(abi.OP_GREATER_THAN, o, abi.BOOLEAN)
    def __ge__(self, o): return self._bin(abi.OP_GREATER_THAN_OR_EQUAL, o, abi.BOOLEAN)
    def is_null_(self): return special(abi.FORM_IS_NULL, abi.BOOLEAN, self)
    def between(self, lo, hi): return special(abi.FORM_BETWEEN, abi.BOOLEAN, self, _lift(lo, self.type), _lift(hi, self.type))
    def isin(self, *values): return special(abi.FORM_IN, abi.BOOLEAN, self, *[_lift(v, self.type) for v in values])
    def cast(self, type_): return call(abi.OP_CAST, type_, self)


def decimal_result_type(op, a, b):
    """The result type the reference's signatures derive for decimal arithmetic (core/trino-main/src/main/java/io/trino/type/
    DecimalOperators.java:76-84 add / subtract: precision min(38, max(p1 - s1, p2 - s2) + max(s1, s2) + 1), scale max(s1, s2);
    :243-249 multiply: precision min(38, p1 + p2), scale s1 + s2)."""
    if op in (abi.OP_ADD, abi.OP_SUBTRACT):
        scale = max(a.scale, b.scale)
        return abi.decimal(min(38, max(a.precision - a.scale, b.precision - b.scale) + scale + 1), scale)
    if op == abi.OP_MULTIPLY:
        return abi.decimal(min(38, a.precision + b.precision), a.scale + b.scale)
    raise ValueError("decimal operator %d is not on the device path" % op)


def _lift(v, type_):
    if isinstance(v, RowExpression):
        return v
    return constant(v, type_)


def field(channel, type_):
    """Expressions.field(index, type) -> InputReferenceExpression"""
    return RowExpression(abi.EXPR_INPUT_REF, type_, channel=channel)


def constant(value, type_):
    """Expressions.constant(value, type) -> ConstantExpression; value None = typed NULL"""
    if isinstance(value, str):
        value = value.encode("utf-8")
    return RowExpression(abi.EXPR_CONSTANT, type_, value=value, is_null=value is None)


def call(op, type_, *args):
    return RowExpression(abi.EXPR_CALL, type_, op=op, args=args)


def special(form, type_, *args):
    return RowExpression(abi.EXPR_SPECIAL, type_, op=form, args=args)


def and_(*args): return special(abi.FORM_AND, abi.BOOLEAN, *args)
def or_(*args): return special(abi.FORM_OR, abi.BOOLEAN, *args)
def not_(a): return call(abi.OP_NOT, abi.BOOLEAN, a)
def if_(c, a, b): return special(abi.FORM_IF, a.type, c, a, b)
def coalesce(*args): return special(abi.FORM_COALESCE, args[0].type, *args)


def serialize(expr):
    """Flattens to (pa_expr, keepalive)."""
    nodes = []
    args = []
    keep = []

    def walk(e):
        child_ids = [walk(a) for a in e.args]
        n = abi.pa_expr_node()
        n.kind = e.kind
        n.op = e.op
        n.type = e.type
        n.channel = e.channel
        n.is_null = 1 if e.is_null else 0
        n.nargs = len(child_ids)
        n.first_arg = len(args)
        args.extend(child_ids)
        if e.kind == abi.EXPR_CONSTANT and not e.is_null:
            if e.type in (abi.DOUBLE, abi.REAL):  # a REAL constant travels as the double it converts to exactly
                n.f64 = float(e.value)
            elif e.type == abi.VARCHAR:
                buf = C.create_string_buffer(bytes(e.value), max(len(e.value), 1))
                keep.append(buf)
                n.str = C.cast(buf, C.c_char_p)
                n.str_len = len(e.value)
            elif e.type == abi.BOOLEAN:
                n.i64 = 1 if e.value else 0
            elif e.type == abi.LONG_DECIMAL:   # two's complement halves of the unscaled value: low in i64, high in the bits of f64
                v = int(e.value) & ((1 << 128) - 1)
                n.i64 = C.c_int64(v & 0xFFFFFFFFFFFFFFFF).value
                n.f64 = C.cast(C.pointer(C.c_uint64(v >> 64)), C.POINTER(C.c_double))[0]
            else:
                n.i64 = int(e.value)
        if isinstance(e.type, abi.DecimalType):
            n.str_len = e.type.param
        nodes.append(n)
        return len(nodes) - 1

    root = walk(expr)
    node_arr = (abi.pa_expr_node * len(nodes))(*nodes)
    arg_arr = abi.int32_array(args)
    out = abi.pa_expr()
    out.node_count = len(nodes)
    out.root = root
    out.nodes = C.cast(node_arr, C.POINTER(abi.pa_expr_node))
    out.arg_count = len(args)
    out.args = C.cast(arg_arr, C.POINTER(C.c_int32))
    keep += [node_arr, arg_arr]
    return out, keep


def serialize_many(exprs):
    """Array of pa_expr for a projection list: (pa_expr array, keepalive)."""
    keep = []
    arr = (abi.pa_expr * max(len(exprs), 1))()
    for i, e in enumerate(exprs):
        s, k = serialize(e)
        arr[i] = s
        keep.append(k)
    return arr, keep
