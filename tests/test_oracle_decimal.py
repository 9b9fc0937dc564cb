"""DECIMAL in the oracle, pinned by the reference's own known answers:
  TestDecimalOperators.java:30-190   add / subtract / multiply results, their derived types and the overflow cases
  TestDecimalSumAggregation.java:40-131, TestDecimalAverageAggregation.java:44-150   the accumulator state after every input
(SQL literals are restated as typed constants: DECIMAL '137.7' is DECIMAL(4, 1) with unscaled value 1377, and the expected
value's leading zeros spell out the precision of the result type, as in the reference's decimal("0154.8").)"""
import pytest

from presto_amd import abi
from presto_amd.expr import constant, field
from presto_amd.page import Block, Page


def lit(text):
    """DECIMAL 'text' -> (type, unscaled)"""
    neg = text.startswith("-")
    digits = text.lstrip("+-")
    whole, _, frac = digits.partition(".")
    precision = max(len(whole) + len(frac), 1)
    value = int((whole + frac) or "0")
    return abi.decimal(precision, len(frac)), -value if neg else value


def evaluate(oracle, expr):
    page = Page([Block.bigint([0])], 1)
    out = oracle.filter_project(page, None, [expr])
    return out.blocks[0].to_pylist()[0], out.blocks[0].type


ADD = [("137.7", "17.1", "0154.8"), ("-1", "-2", "-03"), ("1", "2", "03"), (".1234567890123456", ".1234567890123456", "0.2469135780246912"),
       ("-.1234567890123456", "-.1234567890123456", "-0.2469135780246912"), ("1234567890123456", "1234567890123456", "02469135780246912"),
       ("123456789012345678", "123456789012345678", "0246913578024691356"), (".123456789012345678", ".123456789012345678", "0.246913578024691356"),
       ("1234567890123456789", "1234567890123456789", "02469135780246913578"),
       ("12345678901234567890123456789012345678", "12345678901234567890123456789012345678", "24691357802469135780246913578024691356"),
       ("-99999999999999999999999999999999999999", "99999999999999999999999999999999999999", "00000000000000000000000000000000000000"),
       ("12345678901234567890", "-12345678901234567890", "000000000000000000000"), ("-12345678901234567890", "-12345678901234567890", "-024691357802469135780"),
       ("12345678901234567890", "-12345678901234567891", "-000000000000000000001"), ("999999999999999999", "999999999999999999", "1999999999999999998"),
       ("999999999999999999", ".999999999999999999", "0999999999999999999.999999999999999999"),
       ("123456789012345678901234567890", ".12345678", "123456789012345678901234567890.12345678"),
       (".123456789012345678901234567890", "12345678", "12345678.123456789012345678901234567890"),
       ("17014000000000000000000000000000000000", "-7014000000000000000000000000000000000.1", "9999999999999999999999999999999999999.9")]
ADD_OVERFLOW = [("99999999999999999999999999999999999999", "1"), (".1", "99999999999999999999999999999999999999"),
                ("99999999999999999999999999999999999999", "99999999999999999999999999999999999999"),
                ("-99999999999999999999999999999999999999", "-99999999999999999999999999999999999999"),
                ("17015000000000000000000000000000000000", "-7015000000000000000000000000000000000.1")]
SUBTRACT = [("107.7", "17.1", "0090.6"), ("-1", "-2", "01"), ("1", "2", "-01"), (".1234567890123456", ".1234567890123456", "0.0000000000000000"),
            ("1234567890123456789", "1234567890123456789", "00000000000000000000"), ("-12345678901234567890", "12345678901234567890", "-024691357802469135780"),
            ("12345678901234567890", "12345678901234567891", "-000000000000000000001"), ("999999999999999999", ".999999999999999999", "0999999999999999998.000000000000000001"),
            ("123456789012345678901234567890", ".00000001", "123456789012345678901234567889.99999999"),
            (".000000000000000000000000000001", "87654321", "-87654320.999999999999999999999999999999")]
SUBTRACT_OVERFLOW = [("-99999999999999999999999999999999999999", "1"), (".1", "99999999999999999999999999999999999999"),
                     ("99999999999999999999999999999999999999", ".1")]
# TestDecimalOperators.testMultiply (:138-176)
MULTIPLY = [("12", "3", "036"), ("12", "-3", "-036"), ("-12", "3", "-036"), ("1234567890123456", "3", "03703703670370368"),
            (".1234567890123456", "3", "0.3703703670370368"), (".1234567890123456", ".3", ".03703703670370368"),
            ("12345678901234567", "123456789012345670", "01524157875323883455265967556774890"),
            ("-12345678901234567", "123456789012345670", "-01524157875323883455265967556774890"),
            ("-12345678901234567", "-123456789012345670", "01524157875323883455265967556774890"),
            (".12345678901234567", ".123456789012345670", ".01524157875323883455265967556774890"),
            ("12345678901234567890123456789012345678", "3", "37037036703703703670370370367037037034"),
            ("1234567890123456789.0123456789012345678", "3", "3703703670370370367.0370370367037037034"),
            (".12345678901234567890123456789012345678", "3", ".37037036703703703670370370367037037034"),
            ("3", "12345678901234567890123456789012345678", "37037036703703703670370370367037037034"),
            ("3", "1234567890123456789.0123456789012345678", "3703703670370370367.0370370367037037034"),
            ("3", ".12345678901234567890123456789012345678", ".37037036703703703670370370367037037034"),
            (".1234567890123456789", ".1234567890123456789", ".01524157875323883675019051998750190521")]
MULTIPLY_OVERFLOW = [("12345678901234567890123456789012345678", "9"), (".12345678901234567890123456789012345678", "9"),
                     ("12345678901234567890123456789012345678", "-9"), (".12345678901234567890123456789012345678", "-9")]


@pytest.mark.parametrize("op,cases", [(abi.OP_ADD, ADD), (abi.OP_SUBTRACT, SUBTRACT), (abi.OP_MULTIPLY, MULTIPLY)])
def test_decimal_operator_known_answers(oracle, op, cases):
    for a, b, expected in cases:
        (ta, va), (tb, vb), (te, ve) = lit(a), lit(b), lit(expected)
        e = constant(va, ta)._bin(op, constant(vb, tb))
        assert (e.type.precision, e.type.scale) == (te.precision, te.scale), (a, b, e.type, te)   # the type the signature derives
        got, got_type = evaluate(oracle, e)
        assert got == ve and got_type == int(te), (a, b, got, ve)


@pytest.mark.parametrize("op,cases", [(abi.OP_ADD, ADD_OVERFLOW), (abi.OP_SUBTRACT, SUBTRACT_OVERFLOW), (abi.OP_MULTIPLY, MULTIPLY_OVERFLOW)])
def test_decimal_operator_overflows(oracle, op, cases):
    for a, b in cases:
        (ta, va), (tb, vb) = lit(a), lit(b)
        with pytest.raises(oracle.OracleError) as err:
            evaluate(oracle, constant(va, ta)._bin(op, constant(vb, tb)))
        assert err.value.status == abi.ERR_NUMERIC_VALUE_OUT_OF_RANGE


def test_decimal_sum_state_known_answers(oracle):
    """TestDecimalSumAggregation: testOverflow, testUnderflow, testUnderflowAfterOverflow, the two combine cases (a combine adds the
    other state's sum and its overflow count: the same arithmetic as adding its inputs one by one)."""
    two = lambda k: 1 << k
    assert oracle.decimal_state_after([two(126)]) == (0, False, two(126))
    assert oracle.decimal_state_after([two(126), two(126)]) == (1, False, 0)
    assert oracle.decimal_state_after([-two(126)]) == (0, True, two(126))
    overflow, negative, magnitude = oracle.decimal_state_after([-two(126), -two(126)])
    assert overflow == -1 and magnitude == 0                       # compare(state, 0) == 0: a zero magnitude (with the sign bit set)
    assert oracle.decimal_state_after([two(126), two(126), two(125)]) == (1, False, two(125))
    assert oracle.decimal_state_after([two(126), two(126), two(125), -two(126), -two(126), -two(126)]) == (0, True, two(125))
    assert oracle.decimal_state_after([two(125), two(126), two(125), two(126)]) == (1, False, two(126))        # testCombineOverflow
    assert oracle.decimal_state_after([-two(125), -two(126), -two(125), -two(126)]) == (-1, True, two(126))    # testCombineUnderflow


def test_decimal_average_known_answers(oracle):
    """TestDecimalAverageAggregation: the average sees overflow * 2^127 + the 127-bit sum, divided with ROUND_HALF_UP."""
    two = lambda k: 1 << k
    assert oracle.decimal_state_after([two(126), two(126)], want_average=True)[3] == two(126)
    assert oracle.decimal_state_after([-two(126), -two(126)], want_average=True)[3] == -two(126)
    assert oracle.decimal_state_after([two(126), two(126), two(125), -two(126), -two(126), -two(126)], want_average=True)[3] == -(two(125) // 6)
    # (the reference compares with new BigDecimal(TWO.pow(125).negate().divide(6)): BigInteger.divide truncates, and ROUND_HALF_UP of
    #  -7089215977519551322153637654828504405.33... is that truncated quotient)
    assert oracle.decimal_state_after([two(125), two(126), two(125), two(126)], want_average=True)[3] == (3 * two(126)) // 4
    assert oracle.decimal_state_after([1, 2], want_average=True)[3] == 2          # 1.5 rounds half up
    assert oracle.decimal_state_after([-1, -2], want_average=True)[3] == -2       # ... away from zero
    assert oracle.decimal_state_after([1, 1, 2], want_average=True)[3] == 1       # 1.33


def test_decimal_aggregation_operator(oracle):
    """sum / avg / min / max / count over short decimals through the aggregation operator: sum is a DECIMAL(38, s), avg keeps the
    input type and rounds half up, NULLs are skipped, an empty group is NULL; a sum beyond 38 digits is NUMERIC_VALUE_OUT_OF_RANGE."""
    d = abi.decimal(12, 2)
    keys = [1, 1, 2, 2, 2, 3]
    vals = [1050, 251, -399, None, -2, None]
    page = Page([Block.bigint(keys), Block.decimal([0 if v is None else v for v in vals], [v is None for v in vals])], 6)
    aggs = [(abi.AGG_SUM, 1, d), (abi.AGG_AVG, 1, d), (abi.AGG_MIN, 1, d), (abi.AGG_MAX, 1, d), (abi.AGG_COUNT, 1, d)]
    agg = oracle.HashAggregation([abi.BIGINT, d], [0], aggs)
    agg.add_page(page)
    out = agg.build_result()
    assert [b.type for b in out.blocks] == [abi.BIGINT, abi.LONG_DECIMAL, abi.DECIMAL, abi.DECIMAL, abi.DECIMAL, abi.BIGINT]
    assert sorted(out.to_rows()) == [(1, 1301, 651, 251, 1050, 2), (2, -401, -201, -399, -2, 2), (3, None, None, None, None, 0)]
    # 650.5 -> 651 and -200.5 -> -201: half up, away from zero
    big = abi.decimal(38, 0)
    top = 10 ** 38 - 1
    agg = oracle.HashAggregation([big], [], [(abi.AGG_SUM, 0, big)])
    agg.add_page(Page([Block.long_decimal([top, 1])], 2))
    with pytest.raises(oracle.OracleError) as err:
        agg.build_result()
    assert err.value.status == abi.ERR_NUMERIC_VALUE_OUT_OF_RANGE
    agg = oracle.HashAggregation([big], [], [(abi.AGG_SUM, 0, big), (abi.AGG_AVG, 0, big)])
    agg.add_page(Page([Block.long_decimal([top, -top, 7, None])], 4))
    assert agg.build_result().to_rows() == [(7, 2)]   # 7 / 3 = 2.33


def test_tpch_q6_and_q1_expressions_over_decimal_columns(oracle):
    """The TPC-H expressions with DECIMAL(12, 2) columns: extendedprice * discount is a DECIMAL(24, 4); 1 - discount a DECIMAL(13, 2)
    (the literal 1 coerced to DECIMAL(10, 0) would give (13, 2) too: INTEGER -> DECIMAL(10, 0), DecimalOperators' subtract), price *
    (1 - discount) a DECIMAL(25, 4), times (1 + tax) a DECIMAL(38, 6) -- exact integers end to end."""
    d = abi.decimal(12, 2)
    price, disc, tax = field(0, d), field(1, d), field(2, d)
    one = constant(1, abi.decimal(10, 0))
    revenue = price * disc
    disc_price = price * (one - disc)
    charge = disc_price * (one + tax)
    assert (revenue.type.precision, revenue.type.scale) == (24, 4)
    assert (disc_price.type.precision, disc_price.type.scale) == (25, 4)
    assert (charge.type.precision, charge.type.scale) == (38, 6)
    page = Page([Block.decimal([10494950, 90100, 5]), Block.decimal([10, 0, 7]), Block.decimal([8, 0, 3])], 3)
    out = oracle.filter_project(page, disc >= constant(5, d), [revenue, disc_price, charge])
    assert out.to_rows() == [(104949500, 944545500, 102010914000), (35, 465, 47895)]
