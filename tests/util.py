"""Comparison helpers shared by the parity tests."""
import math


def rows_equal_ignore_order(actual, expected, rel=0.0):
    """assertPagesEqualIgnoreOrder (…/operator/OperatorAssertion.java) with an explicit DOUBLE tolerance."""
    assert len(actual) == len(expected), (len(actual), len(expected))
    key = lambda r: tuple((0, "") if v is None else (1, repr(v)) if not isinstance(v, (float, tuple)) else (1, "") for v in r)
    # sort on the non-float columns (keys / counts), then compare row by row
    a = sorted(actual, key=key)
    e = sorted(expected, key=key)
    def same(va, ve, ra, re_):
        if isinstance(ve, tuple) and isinstance(va, tuple):  # a RowBlock position: field by field
            assert len(va) == len(ve), (va, ve, ra, re_)
            for x, y in zip(va, ve):
                same(x, y, ra, re_)
        elif isinstance(ve, float) and isinstance(va, float):
            if math.isnan(ve):
                assert math.isnan(va)
            else:
                assert va == ve or abs(va - ve) <= rel * max(abs(va), abs(ve)), (va, ve, ra, re_)
        else:
            assert va == ve, (va, ve, ra, re_)

    for ra, re_ in zip(a, e):
        assert len(ra) == len(re_)
        for va, ve in zip(ra, re_):
            same(va, ve, ra, re_)
