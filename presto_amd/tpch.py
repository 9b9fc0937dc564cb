"""TPC-H-shaped workloads of the hot path: query operator descriptors and device-resident synthetic pages.

Query constants follow SURVEY.md section 8d (DATE = int32 days since 1970-01-01):
  Q6  testing/trino-benchmark/src/main/java/io/trino/benchmark/SqlTpchQuery6.java:26-32,
      hand-written twin HandTpchQuery6.java:95-141
  Q1  testing/trino-benchmark/src/main/java/io/trino/benchmark/SqlTpchQuery1.java:29-40,
      hand-written twin HandTpchQuery1.java:241-330
Column shapes follow plugin/trino-tpch/src/main/java/io/trino/plugin/tpch/TpchMetadata.java:587-603
(money/quantity DOUBLE, dates DATE, identifiers BIGINT, flags VARCHAR(1)).
"""
import ctypes as C

from . import abi
from ._lib import DeviceAllocation, check, lib
from .expr import and_, constant, field
from .page import Block, DeviceBuffer, Page

LINEITEM_ROWS_PER_SF = 6001215
ORDERS_ROWS_PER_SF = 1500000
CUSTOMER_ROWS_PER_SF = 150000
DEFAULT_SEED = 0x5EED0000

# ---- Q6 ------------------------------------------------------------------------------------------------
# input page channels: 0 shipdate DATE, 1 discount DOUBLE, 2 quantity DOUBLE, 3 extendedprice DOUBLE
Q6_COLUMNS = [abi.L_SHIPDATE, abi.L_DISCOUNT, abi.L_QUANTITY, abi.L_EXTENDEDPRICE]
Q6_TYPES = [abi.DATE, abi.DOUBLE, abi.DOUBLE, abi.DOUBLE]
Q6_BYTES_PER_ROW = 28  # shipdate 4 + discount 8 + quantity 8 + extendedprice 8


def q6_filter():
    shipdate, discount, quantity = field(0, abi.DATE), field(1, abi.DOUBLE), field(2, abi.DOUBLE)
    return and_(shipdate >= constant(8766, abi.DATE),      # 1994-01-01
                shipdate < constant(9131, abi.DATE),       # 1995-01-01
                discount >= constant(0.05, abi.DOUBLE),
                discount <= constant(0.07, abi.DOUBLE),
                quantity < constant(24.0, abi.DOUBLE))


def q6_projections():
    return [field(3, abi.DOUBLE) * field(1, abi.DOUBLE)]   # extendedprice * discount


Q6_AGGREGATES = [(abi.AGG_SUM, 0, abi.DOUBLE)]             # sum(revenue)

# ---- Q1 ------------------------------------------------------------------------------------------------
# input page channels: 0 returnflag, 1 linestatus, 2 quantity, 3 extendedprice, 4 discount, 5 tax, 6 shipdate
Q1_COLUMNS = [abi.L_RETURNFLAG, abi.L_LINESTATUS, abi.L_QUANTITY, abi.L_EXTENDEDPRICE, abi.L_DISCOUNT, abi.L_TAX,
              abi.L_SHIPDATE]
Q1_TYPES = [abi.VARCHAR, abi.VARCHAR, abi.DOUBLE, abi.DOUBLE, abi.DOUBLE, abi.DOUBLE, abi.DATE]
Q1_TYPE_PARAMS = [1, 1, 0, 0, 0, 0, 0]                     # returnflag / linestatus are VARCHAR(1)
Q1_BYTES_PER_ROW = 46  # 2 x (1 + 4 offset) + 4 x 8 + 4


def q1_filter():
    return field(6, abi.DATE) <= constant(10471, abi.DATE)  # shipdate <= 1998-09-02


def q1_projections():
    rf, ls = field(0, abi.VARCHAR), field(1, abi.VARCHAR)
    qty, price, disc, tax = (field(c, abi.DOUBLE) for c in (2, 3, 4, 5))
    one = constant(1.0, abi.DOUBLE)
    disc_price = price * (one - disc)
    charge = price * (one - disc) * (one + tax)
    return [rf, ls, qty, price, disc_price, charge, disc]


Q1_GROUP_BY = [0, 1]
# sum_qty, sum_base_price, sum_disc_price, sum_charge, avg_qty, avg_price, avg_disc, count_order
Q1_AGGREGATES = [
    (abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.DOUBLE), (abi.AGG_SUM, 4, abi.DOUBLE), (abi.AGG_SUM, 5, abi.DOUBLE),
    (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.DOUBLE), (abi.AGG_AVG, 6, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None),
]


# ---- device-resident synthetic tables ----------------------------------------------------------------------
class DeviceColumns:
    """Whole columns of a synthetic table resident in HBM; pages are zero-copy row ranges of them."""

    def __init__(self, columns, scale_factor, rows, seed=DEFAULT_SEED, stream=None, allocator=None, first_row=0):
        self.columns = list(columns)
        self.rows = int(rows)
        self.first_row = int(first_row)
        self.scale_factor = scale_factor
        self._bufs = {}
        alloc = allocator or (lambda nbytes: DeviceAllocation(nbytes))
        for col in self.columns:
            t = abi.TPCH_COLUMN_TYPE[col]
            if t == abi.VARCHAR:
                width = 1 if col in (abi.L_RETURNFLAG, abi.L_LINESTATUS) else 10
                values = alloc(max(self.rows * width, 16))
                offsets = alloc(4 * (self.rows + 1) + 16)
                optr = _ptr_of(offsets)
            else:
                values = alloc(max(self.rows * abi.TYPE_WIDTH[t], 16))
                offsets = None
                optr = None
            check(lib().pa_tpch_generate(col, scale_factor, self.first_row, self.rows, seed, _ptr_of(values), optr, stream))
            self._bufs[col] = (values, offsets)
        check(lib().pa_stream_synchronize(stream))

    def page(self, first_row=0, row_count=None):
        """PA_MEM_DEVICE Page over rows [first_row, first_row + row_count): Page.getRegion, no copy."""
        if row_count is None:
            row_count = self.rows - first_row
        blocks = []
        for col in self.columns:
            t = abi.TPCH_COLUMN_TYPE[col]
            values, offsets = self._bufs[col]
            if t == abi.VARCHAR:
                blocks.append(Block(t, abi.VARWIDTH, row_count, values=DeviceBuffer(_ptr_of(values), 0, values),
                                    offsets=DeviceBuffer(_ptr_of(offsets) + 4 * first_row, 4 * (row_count + 1), offsets)))
            else:
                w = abi.TYPE_WIDTH[t]
                blocks.append(Block(t, abi.FLAT, row_count,
                                    values=DeviceBuffer(_ptr_of(values) + w * first_row, w * row_count, values)))
        return Page(blocks, row_count, abi.MEM_DEVICE, stable=True)  # the table outlives the operators that read it

    def pages(self, page_rows):
        page_rows = max(4, page_rows - page_rows % 4)  # keeps every page 16-byte aligned
        first = 0
        while first < self.rows:
            n = min(page_rows, self.rows - first)
            yield self.page(first, n)
            first += n


def _ptr_of(buf):
    if buf is None:
        return None
    if hasattr(buf, "data_ptr"):
        return buf.data_ptr()
    return buf.ptr


def lineitem_rows(scale_factor):
    return int(LINEITEM_ROWS_PER_SF * scale_factor)


# ---- Q3 ------------------------------------------------------------------------------------------------
# testing/trino-benchto-benchmarks/src/main/resources/sql/presto/tpch/q03.sql; plan shape lineitem JOIN (orders JOIN customer)
CUSTOMER_COLUMNS = [abi.C_CUSTKEY, abi.C_MKTSEGMENT]
CUSTOMER_TYPES = [abi.BIGINT, abi.VARCHAR]
ORDERS_COLUMNS = [abi.O_ORDERKEY, abi.O_CUSTKEY, abi.O_ORDERDATE, abi.O_SHIPPRIORITY]
ORDERS_TYPES = [abi.BIGINT, abi.BIGINT, abi.DATE, abi.INTEGER]
Q3_LINEITEM_COLUMNS = [abi.L_ORDERKEY, abi.L_EXTENDEDPRICE, abi.L_DISCOUNT, abi.L_SHIPDATE]
Q3_LINEITEM_TYPES = [abi.BIGINT, abi.DOUBLE, abi.DOUBLE, abi.DATE]
Q3_DATE = 9204  # 1995-03-15


def q3_customer_filter():
    return field(1, abi.VARCHAR).eq(constant("BUILDING", abi.VARCHAR))


def q3_orders_filter():
    return field(2, abi.DATE) < constant(Q3_DATE, abi.DATE)


def q3_lineitem_filter():
    return field(3, abi.DATE) > constant(Q3_DATE, abi.DATE)


def q3_lineitem_projections():
    return [field(0, abi.BIGINT), field(1, abi.DOUBLE) * (constant(1.0, abi.DOUBLE) - field(2, abi.DOUBLE))]


def customer_rows(scale_factor):
    n = int(CUSTOMER_ROWS_PER_SF * scale_factor)
    return n - n % 5


def orders_rows(scale_factor):
    return int(ORDERS_ROWS_PER_SF * scale_factor)


# ---- Q6 / Q1 over DECIMAL(12, 2) money and quantity columns (the types TPC-H gives them; DOUBLE above is what the reference's
#      tpch connector produces by default, TpchMetadata.java:587-603) -------------------------------------------------------------
DEC = abi.decimal(12, 2)
Q6_DECIMAL_TYPES = [abi.DATE, DEC, DEC, DEC]
Q1_DECIMAL_TYPES = [abi.VARCHAR, abi.VARCHAR, DEC, DEC, DEC, DEC, abi.DATE]


def q6_decimal_filter():
    shipdate, discount, quantity = field(0, abi.DATE), field(1, DEC), field(2, DEC)
    return and_(shipdate >= constant(8766, abi.DATE), shipdate < constant(9131, abi.DATE),
                discount >= constant(5, DEC), discount <= constant(7, DEC),        # 0.05, 0.07 as DECIMAL(12, 2)
                quantity < constant(2400, DEC))                                    # 24


def q6_decimal_projections():
    return [field(3, DEC) * field(1, DEC)]      # extendedprice * discount: DECIMAL(24, 4)


def q6_decimal_aggregates():
    return [(abi.AGG_SUM, 0, q6_decimal_projections()[0].type)]


def q1_decimal_projections():
    rf, ls = field(0, abi.VARCHAR), field(1, abi.VARCHAR)
    qty, price, disc, tax = (field(c, DEC) for c in (2, 3, 4, 5))
    one = constant(1, abi.decimal(10, 0))   # the INTEGER literal 1 coerced to DECIMAL(10, 0)
    disc_price = price * (one - disc)       # DECIMAL(25, 4)
    charge = disc_price * (one + tax)       # DECIMAL(38, 6)
    return [rf, ls, qty, price, disc_price, charge, disc]


def q1_decimal_aggregates():
    p = q1_decimal_projections()
    return [(abi.AGG_SUM, 2, p[2].type), (abi.AGG_SUM, 3, p[3].type), (abi.AGG_SUM, 4, p[4].type), (abi.AGG_SUM, 5, p[5].type),
            (abi.AGG_AVG, 2, p[2].type), (abi.AGG_AVG, 3, p[3].type), (abi.AGG_AVG, 6, p[6].type), (abi.AGG_COUNT_STAR, -1, None)]
