"""presto_amd -- MI355X-native page processing for Trino's operator hot path.

Only what the path needs: the C-ABI library (csrc/, built into libpresto_amd.so) and a thin host-side
mirror of the reference's Operator / Page / Block / RowExpression interfaces used by tests and bench.
"""
from . import abi  # noqa: F401
from .page import Block, Page, DeviceBuffer, sequence_page  # noqa: F401
from . import expr  # noqa: F401

__version__ = "0.1.0"
