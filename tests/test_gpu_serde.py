"""Page wire format on device (pa_page_serialize / pa_page_deserialize) against the oracle's restatement of PagesSerde:
byte-identical frames, and round trips in both directions (device-written bytes read by the oracle, oracle-written bytes
read by the device)."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import download_page, upload_page
from presto_amd.page import Block, Page, deserialize_page, sequence_page, serialize_page

pytestmark = pytest.mark.gpu


def random_page(rng, n, with_nulls=True):
    words = [b"", b"a", b"BUILDING", b"0123456789abcdefghij", None, b"\xc3\xa9t\xc3\xa9"]

    def nulls(p):
        return rng.random(n) < p if with_nulls else None
    strings = [words[i] for i in rng.integers(0, len(words), n)]
    if not with_nulls:
        strings = [s or b"" for s in strings]
    return Page([Block.bigint(rng.integers(-2 ** 62, 2 ** 62, n), nulls(0.3)), Block.double(rng.standard_normal(n), nulls(0.1)),
                 Block.integer(rng.integers(-2 ** 31, 2 ** 31 - 1, n), nulls(0.5)), Block.date(rng.integers(0, 20000, n)),
                 Block.boolean(rng.random(n) < 0.5, nulls(0.2)), Block.varchar(strings), Block.bigint(np.arange(n))], n)


def wire_rows(page):
    """rows as the wire format types them: DOUBLE as its long bits, NULL positions by flag"""
    rows = []
    for r in page.to_rows():
        rows.append(tuple(np.float64(v).view(np.int64).item() if isinstance(v, float) else (int(v) if isinstance(v, (bool, np.bool_)) else v) for v in r))
    return rows


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, 65537])
@pytest.mark.parametrize("with_nulls", [True, False])
def test_serialized_bytes_equal_the_oracle(gpu, oracle, n, with_nulls):
    rng = np.random.default_rng(n + 17)
    page = random_page(rng, n, with_nulls)
    expected = oracle.serialize_page(page)
    assert serialize_page(page) == expected                 # host page (uploaded by the library)
    assert serialize_page(upload_page(page)) == expected    # device-resident page
    # and back: the device reads the oracle's bytes, the oracle reads the device's
    back = download_page(deserialize_page(expected))
    assert wire_rows(back) == wire_rows(page)
    assert wire_rows(oracle.deserialize_page(serialize_page(page))) == wire_rows(page)


def test_known_frame_layout(gpu, oracle):
    """A hand-computed frame: (BIGINT 1, NULL, 3) and (VARCHAR 'ab', '', NULL)."""
    import struct
    page = Page([Block.bigint([1, 0, 3], [0, 1, 0]), Block.varchar([b"ab", b"", None])], 3)
    payload = struct.pack("<i", 2)
    payload += struct.pack("<i", 10) + b"LONG_ARRAY" + struct.pack("<i", 3) + bytes([1, 0b01000000]) + struct.pack("<iqq", 2, 1, 3)
    payload += struct.pack("<i", 14) + b"VARIABLE_WIDTH" + struct.pack("<i", 3) + struct.pack("<iii", 2, 2, 2) + bytes([1, 0b00100000]) + struct.pack("<i", 2) + b"ab"
    frame = struct.pack("<ibii", 3, 0, len(payload), len(payload)) + payload
    assert oracle.serialize_page(page) == frame
    assert serialize_page(page) == frame


def test_sliced_device_page_and_sequence_kat(gpu, oracle):
    # a region of a device page (offsets not starting at 0), as FilterAndProject's zero-copy outputs are
    page = sequence_page(100, [(abi.VARCHAR, 0), (abi.BIGINT, 0), (abi.DOUBLE, 0)])
    region = page.get_region(37, 41)
    assert serialize_page(region) == oracle.serialize_page(region)
    assert wire_rows(download_page(deserialize_page(serialize_page(region)))) == wire_rows(region)
