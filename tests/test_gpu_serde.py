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


def test_lz4_frames_both_ways(gpu, oracle):
    """PageCodecMarker.COMPRESSED (PagesSerde.java:74-95, 139-156): the library's LZ4 frame inflates -- by the oracle's
    independent decoder -- to the uncompressed payload, byte for byte; frames compressed by the oracle's (different) encoder are
    read by the library; a page that does not shrink to 0.8 stays uncompressed."""
    import struct
    page = sequence_page(5000, [(abi.VARCHAR, 0), (abi.BIGINT, 7), (abi.DOUBLE, 0)])   # regular data: compresses well
    plain = serialize_page(page)
    packed = serialize_page(page, compress=True)
    positions, markers, uncompressed, size = struct.unpack_from("<ibii", packed, 0)
    assert markers == 1 and positions == 5000 and uncompressed == len(plain) - 13 and size == len(packed) - 13 and size <= 0.8 * uncompressed
    assert oracle.lz4_decompress(packed[13:], uncompressed) == plain[13:]
    types = [abi.VARCHAR, abi.BIGINT, abi.DOUBLE]
    assert download_page(deserialize_page(packed, types=types)).to_rows() == page.to_rows()
    other = oracle.compress_frame(plain)
    assert other[4] == 1 and other != packed
    assert download_page(deserialize_page(other, types=types)).to_rows() == page.to_rows()
    rng = np.random.default_rng(1)
    noise = Page([Block.bigint(rng.integers(-2 ** 62, 2 ** 62, 3000))], 3000)          # incompressible
    assert serialize_page(noise, compress=True) == serialize_page(noise)
    # edge sizes through both codecs
    for n in (0, 1, 12, 13, 14, 300):
        p = sequence_page(n, [(abi.BIGINT, 0)]) if n else Page([Block.bigint([])], 0)
        f = serialize_page(p, compress=True)
        assert download_page(deserialize_page(f, types=[abi.BIGINT])).to_rows() == p.to_rows()


def test_deserialize_types_blocks_as_declared(gpu):
    page = Page([Block.double([1.5, -0.0, 3.25]), Block.date([1, 2, 3]), Block.bigint([4, 5, 6]), Block.integer([7, 8, 9], [0, 1, 0])], 3)
    frame = serialize_page(page)
    back = download_page(deserialize_page(frame, types=[abi.DOUBLE, abi.DATE, abi.BIGINT, abi.INTEGER]))
    assert [b.type for b in back.blocks] == [abi.DOUBLE, abi.DATE, abi.BIGINT, abi.INTEGER]
    assert back.to_rows() == page.to_rows()
    from presto_amd._lib import PrestoAmdError
    with pytest.raises(PrestoAmdError):
        deserialize_page(frame, types=[abi.BIGINT, abi.DATE, abi.VARCHAR, abi.INTEGER])   # LONG_ARRAY cannot be a VARCHAR
    with pytest.raises(PrestoAmdError):
        deserialize_page(frame, types=[abi.DOUBLE, abi.DATE])                               # channel count


def test_malformed_frames_are_refused(gpu):
    """Frames come from other workers: end offsets that are not ascending / run past the block's bytes, a payload length that
    reaches behind the frame, and an LZ4 block that does not inflate to its declared size are refused, not trusted."""
    import struct
    from presto_amd._lib import PrestoAmdError
    page = Page([Block.varchar([b"ab", b"cde", b"f"])], 3)
    frame = bytearray(serialize_page(page))
    ends_at = frame.index(b"VARIABLE_WIDTH") + len(b"VARIABLE_WIDTH") + 4
    assert struct.unpack_from("<iii", frame, ends_at) == (2, 5, 6)
    for bad in ((2, 1, 6), (2, 5, 7), (-1, 5, 6), (2, 5, 5)):
        f = bytearray(frame)
        struct.pack_into("<iii", f, ends_at, *bad)
        with pytest.raises(PrestoAmdError):
            deserialize_page(bytes(f))
    f = bytearray(frame)
    struct.pack_into("<i", f, 9, len(frame))            # sizeInBytes larger than what follows
    struct.pack_into("<i", f, 5, len(frame))
    with pytest.raises(PrestoAmdError):
        deserialize_page(bytes(f))
    big = sequence_page(2000, [(abi.BIGINT, 0)])
    packed = bytearray(serialize_page(big, compress=True))
    assert packed[4] == 1
    struct.pack_into("<i", packed, 5, struct.unpack_from("<i", packed, 5)[0] + 8)   # claims 8 more uncompressed bytes
    with pytest.raises(PrestoAmdError):
        deserialize_page(bytes(packed))
    with pytest.raises(PrestoAmdError):
        deserialize_page(bytes(serialize_page(big, compress=True)[:-3]))             # truncated
