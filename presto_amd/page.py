"""Host-side mirror of the reference's Page / Block data model.

Mirrors io.trino.spi.Page (core/trino-spi/src/main/java/io/trino/spi/Page.java:33-398) and the
flat / variable-width / dictionary / RLE blocks (core/trino-spi/src/main/java/io/trino/spi/block/
LongArrayBlock.java:32-130, IntArrayBlock.java, ByteArrayBlock.java, VariableWidthBlock.java:34-110,
DictionaryBlock.java, RunLengthEncodedBlock.java).  A Block only *describes* buffers (numpy arrays on
the host, raw HBM pointers on the device); no arithmetic happens here.
"""
import ctypes as C

import numpy as np

from . import abi

_NP_DTYPE = {abi.BIGINT: np.int64, abi.INTEGER: np.int32, abi.DATE: np.int32, abi.DOUBLE: np.float64,
             abi.BOOLEAN: np.uint8, abi.REAL: np.float32, abi.DECIMAL: np.int64}

# LongDecimalType: 16 B per position -- the low 64 bits of the magnitude, then the high 63 bits with the sign in the top bit
# (UnscaledDecimal128Arithmetic.java: SIGN_LONG_MASK); held as an (n, 2) array of uint64
_SIGN = 1 << 63


def long_decimal_words(value):
    """unscaled Python int -> (low, high) words of the reference's layout"""
    mag = -value if value < 0 else value
    if mag >> 127:
        raise OverflowError("unscaled value beyond 127 bits")
    return mag & 0xFFFFFFFFFFFFFFFF, (mag >> 64) | (_SIGN if value < 0 else 0)


def long_decimal_value(low, high):
    mag = ((int(high) & (_SIGN - 1)) << 64) | int(low)
    return -mag if int(high) & _SIGN else mag


class DeviceBuffer:
    """A span of HBM described by (pointer, bytes); `owner` keeps the allocation alive
    (a torch tensor, or a presto_amd._lib.DeviceAllocation)."""

    def __init__(self, ptr, nbytes, owner=None):
        self.ptr = int(ptr)
        self.nbytes = int(nbytes)
        self.owner = owner


def _ptr(buf):
    if buf is None:
        return None
    if isinstance(buf, DeviceBuffer):
        return buf.ptr
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data
    if hasattr(buf, "data_ptr"):  # torch tensor
        return buf.data_ptr()
    raise TypeError(type(buf))


class Block:
    def __init__(self, type_, encoding, position_count, values=None, offsets=None, nulls=None, ids=None,
                 dictionary=None, fields=None):
        self.fields = fields  # RowBlock: the field blocks (encoding ROW_FIELDS)
        self.type = type_
        self.encoding = encoding
        self.position_count = int(position_count)
        self.values = values
        self.offsets = offsets
        self.nulls = nulls
        self.ids = ids
        self.dictionary = dictionary

    # ---- host constructors -------------------------------------------------------------
    @staticmethod
    def flat(type_, values, nulls=None):
        arr = np.ascontiguousarray(np.asarray(values, dtype=_NP_DTYPE[type_]))
        nl = None
        if nulls is not None:
            nl = np.ascontiguousarray(np.asarray(nulls, dtype=np.uint8))
            if not nl.any():
                nl = None  # LongArrayBlock: valueIsNull == null when mayHaveNull is false
        return Block(type_, abi.FLAT, len(arr), values=arr, nulls=nl)

    @staticmethod
    def decimal(values, nulls=None):
        """ShortDecimalType block: unscaled values (LongArrayBlock)."""
        return Block.flat(abi.DECIMAL, values, nulls)

    @staticmethod
    def long_decimal(values, nulls=None):
        """LongDecimalType block from unscaled Python ints (None = NULL)."""
        values = list(values)
        if nulls is None and any(v is None for v in values):
            nulls = [v is None for v in values]
        arr = np.zeros((len(values), 2), dtype=np.uint64)
        for i, v in enumerate(values):
            if v is not None:
                arr[i, 0], arr[i, 1] = long_decimal_words(int(v))
        nl = None
        if nulls is not None:
            nl = np.ascontiguousarray(np.asarray(nulls, dtype=np.uint8))
            if not nl.any():
                nl = None
        return Block(abi.LONG_DECIMAL, abi.FLAT, len(values), values=np.ascontiguousarray(arr), nulls=nl)

    @staticmethod
    def bigint(values, nulls=None):
        return Block.flat(abi.BIGINT, values, nulls)

    @staticmethod
    def integer(values, nulls=None):
        return Block.flat(abi.INTEGER, values, nulls)

    @staticmethod
    def date(values, nulls=None):
        return Block.flat(abi.DATE, values, nulls)

    @staticmethod
    def double(values, nulls=None):
        return Block.flat(abi.DOUBLE, values, nulls)

    @staticmethod
    def real(values, nulls=None):
        """RealType: IEEE single values (the IntArrayBlock of their raw bits); to_pylist gives them as Python floats."""
        return Block.flat(abi.REAL, values, nulls)

    @staticmethod
    def boolean(values, nulls=None):
        return Block.flat(abi.BOOLEAN, np.asarray(values, dtype=np.uint8), nulls)

    @staticmethod
    def varchar(strings):
        """strings: iterable of bytes / str / None (None = NULL)."""
        data = bytearray()
        offsets = [0]
        nulls = []
        for s in strings:
            if s is None:
                nulls.append(1)
            else:
                if isinstance(s, str):
                    s = s.encode("utf-8")
                data += s
                nulls.append(0)
            offsets.append(len(data))
        vals = np.frombuffer(bytes(data) if data else b"\0", dtype=np.uint8).copy()
        nl = np.asarray(nulls, dtype=np.uint8)
        return Block(abi.VARCHAR, abi.VARWIDTH, len(nulls), values=vals,
                     offsets=np.asarray(offsets, dtype=np.int32), nulls=nl if nl.any() else None)

    @staticmethod
    def varwidth(values_bytes, offsets, nulls=None):
        offsets = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32))
        return Block(abi.VARCHAR, abi.VARWIDTH, len(offsets) - 1,
                     values=np.ascontiguousarray(np.asarray(values_bytes, dtype=np.uint8)), offsets=offsets,
                     nulls=nulls)

    @staticmethod
    def row(fields, nulls=None):
        """RowBlock.fromFieldBlocks (core/trino-spi/src/main/java/io/trino/spi/block/RowBlock.java)"""
        n = fields[0].position_count
        assert all(f.position_count == n for f in fields)
        return Block(abi.ROW, abi.ROW_FIELDS, n, nulls=nulls, fields=list(fields))

    @staticmethod
    def dictionary_block(dictionary, ids):
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int32))
        return Block(dictionary.type, abi.DICTIONARY, len(ids), ids=ids, dictionary=dictionary)

    @staticmethod
    def rle(value_block, position_count):
        assert value_block.position_count == 1
        return Block(value_block.type, abi.RLE, position_count, dictionary=value_block)

    # ---- materialisation (host blocks only) ----------------------------------------------
    def to_pylist(self):
        """Decoded python values (None for NULL); VARCHAR as bytes."""
        if self.encoding == abi.DICTIONARY:
            d = self.dictionary.to_pylist()
            return [d[i] for i in self.ids.tolist()]
        if self.encoding == abi.RLE:
            return self.dictionary.to_pylist() * self.position_count
        if self.encoding == abi.ROW_FIELDS:
            cols = [f.to_pylist() for f in self.fields]
            rows = [tuple(c[i] for c in cols) for i in range(self.position_count)]
            return [None if (self.nulls is not None and self.nulls[i]) else r for i, r in enumerate(rows)]
        n = self.position_count
        nulls = self.nulls
        if self.type == abi.VARCHAR:
            raw = self.values.tobytes()
            off = self.offsets.tolist()
            out = [raw[off[i]:off[i + 1]] for i in range(n)]
        elif self.type == abi.LONG_DECIMAL:
            out = [long_decimal_value(lo, hi) for lo, hi in self.values[:n].tolist()]
        else:
            out = self.values[:n].tolist()
            if self.type == abi.BOOLEAN:
                out = [bool(v) for v in out]
        if nulls is not None:
            out = [None if nulls[i] else out[i] for i in range(n)]
        return out

    def fill_c(self, col, keep):
        col.type = self.type
        col.encoding = self.encoding
        col.values = _ptr(self.values)
        col.offsets = _ptr(self.offsets)
        col.nulls = _ptr(self.nulls)
        col.ids = _ptr(self.ids)
        col.dictionary_size = 0
        if self.fields is not None:
            arr = (abi.pa_column * len(self.fields))()
            for i, f in enumerate(self.fields):
                f.fill_c(arr[i], keep)
            keep.append(arr)
            col.dictionary = C.cast(arr, C.POINTER(abi.pa_column))
            col.dictionary_size = len(self.fields)
        if self.dictionary is not None:
            d = abi.pa_column()
            self.dictionary.fill_c(d, keep)
            keep.append(d)
            col.dictionary = C.pointer(d)
            col.dictionary_size = self.dictionary.position_count
        keep.append(self)


_RETAINED = {}   # id(Page) -> [Page, releases still to come]: retained pages the native side has not let go of yet


class Page:
    def __init__(self, blocks, position_count=None, mem=abi.MEM_HOST, stable=False, pinned=False, on_release=None):
        """stable: the buffers outlive the operator the page is given to (PA_PAGE_STABLE) -- true of a Java Page, which is
        immutable and kept alive by its references; here of pages over buffers the caller keeps for the whole query.
        pinned: a host page whose buffers are pinned host memory (PA_PAGE_PINNED).
        on_release: a callable -- the page is handed over as PA_PAGE_RETAINED: its buffers stay valid until the operator calls it
        (once, from inside a later call on the operator, or its close)."""
        self.stable = stable
        self.pinned = pinned
        self.on_release = on_release
        self.native_release = None   # (function pointer, ctx) of a page an operator handed over (device_page_from_c)
        self.blocks = list(blocks)
        if position_count is None:
            position_count = self.blocks[0].position_count if self.blocks else 0
        for b in self.blocks:
            assert b.position_count == position_count, (b.position_count, position_count)
        self.position_count = int(position_count)
        self.mem = mem

    @property
    def channel_count(self):
        return len(self.blocks)

    def to_c(self):
        """Returns (pa_page, keepalive list).  A Page is immutable (Page.java:33), so the C view is built once."""
        cached = getattr(self, "_c", None)
        if cached is not None:
            return cached
        keep = []
        cols = (abi.pa_column * max(len(self.blocks), 1))()
        for i, b in enumerate(self.blocks):
            b.fill_c(cols[i], keep)
        page = abi.pa_page()
        page.position_count = self.position_count
        page.channel_count = len(self.blocks)
        page.columns = C.cast(cols, C.POINTER(abi.pa_column))
        page.mem = self.mem
        page.flags = (abi.PAGE_STABLE if self.stable else 0) | (abi.PAGE_PINNED if self.pinned else 0)
        if self.native_release is not None:   # handed over by its producer: the operator that takes the page owes the release
            page.flags |= abi.PAGE_RETAINED
            page.release, page.release_ctx = self.native_release
        if self.on_release is not None:
            callback = self.on_release
            me = self

            def release(ctx):
                try:
                    # the operator (or the lookup source it built) is done with the page: the hand-over that kept this Page object and
                    # its buffers alive ends (retain_until_released)
                    held = _RETAINED.get(id(me))
                    if held is not None:
                        held[1] -= 1
                        if held[1] <= 0:
                            del _RETAINED[id(me)]
                    callback()
                except Exception:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
            fn = abi.PAGE_RELEASE(release)
            page.flags |= abi.PAGE_RETAINED
            page.release = C.cast(fn, C.c_void_p)
            keep.append(fn)
        keep.append(cols)
        self._c = (page, keep)
        return self._c

    def __del__(self):
        # a handed-over page nobody took: its buffers go back now
        rel = getattr(self, "native_release", None)
        if rel is not None and not getattr(self, "_taken", False):
            try:
                abi.PAGE_RELEASE(rel[0])(rel[1])
            except Exception:
                pass

    def retain_until_released(self):
        if self.native_release is not None:
            if getattr(self, "_taken", False):
                raise ValueError("a handed-over page can be given to ONE operator")
            self._taken = True
        """Called when the page is handed to an operator as PA_PAGE_RETAINED: the native side will call the page's release thunk some
        time later -- from another operator's call, or when a lookup source is destroyed -- so the Page (its buffers' owners and the
        ctypes thunk) must outlive the Python references to it.  One count per hand-over."""
        if self.on_release is None:
            return
        held = _RETAINED.setdefault(id(self), [self, 0])
        held[1] += 1

    def to_rows(self):
        cols = [b.to_pylist() for b in self.blocks]
        return [tuple(c[i] for c in cols) for i in range(self.position_count)]

    def get_region(self, offset, length):
        """Page.getRegion for flat / varwidth host blocks (views, no copy)."""
        out = []
        for b in self.blocks:
            nulls = None if b.nulls is None else b.nulls[offset:offset + length]
            if b.encoding == abi.FLAT:
                out.append(Block(b.type, abi.FLAT, length, values=b.values[offset:offset + length], nulls=nulls))
            elif b.encoding == abi.VARWIDTH:
                out.append(Block(b.type, abi.VARWIDTH, length, values=b.values,
                                 offsets=b.offsets[offset:offset + length + 1], nulls=nulls))
            else:
                raise NotImplementedError
        return Page(out, length, self.mem)


def _host_array(pointer, dtype, count, copy):
    """count items of dtype at a host address a C structure holds (a view of the memory, or a copy of it).  Through a ctypes char array
    at the address: np.ctypeslib.as_array on a POINTER costs ~4 us per call, this ~1 -- twenty arrays per page of a step loop."""
    dtype = np.dtype(dtype)
    address = C.cast(pointer, C.c_void_p).value
    view = np.frombuffer((C.c_char * (count * dtype.itemsize)).from_address(address), dtype=dtype, count=count)
    return view.copy() if copy else view


def page_from_c(cpage, copy=True):
    """Builds a host Page from a pa_page whose pointers are host addresses (copies by default)."""
    n = cpage.position_count
    blocks = []
    for i in range(cpage.channel_count):
        col = cpage.columns[i]
        if col.encoding == abi.ROW_FIELDS:
            sub = abi.pa_page()
            sub.position_count = n
            sub.channel_count = col.dictionary_size
            sub.columns = col.dictionary
            sub.mem = abi.MEM_HOST
            rn = _host_array(col.nulls, np.uint8, n, True) if col.nulls else None
            blocks.append(Block.row(page_from_c(sub, copy).blocks, rn))
            continue
        nulls = _host_array(col.nulls, np.uint8, n, copy) if col.nulls else None
        if col.encoding == abi.VARWIDTH:
            off = _host_array(col.offsets, np.int32, n + 1, copy)
            total = int(off[n]) if n > 0 else 0
            vals = _host_array(col.values, np.uint8, total, copy) if total > 0 else np.zeros(1, dtype=np.uint8)
            blocks.append(Block(col.type, abi.VARWIDTH, n, values=vals, offsets=off, nulls=nulls))
        elif col.encoding == abi.FLAT and col.type == abi.LONG_DECIMAL:
            vals = _host_array(col.values, np.uint64, 2 * n, True).reshape(n, 2) if n > 0 else np.zeros((n, 2), dtype=np.uint64)
            blocks.append(Block(col.type, abi.FLAT, n, values=vals, nulls=nulls))
        elif col.encoding == abi.FLAT:
            dt = np.dtype(_NP_DTYPE[col.type])
            vals = _host_array(col.values, dt, n, copy) if n > 0 else np.zeros(0, dtype=dt)
            blocks.append(Block(col.type, abi.FLAT, n, values=vals, nulls=nulls))
        else:
            raise NotImplementedError("dictionary output")
    return Page(blocks, n, abi.MEM_HOST)


def sequence_page(length, columns):
    """SequencePageBuilder.createSequencePage (core/trino-main/src/test/java/io/trino/
    SequencePageBuilder.java:44-84): column i = start_i .. start_i+length-1; VARCHAR = decimal string."""
    blocks = []
    for type_, start in columns:
        seq = np.arange(start, start + length)
        if type_ == abi.VARCHAR:
            blocks.append(Block.varchar([str(v) for v in seq.tolist()]))
        elif type_ == abi.DOUBLE:
            blocks.append(Block.double(seq.astype(np.float64)))
        elif type_ == abi.BOOLEAN:
            blocks.append(Block.boolean((seq % 2) == 0))
        else:
            blocks.append(Block.flat(type_, seq))
    return Page(blocks, length)


# ---- wire format (PagesSerde) through the C ABI ------------------------------------------------------------------------
def serialize_page(page, stream=None, compress=False):
    """SerializedPage bytes (frame + payload) of a host or device Page: pa_page_serialize, or pa_page_serialize_lz4 (payload as
    one LZ4 block when that pays, PagesSerde.java:74-95)."""
    from ._lib import check, lib
    cpage, keep = page.to_c()
    n = page.position_count
    cap = 64 + sum(64 + 13 * n + (int(b.offsets[n]) - int(b.offsets[0]) if (b.type == abi.VARCHAR and page.mem == abi.MEM_HOST and b.encoding == abi.VARWIDTH) else 0)
                   for b in page.blocks)
    for _ in range(8):
        buf = (C.c_uint8 * cap)()
        rc = (lib().pa_page_serialize_lz4 if compress else lib().pa_page_serialize)(C.byref(cpage), buf, cap, stream)
        if rc == abi.ERR_INSUFFICIENT_RESOURCES:
            cap *= 4  # device VARCHAR bytes are only known to the library
            continue
        check(rc if rc < 0 else 0)
        return bytes(bytearray(buf)[:rc])
    raise MemoryError("serialized page does not fit")


class _PageBuffer:
    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            from ._lib import lib
            lib().pa_page_buffer_free(self.handle)
        except Exception:
            pass


def deserialize_page(data, stream=None, types=None):
    """PA_MEM_DEVICE Page from SerializedPage bytes: pa_page_deserialize (DOUBLE / DATE columns come back as BIGINT / INTEGER
    blocks of the same bits: the wire format carries encodings, not types), or with `types` pa_page_deserialize_typed."""
    from ._lib import check, lib
    from .operators import device_page_from_c
    raw = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) if data else b"\0")
    h = C.c_void_p()
    if types is None:
        check(lib().pa_page_deserialize(raw, len(data), stream, C.byref(h)))
    else:
        check(lib().pa_page_deserialize_typed(raw, len(data), abi.int32_array(types), len(types), stream, C.byref(h)))
    owner = _PageBuffer(h)
    cpage = abi.pa_page()
    check(lib().pa_page_buffer_page(h, C.byref(cpage)))
    return device_page_from_c(cpage, owner=owner)
