"""Pins the oracle's hash arithmetic: XXH64 against the committed public vectors, the 64-bit mixes
against independent pure-Python restatements of the reference's formulas."""
import json
import os
import struct

import numpy as np

from presto_amd import abi
from presto_amd.page import Block, Page

M = (1 << 64) - 1


def s64(v):
    v &= M
    return v - (1 << 64) if v >= (1 << 63) else v


def rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M


def py_hash_bigint(v):  # AbstractLongType.java:126-130
    return s64(rotl(((v & M) * 0xC2B2AE3D27D4EB4F) & M, 31) * 0x9E3779B185EBCA87)


def py_murmur(x):  # PagesHash.java:225-241
    x &= M
    x ^= x >> 33
    x = (x * 0xff51afd7ed558ccd) & M
    x ^= x >> 33
    x = (x * 0xc4ceb9fe1a85ec53) & M
    x ^= x >> 33
    return s64(x)


def golden():
    with open(os.path.join(os.path.dirname(__file__), "golden", "xxh64_vectors.json")) as f:
        return json.load(f)


def test_xxh64_matches_public_vectors(oracle):
    g = golden()
    for v in g["bytes"]:
        assert "%016x" % oracle.xxh64(bytes.fromhex(v["hex"])) == v["xxh64"], v["hex"]
    for v in g["longs"]:
        assert "%016x" % (oracle.xxh64_long(v["value"]) & M) == v["xxh64"], v["value"]


def test_survey_listed_vectors(oracle):
    # Q1 key bytes
    assert oracle.xxh64(b"") == 0xEF46DB3751D8E999
    assert oracle.xxh64(b"A") == 0x13099D40D095B684
    assert oracle.xxh64(b"N") == 0x16B6310EBD34BD7C
    assert oracle.xxh64(b"R") == 0x59AF2DD4153E940D
    assert oracle.xxh64(b"F") == 0xF3CE876B32C937D5
    assert oracle.xxh64(b"O") == 0x3EEFBA32C3918CC3
    assert oracle.xxh64(b"\0" * 8) == 0x34C96ACDCADB1BBB


def test_scalar_mixes(oracle):
    assert oracle.hash_bigint(0) == 0 and oracle.murmur3_fmix(0) == 0 and oracle.combine_hash(0, 5) == 5
    assert oracle.murmur3_fmix(1) == s64(0xB456BCFC34C2CB2C)  # MurmurHash3 fmix64(1)
    rng = np.random.default_rng(1)
    for v in [1, -1, 2 ** 63 - 1, -2 ** 63, 42] + [int(x) for x in rng.integers(-2 ** 62, 2 ** 62, 200)]:
        assert oracle.hash_bigint(v) == py_hash_bigint(v)
        assert oracle.murmur3_fmix(v) == py_murmur(v)
        assert oracle.combine_hash(v, 17) == s64(31 * v + 17)
    # INTEGER / DATE hash the sign-extended value (AbstractIntType.java:141-145)
    assert oracle.hash_integer(-5) == py_hash_bigint(-5)
    # DOUBLE: -0.0 folds onto +0.0, bits of the value otherwise (DoubleType.java:163-170)
    assert oracle.hash_double(-0.0) == oracle.hash_double(0.0) == 0
    assert oracle.hash_double(1.5) == py_hash_bigint(struct.unpack("<q", struct.pack("<d", 1.5))[0])
    assert oracle.hash_double(float("nan")) == py_hash_bigint(0x7ff8000000000000)
    # BOOLEAN has no HASH_CODE operator -> XxHash64.hash(1L / 0L) (BooleanType.java:39-40)
    assert oracle.hash_boolean(True) == oracle.xxh64_long(1) and oracle.hash_boolean(False) == oracle.xxh64_long(0)


def test_array_size(oracle):
    # fastutil HashCommon.arraySize(expected, 0.75f) = max(2, nextPowerOfTwo(ceil(expected / f)))
    assert [oracle.array_size(n) for n in (0, 1, 2, 3, 4, 6, 7, 100, 10000, 98304, 98305)] == [2, 2, 4, 4, 8, 8, 16, 256, 16384, 131072, 262144]


def test_row_hash_and_partitions(oracle):
    page = Page([Block.bigint([1, 2, 3], [0, 1, 0]), Block.varchar([b"A", None, b"xyz"]), Block.double([0.5, -0.0, 2.0])], 3)
    h = oracle.hash_page(page, [0, 1, 2])
    exp = []
    for i in range(3):
        r = 0
        r = s64(31 * r + (0 if i == 1 else py_hash_bigint([1, 2, 3][i])))
        r = s64(31 * r + (0 if i == 1 else s64(oracle.xxh64([b"A", b"", b"xyz"][i]))))
        r = s64(31 * r + oracle.hash_double([0.5, -0.0, 2.0][i]))
        exp.append(r)
    assert h.tolist() == exp
    # local rule: (int) XxHash64.hash(Long.reverse(rawHash)) & (P - 1)   (LocalPartitionGenerator.java:61-65)
    def rev(x):
        return int("{:064b}".format(x & M)[::-1], 2)
    local = oracle.partition_ids(h, 8, local=True).tolist()
    assert local == [(oracle.xxh64_long(rev(x)) & 0xFFFFFFFF) & 7 for x in exp]
    # remote rule: (rawHash & MAX_LONG) % P   (HashGenerator.java:24-35)
    assert oracle.partition_ids(h, 5, local=False).tolist() == [(x & 0x7fffffffffffffff) % 5 for x in exp]
    pos, counts = oracle.partition_positions(np.array([1, 0, 1, 2, 0, 1], dtype=np.int32), 4)
    assert pos.tolist() == [1, 4, 0, 2, 5, 3] and counts.tolist() == [2, 3, 1, 0]


def test_lz4_block_format_hand_computed(oracle):
    """The LZ4 block codec of the oracle (PagesSerde's compressor is the un-vendored io.airlift:aircompressor; parity is
    pinned by the public block format): a block written by hand -- one literal 'a', a match of 14 at offset 1 (the overlapping
    copy replicates the byte), then the mandatory 5 closing literals -- inflates to 20 x 'a'; and the encoder's output, whatever
    its choices, round-trips."""
    block = bytes([0x1A, 0x61, 0x01, 0x00, 0x50]) + b"aaaaa"
    assert oracle.lz4_decompress(block, 20) == b"a" * 20
    # literal-length extension: 15 + 255 + 3 = 273 literals, no match
    block = bytes([0xF0, 0xFF, 0x03]) + bytes(range(256)) + bytes(range(17))
    assert oracle.lz4_decompress(block, 273) == bytes(range(256)) + bytes(range(17))
    for data in (b"", b"x", b"0123456789ab", b"abcd" * 100, bytes(range(256)) * 3, b"\0" * 5000):
        assert oracle.lz4_decompress(oracle.lz4_compress(data), len(data)) == data
