#!/usr/bin/env python3
"""Generates tests/golden/xxh64_vectors.json with the independent python `xxhash` package (3.8.1).

The reference takes XXH64 from the un-vendored io.airlift:slice (XxHash64.hash(Slice), seed 0; version managed by
io.airlift:airbase:110, /root/reference/pom.xml:5-9) and holds no numeric vector for it in its own tests
(SURVEY.md 8c), so the oracle's restatement is pinned against the public algorithm through these vectors.
Inputs cover every branch of the algorithm: < 4, 4..7, 8..31, >= 32 bytes, and the Q1 / Q3 key strings."""
import json
import os
import struct

import xxhash

inputs = [b"", b"a", b"abc", b"A", b"N", b"R", b"F", b"O", b"BUILDING", b"AUTOMOBILE", b"\x00" * 8,
          bytes(range(3)), bytes(range(4)), bytes(range(7)), bytes(range(8)), bytes(range(15)), bytes(range(31)),
          bytes(range(32)), bytes(range(33)), bytes(range(63)), bytes(range(64)), bytes(range(100)), bytes(range(256)) * 3]
longs = [0, 1, -1, 42, 2 ** 63 - 1, -2 ** 63, 0x0123456789ABCDEF, 6000000]
out = {
    "generator": "python xxhash %s, xxh64 seed 0" % xxhash.VERSION,
    "bytes": [{"hex": b.hex(), "xxh64": "%016x" % xxhash.xxh64(b).intdigest()} for b in inputs],
    # XxHash64.hash(long) == XXH64 of the 8 little-endian bytes
    "longs": [{"value": v, "xxh64": "%016x" % xxhash.xxh64(struct.pack("<q", v)).intdigest()} for v in longs],
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "xxh64_vectors.json"), "w") as f:
    json.dump(out, f, indent=1)
