#!/usr/bin/env python3
"""bench_ops.run() on its own (no headline, no CPU twins): `python3 scripts/run_bench_ops.py hash_agg hash_join` -- what a rocprofv3
kernel trace of the operator benchmarks is taken of.  Prints the entries as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from presto_amd import _lib
    import bench_ops
    _lib.init(0)
    only = set(sys.argv[1:]) or None
    print(json.dumps(bench_ops.run(cpu=False, only=only), indent=1))
