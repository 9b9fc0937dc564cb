"""The C++ host mirror (include/presto_amd.hpp) end to end: tests/cpp/test_driver.cpp runs the reference's
FilterAndProject / HashAggregation known-answer cases and a two-operator Driver pipeline through the C ABI on the GPU
and checks them against the oracle.  On a machine without a GPU the same binary must refuse to run (no CPU path)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _exe():
    import __graft_entry__
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return __graft_entry__.build_cpp_driver_test()


@pytest.mark.gpu
def test_cpp_driver_cases():
    r = subprocess.run([_exe()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all cases pass" in r.stdout


def test_cpp_driver_refuses_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([_exe()], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_native_q3_equals_the_python_driven_pipelines():
    """scripts/q3_native.cpp (C++ Driver loop over the C ABI, descriptors built in C++) and presto_amd/q3.py over the same
    synthetic SF1 tables: the same ten rows, bit for bit."""
    import json

    import __graft_entry__
    from presto_amd import _lib, abi, q3, tpch
    exe = __graft_entry__.build_native_q3()
    r = subprocess.run([exe, "--sf", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    native = json.loads(r.stdout.strip().splitlines()[-1])
    assert native["input_rows"] == tpch.customer_rows(1) + tpch.orders_rows(1) + tpch.lineitem_rows(1)
    customer = tpch.DeviceColumns(tpch.CUSTOMER_COLUMNS, 1, tpch.customer_rows(1))
    orders = tpch.DeviceColumns(tpch.ORDERS_COLUMNS, 1, tpch.orders_rows(1))
    lineitem = tpch.DeviceColumns(tpch.Q3_LINEITEM_COLUMNS, 1, tpch.lineitem_rows(1))
    stream = _lib.DeviceStream()
    out, _ = q3.run(customer.pages(1 << 28), orders.pages(1 << 28), lineitem.pages(1 << 28), stream.handle, result_mem=abi.MEM_HOST, top_n=10,
                    with_count=False)
    rows = [list(r) for p in out for r in p.to_rows()]
    stream.destroy()
    assert len(rows) == 10 and [[int(a), int(b), int(c), float(d)] for a, b, c, d in rows] == native["result"]
