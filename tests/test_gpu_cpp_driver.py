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
