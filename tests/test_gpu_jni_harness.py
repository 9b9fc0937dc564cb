"""jni/presto_amd_jni.c executed without a JVM (tests/cpp/jni_harness.c: a C stand-in for the JNI functions the shim calls): the
reference's operator KATs and one fused Q6 pass through the Java_io_trino_gpu_GpuNative_* natives, checked against the oracle;
every Get<Type>ArrayElements meets its Release; pa_status values arrive as GpuNativeException(status, message)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _exe():
    import __graft_entry__
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return __graft_entry__.build_jni_harness()


@pytest.mark.gpu
def test_jni_shim_cases():
    r = subprocess.run([_exe()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all cases pass" in r.stdout


def test_jni_shim_maps_no_device_onto_the_exception():
    """Without a GPU the first native that needs one must leave a GpuNativeException(PA_ERR_NO_DEVICE) pending -- the shim's
    status -> exception path, executed on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([_exe()], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
    assert "GpuNativeException(-9" in r.stderr and "no CPU fallback" in r.stderr
