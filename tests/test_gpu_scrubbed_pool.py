"""The aggregation / join / fused-join suites once more with every recycled HBM block overwritten before it is handed out
(PRESTO_AMD_POOL_SCRUB=0xA5, pool.cpp): code that relies on what a previous owner of a block left behind -- the round-2 group
table bug needed a stale key under a colliding tag -- fails here deterministically instead of one run in sixty."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SUITES = ["tests/test_gpu_fused.py", "tests/test_gpu_fused_join.py", "tests/test_gpu_join.py", "tests/test_gpu_partial_final.py",
          "tests/test_gpu_small_pages.py", "tests/test_gpu_varchar_keys.py"]


def test_suites_on_a_scrubbed_pool(gpu):
    env = dict(os.environ, PRESTO_AMD_POOL_SCRUB="0xA5")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + SUITES, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    tail = r.stdout.decode()[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail
