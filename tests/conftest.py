import os
import sys

import pytest
import torch  # noqa: F401  -- before libpresto_amd.so: the torch wheel bundles its own HIP/HSA runtime, and a process that
#                 loads torch's after /opt/rocm's (through libpresto_amd.so) finds no GPU from torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    """Initialises the HIP library on device 0; fails loudly (no skip) when it is unusable."""
    from presto_amd import _lib
    _lib.init(0)
    return _lib
