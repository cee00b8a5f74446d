import os
import sys

import pytest

# The library's default compositing mode is fast_exp (v_exp_f32).  The test suite runs in the REPRODUCIBLE mode unless a test
# asks otherwise (gaustudio_amd.options(fast_exp=True) / the fast_exp-parametrised tests): that is the mode the CPU oracle
# pins to the bit.  Set through the environment, before libgsrast.so is loaded, so that spawned workers and the bench.py
# subprocesses of the distributed tests inherit it.
os.environ.setdefault("GSR_FAST_EXP", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand from oracle/gsr_oracle.c."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
