import importlib
import os
import sys

import pytest

# the test process is the library's host: device<->host copies on the DMA engines (INTEGRATION.md "runtime settings"), set before
# any HIP runtime comes up
os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("alevin-fry_amd")


@pytest.fixture(scope="session")
def oracle_module():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as ora  # noqa

    ora.lib()
    return ora


class _OracleInDeviceArithmetic:
    """The oracle as the `-m gpu` parity tests see it.  The device EM sums a round's shares in order-free fixed point
    (csrc/afq_em2.hip) unless AFQ_EM_ORDER=canonical selects the sequential f32 sums; the oracle restates both
    (afq_oracle.cpp: em_update = the reference's arithmetic, em_update_fixed), and a GPU test that does not say otherwise
    compares the device with the oracle IN THE ARITHMETIC THE DEVICE RUNS, bit for bit.  For the EM resolutions that comparison
    (em_arith="fixed") is a SELF-CHECK of the device against a restatement of its own arithmetic: it catches a kernel that drops
    or doubles a share, and it carries no parity weight.  What counts toward north_star's 1e-4 is the comparison with the oracle
    in the REFERENCE's arithmetic (em_arith="reference"): tests/test_gpu_em.py (every EM resolution, small cells),
    tests/test_gpu_fullsize.py (PBMC-sized and tailed cells, inside the measured shuffle envelope), bench.py's cpu_baseline legs,
    tests/test_em_arith_cpu.py."""

    def __init__(self, mod):
        self._mod = mod

    def __getattr__(self, name):
        return getattr(self._mod, name)

    def quant(self, *a, **kw):
        kw.setdefault("em_arith", "reference" if os.environ.get("AFQ_EM_ORDER") == "canonical" else "fixed")
        return self._mod.quant(*a, **kw)


@pytest.fixture
def oracle(request, oracle_module):
    """CPU tests get the oracle as it is (the reference's arithmetic); the test_gpu_* modules the view above."""
    if request.module.__name__.startswith("test_gpu"):
        return _OracleInDeviceArithmetic(oracle_module)
    return oracle_module
