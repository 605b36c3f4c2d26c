import os
import sys

import pytest

# The test session runs on the PRODUCT library (vlm_fo1_amd/libfo1hip.so — the one bench.py times and a deployment loads): every
# parity test that needs no determinism pin exercises exactly that binary.  Tests that pin the GEMM tile /
# split-K / GEMV routing or compare A/B kernels ask for the `ab_library` fixture: inside it every call goes through the test / bench
# build (libfo1hip_ab.so, include/fo1_ab.h), outside it through the product library again (vlm_fo1_amd.lib.use_ab).
os.environ.pop("FO1_AB", None)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def _use_ab():
    from vlm_fo1_amd import lib as L
    with L.use_ab() as lib:
        assert lib._name.endswith("libfo1hip_ab.so")
        yield lib
    assert not L.ab_build(), "the session falls back to the product library after an A/B test"


@pytest.fixture
def ab_library():
    """The test / bench build for the duration of ONE test (include/fo1_ab.h switches available through L.load())."""
    yield from _use_ab()


@pytest.fixture(scope="module")
def ab_library_module():
    yield from _use_ab()


@pytest.fixture
def product_library():
    """Asserts — before and after the test — that the active library is the PRODUCT build and that it is really mapped into this
    process (the same check the driver's native-code record makes)."""
    from vlm_fo1_amd import lib as L

    def check():
        lib = L.load()
        assert lib._name.endswith("libfo1hip.so") and not L.ab_build(), lib._name
        for hook in L.SIGNATURES_AB:
            assert not hasattr(lib, hook), hook + " exported by the product library"
        with open("/proc/self/maps") as f:
            assert any(line.rstrip().endswith("vlm_fo1_amd/libfo1hip.so") for line in f), "libfo1hip.so is not mapped"

    check()
    yield L.load()
    check()
