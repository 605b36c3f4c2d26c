import os
import sys

import pytest

# The test session runs on the test / bench build of the library (libfo1hip_ab.so): the parity tests pin the GEMM tile / split-K / GEMV
# routing through include/fo1_ab.h's switches, which the product library (libfo1hip.so) does not have.  tests/test_product_lib_gpu.py
# and __graft_entry__.smoke() / bench.py run the product library.
os.environ.setdefault("FO1_AB", "1")

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
