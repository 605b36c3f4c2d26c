"""The PRODUCT library (vlm_fo1_amd/libfo1hip.so, include/fo1.h only) on the GPU in a process that never maps the test / bench build:
the test session itself runs on the product library too (tests/conftest.py) but maps libfo1hip_ab.so beside it for the pinned
tests, so this file starts a fresh interpreter WITHOUT FO1_AB and runs the driver's own smoke check
(__graft_entry__.smoke(): HFRE against the oracle + one whole hot-path pass + greedy decode against the composed oracles) on the
product library, and checks that none of include/fo1_ab.h's switches exist there."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, sys
assert "FO1_AB" not in os.environ
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as G
from vlm_fo1_amd import lib as L
lib = L.load()
assert lib._name.endswith("libfo1hip.so"), lib._name
for hook in L.SIGNATURES_AB:
    assert not hasattr(lib, hook), hook + " exported by the product library"
G.smoke()
print("PRODUCT_LIB_OK")
'''


def test_smoke_on_the_product_library():
    env = {k: v for k, v in os.environ.items() if k != "FO1_AB"}
    p = subprocess.run([sys.executable, "-c", CODE, ROOT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "PRODUCT_LIB_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
