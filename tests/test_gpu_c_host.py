"""A non-Python host drives the hot path on the GPU: tests/host_emul/gpu_host.c — plain C99, HIP runtime C API for
memory and the stream, dlopen of libfo1hip.so — launches fo1_hfre_region_pool_ex (constant pyramid maps: the ROI mean is the constant,
the bf16 second destination its RNE cast, a bad argument is refused with a message) and fo1_llm_prefill (one-layer decoder with known
answers for the last hidden row and the greedy id).  No torch, no ctypes: the boundary of include/fo1.h is what is exercised."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_launches_hfre_and_llm_prefill(tmp_path):
    exe = tmp_path / "gpu_host"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
           os.path.join(ROOT, "tests", "host_emul", "gpu_host.c"), "-o", str(exe), "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-ldl", "-lm",
           "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    lib = os.path.join(ROOT, "vlm_fo1_amd", "libfo1hip.so")
    p = subprocess.run([str(exe), lib], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, f"gpu_host rc={p.returncode}\n{p.stdout}\n{p.stderr}"
    assert "hfre ok" in p.stdout and "llm ok" in p.stdout and "gpu_host ok" in p.stdout
