"""Builds libfo1hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m vlm_fo1_amd.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo
snapshot (it is git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfo1hip.so")
LIB_AB = os.path.join(HERE, "libfo1hip_ab.so")
ARCH = "gfx950"


# per-file code generation flags: kernels whose MFMA results are consumed by VALU instructions at once keep the accumulators in VGPRs
FILE_FLAGS = {
    "window_attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "vision_ops.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],        # (the channel-attention kernels are its only MFMA code)
}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "fo1.h"))
    deps.append(os.path.join(HERE, "..", "include", "fo1_ab.h"))
    deps.append(os.path.join(CSRC, "exports.map"))
    return deps


def _build_one(lib: str, objdir: str, defines, force: bool, verbose: bool) -> str:
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(d) for d in _deps()):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and all(
            os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + [d for d in _deps() if d.endswith(".h")]
        ):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-fvisibility-inlines-hidden"] + list(defines) + \
            FILE_FLAGS.get(os.path.basename(src), [])
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", lib] + objs
    subprocess.check_call(cmd)
    return lib


def build(force: bool = False, verbose: bool = False, ab: bool = True) -> str:
    """Builds the PRODUCT library libfo1hip.so (include/fo1.h: no A/B switches, no process-global tuning state, none of the
    measured-slower kernel forms) and, with ab=True, the test / bench build libfo1hip_ab.so (-DFO1_ENABLE_AB: the same sources plus
    include/fo1_ab.h's switches and the kernels they select).  Returns the product library's path."""
    lib = _build_one(LIB, os.path.join(HERE, "_obj"), (), force, verbose)
    if ab:
        _build_one(LIB_AB, os.path.join(HERE, "_obj_ab"), ("-DFO1_ENABLE_AB",), force, verbose)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
