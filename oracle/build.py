"""ORACLE — test infrastructure only.  Builds the C restatements under oracle/
into oracle/_build/liboracle.so with gcc (no GPU code, no reference sources)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")
SOURCES = ["roi_align_ref.c", "msda_ref.c"]


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    if not force and os.path.exists(LIB) and all(
        os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs
    ):
        return LIB
    cmd = ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-std=c99", "-o", LIB] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
