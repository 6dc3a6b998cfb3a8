"""Build the plain-C MSDA oracle (oracle/msda_oracle.c) with gcc.

Output: oracle/_build/libmsda_oracle.so (git-ignored, travels to the GPU box).
The upstream reference's native op is CUDA + ATen (models/bricks/ops/cuda/*.cu)
and cannot be compiled in this image (no nvcc), so there is no ``oracle/_ref``
build; the reference's *Python* fallback path is used instead to generate the
golden vectors under tests/golden/ (see tests/golden/make_golden.py).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
SRC = os.path.join(HERE, "msda_oracle.c")
LIB = os.path.join(OUT_DIR, "libmsda_oracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-std=c99", SRC, "-o", LIB, "-lm"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
