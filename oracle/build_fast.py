"""Builds oracle/fast.c -> oracle/_build/liboracle_fast.so with gcc + OpenMP (portable x86-64 flags: the .so travels
from the build container to the GPU box)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fast.c")
OUT = os.path.join(HERE, "_build", "liboracle_fast.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-shared", "-fPIC", "-o", OUT, SRC, "-lm"])
    return OUT


if __name__ == "__main__":
    print(build(True))
