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


CABI_SRC = os.path.join(HERE, "cabi_cpu.c")
CABI_OUT = os.path.join(HERE, "_build", "libmmlrec_cpu.so")


def build_cabi(force=False):
    """oracle/cabi_cpu.c -> oracle/_build/libmmlrec_cpu.so: the CPU restatement of the hot-path subset of
    include/mmlrec.h under the same symbols (test infrastructure; tests/test_cabi_cpu.py)."""
    hdr = os.path.join(os.path.dirname(HERE), "include", "mmlrec.h")
    if (not force and os.path.exists(CABI_OUT) and
            os.path.getmtime(CABI_OUT) >= max(os.path.getmtime(CABI_SRC), os.path.getmtime(hdr))):
        return CABI_OUT
    os.makedirs(os.path.dirname(CABI_OUT), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-Wall", "-shared", "-fPIC", "-o", CABI_OUT, CABI_SRC, "-lm"])
    return CABI_OUT


if __name__ == "__main__":
    print(build(True))
    print(build_cabi(True))
