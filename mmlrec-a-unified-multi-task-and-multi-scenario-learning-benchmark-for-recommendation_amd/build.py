"""Builds csrc/*.hip into lib/libmmlrec_hip.so for gfx950 with hipcc (no cmake, no JIT cache).

The .so is kept in-tree (git-ignored) so it travels to the GPU box with the snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libmmlrec_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIBPATH):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        [os.path.join(os.path.dirname(HERE), "include", "mmlrec.h")]
    return _newest(deps) > os.path.getmtime(LIBPATH)


def build_library(force=False, verbose=True, jobs=None):
    """Compile every HIP source for gfx950 and link the C-ABI shared library. Returns its path."""
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sources()
    if not srcs:
        raise RuntimeError("no HIP sources under " + CSRC)
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    procs, objs = [], []
    pending = list(srcs)
    failed = []

    def reap(block):
        for item in list(procs):
            p, src = item
            if block:
                p.wait()
            if p.poll() is not None:
                procs.remove(item)
                if p.returncode != 0:
                    failed.append(src)

    while pending or procs:
        while pending and len(procs) < jobs:
            src = pending.pop(0)
            obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
            objs.append(obj)
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            stderr = None
            if os.path.basename(src) == "gemm.hip":
                # the pipelined GEMM counts its own VMEM operations (s_waitcnt vmcnt(N)): a register spill would add
                # scratch traffic to that count and silently break the waits -> check the resource report
                cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
                cmd.insert(-4, "-save-temps=obj")  # keeps the device assembly next to the object (checked below)
                stderr = open(obj + ".log", "w")
            if verbose:
                print("[mmlrec build]", " ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd, stderr=stderr), src))
        reap(block=True)
    if failed:
        for src in failed:
            log = os.path.join(objdir, os.path.basename(src)[:-4] + ".o.log")
            if os.path.exists(log):
                sys.stderr.write("".join(l for l in open(log) if "remark:" not in l)[-4000:])
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    check_no_scratch(os.path.join(objdir, "gemm.o.log"))
    check_async_lds(objdir)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBPATH] + objs
    if verbose:
        print("[mmlrec build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIBPATH


def check_no_scratch(log):
    """Every gemm_pipe_kernel instantiation must use no scratch memory (see build_library)."""
    if not os.path.exists(log):
        return
    name, bad = None, []
    for line in open(log):
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize" in line and name and "gemm_pipe_kernel" in name:
            if int(line.split("ScratchSize [bytes/lane]:")[1].split()[0]) != 0:
                bad.append(name)
    if bad:
        raise RuntimeError("register spills in the pipelined GEMM (counted vmcnt waits would break): " + ", ".join(bad))


def check_async_lds(objdir):
    """No register written by an asm ds_read may be read before the hand-placed lgkmcnt wait (tools/check_async_lds.py)."""
    asm = [f for f in glob.glob(os.path.join(objdir, "gemm*gfx950*.s"))]
    tool = os.path.join(os.path.dirname(HERE), "tools", "check_async_lds.py")
    if not asm or not os.path.exists(tool):
        return
    r = subprocess.run([sys.executable, tool, asm[0]], capture_output=True, text=True)
    for f in glob.glob(os.path.join(objdir, "gemm-*")) + glob.glob(os.path.join(objdir, "gemm.hip-*")):
        os.remove(f)  # -save-temps leftovers (tens of MB that would travel with every snapshot)
    if r.returncode != 0:
        raise RuntimeError("pipelined GEMM reads an LDS-loaded register before its wait:\n" + r.stdout[-2000:])


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
