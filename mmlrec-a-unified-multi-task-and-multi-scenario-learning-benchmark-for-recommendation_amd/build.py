"""Builds csrc/*.hip into lib/libmmlrec_hip.so for gfx950 with hipcc (no cmake, no JIT cache).

The .so is kept in-tree (git-ignored) so it travels to the GPU box with the snapshot.
"""
import glob
import time
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libmmlrec_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
GEMM_SOURCES = ("gemm.hip",)  # kernels with hand-counted waits: resource / assembly checks below
# also order their LDS-DMA rings with compile-time counted s_waitcnt vmcnt(N) (gemm_panel.hip: pn_younger counts the
# epilogue stores of each piece): a spill would add scratch VMEM operations to the count -> resource report checked too
COUNTED_WAIT_SOURCES = GEMM_SOURCES + ("gemm_panel.hip", "gemm_ws.hip", "gemm_os.hip")
COUNTED_WAIT_KERNELS = ("gemm_pipe_kernel", "gemm_panel_kernel", "gemm_ws_kernel", "gemm_os_kernel")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function"]
if os.environ.get("MMLREC_BUILD_OS_LAB") == "1":  # the parts-switched-off variants of gemm_os_kernel (tools/lab/os_lab.sh)
    FLAGS.append("-DMML_OS_LAB")
if os.environ.get("MMLREC_BUILD_NTSTORE") == "1":  # nontemporal output stores of the tile kernel's epilogue (csrc/gemm.hip)
    FLAGS.append("-DMML_GEMM_NTSTORE")


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
    """Compile the HIP sources for gfx950 (only those newer than their object) and link the C-ABI shared library.
    Returns its path.  Concurrent callers (one process per GPU) are serialised by a file lock and the library is
    replaced atomically, so no rank ever maps a half-written file."""
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another process built it while we waited
                return LIBPATH
            return _build_locked(force, verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, jobs):
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    all_srcs = sources()
    if not all_srcs:
        raise RuntimeError("no HIP sources under " + CSRC)
    t_build = time.time()  # (the library gets this time stamp: see the end of this function)
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(os.path.dirname(HERE), "include", "mmlrec.h")]
    hdr_time = _newest(hdrs)

    def stale(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        return force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)

    srcs = [s for s in all_srcs if stale(s)]
    jobs = jobs or max(1, min(len(srcs), (os.cpu_count() or 2) - 1))
    procs = []
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in all_srcs]
    pending = list(srcs)
    failed = []

    def reap(block):
        for item in list(procs):
            p, src, t_start, obj = item
            if block:
                p.wait()
            if p.poll() is not None:
                procs.remove(item)
                if p.returncode != 0:
                    failed.append(src)
                elif os.path.exists(obj):
                    # an object is as new as the moment its compile STARTED: a source edited while hipcc ran (minutes for
                    # gemm.hip) must look newer than the object it did not go into
                    os.utime(obj, (t_start, t_start))

    while pending or procs:
        while pending and len(procs) < jobs:
            src = pending.pop(0)
            obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            stderr = None
            if os.path.basename(src) in COUNTED_WAIT_SOURCES:
                # the pipelined GEMM counts its own VMEM operations (s_waitcnt vmcnt(N)): a register spill would add
                # scratch traffic to that count and silently break the waits -> check the resource report
                cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
                if os.path.basename(src) in GEMM_SOURCES:
                    cmd.insert(-4, "-save-temps=obj")  # keeps the device assembly next to the object (checked below)
                stderr = open(obj + ".log", "w")
            if verbose:
                print("[mmlrec build]", " ".join(cmd), flush=True)
            t_start = time.time()
            procs.append((subprocess.Popen(cmd, stderr=stderr), src, t_start, obj))
        reap(block=True)
    if failed:
        for src in failed:
            log = os.path.join(objdir, os.path.basename(src)[:-4] + ".o.log")
            if os.path.exists(log):
                sys.stderr.write("".join(l for l in open(log) if "remark:" not in l)[-4000:])
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    for g in COUNTED_WAIT_SOURCES:
        check_no_scratch(os.path.join(objdir, g[:-4] + ".o.log"))
    check_async_lds(objdir)
    tmp = LIBPATH + ".tmp.%d" % os.getpid()
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print("[mmlrec build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.utime(tmp, (t_build, t_build))  # anything edited since the build began is newer than the library (needs_build)
    os.replace(tmp, LIBPATH)
    return LIBPATH


def check_no_scratch(log):
    """Every instantiation of a kernel with counted vmcnt waits (COUNTED_WAIT_KERNELS) must use no scratch memory (see
    build_library)."""
    if not os.path.exists(log):
        return
    name, bad = None, []
    for line in open(log):
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize" in line and name and any(k in name for k in COUNTED_WAIT_KERNELS):
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
    out, bad = "", False
    for a in asm:
        r = subprocess.run([sys.executable, tool, a], capture_output=True, text=True)
        out += r.stdout
        bad = bad or r.returncode != 0
    for f in set(glob.glob(os.path.join(objdir, "gemm*-hip-*")) + glob.glob(os.path.join(objdir, "gemm*-host-*")) +
                 glob.glob(os.path.join(objdir, "gemm*.hip-*"))):
        os.remove(f)  # -save-temps leftovers (tens of MB that would travel with every snapshot)
    if bad:
        raise RuntimeError("pipelined GEMM reads an LDS-loaded register before its wait:\n" + out[-2000:])


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
