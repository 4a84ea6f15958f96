#!/usr/bin/env python3
"""Timing ablations of gemm_pipe_kernel (results are garbage, only the time is read): each variant is a PATCHED COPY of
csrc/gemm.hip (the product source carries no switches) compiled for one tile width / arithmetic (-DMML_LAB) and linked
with the regular objects of the other sources into tools/lab/lib_<name>.so (tools/lab/run.sh times them all; the patches
address the form that cuts BOTH operands in registers: run.sh sets GEMM_PLANES=0 so that tools/bench_gemm.py does not
hand the launches pre-cut weights).
usage: ablate_gemm.py [BN [EMU]]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = glob.glob(os.path.join(ROOT, "mmlrec-a-unified*_amd"))[0]
SRC = open(os.path.join(PKG, "csrc", "gemm.hip")).read()
BN = sys.argv[1] if len(sys.argv) > 1 else "128"
EMU = sys.argv[2] if len(sys.argv) > 2 else "2"

STEP_BARRIER = '''      epi_left = 0;
    }
    __builtin_amdgcn_s_barrier();
'''
STEP_WAIT = '''      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    } else if (pf.ok && epi_left > 0) {'''
ISSUE = '''    if (pf.ok) {
      issue((sidx + 3) & (PSTAGES - 1));
      if (advance(pf)) setup_ptrs(pf);
      ++issued;
    }
    // next step's fragments'''
READS = '''    read_b(so_next, I0{}, nb0);
    read_a(so_next, I0{}, na0);
    read_a(so_next, I1{}, RA1[Q]);
    if (NI == 2) read_b(so_next, I1{}, RB1[Q]);
'''
CUT_HR = '''  f16_cut_hr2(x, scale, c, 0);
'''
CUT_HR2 = '''  f16_cut_hr2(x, scale, c, 2);
'''
CUT_L = '''  f16_cut_l(c, 0);
  f16_cut_l(c, 1);
  f16_cut_l(c, 2);
  f16_cut_l(c, 3);
  f16_cut_done(c, o.h, o.l);
'''
NOCUT_DONE = '''  { typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ a_ = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    const u32x4_ b_ = {__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
    o.h = __builtin_bit_cast(f16x8, a_); o.l = __builtin_bit_cast(f16x8, b_); }
'''


def sub(s, old, new, count=1):
    """count = None: every occurrence (at least one)"""
    assert (s.count(old) >= 1) if count is None else (s.count(old) == count), (old[:50], s.count(old))
    return s.replace(old, new)


def nomfma(s):
    # the three MFMAs of mma_prep_f16 -> one cheap dependent VALU op each on the accumulator's first word
    for op in ("b.l, a.h", "b.h, a.l", "b.h, a.h"):
        s = sub(s, "  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(%s, acc, 0, 0, 0);\n  __builtin_amdgcn_sched_barrier(0);\n  f16_cut" % op,
                "  acc[0] += __builtin_bit_cast(float, ((__attribute__((ext_vector_type(4))) uint32_t)__builtin_bit_cast(__attribute__((ext_vector_type(4))) uint32_t, %s))[0]);\n  __builtin_amdgcn_sched_barrier(0);\n  f16_cut" % op.split(",")[0], count=None)
    return s


NOCUT_FN = '''
template <bool RC>
__device__ __forceinline__ void mma_prep_f16_nocut(f32x16& acc, const Prep<2>& a, const Prep<2>& b, const RawFrag<RC>& r,
                                                   const float scale, Prep<2>& o) {
  float x[8];
  r.get(x);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.l, a.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.h, acc, 0, 0, 0);
  { typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ a_ = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    const u32x4_ b_ = {__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
    o.h = __builtin_bit_cast(f16x8, a_); o.l = __builtin_bit_cast(f16x8, b_); }
  __builtin_amdgcn_sched_barrier(0);
}

// Keeps a prepared operand's conversion where it was written'''


def nocut_b(s):
    """what pre-cut weight planes could give at best: the column operand's fragments arrive as planes (no cut)"""
    s = sub(s, "\n// Keeps a prepared operand's conversion where it was written", NOCUT_FN)
    s = sub(s, "mma_prep_f16(acc[0][0], PA0[P], PB0[P], RB1[P], cur.sB, PB1);", "mma_prep_f16_nocut(acc[0][0], PA0[P], PB0[P], RB1[P], cur.sB, PB1);")
    s = sub(s, "mma_prep_f16(acc[1][0], PA1, PB0[P], nb0, nsB, PB0[Q]);", "mma_prep_f16_nocut(acc[1][0], PA1, PB0[P], nb0, nsB, PB0[Q]);")
    return s


VARIANTS = {
    "base": lambda s: s,
    "nobar": lambda s: sub(s, STEP_BARRIER, STEP_BARRIER.replace("    __builtin_amdgcn_s_barrier();\n", "")),
    "nowait": lambda s: sub(s, STEP_WAIT, STEP_WAIT.replace('      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");\n', "")),
    "nobar_nowait": lambda s: VARIANTS["nobar"](VARIANTS["nowait"](s)),
    "noload": lambda s: sub(VARIANTS["nowait"](s), ISSUE, ISSUE.replace("      issue((sidx + 3) & (PSTAGES - 1));\n", "")),
    "noreads": lambda s: sub(s, READS, ""),
    "nocut": lambda s: sub(sub(sub(s, CUT_HR, ""), CUT_HR2, ""), CUT_L, NOCUT_DONE),
    "nomfma": nomfma,
    "nocutB": nocut_b,
    "nocut_noreads": lambda s: VARIANTS["nocut"](VARIANTS["noreads"](s)),
    "nocut_nobar_nowait": lambda s: VARIANTS["nocut"](VARIANTS["nobar_nowait"](s)),
}


def main():
    names = os.environ.get("VARIANTS", ",".join(VARIANTS)).split(",")
    objs = [o for o in glob.glob(os.path.join(PKG, "lib", "obj", "*.o")) if not os.path.basename(o).startswith("gemm")]
    for n in names:
        tmp = os.path.join(PKG, "csrc", "_lab_gemm_%s.hip" % n)
        open(tmp, "w").write(VARIANTS[n](SRC))
        obj = os.path.join(ROOT, "tools", "lab", "gemm_%s.o" % n)
        try:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-munsafe-fp-atomics", "-Wno-unused-function", "-Wno-unused-variable", "-DMML_LAB",
                                   "-DMML_LAB_BN=" + BN, "-DMML_LAB_EMU=" + EMU, "-I" + os.path.join(ROOT, "include"),
                                   "-c", tmp, "-o", obj])
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                                   os.path.join(ROOT, "tools", "lab", "lib_%s.so" % n), obj] + objs)
            print("built", n)
        finally:
            os.remove(tmp)
            if os.path.exists(obj):
                os.remove(obj)


if __name__ == "__main__":
    main()
