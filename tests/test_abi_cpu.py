"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/mmlrec.h declares (no compute call is made here -- there is no GPU in the build container)."""
import ctypes
import os
import re

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mmlrec.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mml_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mmlrec.h but not exported"
    assert set(syms) == set(_lib.EXPORTS), set(syms) ^ set(_lib.EXPORTS)
    assert lib.mml_version() >= 100


def test_struct_sizes_match_header():
    """ctypes mirrors of the descriptor structs must have the C layout (checked via a tiny C program)."""
    import subprocess
    import tempfile
    from mmlrec_amd import _lib
    names = {"mml_gemm_fwd_desc": _lib.GemmFwdDesc, "mml_gemm_dgrad_desc": _lib.GemmDgradDesc,
             "mml_gemm_wgrad_desc": _lib.GemmWgradDesc, "mml_gate_desc": _lib.GateDesc,
             "mml_gate_group": _lib.GateGroup, "mml_head_desc": _lib.HeadDesc, "mml_head_group": _lib.HeadGroup,
             "mml_opt_tensor": _lib.OptTensor, "mml_opt_hyper": _lib.OptHyper,
             "mml_copy2d_desc": _lib.Copy2dDesc, "mml_sumprod_desc": _lib.SumProdDesc, "mml_attn2_desc": _lib.Attn2Desc,
             "mml_amax_desc": _lib.AmaxDesc, "mml_rows_reduce_item": _lib.RowsReduceItem, "mml_cast16_desc": _lib.Cast16Desc, "mml_g16_tn_desc": _lib.G16TnDesc,
             "mml_g16_wgrad_desc": _lib.G16WgradDesc, "mml_planes_desc": _lib.PlanesDesc}
    src = '#include <stdio.h>\n#include "mmlrec.h"\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    for n, cls in names.items():
        assert ctypes.sizeof(cls) == sizes[n], (n, ctypes.sizeof(cls), sizes[n])


def test_bad_arguments_are_rejected_without_a_gpu():
    from mmlrec_amd import _lib
    lib = _lib.load()
    rc = lib.mml_gemm_grouped_fwd(None, 3, None)
    assert rc == -1 and b"descriptor" in lib.mml_last_error()
    rc = lib.mml_counter_update(None, 1, 0, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from mmlrec_amd import ops, _lib
    with pytest.raises(_lib.MMLError):
        ops.gather_fwd([torch.zeros(4, 8)], torch.zeros(2, 1), [0])
