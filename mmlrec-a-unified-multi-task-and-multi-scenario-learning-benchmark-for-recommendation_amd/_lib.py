"""ctypes binding of include/mmlrec.h (libmmlrec_hip.so).  No CPU fallback: if the library is missing the
import of any compute entry point fails loudly."""
import ctypes as C
import os

# torch must load ITS bundled HIP runtime before our library is opened: torch's libraries ask for "libamdhip64.so"
# while ours asks for the SONAME "libamdhip64.so.7"; if ours is opened first the process ends up with two HIP
# runtimes (/opt/rocm's and torch's) and every launch through the second one fails with "no ROCm-capable device".
import torch  # noqa: F401,E402

from . import build as _build

MAX_FIELDS, MAX_GROUP, MAX_SRC, MAX_EXPERTS, MAX_GATES, MAX_HEADS, MAX_OPT_TENSORS = 64, 16, 8, 16, 8, 8, 32
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SIGMOID2 = 0, 1, 2, 3
OPT_SGD, OPT_ADAM, OPT_ADAGRAD, OPT_RMSPROP = 0, 1, 2, 3
OPT_KINDS = {"sgd": OPT_SGD, "adam": OPT_ADAM, "adagrad": OPT_ADAGRAD, "rmsprop": OPT_RMSPROP}

fp = C.c_void_p  # device pointers travel as integers
i32, i64 = C.c_int32, C.c_int64


class GemmFwdDesc(C.Structure):
    _fields_ = [("A", fp), ("W", fp), ("bias", fp), ("C", fp), ("lda", i64), ("ldw", i64), ("ldc", i64),
                ("M", i32), ("N", i32), ("K", i32), ("act", i32), ("w_kn", i32), ("pad_", i32),
                ("relu_mask", fp), ("ldmask", i64), ("amax_a", fp), ("amax_w", fp), ("amax_out", fp),
                ("w_planes", fp), ("w_kexp", fp),
                ("mul", fp), ("prod", fp), ("ldmul", i64), ("ldprod", i64), ("amax_prod", fp)]


class AmaxDesc(C.Structure):
    _fields_ = [("x", fp), ("rows", i64), ("ld", i64), ("cols", i32), ("pad_", i32), ("slot", fp)]


class GemmDgradDesc(C.Structure):
    _fields_ = [("dA", fp), ("Y", fp), ("ldda", i64), ("ldy", i64), ("M", i32), ("K", i32), ("act", i32),
                ("n_src", i32), ("accumulate", i32), ("pad_", i32),
                ("dC", fp * MAX_SRC), ("W", fp * MAX_SRC), ("lddc", i64 * MAX_SRC), ("ldw", i64 * MAX_SRC),
                ("N", i32 * MAX_SRC), ("w_kn", i32 * MAX_SRC), ("relu_mask", fp), ("ldmask", i64),
                ("amax_dc", fp * MAX_SRC), ("amax_w", fp * MAX_SRC), ("amax_out", fp),
                ("w_planes", fp * MAX_SRC), ("w_kexp", fp * MAX_SRC),
                ("gate_h", fp), ("gate_g", fp), ("d_h", fp), ("d_g", fp),
                ("ld_h", i64), ("ld_g", i64), ("ld_dh", i64), ("ld_dg", i64),
                ("act_h", i32), ("act_g", i32), ("acc_h", i32), ("acc_g", i32), ("amax_dh", fp), ("amax_dg", fp)]


class PlanesDesc(C.Structure):
    _fields_ = [("W", fp), ("planes", fp), ("rows", i64), ("ld", i64), ("cols", i32), ("layout", i32),
                ("n_amax", i32), ("pad_", i32), ("amax", fp * MAX_SRC), ("kexp", fp), ("ldp", i64),
                ("W2", fp), ("ld2", i64)]


class GemmWgradDesc(C.Structure):
    _fields_ = [("dC", fp), ("A", fp), ("dW", fp), ("dbias", fp), ("lddc", i64), ("lda", i64), ("lddw", i64),
                ("M", i32), ("N", i32), ("K", i32), ("accumulate", i32), ("w_kn", i32), ("pad_", i32),
                ("amax_dc", fp), ("amax_a", fp)]


class Cast16Desc(C.Structure):
    _fields_ = [("src", fp), ("dst", fp), ("rows", i64), ("lds", i64), ("ldd", i64), ("cols", i32), ("transpose", i32)]


class G16TnDesc(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("n_src", i32), ("act", i32), ("A", fp * MAX_SRC), ("B", fp * MAX_SRC),
                ("lda", i64 * MAX_SRC), ("ldb", i64 * MAX_SRC), ("K", i32 * MAX_SRC), ("bias", fp), ("C", fp),
                ("ldc", i64), ("c_bf16", i32), ("accumulate", i32), ("mask_out", fp), ("mask_in", fp), ("ldmask", i64)]


class G16WgradDesc(C.Structure):
    _fields_ = [("dC", fp), ("A", fp), ("dW", fp), ("dbias", fp), ("lddc", i64), ("lda", i64), ("lddw", i64),
                ("M", i32), ("N", i32), ("K", i32), ("accumulate", i32)]


G16_MAX_GROUP = 8


class RowsReduceItem(C.Structure):
    _fields_ = [("kind", i32), ("pad_", i32), ("group", fp), ("workspace", fp), ("workspace_bytes", i64)]


ROWS_REDUCE_HEAD, ROWS_REDUCE_GATE, ROWS_REDUCE_TOWER_HEAD = 0, 1, 2
MAX_REDUCE_SEGS = 40  # csrc/reduce.hpp: segments of one reduction launch
NT_MAX_GROUP = 48     # csrc/gemm_nt.hip: weight-gradient problems one gemm_nt_kernel launch takes
GATE_MIX_BF16, GATE_DE_BF16, GATE_DG_BF16, GATE_E_BF16 = 1, 2, 4, 8


class GateDesc(C.Structure):
    _fields_ = [("G", fp), ("Wg", fp), ("P", fp), ("mix", fp), ("dmix", fp), ("dG", fp), ("dWg", fp),
                ("ldg", i64), ("ldp", i64), ("ldmix", i64), ("lddmix", i64), ("lddg", i64),
                ("Gd", i32), ("ne", i32), ("g_relu", i32), ("active", i32), ("expert", i32 * MAX_EXPERTS)]


class GateGroup(C.Structure):
    _fields_ = [("E", fp * MAX_EXPERTS), ("dE", fp * MAX_EXPERTS), ("lde", i64 * MAX_EXPERTS),
                ("ldde", i64 * MAX_EXPERTS), ("n_experts", i32), ("n_gates", i32), ("H", i32), ("e_relu", i32),
                ("B", i64), ("gate", GateDesc * MAX_GATES), ("amax_mix", fp), ("amax_dE", fp), ("amax_dG", fp),
                ("out_bf16", i32), ("pad_", i32)]


class HeadDesc(C.Structure):
    _fields_ = [("Hin", fp), ("w", fp), ("w2", fp), ("bias", fp), ("bias2", fp), ("dH", fp), ("dw", fp),
                ("dbias", fp), ("ldh", i64), ("lddh", i64), ("H", i32), ("h_relu", i32), ("n_bias2", i32),
                ("mask_col", i32), ("gate", fp), ("dgate", fp), ("ldgate", i64), ("lddgate", i64),
                ("gate_act", i32), ("pad_", i32)]


class HeadGroup(C.Structure):
    _fields_ = [("n_heads", i32), ("dh_bf16", i32), ("B", i64), ("prob", fp), ("ldprob", i64), ("y", fp),
                ("ldy", i64), ("mask", fp), ("ldmask", i64), ("loss", fp), ("dprob", fp), ("lddprob", i64),
                ("head", HeadDesc * MAX_HEADS), ("amax_dH", fp), ("amax_dG", fp)]


class TowerHeadDesc(C.Structure):
    _fields_ = [("A", fp), ("lda", i64), ("amax_a", fp), ("w_planes_fwd", fp), ("w_planes_bwd", fp), ("ldpf", i64),
                ("ldpb", i64), ("kexp_fwd", fp), ("kexp_bwd", fp), ("bias1", fp), ("w", fp), ("hbias", fp),
                ("hbias2", fp), ("dH", fp), ("dA", fp), ("lddh", i64), ("ldda", i64), ("dw", fp), ("dhbias", fp),
                ("amax_dH", fp), ("amax_dA", fp), ("K", i32), ("N", i32), ("n_hbias2", i32), ("mask_col", i32),
                ("head", i32), ("pad_", i32)]


class TowerHeadGroup(C.Structure):
    _fields_ = [("n", i32), ("pad_", i32), ("M", i64), ("prob", fp), ("ldprob", i64), ("y", fp), ("ldy", i64),
                ("mask", fp), ("ldmask", i64), ("loss", fp), ("t", TowerHeadDesc * MAX_HEADS)]


class OptTensor(C.Structure):
    _fields_ = [("param", fp), ("grad", fp), ("state1", fp), ("state2", fp), ("n", i64), ("l1", C.c_float),
                ("l2", C.c_float), ("skip_rows", fp), ("row_elems", i32), ("zero_grads", i32), ("grad_marks", fp)]


class Copy2dDesc(C.Structure):
    _fields_ = [("src", fp), ("lds", i64), ("dst", fp), ("ldd", i64), ("rows", i64), ("cols", i32),
                ("accumulate", i32), ("amax_out", fp)]


class SumProdDesc(C.Structure):
    _fields_ = [("out", fp), ("x", fp * 8), ("y", fp * 8), ("n", i64), ("n_terms", i32), ("accumulate", i32),
                ("deriv_of", fp), ("act", i32), ("pad_", i32), ("amax_out", fp)]


class Attn2Desc(C.Structure):
    _fields_ = [("V", fp * 2), ("K", fp * 2), ("Q", fp * 2), ("ldv", i64 * 2), ("ldk", i64 * 2), ("ldq", i64 * 2),
                ("out", fp), ("ldo", i64), ("A", fp), ("dout", fp), ("lddo", i64), ("dV", fp * 2), ("dK", fp * 2),
                ("dQ", fp * 2), ("lddv", i64 * 2), ("lddk", i64 * 2), ("lddq", i64 * 2), ("B", i64), ("H", i32),
                ("sqrt_h", C.c_float)]


class OptHyper(C.Structure):
    _fields_ = [("kind", i32), ("step", i32), ("step_dev", fp), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("alpha", C.c_float), ("zero_grad", i32), ("max_blocks", i32)]


_PP = C.POINTER
_SIGS = {
    "mml_version": (C.c_int, []),
    "mml_last_error": (C.c_char_p, []),
    "mml_device_caps": (C.c_int, [C.c_int, _PP(i64)]),
    "mml_stream_create_cu_range": (C.c_int, [C.c_int, C.c_int, C.c_int, _PP(C.c_void_p)]),
    "mml_stream_destroy": (C.c_int, [C.c_void_p]),
    "mml_gather_fwd": (C.c_int, [_PP(fp), _PP(i64), _PP(i32), i32, i32, fp, i64, i32, i32, i64, fp, i64, fp, fp]),
    "mml_gather_wgmax_len": (i64, [i32, i32, i32, i64]),
    "mml_gather_fwd_wgmax": (C.c_int, [_PP(fp), _PP(i64), _PP(i32), i32, i32, fp, i64, i32, i32, i64, fp, i64, fp, i64, fp, fp]),
    "mml_gather_fwd_mark": (C.c_int, [_PP(fp), _PP(i64), _PP(i32), i32, i32, fp, i64, i32, i32, i64, fp, i64, fp, fp, fp]),
    "mml_rows_compact": (C.c_int, [_PP(fp), _PP(i64), _PP(i64), i32, fp, fp, i32, fp, fp]),
    "mml_gather_fwd_idx32": (C.c_int, [_PP(fp), _PP(i64), i32, i32, fp, i64, fp, i64, i32, i64, fp, i64, fp, fp]),
    "mml_scatter_bwd": (C.c_int, [_PP(fp), _PP(i64), _PP(i32), i32, i32, fp, i64, i64, fp, i64, _PP(fp), _PP(i64),
                                  fp, fp, i32, fp, fp, fp]),
    "mml_scatter_bwd_det": (C.c_int, [_PP(fp), _PP(i64), _PP(i32), i32, i32, fp, i64, i64, fp, i64, _PP(fp), fp, fp, i32,
                                      fp, fp]),
    "mml_scatter_bwd_idx32": (C.c_int, [_PP(fp), _PP(i64), i32, i32, fp, i64, i64, fp, i64, _PP(fp), _PP(i64),
                                        fp, fp, i32, fp, fp, fp]),
    "mml_index_unique_idx32": (C.c_int, [_PP(i64), i32, i32, fp, i64, i64, _PP(fp), _PP(i64), fp, fp, i32, fp, fp, fp]),
    "mml_route_count": (C.c_int, [fp, i64, fp, i64, _PP(i32), _PP(i64), i32, i64, i32, fp, fp, fp]),
    "mml_route_place": (C.c_int, [fp, i64, fp, i64, _PP(i32), _PP(i64), _PP(i64), i32, i64, i32, fp, fp, fp, fp, fp]),
    "mml_rows_permute": (C.c_int, [fp, i64, fp, i32, i32, i64, fp, fp]),
    "mml_route_list_count": (C.c_int, [fp, fp, i32, _PP(i64), _PP(i64), i32, i32, fp, fp]),
    "mml_route_list_place": (C.c_int, [fp, fp, i32, _PP(i64), _PP(i64), _PP(i64), i32, i32, fp, fp, fp, fp]),
    "mml_lookup_slots": (C.c_int, [fp, i64, _PP(i32), _PP(i64), _PP(i64), i32, i64, fp, fp, fp, fp]),
    "mml_rows_clear": (C.c_int, [fp, fp, i32, _PP(i64), _PP(fp), i32, fp]),
    "mml_shard_rows": (C.c_int, [fp, i64, fp, i64, i32, i32, i32, i32, fp]),
    "mml_amax_batch": (C.c_int, [_PP(AmaxDesc), i32, fp]),
    "mml_amax_reset": (C.c_int, [fp, i64, fp]),
    "mml_gemm_planes_cut": (C.c_int, [C.POINTER(PlanesDesc), i32, fp]),
    "mml_gemm_set_mode": (C.c_int, [i32]),
    "mml_gemm_get_mode": (C.c_int, []),
    "mml_gemm_set_panel": (C.c_int, [i32]),
    "mml_gemm_set_ws": (C.c_int, [i32]),
    "mml_gemm_set_nt": (C.c_int, [i32]),
    "mml_gemm_nt_serves": (C.c_int, [_PP(GemmWgradDesc)]),
    "mml_tower_head_serves": (C.c_int, [_PP(TowerHeadGroup)]),
    "mml_tower_head_workspace_bytes": (i64, [_PP(TowerHeadGroup)]),
    "mml_tower_head_fwd_bwd": (C.c_int, [_PP(TowerHeadGroup), fp, i64, i32, fp]),
    "mml_gemm_last_kernel": (C.c_char_p, []),
    "mml_gather_last_kernel": (C.c_char_p, []),
    "mml_gemm_set_wgrad_lds_pad": (C.c_int, [i32]),
    "mml_cast16_batch": (C.c_int, [_PP(Cast16Desc), i32, fp]),
    "mml_gather16_fwd": (C.c_int, [_PP(fp), _PP(i64), _PP(i32), i32, i32, fp, i64, i32, i32, i64, fp, i64, fp, fp]),
    "mml_g16_tn": (C.c_int, [_PP(G16TnDesc), i32, fp]),
    "mml_g16_wgrad_workspace_bytes": (i64, [_PP(G16WgradDesc), i32]),
    "mml_g16_wgrad": (C.c_int, [_PP(G16WgradDesc), i32, fp, i64, i32, fp]),
    "mml_g16_last_kernel": (C.c_char_p, []),
    "mml_gemm_grouped_fwd": (C.c_int, [_PP(GemmFwdDesc), i32, fp]),
    "mml_gemm_grouped_dgrad": (C.c_int, [_PP(GemmDgradDesc), i32, fp]),
    "mml_pep_gate_fwd": (C.c_int, [_PP(GemmFwdDesc), i32, fp]),
    "mml_pep_gate_bwd": (C.c_int, [_PP(GemmDgradDesc), i32, fp]),
    "mml_star_linear_fwd": (C.c_int, [_PP(GemmFwdDesc), i32, fp]),
    "mml_star_linear_bwd": (C.c_int, [_PP(GemmDgradDesc), i32, fp]),
    "mml_gemm_grouped_wgrad_workspace_bytes": (i64, [_PP(GemmWgradDesc), i32]),
    "mml_gemm_grouped_wgrad": (C.c_int, [_PP(GemmWgradDesc), i32, fp, i64, fp]),
    "mml_gemm_grouped_wgrad_phase": (C.c_int, [_PP(GemmWgradDesc), i32, fp, i64, i32, fp]),
    "mml_gate_mix_fwd": (C.c_int, [_PP(GateGroup), fp]),
    "mml_gate_mix_bwd_workspace_bytes": (i64, [_PP(GateGroup)]),
    "mml_gate_mix_bwd": (C.c_int, [_PP(GateGroup), fp, i64, fp]),
    "mml_gate_mix_bwd_phase": (C.c_int, [_PP(GateGroup), fp, i64, i32, fp]),
    "mml_head_workspace_bytes": (i64, [_PP(HeadGroup)]),
    "mml_head_fwd": (C.c_int, [_PP(HeadGroup), fp]),
    "mml_head_bce_fwd_bwd": (C.c_int, [_PP(HeadGroup), fp, i64, fp]),
    "mml_head_bce_fwd_bwd_phase": (C.c_int, [_PP(HeadGroup), fp, i64, i32, fp]),
    "mml_rows_reduce_batch": (C.c_int, [_PP(RowsReduceItem), i32, fp]),
    "mml_ew_mul": (C.c_int, [fp, fp, fp, i64, fp]),
    "mml_ew_mul_bwd": (C.c_int, [fp, fp, fp, fp, fp, i32, i32, i64, fp]),
    "mml_ew_mul_bwd_act": (C.c_int, [fp, fp, fp, fp, fp, i32, i32, i64, i32, i32, fp]),
    "mml_ew_add_n": (C.c_int, [_PP(fp), i32, fp, i64, fp]),
    "mml_sumprod_batch": (C.c_int, [_PP(SumProdDesc), i32, fp]),
    "mml_copy2d": (C.c_int, [fp, i64, fp, i64, i64, i32, i32, fp]),
    "mml_dropout": (C.c_int, [fp, i64, fp, i64, i64, i32, i64, C.c_float, C.c_uint64, C.c_uint32, fp, i32, i32, fp]),
    "mml_copy2d_batch": (C.c_int, [_PP(Copy2dDesc), i32, fp]),
    "mml_auc_segments": (C.c_int, [fp, i64, fp, i64, i64, i32, i32, fp, fp]),
    "mml_bn_workspace_bytes": (C.c_int64, [i64, i32]),
    "mml_bn_fwd": (C.c_int, [fp, i64, fp, fp, fp, fp, fp, fp, fp, fp, i64, i64, i32, i32, i32, C.c_float, C.c_float, fp,
                             i64, fp]),
    "mml_bn_bwd": (C.c_int, [fp, i64, fp, i64, fp, fp, fp, fp, i64, fp, fp, i32, i64, i32, fp, i64, fp]),
    "mml_domain_bn_update": (C.c_int, [fp, i64, fp, i64, i64, i32, i32, fp, fp, C.c_float, fp]),
    "mml_domain_bn_eval": (C.c_int, [fp, i64, fp, i64, fp, fp, fp, i64, i64, i32, i32, C.c_float, fp]),
    "mml_snr_gate_weights_fwd": (C.c_int, [fp, fp, fp, fp, i32, i64, i32, C.c_float, C.c_float, C.c_float, fp]),
    "mml_snr_gate_weights_bwd": (C.c_int, [fp, fp, fp, fp, fp, fp, i32, i32, i32, i64, i32, C.c_float, C.c_float,
                                           C.c_float, fp, fp]),
    "mml_attn2_fwd": (C.c_int, [_PP(Attn2Desc), fp]),
    "mml_attn2_bwd": (C.c_int, [_PP(Attn2Desc), fp]),
    "mml_esmm_combine": (C.c_int, [fp, i64, fp, i64, fp, i64, fp, i64, fp, i64, fp, i64, fp]),
    "mml_apg_features_fwd": (C.c_int, [fp, i64, fp, i64, fp, i64, i64, i32, i32, i32, fp]),
    "mml_apg_features_bwd": (C.c_int, [fp, i64, fp, i64, fp, i64, i64, i32, i32, i32, fp]),
    "mml_apg_weights": (C.c_int, [fp, fp, fp, fp, i64, i32, i32, i32, i32, i32, i32, fp]),
    "mml_escm_combine": (C.c_int, [fp, i64, fp, i64, fp, i64, fp, i64, fp, i64, fp, i64, C.c_float, C.c_float, fp]),
    "mml_act_bwd": (C.c_int, [fp, fp, fp, i64, i32, fp]),
    "mml_copy_cols": (C.c_int, [_PP(fp), _PP(i64), _PP(fp), _PP(i64), _PP(i32), i32, i64, i32, fp]),
    "mml_opt_step_dense": (C.c_int, [_PP(OptTensor), i32, _PP(OptHyper), fp]),
    "mml_opt_step_rows": (C.c_int, [_PP(fp), _PP(fp), _PP(fp), _PP(fp), _PP(fp), _PP(i64), i32, i32, fp, fp, i32,
                                    _PP(fp), _PP(OptHyper), fp]),
    "mml_opt_catchup_rows": (C.c_int, [_PP(fp), _PP(fp), _PP(fp), _PP(fp), _PP(i64), i32, i32, fp, fp, i32,
                                       _PP(OptHyper), fp]),
    "mml_opt_catchup_dense": (C.c_int, [fp, fp, fp, fp, i64, i32, _PP(OptHyper), fp]),
    "mml_index_unique": (C.c_int, [_PP(i64), _PP(i32), i32, i32, fp, i64, i64, _PP(fp), _PP(i64), fp, fp, i32, fp, fp,
                                   fp]),
    "mml_counter_update": (C.c_int, [fp, i32, i32, fp]),
}
EXPORTS = tuple(_SIGS)

_lib = None


class MMLError(RuntimeError):
    pass


def library_path():
    return _build.LIBPATH


def load():
    """Load libmmlrec_hip.so (building it first when hipcc is available and the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MMLREC_LIB") or _build.LIBPATH  # MMLREC_LIB: an alternative build (kernel ablations, tools/lab)
    if path == _build.LIBPATH and (not os.path.exists(path) or _build.needs_build()) and os.path.exists(_build.HIPCC):
        _build.build_library(verbose=False)
    if not os.path.exists(path):
        raise MMLError(f"HIP extension {path} is missing and cannot be built (no hipcc): the MI355X path has no "
                       "CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError = library / header out of sync: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mml_last_error()
        raise MMLError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device (or host) address of a torch tensor, or None."""
    return None if t is None else t.data_ptr()
