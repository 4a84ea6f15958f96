"""The C ABI of include/mmlrec.h, exercised in the CPU container ("the same symbols are provided by the CPU
restatement library so tests run in both containers").

oracle/_build/libmmlrec_cpu.so (oracle/cabi_cpu.c -- test infrastructure, never loaded by the package) exports the
hot-path subset of the header under the same symbols and signatures.  Here it is bound with the SAME ctypes signature
table and descriptor structs the package uses for the HIP library (mmlrec_amd/_lib.py) and driven through the call
sequence mmlrec_amd/engine.py records for MMoE -- gather, grouped Linear(+ReLU) launches, gate mix, head + summed BCE,
the backward launches, scatter, dense Adam -- on a fixture the unmodified reference produced
(tests/golden/mmoe_ae30.npz): probabilities, loss, every gradient and the parameters after one Adam step.
What this pins on CPU: the descriptor layouts and argument meaning of the header (a layout or stride mistake in
_lib.py / ops.py shows up here without a GPU), and the semantics each entry point documents."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden

RTOL = 1e-4


@pytest.fixture(scope="module")
def cpu():
    from oracle import build_fast
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib as L
    lib = C.CDLL(build_fast.build_cabi())
    for name, (res, args) in L._SIGS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib, L


def test_cpu_library_symbols_are_a_subset_of_the_header_with_matching_signatures(cpu):
    lib, L = cpu
    import re
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib._name]).decode()
    syms = sorted(set(re.findall(r"\bT (mml_[a-z0-9_]+)", out)))
    assert len(syms) >= 18
    hdr = open(os.path.join(ROOT, "include", "mmlrec.h")).read()
    for s in syms:
        assert re.search(r"\b%s\s*\(" % s, hdr), f"{s} exported by the CPU library but not declared in include/mmlrec.h"
        assert s in L._SIGS, s
    assert lib.mml_version() == 100
    assert lib.mml_gemm_grouped_fwd(None, 3, None) == -1 and b"descriptor" in lib.mml_last_error()


def test_cpu_library_dropout_is_the_oracle_mask(cpu):
    """Two independent restatements of the mask contract of mml_dropout (C here, numpy in the oracle) agree bit for bit
    -- and with the HIP kernel (tests/test_dropout_gpu.py compares it with the numpy one)."""
    lib, L = cpu
    from oracle import mmlrec_oracle as orc
    rng = np.random.default_rng(5)
    rows, cols, p, seed, site, step = 300, 37, 0.3, 0xfeedfacecafebeef, 0x9e3779b9, 5
    x = rng.standard_normal((rows, cols + 3)).astype(np.float32)
    out = np.ones((rows, cols), dtype=np.float32)
    sd = np.array([step], dtype=np.int32)
    assert lib.mml_dropout(x.ctypes.data, cols + 3, out.ctypes.data, cols, rows, cols, 11, p, seed, site, sd.ctypes.data,
                           99, 1, None) == 0
    want = x[:, :cols] * orc.dropout_scale(rows, cols, p, seed, step, site, row0=11) + np.float32(1.0)
    assert np.array_equal(out, want.astype(np.float32))
    assert lib.mml_dropout(x.ctypes.data, cols + 3, out.ctypes.data, cols, rows, cols, 0, 1.0, seed, site, None, 0, 0,
                           None) == -1


def test_cpu_library_planes_cut_layout(cpu):
    """The plane image of mml_gemm_planes_cut (include/mmlrec.h) from the C restatement against a numpy one: both layouts,
    a ragged reduction with a wider planes pitch, a group of two magnitudes.  (tests/test_gemm_pipe_gpu.py pins the HIP
    kernel to the same numpy image.)"""
    lib, L = cpu
    rng = np.random.default_rng(4)

    def slot(v):
        s = np.zeros(8, dtype=np.uint32)
        s[3] = np.float32(v).view(np.uint32)
        return s

    def kexp(v):
        e = (np.float32(v).view(np.uint32) >> 23) & 0xff
        return int(np.clip(141 - int(e), -110, 110))

    def image(W, k, layout):
        y = W.astype(np.float32) * np.float32(2.0 ** k)
        h = y.astype(np.float16)
        lo = (y - h.astype(np.float32)).astype(np.float16)
        hb, lb = h.view(np.uint16).astype(np.uint32), lo.view(np.uint16).astype(np.uint32)
        if layout == 1:
            hb, lb = hb.T, lb.T
        out = np.zeros(hb.shape, dtype=np.uint32)
        S = [[4 * hh + (e & 3) + 8 * (e >> 2) for e in range(8)] for hh in range(2)]
        for b in range(hb.shape[1] // 16):
            bh, bl = hb[:, 16 * b:16 * b + 16], lb[:, 16 * b:16 * b + 16]
            for hh in range(2):
                for i in range(4):
                    k0, k1 = S[hh][2 * i], S[hh][2 * i + 1]
                    out[:, 16 * b + 4 * hh + i] = bh[:, k0] | (bh[:, k1] << 16)
                    out[:, 16 * b + 8 + 4 * hh + i] = bl[:, k0] | (bl[:, k1] << 16)
        return out.T.copy() if layout == 1 else out

    N, K, KP = 48, 72, 80
    W = (rng.standard_normal((N, K)) * 3e-4).astype(np.float32)
    s_own, s_big = slot(np.abs(W).max()), slot(7.5)
    arr = (L.PlanesDesc * 3)()
    outs = [np.zeros((N, KP), dtype=np.uint32) for _ in range(3)]
    kx = np.zeros(3, dtype=np.int32)
    for i, (layout, slots) in enumerate(((0, [s_own]), (1, [s_own]), (1, [s_own, s_big]))):
        d = arr[i]
        d.W, d.planes, d.rows, d.ld, d.cols, d.layout, d.ldp = W.ctypes.data, outs[i].ctypes.data, N, K, K, layout, KP
        d.n_amax = len(slots)
        for a, sl in enumerate(slots):
            d.amax[a] = sl.ctypes.data
        d.kexp = kx[i:i + 1].ctypes.data
    assert lib.mml_gemm_planes_cut(arr, 3, None) == 0
    Wpad = np.zeros((N, KP), dtype=np.float32)
    Wpad[:, :K] = W
    k_own, k_grp = kexp(np.abs(W).max()), min(kexp(np.abs(W).max()), kexp(7.5))
    assert kx.tolist() == [k_own, k_own, k_grp]
    assert np.array_equal(outs[0], image(Wpad, k_own, 0))
    assert np.array_equal(outs[1], image(Wpad, k_own, 1))
    assert np.array_equal(outs[2], image(Wpad, k_grp, 1))
    arr[0].ldp = K  # no room for the rounded-up block of the ragged reduction
    assert lib.mml_gemm_planes_cut(arr, 1, None) == -1
    # K6: planes of the element-wise product of two factors (STAR's W_specific (.) W_shared, [K, N] layout), exponent from
    # the product of the factors' bounds
    Kk, Nn = 64, 48
    A = (rng.standard_normal((Kk, Nn)) * 2.0).astype(np.float32)
    B = (rng.standard_normal((Kk, Nn)) * 0.05).astype(np.float32)
    sa, sb = slot(np.abs(A).max()), slot(np.abs(B).max())
    arr = (L.PlanesDesc * 2)()
    outs = [np.zeros((Kk, Nn), dtype=np.uint32) for _ in range(2)]
    kx = np.zeros(2, dtype=np.int32)
    for i, layout in enumerate((1, 0)):  # forward of a [K, N] weight: down the rows; its input gradient: along a row
        d = arr[i]
        d.W, d.W2, d.planes, d.rows, d.ld, d.ld2, d.cols, d.layout = A.ctypes.data, B.ctypes.data, outs[i].ctypes.data, Kk, Nn, Nn, Nn, layout
        d.n_amax = 1
        d.amax[0], d.amax[1] = sa.ctypes.data, sb.ctypes.data
        d.kexp = kx[i:i + 1].ctypes.data
    assert lib.mml_gemm_planes_cut(arr, 2, None) == 0
    kp = kexp(np.float32(np.abs(A).max()) * np.float32(np.abs(B).max()))
    assert kx.tolist() == [kp, kp]
    assert np.array_equal(outs[0], image(A * B, kp, 1))
    assert np.array_equal(outs[1], image(A * B, kp, 0))


def test_cpu_library_named_k7_k6_entry_points(cpu):
    """mml_pep_gate_fwd / _bwd and mml_star_linear_fwd / _bwd (include/mmlrec.h): the grouped GEMMs restricted to the
    descriptors they are named for -- results against numpy, and the refusals."""
    lib, L = cpu
    rng = np.random.default_rng(9)
    M, K, N = 37, 24, 16
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 5).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    mul = rng.standard_normal((M, N)).astype(np.float32)
    C, prod = np.zeros((M, N), np.float32), np.zeros((M, N), np.float32)
    d = (L.GemmFwdDesc * 1)()
    q = d[0]
    q.A, q.W, q.bias, q.C, q.lda, q.ldw, q.ldc, q.M, q.N, q.K, q.act = ptr(A), ptr(W), ptr(b), ptr(C), K, K, N, M, N, K, L.ACT_SIGMOID2
    assert lib.mml_pep_gate_fwd(d, 1, None) == -1          # no mul / prod
    q.mul, q.prod, q.ldmul, q.ldprod = ptr(mul), ptr(prod), N, N
    assert lib.mml_pep_gate_fwd(d, 1, None) == 0
    gate = 2.0 / (1.0 + np.exp(-(A.astype(np.float64) @ W.T.astype(np.float64) + b)))
    assert np.allclose(C, gate, rtol=1e-6) and np.allclose(prod, gate * mul, rtol=1e-6, atol=1e-7)
    assert lib.mml_star_linear_fwd(d, 1, None) == -1       # nn.Linear layout, no planes
    # gate-mode input gradient
    dC = rng.standard_normal((M, N)).astype(np.float32)
    h = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    g = (2.0 / (1.0 + np.exp(-rng.standard_normal((M, K))))).astype(np.float32)
    dh, dg = np.ones((M, K), np.float32), np.zeros((M, K), np.float32)
    e = (L.GemmDgradDesc * 1)()
    r = e[0]
    r.M, r.K, r.n_src = M, K, 1
    r.dC[0], r.W[0], r.lddc[0], r.ldw[0], r.N[0], r.w_kn[0] = ptr(dC), ptr(W), N, K, N, 0
    dA = np.zeros((M, K), np.float32)
    r.dA, r.ldda = ptr(dA), K
    assert lib.mml_pep_gate_bwd(e, 1, None) == -1          # not in gate mode
    assert lib.mml_star_linear_bwd(e, 1, None) == -1
    r.gate_h, r.gate_g, r.d_h, r.d_g, r.ld_h, r.ld_g, r.ld_dh, r.ld_dg = ptr(h), ptr(g), ptr(dh), ptr(dg), K, K, K, K
    r.act_h, r.act_g, r.acc_h, r.acc_g = L.ACT_RELU, L.ACT_SIGMOID2, 1, 0
    assert lib.mml_pep_gate_bwd(e, 1, None) == 0
    v = dC.astype(np.float64) @ W.astype(np.float64)
    assert np.allclose(dh, 1.0 + v * g * (h > 0), rtol=1e-5, atol=1e-6)
    assert np.allclose(dg, v * h * (g.astype(np.float64) * (1 - g.astype(np.float64) / 2)), rtol=1e-5, atol=1e-6)


def ptr(a):
    return a.ctypes.data


def ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def test_mmoe_training_step_through_the_cpu_cabi_matches_the_reference_golden(cpu):
    lib, L = cpu
    g = load_golden("mmoe_ae30")
    P = {k[6:]: np.array(g[k], dtype=np.float32) for k in g.files if k.startswith("state/")}
    X, y = np.ascontiguousarray(g["X0"], dtype=np.float32), np.ascontiguousarray(g["y0"], dtype=np.float32)
    vocab = [int(v) for v in g["vocab"]]
    names = [str(s) for s in g["sparse_names"]]
    B, F, E, T, Ne = X.shape[0], len(vocab), 8, 2, 4
    K0 = F * E
    f32 = lambda *s: np.zeros(s, dtype=np.float32)  # noqa: E731
    amax = np.zeros((64, 8), dtype=np.uint32)  # operand-magnitude slots (raised by every producer below)
    slot = iter(range(64))
    new_slot = lambda: amax[next(slot)]  # noqa: E731

    # ---- K1 gather
    tabs = [P[f"embedding_dict.{n}.weight"] for n in names]
    x0 = f32(B, K0)
    status = np.zeros(1, dtype=np.int32)
    rc = lib.mml_gather_fwd(ptr_array(tabs), (C.c_int64 * F)(*vocab), (C.c_int32 * F)(*range(F)), F, E, ptr(X), X.shape[1],
                            0, 0, B, ptr(x0), K0, ptr(status), None)
    assert rc == 0 and status[0] == 0
    assert np.array_equal(x0, g["dnn_input"])  # index work: bit-exact

    def linear_group(probs):
        """probs: (A, W, b, act) -> outputs, relu masks; one mml_gemm_grouped_fwd call."""
        arr = (L.GemmFwdDesc * len(probs))()
        outs, masks = [], []
        for d, (A, W, b, act) in zip(arr, probs):
            out = f32(A.shape[0], W.shape[0])
            m = np.zeros((A.shape[0], (W.shape[0] + 31) // 32), dtype=np.uint32)
            d.A, d.W, d.bias, d.C = ptr(A), ptr(W), (ptr(b) if b is not None else None), ptr(out)
            d.lda, d.ldw, d.ldc = A.shape[1], W.shape[1], out.shape[1]
            d.M, d.N, d.K, d.act, d.w_kn = A.shape[0], W.shape[0], A.shape[1], act, 0
            d.relu_mask, d.ldmask = ptr(m), m.shape[1]
            d.amax_out = ptr(new_slot())
            outs.append(out)
            masks.append(m)
        assert lib.mml_gemm_grouped_fwd(arr, len(probs), None) == 0
        return outs, masks

    W_ = lambda k: P[k + ".weight"]  # noqa: E731
    b_ = lambda k: P[k + ".bias"]  # noqa: E731
    # ---- experts (two ReLU layers) and gate DNNs (one), as ONE grouped launch per layer like engine.LinearGroupOp
    l1, m1 = linear_group([(x0, W_(f"expert_dnn.{e}.linears.0"), b_(f"expert_dnn.{e}.linears.0"), L.ACT_RELU) for e in range(Ne)] +
                          [(x0, W_(f"gate_dnn.{t}.linears.0"), b_(f"gate_dnn.{t}.linears.0"), L.ACT_RELU) for t in range(T)])
    h1, G, mG = l1[:Ne], l1[Ne:], m1[Ne:]
    h2, m2 = linear_group([(h1[e], W_(f"expert_dnn.{e}.linears.1"), b_(f"expert_dnn.{e}.linears.1"), L.ACT_RELU) for e in range(Ne)])
    H = h2[0].shape[1]
    # ---- K4 gates
    grp = L.GateGroup()
    grp.n_experts, grp.n_gates, grp.H, grp.B, grp.e_relu = Ne, T, H, B, 1
    dE = [f32(B, H) for _ in range(Ne)]
    for e in range(Ne):
        grp.E[e], grp.lde[e], grp.dE[e], grp.ldde[e] = ptr(h2[e]), H, ptr(dE[e]), H
    Pg, mix, dmix, dG, dWg = [], [], [], [], []
    for t in range(T):
        d = grp.gate[t]
        Wg = W_(f"gate_dnn_final_layer.{t}")
        Pg.append(f32(B, Ne)); mix.append(f32(B, H)); dmix.append(f32(B, H)); dG.append(f32(B, G[t].shape[1])); dWg.append(f32(*Wg.shape))
        d.G, d.Wg, d.P, d.mix, d.dmix, d.dG, d.dWg = ptr(G[t]), ptr(Wg), ptr(Pg[t]), ptr(mix[t]), ptr(dmix[t]), ptr(dG[t]), ptr(dWg[t])
        d.ldg, d.ldp, d.ldmix, d.lddmix, d.lddg = G[t].shape[1], Ne, H, H, G[t].shape[1]
        d.Gd, d.ne, d.g_relu, d.active = G[t].shape[1], Ne, 1, 1
        for e in range(Ne):
            d.expert[e] = e
    s_mix, s_dE, s_dG, s_dH = new_slot(), new_slot(), new_slot(), new_slot()
    grp.amax_mix, grp.amax_dE, grp.amax_dG = ptr(s_mix), ptr(s_dE), ptr(s_dG)
    assert lib.mml_gate_mix_fwd(C.byref(grp), None) == 0
    assert float(s_mix.view(np.float32).max()) == float(max(np.abs(m).max() for m in mix))
    # ---- towers + heads (+ summed BCE and its backward)
    tw, mt = linear_group([(mix[t], W_(f"tower_dnn.{t}.linears.0"), b_(f"tower_dnn.{t}.linears.0"), L.ACT_RELU) for t in range(T)])
    hg = L.HeadGroup()
    prob, loss = f32(B, T), f32(1)
    hg.n_heads, hg.B, hg.prob, hg.ldprob, hg.y, hg.ldy, hg.loss = T, B, ptr(prob), T, ptr(y), T, ptr(loss)
    dH, dw, dbias = [], [], []
    for t in range(T):
        d = hg.head[t]
        w = W_(f"tower_dnn_final_layer.{t}")
        dH.append(f32(B, tw[t].shape[1])); dw.append(f32(w.size)); dbias.append(f32(1))
        d.Hin, d.w, d.bias, d.dH, d.dw, d.dbias = ptr(tw[t]), ptr(w), ptr(P[f"out.{t}.bias"]), ptr(dH[t]), ptr(dw[t]), ptr(dbias[t])
        d.ldh, d.lddh, d.H, d.h_relu, d.n_bias2, d.mask_col = tw[t].shape[1], tw[t].shape[1], tw[t].shape[1], 1, 0, -1
    hg.amax_dH = ptr(s_dH)
    assert lib.mml_head_bce_fwd_bwd(C.byref(hg), None, 0, None) == 0
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))  # noqa: E731
    assert rel(prob, g["y_pred"]) < RTOL
    assert abs(float(loss[0]) - float(g["loss"])) / float(g["loss"]) < RTOL

    # ---- backward: weight gradients (one grouped launch per layer) and input gradients
    grads = {}

    def wgrad(items):
        arr = (L.GemmWgradDesc * len(items))()
        for d, (dCm, A, key) in zip(arr, items):
            grads[key + ".weight"], grads[key + ".bias"] = f32(dCm.shape[1], A.shape[1]), f32(dCm.shape[1])
            d.dC, d.A, d.dW, d.dbias = ptr(dCm), ptr(A), ptr(grads[key + ".weight"]), ptr(grads[key + ".bias"])
            d.lddc, d.lda, d.lddw, d.M, d.N, d.K = dCm.shape[1], A.shape[1], A.shape[1], A.shape[0], dCm.shape[1], A.shape[1]
        ws = np.zeros(lib.mml_gemm_grouped_wgrad_workspace_bytes(arr, len(items)), dtype=np.uint8)
        assert lib.mml_gemm_grouped_wgrad(arr, len(items), ptr(ws), ws.size, None) == 0

    def dgrad(problems):
        """problems: (dA, mask or None, [(dC, W)]) -- relu' from the sign mask the forward wrote."""
        arr = (L.GemmDgradDesc * len(problems))()
        for d, (dA, mask, srcs) in zip(arr, problems):
            d.dA, d.ldda, d.M, d.K, d.n_src = ptr(dA), dA.shape[1], dA.shape[0], dA.shape[1], len(srcs)
            d.act = L.ACT_RELU if mask is not None else L.ACT_NONE
            if mask is not None:
                d.relu_mask, d.ldmask = ptr(mask), mask.shape[1]
            for s, (dCm, Wm) in enumerate(srcs):
                d.dC[s], d.W[s], d.lddc[s], d.ldw[s], d.N[s], d.w_kn[s] = ptr(dCm), ptr(Wm), dCm.shape[1], Wm.shape[1], dCm.shape[1], 0
        assert lib.mml_gemm_grouped_dgrad(arr, len(problems), None) == 0

    for t in range(T):
        grads[f"tower_dnn_final_layer.{t}.weight"] = dw[t].reshape(W_(f"tower_dnn_final_layer.{t}").shape)
        grads[f"out.{t}.bias"] = dbias[t]
    wgrad([(dH[t], mix[t], f"tower_dnn.{t}.linears.0") for t in range(T)])
    dgrad([(dmix[t], None, [(dH[t], W_(f"tower_dnn.{t}.linears.0"))]) for t in range(T)])
    ws = np.zeros(max(lib.mml_gate_mix_bwd_workspace_bytes(C.byref(grp)), 8), dtype=np.uint8)
    assert lib.mml_gate_mix_bwd(C.byref(grp), ptr(ws), ws.size, None) == 0
    for t in range(T):
        grads[f"gate_dnn_final_layer.{t}.weight"] = dWg[t]
    wgrad([(dE[e], h1[e], f"expert_dnn.{e}.linears.1") for e in range(Ne)])
    dh1 = [f32(B, h1[e].shape[1]) for e in range(Ne)]
    dgrad([(dh1[e], m1[e], [(dE[e], W_(f"expert_dnn.{e}.linears.1"))]) for e in range(Ne)])
    wgrad([(dh1[e], x0, f"expert_dnn.{e}.linears.0") for e in range(Ne)] + [(dG[t], x0, f"gate_dnn.{t}.linears.0") for t in range(T)])
    dx0 = f32(B, K0)
    dgrad([(dx0, None, [(dh1[e], W_(f"expert_dnn.{e}.linears.0")) for e in range(Ne)] +
            [(dG[t], W_(f"gate_dnn.{t}.linears.0")) for t in range(T)])])
    # ---- K2 scatter
    gt = [np.zeros_like(t) for t in tabs]
    rc = lib.mml_scatter_bwd(ptr_array(gt), (C.c_int64 * F)(*vocab), (C.c_int32 * F)(*range(F)), F, E, ptr(X), X.shape[1], B,
                             ptr(dx0), K0, None, None, None, None, 0, None, ptr(status), None)
    assert rc == 0
    for n, t in zip(names, gt):
        grads[f"embedding_dict.{n}.weight"] = t
    for k in P:
        assert rel(grads[k], g["grad/" + k].astype(np.float64)) < RTOL, k

    # ---- K8 dense Adam, step 1 (torch.optim defaults), every tensor in one call
    keys = list(P)
    arr = (L.OptTensor * len(keys))()
    mom = {k: (np.zeros_like(P[k]), np.zeros_like(P[k])) for k in keys}
    for d, k in zip(arr, keys):
        grads[k] = np.ascontiguousarray(grads[k], dtype=np.float32)
        d.param, d.grad, d.state1, d.state2, d.n = ptr(P[k]), ptr(grads[k]), ptr(mom[k][0]), ptr(mom[k][1]), P[k].size
    hy = L.OptHyper()
    cfg_lr = 0.005
    hy.kind, hy.step, hy.lr, hy.beta1, hy.beta2, hy.eps, hy.alpha = L.OPT_ADAM, 1, cfg_lr, 0.9, 0.999, 1e-8, 0.99
    assert lib.mml_opt_step_dense(arr, len(keys), C.byref(hy), None) == 0
    for k in keys:
        ref = g["adam1/" + k].astype(np.float64)
        dv = np.abs(P[k].astype(np.float64) - ref)
        # first-step Adam turns a noise-level gradient's sign into an lr-sized move: outlier share + absolute bound
        assert (dv > RTOL * max(np.abs(ref).max(), 1e-30)).mean() < 2e-3, k
        assert dv.max() <= 2.5 * cfg_lr, k


def test_bf16_storage_family_on_the_cpu_library(cpu):
    """The bf16-storage entry points (include/mmlrec.h K3': mml_gather16_fwd, mml_cast16_batch, mml_g16_tn forward and input
    gradient, mml_g16_wgrad) through the SAME ctypes descriptors ops.py builds for the HIP library, on host tensors: pins
    the descriptor layouts and argument meaning of the new section of the header without a GPU -- a two-layer chain
    (gather -> Linear + ReLU with sign masks -> Linear) and its backward against float64 torch on the bf16-rounded
    operands."""
    import torch
    lib, L = cpu
    from mmlrec_amd import ops
    g = torch.Generator().manual_seed(0)
    B, F, E, H1, H2 = 128, 8, 16, 128, 128
    vocab = [5, 9, 100, 3, 40, 7, 2, 11]
    tabs = [torch.randn(v, E, generator=g) for v in vocab]
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocab], 1)
    X = idx.float().contiguous()
    x0 = torch.empty(B, F * E, dtype=torch.bfloat16)
    vs = (L.i64 * F)(*vocab)
    col = (L.i32 * F)(*range(F))
    status = torch.zeros(1, dtype=torch.int32)
    assert lib.mml_gather16_fwd(ops._ptr_array(tabs), vs, col, F, E, X.data_ptr(), F, 0, 0, B, x0.data_ptr(), F * E,
                                status.data_ptr(), None) == 0 and int(status) == 0
    ref0 = torch.cat([tabs[f][idx[:, f]] for f in range(F)], 1).to(torch.bfloat16)
    assert torch.equal(x0, ref0)
    W1, b1 = torch.randn(H1, F * E, generator=g) / (F * E) ** 0.5, torch.randn(H1, generator=g) * 0.1
    W2 = torch.randn(H2, H1, generator=g) / H1 ** 0.5
    W1b, W2b = torch.empty(H1, F * E, dtype=torch.bfloat16), torch.empty(H2, H1, dtype=torch.bfloat16)
    W2t = torch.empty(H1, H2, dtype=torch.bfloat16)
    arr = ops.make_cast16_descs([(W1, W1b, False), (W2, W2b, False), (W2, W2t, True)])
    assert lib.mml_cast16_batch(arr, 3, None) == 0
    assert torch.equal(W1b, W1.to(torch.bfloat16)) and torch.equal(W2t, W2.t().contiguous().to(torch.bfloat16))
    h1 = torch.empty(B, H1, dtype=torch.bfloat16)
    mask = torch.zeros(B, H1 // 32, dtype=torch.int32)
    y = torch.empty(B, H2)
    d1 = ops.make_g16_tn_descs([dict(srcs=[(x0, W1b)], C=h1, bias=b1, act=L.ACT_RELU, mask_out=mask)])
    assert lib.mml_g16_tn(d1, 1, None) == 0
    d2 = ops.make_g16_tn_descs([dict(srcs=[(h1, W2b)], C=y)])
    assert lib.mml_g16_tn(d2, 1, None) == 0
    r1 = torch.relu(x0.double() @ W1b.double().t() + b1.double())
    assert (h1.double() - r1).abs().max() <= 2.0 ** -8 * r1.abs().max()
    r2 = h1.double() @ W2b.double().t()
    assert (y.double() - r2).abs().max() <= 1e-5 * r2.abs().max()
    bits = ((mask.unsqueeze(-1) >> torch.arange(32)) & 1).reshape(B, -1).bool()
    assert torch.equal(bits, h1.float() > 0)
    # backward: dy -> (dW2, db2), dh1 (ReLU derivative from the sign bits, bf16), then dW1
    dy = torch.randn(B, H2, generator=g).to(torch.bfloat16)
    dW2, db2 = torch.full((H2, H1), 7.0), torch.full((H2,), 7.0)
    wd = ops.make_g16_wgrad_descs([dict(dC=dy, A=h1, dW=dW2, dbias=db2)])
    ws = torch.empty(256, dtype=torch.uint8)
    assert lib.mml_g16_wgrad_workspace_bytes(wd, 1) >= 0
    assert lib.mml_g16_wgrad(wd, 1, ws.data_ptr(), ws.numel(), 0, None) == 0
    rw = dy.double().t() @ h1.double()
    assert (dW2.double() - rw).abs().max() <= 1e-5 * rw.abs().max()
    assert (db2.double() - dy.double().sum(0)).abs().max() <= 1e-5 * dy.double().sum(0).abs().max()
    dh1 = torch.empty(B, H1, dtype=torch.bfloat16)
    dd = ops.make_g16_tn_descs([dict(srcs=[(dy, W2t)], C=dh1, mask_in=mask)])
    assert lib.mml_g16_tn(dd, 1, None) == 0
    rd = (dy.double() @ W2t.double().t()) * bits
    assert (dh1.double() - rd).abs().max() <= 2.0 ** -8 * rd.abs().max()
    assert bool((dh1.float()[~bits] == 0).all())
    # the row kernels' bf16 outputs exist in the HIP library only: the CPU library says so instead of writing fp32
    grp = L.GateGroup()
    grp.n_experts, grp.n_gates, grp.H, grp.B, grp.out_bf16 = 1, 1, 4, 0, 1
    assert lib.mml_gate_mix_fwd(C.byref(grp), None) == -3


@pytest.mark.parametrize("B,H,T_,gate_act", [(333, 64, 2, "none"), (1000, 128, 4, "sigmoid2"), (77, 16, 3, "sigmoid")])
def test_cpu_library_gated_heads(cpu, B, H, T_, gate_act):
    """Gated heads of include/mmlrec.h (mml_head_desc.gate, round 6; reference model/pepnet.py:72-78: the last PPNet layer
    reads h (.) 2 sigmoid(gate)) through the CPU restatement with the package's descriptor structs: probabilities, loss, dH,
    dgate, dw, dbias against float64 -- the same numpy reference tests/test_kernels_gpu.py::test_gated_head_bce holds the
    HIP kernel to."""
    lib, L = cpu
    from oracle import mmlrec_oracle as orc
    rng = np.random.default_rng(B + H)
    acts = {"none": L.ACT_NONE, "sigmoid": L.ACT_SIGMOID, "sigmoid2": L.ACT_SIGMOID2}
    y = (rng.random((B, T_)) < 0.4).astype(np.float32)
    prob = np.zeros((B, T_), np.float32)
    loss = np.zeros(1, np.float32)
    hg = L.HeadGroup()
    hg.n_heads, hg.B, hg.prob, hg.ldprob, hg.y, hg.ldy, hg.loss = T_, B, prob.ctypes.data, T_, y.ctypes.data, T_, loss.ctypes.data
    keep, ref_loss = [], 0.0
    for t in range(T_):
        Hin = np.maximum(rng.standard_normal((B, H)), 0).astype(np.float32)
        z = rng.standard_normal((B, H))
        if gate_act == "none":
            g, dact = z.astype(np.float32), np.ones_like(z)
        elif gate_act == "sigmoid":
            g = (1 / (1 + np.exp(-z))).astype(np.float32)
            dact = g.astype(np.float64) * (1 - g.astype(np.float64))
        else:
            g = (2 / (1 + np.exp(-z))).astype(np.float32)
            dact = g.astype(np.float64) * (1 - g.astype(np.float64) / 2)
        w = (rng.standard_normal(H) * 0.5 / np.sqrt(H)).astype(np.float32)
        bias = rng.standard_normal(1).astype(np.float32)
        he = Hin.astype(np.float64) * g.astype(np.float64)
        logit = (he @ w + bias).astype(np.float32)
        p = orc.sigmoid(logit)
        ref_loss += orc.bce_sum(p, y[:, t])
        p64 = p.astype(np.float64)
        dlogit = (p64 - y[:, t]) / np.maximum(p64 * (1 - p64), 1e-12) * p64 * (1 - p64)
        dH, dG = np.full((B, H), np.nan, np.float32), np.full((B, H), np.nan, np.float32)
        dw, db = np.zeros(H, np.float32), np.zeros(1, np.float32)
        d = hg.head[t]
        d.Hin, d.ldh, d.H, d.w, d.bias = Hin.ctypes.data, H, H, w.ctypes.data, bias.ctypes.data
        d.dH, d.lddh, d.dw, d.dbias, d.h_relu, d.mask_col = dH.ctypes.data, H, dw.ctypes.data, db.ctypes.data, 1, -1
        d.gate, d.ldgate, d.dgate, d.lddgate, d.gate_act = g.ctypes.data, H, dG.ctypes.data, H, acts[gate_act]
        keep.append((Hin, g, w, bias, dH, dG, dw, db, p, dlogit[:, None] * w[None, :] * g * (Hin > 0),
                     dlogit[:, None] * w[None, :] * Hin * dact, dlogit @ he, dlogit.sum()))
    assert lib.mml_head_bce_fwd_bwd(C.byref(hg), None, 0, None) == 0

    def rel(a, b):
        return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)
    assert abs(float(loss[0]) - ref_loss) / ref_loss < 1e-4
    for t, (Hin, g, w, bias, dH, dG, dw, db, p, rH, rG, rw, rb) in enumerate(keep):
        assert rel(prob[:, t], p) < 1e-5 and rel(dH, rH) < 2e-5 and rel(dG, rG) < 2e-5 and rel(dw, rw) < 2e-5
        assert abs(float(db[0]) - rb) < 2e-5 * max(abs(rb), 1.0)
