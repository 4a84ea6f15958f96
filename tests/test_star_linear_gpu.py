"""K6: STAR's derived weight W_specific (.) W_shared (reference model/utils.py:171-218, SharedSpecificLinear in its [K, N]
layout) reaching the GEMMs as planes cut straight from its two factors: mml_gemm_planes_cut with W2, mml_star_linear_fwd,
mml_star_linear_bwd -- against the cut of the materialised product (bit for bit) and float64."""
import numpy as np
import pytest

from test_gemm_pipe_gpu import _kexp, _planes_ref

pytestmark = pytest.mark.gpu
RTOL = 2e-6


@pytest.fixture()
def env():
    import torch
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib as L, ops
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    lib = L.load()
    mode0 = lib.mml_gemm_get_mode()
    yield torch, L, ops, lib
    lib.mml_gemm_set_mode(mode0)


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("scale", [(1e-4, 1.0), (1.0, 1.0), (3e3, 40.0)])
def test_product_planes_are_the_planes_of_the_fp32_product(env, scale):
    """Both layouts, pitched factors, a group of two products sharing one exponent.  The exponent comes from the PRODUCT OF
    THE FACTORS' BOUNDS (>= the product's own magnitude), the planes from the fp32 product the element-wise kernel stores."""
    torch, L, ops, lib = env
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(8)
    K, N = 96, 80
    bufa = torch.zeros(K, N + 16, device=dev)
    Wa = bufa[:, :N]
    Wa.copy_((torch.randn(K, N, generator=g) * scale[0]).to(dev))
    Wb = (torch.randn(K, N, generator=g) * scale[1]).to(dev)
    Va = (torch.randn(48, 64, generator=g) * scale[0] * 5).to(dev)
    Vb = (torch.randn(48, 64, generator=g) * scale[1]).to(dev)
    slots = ops.amax_slots(4, dev)
    ops.amax_batch([(Wa, slots[0]), (Wb, slots[1]), (Va, slots[2]), (Vb, slots[3])])
    pr = torch.zeros(K, N, dtype=torch.int32, device=dev)
    pc = torch.zeros(K, N + 16, dtype=torch.int32, device=dev)[:, :N]
    pv = torch.zeros(48, 64, dtype=torch.int32, device=dev)
    kx = torch.full((3,), 777, dtype=torch.int32, device=dev)
    ops.planes_cut([((Wa, Wb), pr, ops.PLANES_ROWS, [(slots[0], slots[1])], kx[0:1]),
                    ((Wa, Wb), pc, ops.PLANES_COLS, [(slots[0], slots[1]), (slots[2], slots[3])], kx[1:2]),
                    ((Va, Vb), pv, ops.PLANES_COLS, [(slots[2], slots[3]), (slots[0], slots[1])], kx[2:3])])
    torch.cuda.synchronize()
    f32 = np.float32
    bw = f32(float(Wa.abs().max())) * f32(float(Wb.abs().max()))
    bv = f32(float(Va.abs().max())) * f32(float(Vb.abs().max()))
    k_own = _kexp(bw)
    k_grp = min(k_own, _kexp(bv))
    assert kx.tolist() == [k_own, k_grp, k_grp]
    P = (Wa * Wb).cpu().numpy()       # the fp32 product
    Q = (Va * Vb).cpu().numpy()
    assert np.array_equal(pr.cpu().numpy().view(np.uint32), _planes_ref(P, k_own, 0))
    assert np.array_equal(pc.cpu().numpy().view(np.uint32), _planes_ref(P, k_grp, 1))
    assert np.array_equal(pv.cpu().numpy().view(np.uint32), _planes_ref(Q, k_grp, 1))
    # the same planes as the cut of the stored product under the same exponent: force it by giving the product's cut a
    # slot that holds the bound
    bound = ops.amax_slots(1, dev)
    bound[0, 0] = int(np.float32(bw).view(np.int32))
    Pm = Wa * Wb
    pm = torch.zeros(K, N, dtype=torch.int32, device=dev)
    ops.planes_cut([(Pm, pm, ops.PLANES_ROWS, [bound[0]], kx[0:1])])
    torch.cuda.synchronize()
    assert torch.equal(pm, pr)
    with pytest.raises(L.MMLError):   # factors of different shapes
        ops.planes_cut([((Wa, Vb), pr, ops.PLANES_ROWS, [(slots[0], slots[3])], kx[0:1])])


def _star_layer(torch, ops, dev, g, K, N, scale_w=1.0):
    ws = (torch.randn(K, N, generator=g) * scale_w).to(dev)
    wsh = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)
    return ws, wsh


@pytest.mark.parametrize("mode", [4, 2])
@pytest.mark.parametrize("M,K,Ns", [(1000, 304 - 64, [256]), (8192 + 77, 256, [128, 128, 128]), (515, 128, [64, 64])])
def test_star_linear_fwd(env, mode, M, K, Ns):
    """y_d = relu(x (W_spec,d (.) W_shared) + b) for the domains of one layer, weights in the [K, N] layout, from planes cut
    out of the factors -- float64, and the same bits as the launch that reads the materialised product."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(mode)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(dev)
    nslot = 1 + 3 * len(Ns)
    slots = ops.amax_slots(nslot, dev)
    ops.amax_batch([(A, slots[0])])
    probs, items, mats = [], [], []
    for i, N in enumerate(Ns):
        ws, wsh = _star_layer(torch, ops, dev, g, K, N, scale_w=1.0 + i)
        weff = ws * wsh
        b = torch.randn(N, generator=g).to(dev)
        sa, sb, sw = slots[1 + 3 * i], slots[2 + 3 * i], slots[3 + 3 * i]
        ops.amax_batch([(ws, sa), (wsh, sb)])
        # the magnitude the GEMM is told for the weight: the factors' bound (what the planes were scaled against)
        sw[0] = int((np.float32(float(ws.abs().max())) * np.float32(float(wsh.abs().max()))).view(np.int32))
        planes = torch.zeros(K, N, dtype=torch.int32, device=dev)
        kexp = torch.zeros(1, dtype=torch.int32, device=dev)
        items.append(((ws, wsh), planes, ops.PLANES_COLS, [(sa, sb)], kexp))   # [K, N]: the reduction runs down the rows
        probs.append(dict(A=A, W=weff, bias=b, act=L.ACT_RELU, w_kn=1, amax_a=slots[0], amax_w=sw, w_planes=planes,
                          w_kexp=kexp, C=torch.full((M, N), float("nan"), device=dev)))
        mats.append(weff)
    ops.planes_cut(items)
    arr = ops.make_fwd_descs(probs)
    L.check(lib.mml_star_linear_fwd(arr, len(probs), ops._stream()), "mml_star_linear_fwd")
    torch.cuda.synchronize()
    name = lib.mml_gemm_last_kernel().decode()
    assert "gemm_pipe_kernel" in name or name.startswith("gemm_ws_kernel"), name  # (8 269 x 256 -> 128: the weight-stationary kernel)
    outs = [p["C"].clone() for p in probs]
    for p, C in zip(probs, outs):
        ref = torch.relu(A.double() @ p["W"].double() + p["bias"].double())
        assert rel(C, ref) < RTOL
    # the same launch from the stored product (in-kernel cut against the same bound): same bits
    for p in probs:
        p.pop("w_planes"), p.pop("w_kexp")
        p["C"] = torch.full_like(p["C"], float("nan"))
    ops.gemm_fwd(probs)
    torch.cuda.synchronize()
    for p, C in zip(probs, outs):
        assert torch.equal(p["C"], C)
    # a layer in nn.Linear layout, or without planes, is refused by the named entry point
    plain = ops.make_fwd_descs([dict(A=A, W=mats[0], bias=None, C=outs[0], act=L.ACT_NONE, w_kn=1)])
    assert lib.mml_star_linear_fwd(plain, 1, ops._stream()) != 0


@pytest.mark.parametrize("mode", [4, 2])
@pytest.mark.parametrize("M,K,srcNs", [(1000, 240, [256]), (8192 + 5, 256, [128, 128]), (515, 64, [64, 32, 16])])
def test_star_linear_bwd(env, mode, M, K, srcNs):
    """dx = sum_d dy_d (W_spec,d (.) W_shared)^T * relu'(x) with every source's weight as product planes of ONE exponent."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(mode)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(M + K + 1)
    n = len(srcNs)
    slots = ops.amax_slots(4 * n, dev)
    Y = torch.relu(torch.randn(M, K, generator=g)).to(dev)
    kexp = torch.zeros(1, dtype=torch.int32, device=dev)
    srcs, pairs, items, v = [], [], [], torch.zeros(M, K, dtype=torch.float64, device=dev)
    facs = []
    for i, N in enumerate(srcNs):
        ws, wsh = _star_layer(torch, ops, dev, g, K, N, scale_w=0.5 + i)
        dC = torch.randn(M, N, generator=g).to(dev)
        sa, sb, sw, sd = slots[4 * i], slots[4 * i + 1], slots[4 * i + 2], slots[4 * i + 3]
        ops.amax_batch([(ws, sa), (wsh, sb), (dC, sd)])
        sw[0] = int((np.float32(float(ws.abs().max())) * np.float32(float(wsh.abs().max()))).view(np.int32))
        pairs.append((sa, sb))
        facs.append((ws, wsh, dC, sw, sd))
        v += dC.double() @ (ws * wsh).double().t()
    for ws, wsh, dC, sw, sd in facs:
        planes = torch.zeros(ws.shape, dtype=torch.int32, device=dev)
        items.append(((ws, wsh), planes, ops.PLANES_ROWS, pairs, kexp))  # [K, N] read by the input gradient: along a row
        srcs.append((dC, ws * wsh, 1, sd, sw, planes, kexp))
    ops.planes_cut(items)
    dA = torch.full((M, K), float("nan"), device=dev)
    arr = ops.make_dgrad_descs([dict(dA=dA, Y=Y, act=L.ACT_RELU, srcs=srcs)])
    L.check(lib.mml_star_linear_bwd(arr, 1, ops._stream()), "mml_star_linear_bwd")
    torch.cuda.synchronize()
    assert "gemm_pipe_kernel" in lib.mml_gemm_last_kernel().decode()
    ref = v * (Y > 0).double()
    assert rel(dA, ref) < RTOL
    plain = ops.make_dgrad_descs([dict(dA=dA, Y=Y, act=L.ACT_RELU, srcs=[s[:5] for s in srcs])])
    assert lib.mml_star_linear_bwd(plain, 1, ops._stream()) != 0


def test_star_model_steps_from_product_planes(env, monkeypatch):
    """Amazon-8 STAR at a batch where the two-plane arithmetic is on: the plan cuts the derived weights from their
    factors (no [K, N] weight is cut inside a GEMM), and one fused step gives the loss and parameters of the same step
    with pre-cut planes off to fp32 noise."""
    torch, L, ops, lib = env
    from mmlrec_amd import workloads as W
    dev = torch.device("cuda:0")
    B = 8192
    monkeypatch.setenv("MMLREC_AMAX", "1")

    def one(planes):
        monkeypatch.setenv("MMLREC_GEMM_PLANES", "1" if planes else "0")
        monkeypatch.setenv("MMLREC_STAR_PLANES", "1" if planes else "0")
        torch.manual_seed(3)
        model, cfg, vocab, dense = W.build_model("star_amazon", dev)
        T = W.num_tasks(cfg)
        X, y = W.synth_batch(vocab, len(dense), B, T, seed=5)
        model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], ["auc"])
        model.train()
        step = model.train_step_runner(B, use_graph=False)
        step.plan.X.copy_(X.to(dev))
        step.plan.y.copy_(y.to(dev))
        step.run()
        torch.cuda.synchronize()
        n_prod = sum(isinstance(it[0], tuple) for it in step.plan.planes_items)
        return float(step.plan.loss.item()), {k: v.clone() for k, v in model.state_dict().items()}, n_prod

    loss1, sd1, n1 = one(True)
    loss0, sd0, n0 = one(False)
    assert n1 > 0 and n0 == 0, (n1, n0)
    assert abs(loss1 - loss0) / abs(loss0) < 1e-5
    for k in sd0:
        a, b = sd1[k].double(), sd0[k].double()
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-30) + 1e-7, k
