"""K7: the PepNet gate products fused into the GEMM epilogues (mml_pep_gate_fwd / mml_pep_gate_bwd = the mul / prod and
gate-mode fields of the grouped GEMM descriptors; reference model/pepnet.py:31-32, :72-78, :139-140) against float64."""
import numpy as np
import pytest

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


@pytest.mark.parametrize("mode", [3, 2])
@pytest.mark.parametrize("M,K,Ns", [
    (1000, 64, [64, 64, 64, 64]),      # four tasks' gates of one layer: 128 x 64 tiles
    (8192 + 5, 80, [256, 256]),        # 128 x 128 tiles, ragged batch
    (515, 128, [128, 64, 4]),          # a 4-column problem in the group
])
def test_pep_gate_fwd(env, mode, M, K, Ns):
    """C = 2 sigmoid(A W^T + b) and prod = C * mul from one launch."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(mode)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(dev)
    probs = []
    for i, N in enumerate(Ns):
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        mul = torch.randn(M, N, generator=g).to(dev)
        probs.append(dict(A=A, W=W, bias=b, C=torch.full((M, N), float("nan"), device=dev), act=L.ACT_SIGMOID2,
                          mul=mul, prod=torch.full((M, N), float("nan"), device=dev),
                          amax_prod=ops.amax_slots(1, dev)[0]))
    if mode == 2:
        cache = ops._measured([p["A"] for p in probs] + [p["W"] for p in probs], {})
        key = lambda t: (t.data_ptr(), tuple(t.shape), t.stride(0))  # noqa: E731
        for p in probs:
            p.update(amax_a=cache[key(p["A"])], amax_w=cache[key(p["W"])])
    arr = ops.make_fwd_descs(probs)
    L.check(lib.mml_pep_gate_fwd(arr, len(probs), ops._stream()), "mml_pep_gate_fwd")
    torch.cuda.synchronize()
    assert "gemm_pipe_kernel" in lib.mml_gemm_last_kernel().decode()
    for p in probs:
        z = A.double() @ p["W"].double().t() + p["bias"].double()
        c = 2 * torch.sigmoid(z)
        assert rel(p["C"], c) < RTOL
        assert rel(p["prod"], c * p["mul"].double()) < RTOL
        am = float(torch.max(p["amax_prod"].view(torch.float32)))
        assert am >= float(p["prod"].abs().max()) and am <= float(p["prod"].abs().max()) * (1 + 1e-6)
    # a descriptor without mul / prod is refused by the named entry point
    plain = ops.make_fwd_descs([dict(A=A, W=probs[0]["W"], bias=None, C=probs[0]["C"], act=L.ACT_NONE)])
    assert lib.mml_pep_gate_fwd(plain, 1, ops._stream()) != 0


@pytest.mark.parametrize("mode", [3, 2])
@pytest.mark.parametrize("M,K,srcNs,act_h,act_g,acc_h,acc_g", [
    (1000, 64, [256], "none", "sigmoid2", 1, 0),      # first PPNet layer: h = the gated input (shared: accumulate)
    (8192 + 5, 256, [128], "relu", "sigmoid2", 0, 0),   # second layer: h = relu output of the layer before
    (515, 128, [64, 32], "relu", "none", 0, 1),         # two sources
])
def test_pep_gate_bwd(env, mode, M, K, srcNs, act_h, act_g, acc_h, acc_g):
    """v = sum_s dC_s W_s is not stored: d_h (+)= v g act_h'(h), d_g (+)= v h act_g'(g)."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(mode)
    dev = torch.device("cuda:0")
    acts = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "sigmoid2": L.ACT_SIGMOID2}
    g = torch.Generator(device="cpu").manual_seed(M + K)
    srcs, v = [], torch.zeros(M, K, dtype=torch.float64, device=dev)
    for N in srcNs:
        dC = torch.randn(M, N, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) / N ** 0.5).to(dev)
        srcs.append((dC, W, 0))
        v += dC.double() @ W.double()
    h = torch.randn(M, K, generator=g).to(dev)
    if act_h == "relu":
        h = torch.relu(h)
    gt = (2 * torch.sigmoid(torch.randn(M, K, generator=g))).to(dev)
    old_h, old_g = torch.randn(M, K, generator=g).to(dev), torch.randn(M, K, generator=g).to(dev)
    dh = old_h.clone() if acc_h else torch.full((M, K), float("nan"), device=dev)
    dg = old_g.clone() if acc_g else torch.full((M, K), float("nan"), device=dev)
    ref_h = v * gt.double() * ((h > 0).double() if act_h == "relu" else 1.0) + (old_h.double() if acc_h else 0.0)
    gd = gt.double()
    ref_g = v * h.double() * (gd * (1 - gd / 2) if act_g == "sigmoid2" else 1.0) + (old_g.double() if acc_g else 0.0)
    slots = ops.amax_slots(2, dev)
    prob = dict(dA=None, Y=None, act=L.ACT_NONE, srcs=srcs,
                gate=dict(h=h, g=gt, dh=dh, dg=dg, act_h=acts[act_h], act_g=acts[act_g], acc_h=acc_h, acc_g=acc_g,
                          amax_dh=slots[0], amax_dg=slots[1]))
    if mode == 2:
        cache = ops._measured([t for sr in srcs for t in sr[:2]], {})
        key = lambda t: (t.data_ptr(), tuple(t.shape), t.stride(0))  # noqa: E731
        prob["srcs"] = [(dC, W, kn, cache[key(dC)], cache[key(W)]) for dC, W, kn in srcs]
    arr = ops.make_dgrad_descs([prob])
    L.check(lib.mml_pep_gate_bwd(arr, 1, ops._stream()), "mml_pep_gate_bwd")
    torch.cuda.synchronize()
    assert "gemm_pipe_kernel" in lib.mml_gemm_last_kernel().decode()
    assert rel(dh, ref_h) < RTOL
    assert rel(dg, ref_g) < RTOL
    for t, s in ((dh, slots[0]), (dg, slots[1])):
        am = float(torch.max(s.view(torch.float32)))
        assert am >= float(t.abs().max()) and am <= float(t.abs().max()) * (1 + 1e-6)


def test_pepnet_with_fused_gate_products_matches_the_reference(env, monkeypatch):
    """The whole PepNet model with MMLREC_PEP_FUSE=1 (products of the hidden PPNet layers from the GEMM epilogues, their
    backward in the gate-mode input-gradient launches) against the reference-made fixture: loss, every gradient, and
    three fused Adagrad steps."""
    torch, L, ops, lib = env
    from conftest import load_golden
    from test_models_gpu import RTOL as MT, build, load_state, rel as mrel
    monkeypatch.setenv("MMLREC_PEP_FUSE", "1")
    g = load_golden("pepnet_amazon")
    model, cfg = build(g)
    load_state(model, g)
    model.train()
    X, y = torch.from_numpy(g["X0"]).cuda(), torch.from_numpy(g["y0"]).cuda()
    yp = model(X)
    plan = model._get_plan(X.shape[0], True, False)
    assert any(q.get("mul") is not None for op in plan.ops if hasattr(op, "p") and isinstance(op.p, list)
               for q in op.p if isinstance(q, dict)), "the fused path did not engage"
    bce = torch.nn.functional.binary_cross_entropy
    loss = sum(bce(yp[:, i], y[:, i], reduction="sum") for i in range(yp.shape[1]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) / float(g["loss"]) < MT
    for n, p in model.named_parameters():
        if "grad/" + n in g.files:
            assert mrel(p.grad.cpu().numpy(), g["grad/" + n]) < MT, n
    # fused steps
    model2, cfg2 = build(g, table_update="sparse_rows")
    load_state(model2, g)
    model2.compile("adagrad", cfg2["optim_config"]["loss"], ["auc"])
    model2.train()
    losses = []
    for i in range(3):
        step = model2.train_step_runner(64, use_graph=True)
        step.plan.X.copy_(torch.from_numpy(g[f"X{i}"]).cuda())
        step.plan.y.copy_(torch.from_numpy(g[f"y{i}"]).cuda())
        step.run()
        losses.append(float(step.plan.loss.item()))
    assert np.allclose(losses, g["adagrad_losses"], rtol=MT), losses
    sd = model2.state_dict()
    lr = cfg2["optim_config"]["lr"]
    for k in sd:
        ref = g[f"adagrad3/{k}"].astype(np.float64)
        dv = np.abs(sd[k].cpu().numpy().astype(np.float64) - ref)
        assert (dv > MT * max(np.abs(ref).max(), 1e-30)).mean() < 2e-3, k
        assert dv.max() <= 2.5 * lr * 3, k
