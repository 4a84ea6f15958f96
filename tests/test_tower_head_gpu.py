"""K5' (csrc/tower_head.hip, round 6): last tower layer + head + summed BCE + their backward in one launch (reference
model/mmoe.py:93-108, model/utils.py:146-161, :242-248, model/basemodel.py:294-296) against float64, and against the three
launches it replaces (mml_gemm_grouped_fwd -> mml_head_bce_fwd_bwd -> mml_gemm_grouped_dgrad)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def env():
    import torch
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib as L, ops
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    lib = L.load()
    mode0 = lib.mml_gemm_get_mode()
    lib.mml_gemm_set_mode(4)
    yield torch, L, ops, lib
    lib.mml_gemm_set_mode(mode0)


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def build(torch, L, ops, M, K, N, T, masked, seed, scale=1.0):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    y = (torch.rand(M, T, generator=g) < 0.4).float().to(dev)
    mask = (torch.rand(M, 2, generator=g) < 0.6).float().to(dev) if masked else None
    tasks, items = [], []
    for t in range(T):
        A = (torch.randn(M, K, generator=g) * scale).to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5 / scale).to(dev)
        b1 = (torch.randn(N, generator=g) * 0.1).to(dev)
        w = (torch.randn(N, generator=g) * 0.5 / N ** 0.5).to(dev)
        hb = torch.randn(1, generator=g).to(dev)
        slots = ops.amax_slots(4, dev)
        ops.amax_batch([(A, slots[0]), (W, slots[1])])
        pf, pb = torch.zeros(N, K, dtype=torch.int32, device=dev), torch.zeros(N, K, dtype=torch.int32, device=dev)
        kf, kb = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        items += [(W, pf, ops.PLANES_ROWS, [slots[1]], kf), (W, pb, ops.PLANES_COLS, [slots[1]], kb)]
        tasks.append(dict(A=A, W=W, amax_a=slots[0], amax_w=slots[1], planes_fwd=pf, kexp_fwd=kf, planes_bwd=pb, kexp_bwd=kb,
                          bias1=b1, w=w, hbias=hb, dH=torch.full((M, N), float("nan"), device=dev),
                          dA=torch.full((M, K), float("nan"), device=dev), dw=torch.full((N,), float("nan"), device=dev),
                          dhbias=torch.full((1,), float("nan"), device=dev), amax_dH=slots[2], amax_dA=slots[3],
                          mask_col=(t % 2 if masked else -1), head=t))
    ops.planes_cut(items)
    return y, mask, tasks


def reference(torch, y, mask, q, t):
    A, W = q["A"].double(), q["W"].double()
    h = torch.relu(A @ W.t() + q["bias1"].double())
    logit = (h @ q["w"].double() + q["hbias"].double()).float()           # the kernel forms the probability in fp32
    p = (1.0 / (1.0 + torch.exp(-logit))).double()
    m = mask[:, q["mask_col"]].double() if q["mask_col"] >= 0 else torch.ones_like(p)
    pm = (p * m).float().double()
    yy = y[:, t].double()
    lp = torch.clamp(torch.log(pm.float()), min=-100).double()
    l1p = torch.clamp(torch.log1p(-pm.float()), min=-100).double()
    loss = float(-(yy * lp + (1 - yy) * l1p).sum())
    dlogit = (pm - yy) / torch.clamp(pm * (1 - pm), min=1e-12) * m * p * (1 - p)
    dH = dlogit[:, None] * q["w"].double()[None, :] * (h > 0)
    return pm, loss, dH, dH @ W, dlogit @ h, float(dlogit.sum())


@pytest.mark.parametrize("M,K,T,masked,scale", [(65536, 128, 2, True, 1.0), (8192 + 77, 128, 2, False, 1e-3), (333, 128, 4, True, 1.0),
                                                (16384, 64, 3, False, 30.0), (32, 128, 1, False, 1.0)])
def test_tower_head_matches_float64(env, M, K, T, masked, scale):
    torch, L, ops, lib = env
    dev = torch.device("cuda:0")
    N = 64
    y, mask, tasks = build(torch, L, ops, M, K, N, T, masked, seed=M + K + T, scale=scale)
    prob = torch.full((M, T), float("nan"), device=dev)
    loss = torch.full((1,), float("nan"), device=dev)
    grp = ops.make_tower_head_group(tasks, prob, y, mask=mask, loss=loss)
    assert lib.mml_tower_head_serves(grp) == 1
    ops.tower_head_fwd_bwd(grp, dev)
    torch.cuda.synchronize()
    total = 0.0
    for t, q in enumerate(tasks):
        pm, ls, dH, dA, dw, db = reference(torch, y, mask, q, t)
        total += ls
        assert rel(prob[:, t], pm) < 1e-5
        assert rel(q["dH"], dH) < 1e-5, (t, rel(q["dH"], dH))
        assert rel(q["dA"], dA) < 1e-5, (t, rel(q["dA"], dA))
        assert rel(q["dw"], dw) < 2e-5
        assert abs(float(q["dhbias"]) - db) < 2e-5 * max(abs(db), 1.0)
        for buf, slot in ((q["dH"], q["amax_dH"]), (q["dA"], q["amax_dA"])):
            am = float(torch.max(slot.view(torch.float32)))
            assert am >= float(buf.abs().max()) and am <= float(buf.abs().max()) * (1 + 1e-6)
    assert abs(float(loss) - total) / total < 1e-4
    # the two-launch form: the same bits
    keep = [(q["dw"].clone(), q["dhbias"].clone(), q["dH"].clone(), q["dA"].clone()) for q in tasks]
    for q in tasks:
        q["dw"].fill_(float("nan"))
        q["dhbias"].fill_(float("nan"))
    loss2 = torch.full((1,), float("nan"), device=dev)
    grp2 = ops.make_tower_head_group(tasks, prob, y, mask=mask, loss=loss2)
    ops.tower_head_fwd_bwd(grp2, dev, phases=True)
    assert float(loss2) == float(loss)
    for q, (w_, b_, h_, a_) in zip(tasks, keep):
        assert torch.equal(q["dw"], w_) and torch.equal(q["dhbias"], b_) and torch.equal(q["dH"], h_) and torch.equal(q["dA"], a_)


def test_tower_head_against_the_three_launches_it_replaces(env):
    torch, L, ops, lib = env
    dev = torch.device("cuda:0")
    M, K, N, T = 16384, 128, 64, 2
    y, mask, tasks = build(torch, L, ops, M, K, N, T, True, seed=5)
    prob = torch.empty(M, T, device=dev)
    loss = torch.zeros(1, device=dev)
    ops.tower_head_fwd_bwd(ops.make_tower_head_group(tasks, prob, y, mask=mask, loss=loss), dev)
    # unfused: tower forward (weight-stationary kernel), head + BCE, tower input gradient
    hs = [torch.empty(M, N, device=dev) for _ in tasks]
    ops.gemm_fwd([dict(A=q["A"], W=q["W"], bias=q["bias1"], C=h_, act=L.ACT_RELU, amax_a=q["amax_a"], amax_w=q["amax_w"],
                       w_planes=q["planes_fwd"], w_kexp=q["kexp_fwd"]) for q, h_ in zip(tasks, hs)])
    prob2, loss2 = torch.empty(M, T, device=dev), torch.zeros(1, device=dev)
    heads = [dict(Hin=h_, w=q["w"], bias=q["hbias"], dH=torch.empty(M, N, device=dev), dw=torch.empty(N, device=dev),
                  dbias=torch.empty(1, device=dev), h_relu=1, mask_col=q["mask_col"]) for q, h_ in zip(tasks, hs)]
    ops.head_bce_fwd_bwd(ops.make_head_group(heads, prob2, y=y, mask=mask, loss=loss2), dev)
    assert float((prob - prob2).abs().max()) < 1e-6
    assert abs(float(loss) - float(loss2)) / float(loss2) < 1e-5
    for q, hd in zip(tasks, heads):
        assert rel(q["dH"], hd["dH"].double()) < 1e-5
        assert rel(q["dw"], hd["dw"].double()) < 1e-5
        ref = hd["dH"].double() @ q["W"].double()
        assert rel(q["dA"], ref) < 1e-5


def test_tower_head_refuses_what_it_does_not_serve(env):
    torch, L, ops, lib = env
    dev = torch.device("cuda:0")
    y, mask, tasks = build(torch, L, ops, 256, 128, 64, 1, False, seed=1)
    tasks[0]["dH"] = torch.empty(256, 128, device=dev)      # a 128-wide tower: not instantiated
    prob = torch.empty(256, 1, device=dev)
    grp = ops.make_tower_head_group(tasks, prob, y)
    assert lib.mml_tower_head_serves(grp) == 0
    with pytest.raises(L.MMLError):
        ops.tower_head_fwd_bwd(grp, dev)
