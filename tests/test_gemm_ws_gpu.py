"""The weight-stationary streaming kernel (csrc/gemm_ws.hip) against float64 and, bit for bit, against the tile kernel it
replaces for the second expert layers, the towers and their input gradients (reference model/mmoe.py:69-119,
model/utils.py:146-161; STAR's [K, N] layout: model/utils.py:171-218)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 2e-6  # max-norm, against float64 (the two-plane fp16 arithmetic measures 3.3e-7)


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
    lib.mml_gemm_set_ws(1)


def unpack(words, n):
    w = words.cpu().numpy().view(np.uint32)
    return ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(w.shape[0], -1)[:, :n].astype(bool)


def fwd_launch(torch, L, ops, M, K, N, nprob, kn=False, acts=None, bias=True, seed=0, scale=1.0):
    """K, N: one value for all problems, or a list with one value per problem."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    slots = ops.amax_slots(3 * nprob, dev)
    probs, items = [], []
    Ks = K if isinstance(K, (list, tuple)) else [K] * nprob
    Ns = N if isinstance(N, (list, tuple)) else [N] * nprob
    for i in range(nprob):
        K, N = Ks[i], Ns[i]
        A = (torch.randn(M, K, generator=g) * scale * (1 + i)).to(dev)
        W = (torch.randn(*((K, N) if kn else (N, K)), generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev) if (bias and i % 3 != 2) else None
        sa, sw, so = slots[3 * i], slots[3 * i + 1], slots[3 * i + 2]
        ops.amax_batch([(A, sa), (W, sw)])
        planes = torch.zeros(W.shape, dtype=torch.int32, device=dev)
        kexp = torch.zeros(1, dtype=torch.int32, device=dev)
        items.append((W, planes, ops.PLANES_COLS if kn else ops.PLANES_ROWS, [sw], kexp))
        probs.append(dict(A=A, W=W, bias=b, act=(acts[i] if acts else L.ACT_RELU), w_kn=int(kn), amax_a=sa, amax_w=sw,
                          w_planes=planes, w_kexp=kexp, amax_out=so))
    ops.planes_cut(items)
    return probs


def run_fwd(torch, ops, lib, probs, ws, masks):
    dev = torch.device("cuda:0")
    lib.mml_gemm_set_ws(1 if ws else 0)
    for p in probs:
        M = p["A"].shape[0]
        N = p["W"].shape[1] if p["w_kn"] else p["W"].shape[0]
        p["C"] = torch.full((M, N), float("nan"), device=dev)
        if masks:
            p["mask"] = torch.full((M, (N + 31) // 32), 0x55555555, dtype=torch.int32, device=dev)
        else:
            p.pop("mask", None)
        p["amax_out"].zero_()
    ops.gemm_fwd(probs)
    torch.cuda.synchronize()
    name = lib.mml_gemm_last_kernel().decode()
    return name, [(p["C"].clone(), p["mask"].clone() if masks else None, p["amax_out"].clone()) for p in probs]


@pytest.mark.parametrize("M,K,N,nprob,kn,masks", [
    (65536, 256, 128, 4, False, True),     # AE-30's second expert layer at the benchmark batch
    (65536, 128, 64, 2, False, True),      # its towers
    (8192 + 77, 256, 128, 3, False, False),  # ragged batch, three problems (85 workgroups each, one idle), inference
    (8192, 64, 128, 1, False, True),       # one group of four k-steps
    (16384 + 5, 192, 64, 5, True, True),   # STAR's [K, N] layout (planes cut down the rows), five domains
    (40000, 128, 128, 16, False, False),   # sixteen problems
    (8192 + 64, 128, 256, 2, False, True),  # 256 output columns: eight sub-tiles of accumulators per wave
])
def test_ws_fwd_matches_float64_and_the_tile_kernel(env, M, K, N, nprob, kn, masks):
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    probs = fwd_launch(torch, L, ops, M, K, N, nprob, kn=kn, seed=M + K)
    name_w, out_w = run_fwd(torch, ops, lib, probs, True, masks)
    assert name_w.startswith("gemm_ws_kernel"), name_w
    name_t, out_t = run_fwd(torch, ops, lib, probs, False, masks)
    assert "gemm_pipe_kernel" in name_t, name_t
    for p, (C, mk, am), (Ct, mkt, amt) in zip(probs, out_w, out_t):
        Wd = p["W"].double() if kn else p["W"].double().t()
        z = p["A"].double() @ Wd
        if p["bias"] is not None:
            z = z + p["bias"].double()
        ref = torch.relu(z)
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        assert err < RTOL, err
        assert torch.equal(C, Ct)                       # same planes, same product and k order: same bits
        if masks:
            assert np.array_equal(unpack(mk, N), (C > 0).cpu().numpy())
            assert torch.equal(mk, mkt)
        amax = float(torch.max(am.view(torch.float32)))
        assert amax >= float(C.abs().max()) and amax <= float(C.abs().max()) * (1 + 1e-6)


def test_ws_fwd_linear_layers_and_scales(env):
    """No activation (negative outputs, magnitude of |v|), operands far outside fp16's own range."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    for scale in (3e-9, 1.0, 2e20):
        probs = fwd_launch(torch, L, ops, 8192 + 33, 128, 128, 2, acts=[L.ACT_NONE, L.ACT_RELU], seed=3, scale=scale)
        name, out = run_fwd(torch, ops, lib, probs, True, False)
        assert name.startswith("gemm_ws_kernel")
        for p, (C, _, am) in zip(probs, out):
            z = p["A"].double() @ p["W"].double().t()
            if p["bias"] is not None:
                z = z + p["bias"].double()
            ref = torch.relu(z) if p["act"] == L.ACT_RELU else z
            err = float((C.double() - ref).abs().max() / ref.abs().max())
            assert err < RTOL, (scale, err)
            amax = float(torch.max(am.view(torch.float32)))
            assert amax >= float(C.abs().max()) and amax <= float(C.abs().max()) * (1 + 1e-6)


def test_ws_fwd_mixed_widths_and_gate_activations(env):
    """PepNet's / PLE's sibling layers: 128- and 64-wide problems with different reductions in ONE call (served as one
    launch per width), and the gate networks' sigmoid / 2 sigmoid outputs -- float64, and the tile kernel's bits."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    M = 8192 + 32
    Ns = [64, 128, 128, 64, 256, 128, 64]
    Ks = [64, 64, 128, 192, 128, 80, 160]          # (80, 160: groups of five k-steps -- PepNet's 72 -> 80 padded inputs)
    acts = [L.ACT_RELU, L.ACT_SIGMOID2, L.ACT_SIGMOID, L.ACT_NONE, L.ACT_SIGMOID2, L.ACT_RELU, L.ACT_RELU]
    probs = fwd_launch(torch, L, ops, M, Ks, Ns, 7, acts=acts, seed=21)
    name_w, out_w = run_fwd(torch, ops, lib, probs, True, False)
    assert name_w.startswith("gemm_ws_kernel"), name_w
    name_t, out_t = run_fwd(torch, ops, lib, probs, False, False)
    assert "gemm_pipe_kernel" in name_t
    for p, (C, _, am), (Ct, _, _) in zip(probs, out_w, out_t):
        z = p["A"].double() @ p["W"].double().t()
        if p["bias"] is not None:
            z = z + p["bias"].double()
        ref = {L.ACT_RELU: torch.relu(z), L.ACT_NONE: z, L.ACT_SIGMOID: torch.sigmoid(z),
               L.ACT_SIGMOID2: 2 * torch.sigmoid(z)}[p["act"]]
        assert float((C.double() - ref).abs().max() / ref.abs().max()) < RTOL
        assert torch.equal(C, Ct)
        amax = float(torch.max(am.view(torch.float32)))
        assert amax >= float(C.abs().max()) and amax <= float(C.abs().max()) * (1 + 1e-6)


def test_launches_the_ws_kernel_does_not_serve_fall_back(env):
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    # a weight beyond the LDS (twice), a reduction that is neither a multiple of 64 nor of 80, a batch below the threshold, an output
    # width that is not instantiated
    for M, K, N, acts in ((8192, 512, 128, None), (8192, 208, 128, None), (4096, 256, 128, None),
                          (8192, 256, 256, None), (8192, 128, 192, None)):
        probs = fwd_launch(torch, L, ops, M, K, N, 1, acts=acts)
        name, out = run_fwd(torch, ops, lib, probs, True, False)
        assert not name.startswith("gemm_ws_kernel"), (M, K, N, name)
        z = probs[0]["A"].double() @ probs[0]["W"].double().t() + probs[0]["bias"].double()
        ref = torch.relu(z)
        assert float((out[0][0].double() - ref).abs().max() / ref.abs().max()) < RTOL


def dgrad_launch(torch, L, ops, M, Nred, K, nprob, kn=False, relu=True, seed=0):
    """dA [M, K] = dC [M, Nred] W, W = [Nred, K] (nn.Linear) or [K, Nred] (kn)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    slots = ops.amax_slots(3 * nprob, dev)
    probs, items = [], []
    for i in range(nprob):
        dC = (torch.randn(M, Nred, generator=g) * (1 + i)).to(dev)
        W = (torch.randn(*((K, Nred) if kn else (Nred, K)), generator=g) / Nred ** 0.5).to(dev)
        Y = torch.relu(torch.randn(M, K, generator=g)).to(dev)
        sd, sw, so = slots[3 * i], slots[3 * i + 1], slots[3 * i + 2]
        ops.amax_batch([(dC, sd), (W, sw)])
        planes = torch.zeros(W.shape, dtype=torch.int32, device=dev)
        kexp = torch.zeros(1, dtype=torch.int32, device=dev)
        items.append((W, planes, ops.PLANES_ROWS if kn else ops.PLANES_COLS, [sw], kexp))
        bits = (Y > 0).cpu().numpy()
        words = np.packbits(bits.reshape(M, K // 32, 32), axis=2, bitorder="little").view(np.uint32).reshape(M, K // 32)
        mask = torch.from_numpy(words.view(np.int32).copy()).to(dev)
        probs.append(dict(Y=Y, act=L.ACT_RELU if relu else L.ACT_NONE, mask=mask if relu else None, amax_out=so,
                          srcs=[(dC, W, int(kn), sd, sw, planes, kexp)]))
    ops.planes_cut(items)
    return probs


def run_dgrad(torch, ops, lib, probs, ws, accumulate):
    dev = torch.device("cuda:0")
    lib.mml_gemm_set_ws(1 if ws else 0)
    g = torch.Generator(device="cpu").manual_seed(77)
    olds = []
    for p in probs:
        M, K = p["Y"].shape
        old = torch.randn(M, K, generator=g).to(dev)
        olds.append(old)
        p["dA"] = old.clone() if accumulate else torch.full((M, K), float("nan"), device=dev)
        p["accumulate"] = int(accumulate)
        p["amax_out"].zero_()
    ops.gemm_dgrad(probs)
    torch.cuda.synchronize()
    name = lib.mml_gemm_last_kernel().decode()
    return name, olds, [(p["dA"].clone(), p["amax_out"].clone()) for p in probs]


@pytest.mark.parametrize("M,Nred,K,nprob,kn,relu,acc", [
    (65536, 128, 256, 4, False, True, False),    # AE-30: input gradient of the second expert layer (eight sub-tiles per wave)
    (65536, 64, 128, 2, False, False, False),    # of the towers (no derivative: the mixed expert outputs)
    (8192 + 77, 128, 256, 3, False, True, True),   # ragged, accumulating
    (16384 + 5, 192, 128, 5, True, True, False),   # STAR's [K, N] layout
    (8192, 64, 256, 1, False, False, True),
    (8192 + 96, 128, 64, 4, False, True, False),   # 64 output columns (PepNet's gate networks)
    (8192, 80, 128, 2, False, True, False),        # a reduction of 80: one group of five k-steps
])
def test_ws_dgrad_matches_float64_and_the_tile_kernel(env, M, Nred, K, nprob, kn, relu, acc):
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    probs = dgrad_launch(torch, L, ops, M, Nred, K, nprob, kn=kn, relu=relu, seed=M + K)
    name_w, olds, out_w = run_dgrad(torch, ops, lib, probs, True, acc)
    assert name_w.startswith("gemm_ws_kernel"), name_w
    name_t, _, out_t = run_dgrad(torch, ops, lib, probs, False, acc)
    assert "gemm_pipe_kernel" in name_t, name_t
    for p, old, (dA, am), (dAt, amt) in zip(probs, olds, out_w, out_t):
        dC, W = p["srcs"][0][:2]
        v = dC.double() @ (W.double().t() if kn else W.double())
        if relu:
            v = v * (p["Y"] > 0).double()
        if acc:
            v = v + old.double()
        err = float((dA.double() - v).abs().max() / v.abs().max())
        assert err < RTOL, err
        assert torch.equal(dA, dAt)
        amax = float(torch.max(am.view(torch.float32)))
        assert amax >= float(dA.abs().max()) and amax <= float(dA.abs().max()) * (1 + 1e-6)


def test_dgrad_launches_the_ws_kernel_does_not_serve_fall_back(env):
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    # two sources; a derivative taken from the stored outputs (no sign mask); an output width that is not instantiated
    probs = dgrad_launch(torch, L, ops, 8192, 128, 256, 2, seed=5)
    two = [dict(probs[0], srcs=probs[0]["srcs"] + probs[1]["srcs"])]
    name, _, out = run_dgrad(torch, ops, lib, two, True, False)
    assert not name.startswith("gemm_ws_kernel")
    v = sum(s[0].double() @ s[1].double() for s in two[0]["srcs"]) * (two[0]["Y"] > 0).double()
    assert float((out[0][0].double() - v).abs().max() / v.abs().max()) < RTOL
    noy = [dict(probs[0], mask=None)]
    name, _, out = run_dgrad(torch, ops, lib, noy, True, False)
    assert not name.startswith("gemm_ws_kernel")
    probs = dgrad_launch(torch, L, ops, 8192, 64, 192, 1, seed=6)
    name, _, _ = run_dgrad(torch, ops, lib, probs, True, False)
    assert not name.startswith("gemm_ws_kernel")


def test_ws_kernels_are_repeatable_on_a_full_chip(env):
    """Race screen: the benchmark's launches twenty times each against the tile kernel's results, bit for bit."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    probs = fwd_launch(torch, L, ops, 65536, 256, 128, 4, seed=11)
    _, ref = run_fwd(torch, ops, lib, probs, False, True)
    for rep in range(20):
        name, out = run_fwd(torch, ops, lib, probs, True, True)
        assert name.startswith("gemm_ws_kernel")
        for i, ((C, mk, am), (Ct, mkt, amt)) in enumerate(zip(out, ref)):
            assert torch.equal(C, Ct), (rep, i, int((C != Ct).sum()))
            assert torch.equal(mk, mkt), (rep, i)
            assert float(torch.max(am.view(torch.float32))) == float(torch.max(amt.view(torch.float32))), (rep, i)
    dprobs = dgrad_launch(torch, L, ops, 65536, 128, 256, 4, seed=12)
    _, _, dref = run_dgrad(torch, ops, lib, dprobs, False, False)
    for rep in range(20):
        name, _, out = run_dgrad(torch, ops, lib, dprobs, True, False)
        assert name.startswith("gemm_ws_kernel")
        for i, ((dA, am), (dAt, amt)) in enumerate(zip(out, dref)):
            assert torch.equal(dA, dAt), (rep, i, int((dA != dAt).sum()))


# ---------------------------------------------------------------------------------------------------------------
# K7 (round 6): PepNet's gate products in the weight-stationary kernel's turn (reference model/pepnet.py:64-78, :139-140)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,nprob", [
    (65536, 64, 64, 4),        # the four tasks' first-layer gates of Amazon-8 PepNet (gate hidden 64 -> K0 = 64)
    (8192 + 77, 128, 128, 3),  # hidden-layer gates (128 -> 128), ragged batch
    (16384, 80, 64, 2),        # a reduction of 80 (one group of five k-steps)
])
def test_ws_fwd_with_gate_product(env, M, K, N, nprob):
    """C = 2 sigmoid(A W^T + b) and prod = C (.) mul from the same turn: against float64, and C / prod / both magnitude
    slots bit for bit against the tile kernel's K7 epilogue."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    dev = torch.device("cuda:0")
    probs = fwd_launch(torch, L, ops, M, K, N, nprob, acts=[L.ACT_SIGMOID2] * nprob, seed=M + N)
    g = torch.Generator(device="cpu").manual_seed(3)
    for p in probs:
        p["mul"] = torch.randn(M, N, generator=g).to(dev)
        p["amax_prod"] = ops.amax_slots(1, dev)[0]
    res = {}
    for ws in (True, False):
        for p in probs:
            p["prod"] = torch.full((M, N), float("nan"), device=dev)
            p["amax_prod"].zero_()
        name, out = run_fwd(torch, ops, lib, probs, ws, False)
        assert name.startswith("gemm_ws_kernel") == ws, name
        res[ws] = [(C, p["prod"].clone(), am, p["amax_prod"].clone()) for p, (C, _, am) in zip(probs, out)]
    for p, (C, prod, am, am2), (Ct, prodt, amt, am2t) in zip(probs, res[True], res[False]):
        z = p["A"].double() @ p["W"].double().t() + (p["bias"].double() if p["bias"] is not None else 0.0)
        c = 2 * torch.sigmoid(z)
        assert float((C.double() - c).abs().max() / c.abs().max()) < RTOL
        pr = c * p["mul"].double()
        assert float((prod.double() - pr).abs().max() / pr.abs().max()) < RTOL
        assert torch.equal(C, Ct) and torch.equal(prod, prodt)
        a2 = float(torch.max(am2.view(torch.float32)))
        assert a2 >= float(prod.abs().max()) and a2 <= float(prod.abs().max()) * (1 + 1e-6)
        assert float(torch.max(am.view(torch.float32))) == float(torch.max(amt.view(torch.float32)))
        assert a2 == float(torch.max(am2t.view(torch.float32)))


@pytest.mark.parametrize("M,Nred,K,nprob,act_h,acc_h,acc_g", [
    (65536, 128, 64, 4, "none", 0, 0),     # first PPNet layer of four tasks: h = the gated input, K0 = 64
    (8192 + 77, 128, 128, 3, "relu", 0, 0),  # second layer: h = the ReLU output of the layer before
    (8192, 128, 64, 1, "none", 1, 1),      # accumulating into both factors' gradients
    (16384, 80, 128, 2, "relu", 1, 0),     # a reduction of 80
])
def test_ws_dgrad_gate_mode(env, M, Nred, K, nprob, act_h, acc_h, acc_g):
    """v = dC W is not stored: dH (+)= v g act_h'(h), dG (+)= v h act_g'(g) -- against float64 and, bit for bit, against
    the tile kernel's gate mode; both magnitude slots."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    dev = torch.device("cuda:0")
    probs = dgrad_launch(torch, L, ops, M, Nred, K, nprob, relu=False, seed=M + K)
    g = torch.Generator(device="cpu").manual_seed(9)
    gates = []
    for p in probs:
        h = torch.randn(M, K, generator=g)
        if act_h == "relu":
            h = torch.relu(h)
        gt = 2 * torch.sigmoid(torch.randn(M, K, generator=g))
        gates.append(dict(h=h.to(dev), g=gt.to(dev), old_h=torch.randn(M, K, generator=g).to(dev),
                          old_g=torch.randn(M, K, generator=g).to(dev), slots=ops.amax_slots(2, dev)))
    res = {}
    for ws in (True, False):
        lib.mml_gemm_set_ws(1 if ws else 0)
        launch = []
        for p, q in zip(probs, gates):
            q["dh"] = q["old_h"].clone() if acc_h else torch.full((M, K), float("nan"), device=dev)
            q["dg"] = q["old_g"].clone() if acc_g else torch.full((M, K), float("nan"), device=dev)
            q["slots"].zero_()
            launch.append(dict(dA=None, Y=None, act=L.ACT_NONE, srcs=p["srcs"],
                               gate=dict(h=q["h"], g=q["g"], dh=q["dh"], dg=q["dg"],
                                         act_h=L.ACT_RELU if act_h == "relu" else L.ACT_NONE, act_g=L.ACT_SIGMOID2,
                                         acc_h=acc_h, acc_g=acc_g, amax_dh=q["slots"][0], amax_dg=q["slots"][1])))
        arr = ops.make_dgrad_descs(launch)
        L.check(lib.mml_pep_gate_bwd(arr, len(launch), ops._stream()), "mml_pep_gate_bwd")
        torch.cuda.synchronize()
        name = lib.mml_gemm_last_kernel().decode()
        assert name.startswith("gemm_ws_kernel") == ws, name
        res[ws] = [(q["dh"].clone(), q["dg"].clone(), q["slots"].clone()) for q in gates]
    for p, q, (dh, dg, sl), (dht, dgt, slt) in zip(probs, gates, res[True], res[False]):
        dC, W = p["srcs"][0][:2]
        v = dC.double() @ W.double()
        gd, hd = q["g"].double(), q["h"].double()
        ref_h = v * gd * ((hd > 0).double() if act_h == "relu" else 1.0) + (q["old_h"].double() if acc_h else 0.0)
        ref_g = v * hd * (gd * (1 - gd / 2)) + (q["old_g"].double() if acc_g else 0.0)
        assert float((dh.double() - ref_h).abs().max() / ref_h.abs().max()) < RTOL
        assert float((dg.double() - ref_g).abs().max() / ref_g.abs().max()) < RTOL
        assert torch.equal(dh, dht) and torch.equal(dg, dgt)
        for t, s in ((dh, sl[0]), (dg, sl[1])):
            am = float(torch.max(s.view(torch.float32)))
            assert am >= float(t.abs().max()) and am <= float(t.abs().max()) * (1 + 1e-6)
