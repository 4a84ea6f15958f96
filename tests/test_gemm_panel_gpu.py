"""The activation-stationary forward kernel (csrc/gemm_panel.hip) against float64 and, bit for bit, against the tile
kernel it replaces for the shared-input first layers (reference model/mmoe.py:69-79, model/utils.py:146-161)."""
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
    lib.mml_gemm_set_ws(0)     # (this file compares the panel kernel with the TILE kernel: the weight-stationary one off)
    yield torch, L, ops, lib
    lib.mml_gemm_set_mode(mode0)
    lib.mml_gemm_set_panel(1)
    lib.mml_gemm_set_ws(1)


def make_launch(torch, L, ops, M, K, Ns, masks=True, bias=True, acts=None, seed=0, scale=1.0):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = (torch.randn(M, K, generator=g) * scale).to(dev)
    slots = ops.amax_slots(1 + 2 * len(Ns), dev)
    ops.amax_batch([(A, slots[0])])
    probs, items = [], []
    for i, N in enumerate(Ns):
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev) if (bias and i % 3 != 2) else None
        ops.amax_batch([(W, slots[1 + 2 * i])])
        planes = torch.zeros(W.shape, dtype=torch.int32, device=dev)
        kexp = torch.zeros(1, dtype=torch.int32, device=dev)
        items.append((W, planes, ops.PLANES_ROWS, [slots[1 + 2 * i]], kexp))
        act = (acts[i] if acts else L.ACT_RELU)
        p = dict(A=A, W=W, bias=b, act=act, amax_a=slots[0], amax_w=slots[1 + 2 * i], w_planes=planes, w_kexp=kexp,
                 amax_out=slots[2 + 2 * i])
        probs.append(p)
    ops.planes_cut(items)
    return A, probs, slots


def run(torch, ops, lib, probs, M, panel, masks):
    dev = torch.device("cuda:0")
    lib.mml_gemm_set_panel(1 if panel else 0)
    out = []
    for p in probs:
        N = p["W"].shape[0]
        p["C"] = torch.full((M, N), float("nan"), device=dev)
        if masks:
            p["mask"] = torch.full((M, (N + 31) // 32), 0x55555555, dtype=torch.int32, device=dev)
        p["amax_out"].zero_()
    ops.gemm_fwd(probs)
    torch.cuda.synchronize()
    name = lib.mml_gemm_last_kernel().decode()
    for p in probs:
        out.append((p["C"].clone(), p["mask"].clone() if masks else None, p["amax_out"].clone()))
    return name, out


@pytest.mark.parametrize("M,K,Ns,masks", [
    (8192, 240, [256, 256, 256, 256, 64, 64], True),     # AE-30's first layer: 4 experts + 2 gates, two gates in one tile
    (8192, 240, [256, 64, 64], False),                   # inference: no masks; a problem without bias
    (8192 + 128, 160, [128, 64, 64], True),              # the shortest reduction the kernel takes, 65 panels
    (40064, 208, [64, 64], True),                        # one tile only; more panels than workgroups (uneven shares)
    (65536, 240, [256, 256, 256, 256, 64, 64], True),    # the benchmark's launch: two panels of nine tiles per workgroup
    (3 * 32768 + 128, 240, [128, 64, 64], False),        # three or four panels per workgroup
    (65536, 304, [256, 256, 256, 256, 64, 64], True),    # AE-30 with the 63 dense columns (K0 = 303 zero-padded to 304)
    (8192, 320, [128, 64, 64], True),                    # the longest reduction the registers hold
    (8192, 176, [128, 128], True),                       # every k-block count between 10 and 20 is instantiated
    (8192, 192, [64, 64], False),
    (8192, 224, [128, 64, 64], True),
    (8192, 256, [256, 128], True),
    (8192, 272, [128, 128], False),
    (8192, 288, [128, 64, 64], True),
    (16384 + 77, 240, [256, 64, 64], True),              # ragged batch: whole panels here, the last 77 rows through the tile kernel
    (8192 + 3, 304, [128, 128], False),
])
def test_panel_fwd_matches_float64_and_the_tile_kernel(env, M, K, Ns, masks):
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    A, probs, _ = make_launch(torch, L, ops, M, K, Ns, masks=masks, seed=M + K)
    name_p, out_p = run(torch, ops, lib, probs, M, True, masks)
    assert name_p == "gemm_panel_kernel", name_p
    name_t, out_t = run(torch, ops, lib, probs, M, False, masks)
    assert "gemm_pipe_kernel" in name_t and ", 2, " in name_t, name_t
    for p, (C, mk, am), (Ct, mkt, amt) in zip(probs, out_p, out_t):
        z = A.double() @ p["W"].double().t()
        if p["bias"] is not None:
            z = z + p["bias"].double()
        ref = torch.relu(z) if p["act"] == L.ACT_RELU else z
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        assert err < RTOL, err
        assert torch.equal(C, Ct)                       # same planes, same product and k order: same bits
        if masks:
            N = p["W"].shape[0]
            bits = (C > 0).cpu().numpy()
            words = mk.cpu().numpy().view(np.uint32)
            got = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(M, -1)[:, :N].astype(bool)
            assert np.array_equal(got, bits)
            assert torch.equal(mk, mkt)
        # the published magnitude bounds what was stored
        amax = float(torch.max(am.view(torch.float32)))
        assert amax >= float(C.abs().max()) and amax <= float(C.abs().max()) * (1 + 1e-6)


def test_panel_fwd_is_scale_invariant(env):
    """Operands far outside fp16's own range (the reference initialises weights at 1e-4; activations of 3e-9 or 2e20)."""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    for scale in (3e-9, 1.0, 2e20):
        A, probs, _ = make_launch(torch, L, ops, 8192, 240, [128, 64, 64], masks=False, bias=False, seed=3, scale=scale)
        name, out = run(torch, ops, lib, probs, 8192, True, False)
        assert name == "gemm_panel_kernel"
        for p, (C, _, _) in zip(probs, out):
            z = A.double() @ p["W"].double().t()
            ref = torch.relu(z) if p["act"] == L.ACT_RELU else z
            err = float((C.double() - ref).abs().max() / ref.abs().max())
            assert err < RTOL, (scale, err)


def test_launches_the_panel_kernel_does_not_serve_fall_back(env):
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    # K beyond the registers, K below the ring's depth, a batch below the threshold (also after the ragged rows are taken
    # off), a sigmoid, a linear layer, an odd number of half tiles
    for M, K, Ns, acts in ((8192, 336, [128], None), (8192, 144, [128], None), (4096, 240, [128], None),
                           (8192 + 77 - 128, 240, [128], None), (8192, 240, [128], [L.ACT_SIGMOID]),
                           (8192, 240, [128], [L.ACT_NONE]), (8192, 240, [192], None)):
        A, probs, _ = make_launch(torch, L, ops, M, K, Ns, masks=False, acts=acts)
        name, out = run(torch, ops, lib, probs, M, True, False)
        assert "gemm_pipe_kernel" in name, (M, K, name)
        z = A.double() @ probs[0]["W"].double().t() + probs[0]["bias"].double()
        ref = torch.relu(z) if not acts else (torch.sigmoid(z) if acts[0] == L.ACT_SIGMOID else z)
        assert float((out[0][0].double() - ref).abs().max() / ref.abs().max()) < RTOL


def test_panel_fwd_is_repeatable_on_a_full_chip(env):
    """Race screen: the benchmark's launch (every CU busy, two panels per workgroup) forty times against the tile kernel's
    result -- outputs, sign masks and magnitudes bit for bit every time.  (The first hand-placed form read a tile's bias
    one k-step after its LDS-DMA was issued: wrong bias in ~3 % of the panels, only with every CU loaded.)"""
    torch, L, ops, lib = env
    lib.mml_gemm_set_mode(4)
    M, K, Ns = 65536, 240, [256, 256, 256, 256, 64, 64]
    A, probs, _ = make_launch(torch, L, ops, M, K, Ns, masks=True, seed=11)
    name_t, ref = run(torch, ops, lib, probs, M, False, True)
    assert "gemm_pipe_kernel" in name_t
    for rep in range(40):
        name_p, out = run(torch, ops, lib, probs, M, True, True)
        assert name_p == "gemm_panel_kernel"
        for i, ((C, mk, am), (Ct, mkt, amt)) in enumerate(zip(out, ref)):
            assert torch.equal(C, Ct), (rep, i, int((C != Ct).sum()))
            assert torch.equal(mk, mkt), (rep, i)
            assert float(torch.max(am.view(torch.float32))) == float(torch.max(amt.view(torch.float32))), (rep, i)
