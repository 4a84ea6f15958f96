"""The cut-once weight-gradient kernel (csrc/gemm_nt.hip: gemm_nt_kernel -- planes cut once per workgroup, fragments
through ds_read_b64_tr_b16) against float64 torch, at the layer shapes of the AliExpress MMoE (reference model/mmoe.py:65-119:
first layers 4 x (240 -> 256) + 2 x (240 -> 64) on ONE input, second layers 4 x (256 -> 128), towers 2 x (128 -> 64);
AE-30d: 303 + 1 padded input columns) and at edge shapes: problems narrower than a tile, K that ends inside a tile, slabs of
unequal length, operands of very different magnitude.  Tolerance: the fp32 contract (1e-4 of the largest output; measured
~3e-7) -- the same products and fp32 accumulation as the tile kernel in another summation order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def env():
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib as L, ops
    lib = L.load()
    mode0 = lib.mml_gemm_get_mode()
    lib.mml_gemm_set_mode(4)
    yield L, ops, lib
    lib.mml_gemm_set_mode(mode0)
    lib.mml_gemm_set_nt(1)   # (the library's default)


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def make(shapes, M, seed, shared_A=True, scale_c=1.0, scale_a=1.0):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    probs, As = [], {}
    for N, K in shapes:
        key = K if shared_A else len(probs)
        if key not in As:
            As[key] = (torch.randn(M, K, generator=g) * scale_a).to(dev)
        dC = (torch.randn(M, N, generator=g) * scale_c).to(dev)
        probs.append(dict(dC=dC, A=As[key], dW=torch.full((N, K), float("nan"), device=dev),
                          dbias=torch.full((N,), float("nan"), device=dev)))
    return probs


def check(probs, tol=RTOL):
    worst = 0.0
    for p in probs:
        ref = p["dC"].double().t() @ p["A"].double()
        worst = max(worst, rel(p["dW"], ref), rel(p["dbias"], p["dC"].double().sum(0)))
        assert rel(p["dW"], ref) < tol, (tuple(ref.shape), rel(p["dW"], ref))
        assert rel(p["dbias"], p["dC"].double().sum(0)) < tol
    return worst


@pytest.mark.parametrize("M,shapes,shared", [
    (65536, [(256, 240)] * 4 + [(64, 240)] * 2, True),     # AE-30 first layers: six problems on one input, K ends in a tile
    (65536, [(128, 256)] * 4, False),                      # second expert layers: own inputs
    (65536, [(64, 128)] * 2, False),                       # towers: problems half a tile wide
    (16384, [(256, 304), (64, 304)], True),                # AE-30d: 303 input columns padded to 304
    (16384 + 32 * 5, [(96, 100), (32, 4)], False),         # slabs of unequal length, ragged tiles, a 4-column problem
    (32768, [(512, 512), (128, 512)], True),               # KuaiRec-32 first layers (several tiles each way)
    # round 6: 48 problems per launch (PepNet's weight gradients were three launches of <= 16)
    (16384, [(64, 72), (96, 80), (32, 64), (128, 128)] * 12, False),
    (32768, [(32, 8)] * 41, False),
])
def test_nt_wgrad_matches_float64(env, M, shapes, shared):
    L, ops, lib = env
    lib.mml_gemm_set_nt(1)
    probs = make(shapes, M, seed=M + len(shapes), shared_A=shared)
    ops.gemm_wgrad(probs, amax=True)
    torch.cuda.synchronize()
    assert lib.mml_gemm_last_kernel().decode() == "gemm_nt_kernel"
    worst = check(probs)
    assert worst < 5e-6, worst   # (measured ~3e-7: fp32-equivalent, not merely inside the 1e-4 contract)


@pytest.mark.parametrize("M,shapes,shared", [
    (65536, [(256, 240)] * 4 + [(64, 240)] * 2, True),
    (16384 + 32 * 5, [(96, 100), (32, 4)], False),
    (16384, [(64, 72), (96, 80), (32, 64), (128, 128)] * 12, False),
])
def test_nt_wgrad_lab_variant_with_direct_gradient_fragments(env, M, shapes, shared):
    """mml_gemm_set_nt(3): gemm_ntd_kernel (the dC fragments straight from global memory; measured slower, kept for the record)
    computes the same products -- the weight gradients are BITWISE those of gemm_nt_kernel, the bias gradients (another
    summation order) agree to fp32 noise."""
    L, ops, lib = env
    out = {}
    for mode in (1, 3):
        lib.mml_gemm_set_nt(mode)
        probs = make(shapes, M, seed=7, shared_A=shared)
        ops.gemm_wgrad(probs, amax=True)
        torch.cuda.synchronize()
        assert lib.mml_gemm_last_kernel().decode() == "gemm_nt_kernel"
        check(probs)
        out[mode] = probs
    lib.mml_gemm_set_nt(1)
    for a, b in zip(out[1], out[3]):
        assert torch.equal(a["dW"], b["dW"])
        assert rel(a["dbias"], b["dbias"].double()) < 1e-5


def test_nt_wgrad_against_the_tile_kernel_accumulate_and_phases(env):
    """Same arithmetic as gemm_pipe_kernel (two fp16 planes, three products, fp32 accumulation): the two kernels agree to
    summation order; accumulate adds to what dW / dbias hold; the two-launch form equals the one-call form bit for bit;
    two runs are bitwise equal (fixed slab order)."""
    L, ops, lib = env
    M, shapes = 32768, [(256, 240), (64, 240)]
    probs = make(shapes, M, seed=3)
    lib.mml_gemm_set_nt(0)
    ops.gemm_wgrad(probs, amax=True)
    torch.cuda.synchronize()
    assert "gemm_pipe_kernel" in lib.mml_gemm_last_kernel().decode()
    tile = [(p["dW"].clone(), p["dbias"].clone()) for p in probs]
    lib.mml_gemm_set_nt(1)
    ops.gemm_wgrad(probs, amax=True)
    torch.cuda.synchronize()
    assert lib.mml_gemm_last_kernel().decode() == "gemm_nt_kernel"
    for p, (tw, tb) in zip(probs, tile):
        assert rel(p["dW"], tw.double()) < 2e-6 and rel(p["dbias"], tb.double()) < 2e-6
        assert not torch.equal(p["dW"], tw)          # (another summation order: really the other kernel)
    first = [(p["dW"].clone(), p["dbias"].clone()) for p in probs]
    ops.gemm_wgrad(probs, amax=True)
    for p, (w0, b0) in zip(probs, first):
        assert torch.equal(p["dW"], w0) and torch.equal(p["dbias"], b0)
    # accumulate
    for p in probs:
        p["accumulate"] = 1
    ops.gemm_wgrad(probs, amax=True)
    for p, (w0, b0) in zip(probs, first):
        assert rel(p["dW"], 2 * w0.double()) < 1e-6 and rel(p["dbias"], 2 * b0.double()) < 1e-6
    # phases 1 + 2 through the C ABI
    for p in probs:
        p["accumulate"] = 0
    cache = ops._measured([p["dC"] for p in probs] + [p["A"] for p in probs], {})
    k = lambda t: (t.data_ptr(), tuple(t.shape), t.stride(0))  # noqa: E731
    withmax = [dict(p, amax_dc=cache[k(p["dC"])], amax_a=cache[k(p["A"])]) for p in probs]
    arr = ops.make_wgrad_descs(withmax)
    n = lib.mml_gemm_grouped_wgrad_workspace_bytes(arr, len(probs))
    ws = torch.empty(int(n), dtype=torch.uint8, device="cuda:0")
    for ph in (1, 2):
        L.check(lib.mml_gemm_grouped_wgrad_phase(arr, len(probs), ws.data_ptr(), ws.numel(), ph,
                                                 torch.cuda.current_stream().cuda_stream), "wgrad phase")
    for p, (w0, b0) in zip(probs, first):
        assert torch.equal(p["dW"], w0) and torch.equal(p["dbias"], b0)


@pytest.mark.parametrize("scale_c,scale_a", [(1e-6, 1e3), (3e4, 1e-5), (1e-12, 1e-12), (1e15, 1e12)])
def test_nt_wgrad_is_scale_invariant(env, scale_c, scale_a):
    """The planes are cut from operands scaled by powers of two taken from their magnitude slots: the relative error does
    not depend on the operands' magnitudes (gradients of 1e-6, activations of 1e3: what a step really holds)."""
    L, ops, lib = env
    lib.mml_gemm_set_nt(1)
    probs = make([(128, 256), (64, 256)], 16384, seed=11, scale_c=scale_c, scale_a=scale_a)
    ops.gemm_wgrad(probs, amax=True)
    torch.cuda.synchronize()
    assert lib.mml_gemm_last_kernel().decode() == "gemm_nt_kernel"
    assert check(probs) < 5e-6


def test_nt_wgrad_leaves_other_launches_to_the_tile_kernel(env):
    L, ops, lib = env
    lib.mml_gemm_set_nt(1)
    for M, shapes, amax in ((8192, [(128, 256)], True),          # small batch
                            (16384, [(32, 16)] * 49, True),      # more than 48 problems in one call
                            (16384 + 16, [(128, 256)], True),    # M % 32
                            (16384, [(100, 48)], True),          # N % 32
                            (16384, [(128, 256)], False)):       # no magnitudes
        probs = make(shapes, M, seed=5)
        ops.gemm_wgrad(probs, amax=amax)
        torch.cuda.synchronize()
        assert "gemm_pipe_kernel" in lib.mml_gemm_last_kernel().decode(), (M, shapes, amax)
        check(probs)
