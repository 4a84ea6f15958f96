"""The output-stationary input-gradient kernel (csrc/gemm_os.hip) against float64 and, bit for bit, against the tile kernel
it replaces for d(dnn_input): the one wide gradient every expert's and every gate's first layer adds to (reference
model/mmoe.py:69-79 calls them all on the combined input; model/utils.py:146-161 is the Linear whose `mm` backward this is)."""
import pytest

pytestmark = pytest.mark.gpu
RTOL = 2e-6  # max-norm, against float64 (the two-plane fp16 arithmetic measures 3.3e-7)


@pytest.fixture()
def env(monkeypatch):
    import torch
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib as L, ops
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    lib = L.load()
    mode0 = lib.mml_gemm_get_mode()
    lib.mml_gemm_set_mode(4)
    monkeypatch.delenv("MMLREC_GEMM_OS", raising=False)
    yield torch, L, ops, lib, monkeypatch
    lib.mml_gemm_set_mode(mode0)


def launch(torch, L, ops, M, K, Ns, seed=0, scales=None, ldpad=0):
    """dA [M, K] = sum_s dC_s [M, N_s] W_s [N_s, K]; the weights cut as ONE group (one exponent), like engine.py does
    for the layers that feed one input gradient."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    slots = ops.amax_slots(2 * len(Ns) + 1, dev)
    kexp = torch.zeros(1, dtype=torch.int32, device=dev)
    srcs, items, wslots = [], [], []
    for s, N in enumerate(Ns):
        sc = scales[s] if scales else (1.0 + s)
        dC = (torch.randn(M, N, generator=g) * sc).to(dev)
        W = torch.zeros(N, K + ldpad, device=dev)[:, :K]
        W.copy_((torch.randn(N, K, generator=g) / sum(Ns) ** 0.5).to(dev))
        sd, sw = slots[2 * s], slots[2 * s + 1]
        ops.amax_batch([(dC, sd), (W, sw)])
        planes = torch.zeros((N, K + ldpad), dtype=torch.int32, device=dev)[:, :K]
        wslots.append(sw)
        srcs.append([dC, W, 0, sd, sw, planes, kexp])
    for (dC, W, _, sd, sw, planes, _) in srcs:
        items.append((W, planes, ops.PLANES_COLS, wslots, kexp))
    ops.planes_cut(items)
    return dict(Y=None, act=L.ACT_NONE, mask=None, amax_out=slots[-1], srcs=[tuple(s) for s in srcs]), K


def run(torch, ops, lib, mp, prob, M, K, on, accumulate=False, ldda=None):
    dev = torch.device("cuda:0")
    mp.setenv("MMLREC_GEMM_OS", "1" if on else "0")
    g = torch.Generator(device="cpu").manual_seed(77)
    old = torch.randn(M, K, generator=g).to(dev)
    buf = torch.full((M, ldda or K), float("nan"), device=dev)
    dA = buf[:, :K]
    if accumulate:
        dA.copy_(old)
    prob = dict(prob, dA=dA, accumulate=int(accumulate))
    prob["amax_out"].zero_()
    ops.gemm_dgrad([prob])
    torch.cuda.synchronize()
    return lib.mml_gemm_last_kernel().decode(), old, dA.clone(), prob["amax_out"].clone(), buf


@pytest.mark.parametrize("M,K,Ns,acc", [
    (65536, 240, [256, 256, 256, 256, 64, 64], False),   # AE-30: four experts + two gates feed d(dnn_input)
    (16384, 240, [256, 256, 256, 256, 64, 64], True),    # ... accumulating onto what the buffer holds
    (16384 + 77, 240, [256, 64], False),                 # ragged last panel
    (16384, 256, [128, 128, 128], False),                # every column of the panel in use
    (32768 + 256, 192, [512, 64], False),                # the narrowest gradient it takes; more panels than one round
    (16384, 208, [256, 256, 256, 256, 64, 64, 64, 64], False),   # eight sources (MML_MAX_SRC), K = 208 (AE's real layout)
    (70000, 224, [16, 48, 256], False),                  # short sources: the cursor changes source inside the ring
])
def test_os_dgrad_matches_float64_and_the_tile_kernel(env, M, K, Ns, acc):
    torch, L, ops, lib, mp = env
    prob, K = launch(torch, L, ops, M, K, Ns, seed=M + K)
    name_o, old, dA, am, _ = run(torch, ops, lib, mp, prob, M, K, True, acc)
    assert name_o == "gemm_os_kernel", name_o
    name_t, _, dAt, amt, _ = run(torch, ops, lib, mp, prob, M, K, False, acc)
    assert "gemm_pipe_kernel" in name_t and ", 2, " in name_t, name_t
    v = sum(s[0].double() @ s[1].double() for s in prob["srcs"])
    if acc:
        v = v + old.double()
    err = float((dA.double() - v).abs().max() / v.abs().max())
    assert err < RTOL, err
    assert torch.equal(dA, dAt)                      # same planes, same product, source and k order: same bits
    amax = float(torch.max(am.view(torch.float32)))
    assert amax >= float(dA.abs().max()) and amax <= float(dA.abs().max()) * (1 + 1e-6)


def test_os_dgrad_padded_pitches_and_unequal_magnitudes(env):
    """The zero-padded operand pair of a reduction that is not a multiple of 16 on the OUTPUT side (K0 = 303 in rows of
    304: engine.Val.grad_cols narrows the launch to the 240 embedding columns of buffers pitched 304), and sources whose
    gradients differ by 2^20 in magnitude (one common scale: the small one keeps fewer bits, like in the tile kernel)."""
    torch, L, ops, lib, mp = env
    M, K = 16384, 240
    prob, K = launch(torch, L, ops, M, K, [256, 64, 64], seed=5, scales=[1.0, 2.0 ** -20, 3.0], ldpad=64)
    name_o, _, dA, am, buf = run(torch, ops, lib, mp, prob, M, K, True, ldda=304)
    assert name_o == "gemm_os_kernel", name_o
    assert torch.isnan(buf[:, K:]).all()             # the columns behind the gradient are not written
    name_t, _, dAt, _, _ = run(torch, ops, lib, mp, prob, M, K, False, ldda=304)
    assert "gemm_pipe_kernel" in name_t
    assert torch.equal(dA, dAt)
    v = sum(s[0].double() @ s[1].double() for s in prob["srcs"])
    assert float((dA.double() - v).abs().max() / v.abs().max()) < RTOL


def test_launches_the_os_kernel_does_not_serve_fall_back(env):
    torch, L, ops, lib, mp = env
    # one source (the weight-stationary kernel's launch), a small batch, a short reduction, more than 256 columns, a narrow
    # gradient (half of the workgroup's 256 columns would be idle)
    for M, K, Ns in ((16384, 256, [128]), (8192, 240, [256, 64]), (16384, 240, [64, 64]), (16384, 288, [256, 256]),
                     (16384, 128, [256, 256])):
        prob, K = launch(torch, L, ops, M, K, Ns, seed=3)
        name, _, dA, _, _ = run(torch, ops, lib, mp, prob, M, K, True)
        assert name != "gemm_os_kernel", (M, K, Ns, name)
        v = sum(s[0].double() @ s[1].double() for s in prob["srcs"])
        assert float((dA.double() - v).abs().max() / v.abs().max()) < RTOL


def test_os_dgrad_is_repeatable_on_a_full_chip(env):
    """Race screen for the ring (one barrier per k-step, counted waits): the benchmark's launch twenty times, the tile
    kernel's bits every time."""
    torch, L, ops, lib, mp = env
    M, K, Ns = 65536, 240, [256, 256, 256, 256, 64, 64]
    prob, K = launch(torch, L, ops, M, K, Ns, seed=11)
    _, _, ref, amt, _ = run(torch, ops, lib, mp, prob, M, K, False)
    for rep in range(20):
        name, _, dA, am, _ = run(torch, ops, lib, mp, prob, M, K, True)
        assert name == "gemm_os_kernel"
        assert torch.equal(dA, ref), (rep, int((dA != ref).sum()))
        assert float(torch.max(am.view(torch.float32))) == float(torch.max(amt.view(torch.float32))), rep
