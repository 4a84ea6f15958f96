"""The bf16-STORAGE GEMM family (include/mmlrec.h K3', csrc/gemm16.hip) against float64 products of the SAME bf16 operand
values: the kernels add nothing to the operands' rounding but the fp32 accumulation order, so the tolerance is the fp32
one (1e-5 of the largest output for reductions up to 65 536 terms), plus the bf16 rounding of a bf16 output (2^-9).
Shapes: BASELINE.json configs[1] (MMoE on KuaiRec-shaped batches: experts 512 -> 512 -> 256, gates 512 -> 128, towers
256 -> 128; reference configs_mtl/config_kuairec.json) at a reduced batch, and edge shapes of the tiles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import ops as o
    return o


def rand16(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen, device="cpu") * scale).to(torch.bfloat16).to(dev())


def f64(t):
    return t.to(torch.float64)


def maxrel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def test_cast16_and_transposed_copy(ops):
    g = torch.Generator().manual_seed(0)
    for rows, cols in ((512, 512), (128, 512), (4, 130), (257, 33), (1, 7)):
        big = torch.randn(rows, cols + 5, generator=g).to(dev())
        src = big[:, :cols]                       # (a pitched view: row pitch cols + 5)
        assert torch.equal(ops.cast16(src), src.to(torch.bfloat16))
        assert torch.equal(ops.cast16(src, transpose=True), src.t().contiguous().to(torch.bfloat16))
    # round to nearest even, NaN / Inf / zero bit patterns like torch's conversion
    edge = torch.tensor([[1.0, 1.00390625, 1.01171875, -0.0, float("inf"), 3.3895314e38, 65504.0]], device=dev())
    assert torch.equal(ops.cast16(edge).view(torch.int16), edge.to(torch.bfloat16).view(torch.int16))


def test_gather16_is_the_gather_rounded(ops):
    rng = np.random.default_rng(3)
    E, nd, B = 16, 4, 3001
    vocab = [2, 63, 800, 7000, 10000, 5]
    F = len(vocab)
    tabs = [torch.from_numpy(rng.standard_normal((v, E)).astype(np.float32)).to(dev()) for v in vocab]
    idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
    idx[0, :] = 0
    idx[1, :] = np.array(vocab) - 1
    X = torch.from_numpy(np.concatenate([idx.astype(np.float32), rng.random((B, nd), dtype=np.float32)], 1)).to(dev())
    status = ops.new_status(dev())
    out = ops.gather16_fwd(tabs, X, list(range(F)), F, nd, status=status)
    ops.check_status(status)
    ref = ops.gather_fwd(tabs, X, list(range(F)), F, nd)
    assert torch.equal(out, ref.to(torch.bfloat16))
    X[7, 2] = 800.0
    ops.gather16_fwd(tabs, X, list(range(F)), F, nd, status=status)
    with pytest.raises(IndexError):
        ops.check_status(status)


@pytest.mark.parametrize("M", [128, 4096])
def test_forward_kuairec_first_layer_group(ops, M):
    """Six problems on ONE input: 4 x (512 -> 512) + 2 x (512 -> 128), bias + ReLU, sign masks; expert outputs stored as
    bf16, gate-network outputs as fp32 (a row kernel reads them)."""
    from mmlrec_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    A = rand16(g, M, 512)
    probs, refs = [], []
    for N, c16 in ((512, True), (512, True), (512, True), (512, True), (128, False), (128, False)):
        W = rand16(g, N, 512, scale=512 ** -0.5)
        b = torch.randn(N, generator=g).to(dev())
        C = torch.empty(M, N, dtype=torch.bfloat16 if c16 else torch.float32, device=dev())
        mask = torch.zeros(M, N // 32, dtype=torch.int32, device=dev())
        probs.append(dict(srcs=[(A, W)], C=C, bias=b, act=L.ACT_RELU, mask_out=mask))
        refs.append(torch.relu(f64(A) @ f64(W).t() + f64(b)))
    ops.g16_tn(probs)
    assert L.load().mml_g16_last_kernel().decode() == "g16_tn_kernel<128>"
    for q, ref in zip(probs, refs):
        tol = 1e-5 + (2.0 ** -8 if q["C"].dtype == torch.bfloat16 else 0.0)
        assert maxrel(q["C"], ref) < tol
        bits = ((q["mask_out"].view(torch.int32).unsqueeze(-1) >> torch.arange(32, device=dev())) & 1).reshape(M, -1).bool()
        # the mask is the sign of what the kernel computed in fp32: compare where the reference is clear of zero
        clear = ref.abs() > 1e-4 * ref.abs().max()
        assert torch.equal(bits[clear], (ref > 0)[clear])
        assert bool((f64(q["C"])[~bits] == 0).all())


@pytest.mark.parametrize("M", [8192, 16384 + 128])
def test_wide_tiles_forward_and_input_gradient(ops, M):
    """Round 6: problems whose width is a multiple of 256 AND whose reduction is at least 1 536 columns long run 128 x 256
    tiles (a wave's tile 64 x 128: 0.75 LDS fragment reads per MFMA instead of 1.0) from 8 192 rows on, in a launch of
    their own -- here: KuaiRec-32's first-layer group and second layers stay on the 128-wide tiles (reductions of 512);
    the multi-source input gradient of the first layers (2 304 reduction columns) into an accumulating fp32 output and a
    long masked bf16 input gradient + a forward with bias / ReLU / sign masks over a 2 048-column reduction run the wide
    ones.  Same tolerances as the 128-wide tiles."""
    from mmlrec_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(21)
    A = rand16(g, M, 512)
    probs, refs = [], []
    for N, c16 in ((512, True), (512, True), (128, False), (512, True), (512, False), (128, False), (256, True)):
        W = rand16(g, N, 512, scale=512 ** -0.5)
        b = torch.randn(N, generator=g).to(dev())
        C = torch.empty(M, N, dtype=torch.bfloat16 if c16 else torch.float32, device=dev())
        mask = torch.zeros(M, N // 32, dtype=torch.int32, device=dev())
        probs.append(dict(srcs=[(A, W)], C=C, bias=b, act=L.ACT_RELU, mask_out=mask))
        refs.append(torch.relu(f64(A) @ f64(W).t() + f64(b)))
    ops.g16_tn(probs)
    assert lib.mml_g16_last_kernel().decode() == "g16_tn_kernel<128>"    # (reductions of 512: below the wide tiles' 1 536)
    for q, ref in zip(probs, refs):
        tol = 1e-5 + (2.0 ** -8 if q["C"].dtype == torch.bfloat16 else 0.0)
        assert maxrel(q["C"], ref) < tol
        bits = ((q["mask_out"].view(torch.int32).unsqueeze(-1) >> torch.arange(32, device=dev())) & 1).reshape(M, -1).bool()
        clear = ref.abs() > 1e-4 * ref.abs().max()
        assert torch.equal(bits[clear], (ref > 0)[clear])
        assert bool((f64(q["C"])[~bits] == 0).all())
    # input gradient, six sources, fp32 output that accumulates
    srcs, ref = [], torch.zeros(M, 512, dtype=torch.float64, device=dev())
    for N in (512, 512, 512, 512, 128, 128):
        dC = rand16(g, M, N)
        Wt = rand16(g, 512, N, scale=N ** -0.5)
        srcs.append((dC, Wt))
        ref += f64(dC) @ f64(Wt).t()
    out = torch.full((M, 512), 3.0, device=dev())
    ops.g16_tn([dict(srcs=srcs, C=out)])
    assert lib.mml_g16_last_kernel().decode() == "g16_tn_kernel<256>"
    assert maxrel(out, ref) < 1e-5
    ops.g16_tn([dict(srcs=srcs, C=out, accumulate=1)])
    assert maxrel(out, 2 * ref) < 1e-5
    # masked bf16 input gradient over a long reduction (wide tiles), and the second layers' own (narrow tiles)
    for K2, want in ((2048, "g16_tn_kernel<256>"), (256, "g16_tn_kernel<128>")):
        dE = rand16(g, M, K2)
        W2t = rand16(g, 512, K2, scale=K2 ** -0.5)
        mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (M, 16), dtype=torch.int32, generator=g).to(dev())
        dH = torch.empty(M, 512, dtype=torch.bfloat16, device=dev())
        ops.g16_tn([dict(srcs=[(dE, W2t)], C=dH, mask_in=mask)])
        assert lib.mml_g16_last_kernel().decode() == want
        bits = ((mask.unsqueeze(-1) >> torch.arange(32, device=dev())) & 1).reshape(M, -1).bool()
        ref2 = (f64(dE) @ f64(W2t).t()) * bits
        assert maxrel(dH, ref2) < 1e-5 + 2.0 ** -8
        assert bool((f64(dH)[~bits] == 0).all())
    # forward with bias + ReLU + sign masks on the wide tiles: bf16 and fp32 outputs, 256 and 512 columns
    A2 = rand16(g, M, 2048)
    probs2, refs2 = [], []
    for N, c16 in ((256, True), (512, False), (512, True)):
        W = rand16(g, N, 2048, scale=2048 ** -0.5)
        b = torch.randn(N, generator=g).to(dev())
        C = torch.empty(M, N, dtype=torch.bfloat16 if c16 else torch.float32, device=dev())
        mk = torch.zeros(M, N // 32, dtype=torch.int32, device=dev())
        probs2.append(dict(srcs=[(A2, W)], C=C, bias=b, act=L.ACT_RELU, mask_out=mk))
        refs2.append(torch.relu(f64(A2) @ f64(W).t() + f64(b)))
    ops.g16_tn(probs2)
    assert lib.mml_g16_last_kernel().decode() == "g16_tn_kernel<256>"
    for q, ref in zip(probs2, refs2):
        tol = 1e-5 + (2.0 ** -8 if q["C"].dtype == torch.bfloat16 else 0.0)
        assert maxrel(q["C"], ref) < tol
        bits = ((q["mask_out"].view(torch.int32).unsqueeze(-1) >> torch.arange(32, device=dev())) & 1).reshape(M, -1).bool()
        clear = ref.abs() > 1e-4 * ref.abs().max()
        assert torch.equal(bits[clear], (ref > 0)[clear])
        assert bool((f64(q["C"])[~bits] == 0).all())


def test_forward_64_wide_outputs_and_linear_activation(ops):
    from mmlrec_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    M = 256
    A = rand16(g, M, 128)
    probs, refs = [], []
    for N in (64, 192, 128):
        W = rand16(g, N, 128, scale=0.1)
        C = torch.empty(M, N + 8, dtype=torch.float32, device=dev())[:, :N]   # pitched output
        probs.append(dict(srcs=[(A, W)], C=C, act=L.ACT_NONE))
        refs.append(f64(A) @ f64(W).t())
    ops.g16_tn(probs)
    assert L.load().mml_g16_last_kernel().decode() == "g16_tn_kernel<64>"
    for q, ref in zip(probs, refs):
        assert maxrel(q["C"], ref) < 1e-5


def test_input_gradient_multi_source_mask_and_accumulate(ops):
    """d_in [M, 512] = sum over six sources (4 x [M, 512] . W^T[512, 512] + 2 x [M, 128] . W^T[512, 128]) -- the first
    layer's input gradient of KuaiRec MMoE -- and a masked bf16 result (second layer's input gradient); fp32 results may
    accumulate."""
    from mmlrec_amd import _lib as L
    g = torch.Generator().manual_seed(4)
    M = 512
    srcs, ref = [], torch.zeros(M, 512, dtype=torch.float64, device=dev())
    for N in (512, 512, 512, 512, 128, 128):
        dC = rand16(g, M, N)
        Wt = rand16(g, 512, N, scale=N ** -0.5)     # W^T: [K_in, N], the reduction along its rows
        srcs.append((dC, Wt))
        ref += f64(dC) @ f64(Wt).t()
    out = torch.full((M, 512), 3.0, device=dev())
    ops.g16_tn([dict(srcs=srcs, C=out)])
    assert maxrel(out, ref) < 1e-5
    ops.g16_tn([dict(srcs=srcs, C=out, accumulate=1)])
    assert maxrel(out, 2 * ref) < 1e-5
    # ReLU derivative from the forward's sign bits, bf16 result
    dE = rand16(g, M, 256)
    W2t = rand16(g, 512, 256, scale=1 / 16)
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (M, 16), dtype=torch.int32, generator=g).to(dev())
    dH = torch.empty(M, 512, dtype=torch.bfloat16, device=dev())
    ops.g16_tn([dict(srcs=[(dE, W2t)], C=dH, mask_in=mask)])
    bits = ((mask.unsqueeze(-1) >> torch.arange(32, device=dev())) & 1).reshape(M, -1).bool()
    ref2 = (f64(dE) @ f64(W2t).t()) * bits
    assert maxrel(dH, ref2) < 1e-5 + 2.0 ** -8
    assert bool((f64(dH)[~bits] == 0).all())


@pytest.mark.parametrize("M,phases,extra", [(256, False, None), (65536, True, None), (8192 + 64, False, None),
                                            (8192 + 64, False, (128, 384)), (16384, True, (256, 128))])
def test_weight_gradient_with_bias_gradient(ops, M, phases, extra):
    """dW = dC^T A and dbias = column sums of dC over the batch, for the KuaiRec layer shapes in one launch; the batch
    reduction runs through the transposing LDS reads (ds_read_b64_tr_b16).  M = 8 256: slabs of unequal length.  Round 6:
    launches whose problems all have K % 256 == 0 run 128 x 256 tiles from 8 192 rows on (g16_nt_kernel<256>); one problem
    with K = 384 or 128 keeps the whole launch on the 128 x 128 tiles."""
    from mmlrec_amd import _lib as L_
    g = torch.Generator().manual_seed(5)
    shapes = [(512, 512), (128, 512), (256, 512), (128, 256)] + ([extra] if extra else [])
    probs, refs = [], []
    for N, K in shapes:
        dC = rand16(g, M, N)
        A = rand16(g, M, K)
        dW = torch.full((N, K), 7.0, device=dev())
        db = torch.full((N,), 7.0, device=dev())
        probs.append(dict(dC=dC, A=A, dW=dW, dbias=db))
        refs.append((f64(dC).t() @ f64(A), f64(dC).sum(0)))
    ops.g16_wgrad(probs, phases=phases)
    if not phases:
        want = "g16_nt_kernel<256>" if (M >= 8192 and extra is None) else "g16_nt_kernel<128>"
        assert L_.load().mml_g16_last_kernel().decode() in (want, "g16_reduce_kernel"), L_.load().mml_g16_last_kernel()
    for q, (rw, rb) in zip(probs, refs):
        assert maxrel(q["dW"], rw) < 2e-5, (tuple(rw.shape), maxrel(q["dW"], rw))
        assert maxrel(q["dbias"], rb) < 2e-5
    # asymmetric check: a transposed or mirrored tile cannot pass (dC and A have different widths and random entries);
    # accumulate adds to what the buffers hold
    for q in probs:
        q["accumulate"] = 1
    ops.g16_wgrad(probs)
    for q, (rw, rb) in zip(probs, refs):
        assert maxrel(q["dW"], 2 * rw) < 2e-5
        assert maxrel(q["dbias"], 2 * rb) < 2e-5
    # bitwise reproducible (fixed-order slab reduction)
    a = [q["dW"].clone() for q in probs]
    for q in probs:
        q["accumulate"] = 0
    ops.g16_wgrad(probs)
    b = [q["dW"].clone() for q in probs]
    ops.g16_wgrad(probs)
    assert all(torch.equal(x, q["dW"]) for x, q in zip(b, probs))
    assert not any(torch.equal(x, y) for x, y in zip(a, b))


def test_bad_arguments_are_rejected(ops):
    from mmlrec_amd import _lib as L
    A = torch.zeros(100, 64, dtype=torch.bfloat16, device=dev())
    W = torch.zeros(64, 64, dtype=torch.bfloat16, device=dev())
    with pytest.raises(L.MMLError):
        ops.g16_tn([dict(srcs=[(A, W)], C=torch.zeros(100, 64, device=dev()))])         # M % 128
    A = torch.zeros(128, 48, dtype=torch.bfloat16, device=dev())
    W = torch.zeros(64, 48, dtype=torch.bfloat16, device=dev())
    with pytest.raises(L.MMLError):
        ops.g16_tn([dict(srcs=[(A, W)], C=torch.zeros(128, 64, device=dev()))])         # K % 64
    with pytest.raises(L.MMLError):
        ops.g16_wgrad([dict(dC=torch.zeros(128, 64, dtype=torch.bfloat16, device=dev()),
                            A=torch.zeros(128, 128, dtype=torch.bfloat16, device=dev()),
                            dW=torch.zeros(64, 128, device=dev()))])                     # N % 128


@pytest.mark.parametrize("B,H,Gd", [(333, 200, 72), (4099, 256, 128), (64, 136, 128)])
def test_gate_kernels_with_eight_columns_per_lane_against_float64(ops, B, H, Gd):
    """Round 6: bf16 expert rows of 129..256 columns under gate inputs of at most 128 (4 experts x 2 gates: KuaiRec-32's MMoE
    in the bf16-storage mode) run on 32-lane groups with EIGHT row columns per lane (csrc/rows_fast.hip, HV = 2: 16-byte
    loads of bf16 rows, two samples per wave and trip), forward and backward -- against float64 on the same (bf16-exact)
    expert values, ragged batches and row widths that end inside a lane's eight columns."""
    g = torch.Generator().manual_seed(8)
    Ne, T = 4, 2
    E16 = [torch.randn(B, H, generator=g).relu().to(torch.bfloat16).to(dev()) for _ in range(Ne)]
    Gs = [torch.randn(B, Gd, generator=g).relu().to(dev()) for _ in range(T)]
    Wg = [(torch.randn(Ne, Gd, generator=g) * 0.1).to(dev()) for _ in range(T)]
    dmix = [torch.randn(B, H, generator=g).to(dev()) for _ in range(T)]
    for out_dtype in (torch.bfloat16, torch.float32):
        gates = [dict(G=Gs[t], Wg=Wg[t], P=torch.empty(B, Ne, device=dev()),
                      mix=torch.empty(B, H, dtype=out_dtype, device=dev()), expert=list(range(Ne))) for t in range(T)]
        ops.gate_mix_fwd(ops.make_gate_group(E16, gates, B, H))
        dE = [torch.empty(B, H, dtype=out_dtype, device=dev()) for _ in range(Ne)]
        for t in range(T):
            gates[t].update(dmix=dmix[t], dG=torch.empty(B, Gd, device=dev()), dWg=torch.empty(Ne, Gd, device=dev()),
                            g_relu=1, active=1)
        ops.gate_mix_bwd(ops.make_gate_group(E16, gates, B, H, d_experts=dE, e_relu=True), dev())
        torch.cuda.synchronize()
        Ed = torch.stack([e.double() for e in E16], 1)                         # [B, Ne, H]
        dE_ref = torch.zeros_like(Ed)
        tol = 2.0 ** -7 if out_dtype == torch.bfloat16 else 2e-5
        for t in range(T):
            p = torch.softmax(Gs[t].double() @ Wg[t].double().t(), 1)          # [B, Ne]
            mix = (p[:, :, None] * Ed).sum(1)
            assert float((gates[t]["P"].double() - p).abs().max()) < 1e-5
            assert float((gates[t]["mix"].double() - mix).abs().max()) <= tol * float(mix.abs().max())
            dl = (dmix[t].double()[:, None, :] * Ed).sum(2)                    # [B, Ne]
            dlogit = p * (dl - (p * dl).sum(1, keepdim=True))
            dG = (dlogit @ Wg[t].double()) * (Gs[t] > 0)
            assert float((gates[t]["dG"].double() - dG).abs().max()) <= 2e-5 * float(dG.abs().max())
            dW = dlogit.t() @ Gs[t].double()
            assert float((gates[t]["dWg"].double() - dW).abs().max()) <= 2e-5 * float(dW.abs().max())
            dE_ref += p[:, :, None] * dmix[t].double()[:, None, :]
        dE_ref = dE_ref * (Ed > 0)
        for x in range(Ne):
            assert float((dE[x].double() - dE_ref[:, x]).abs().max()) <= tol * float(dE_ref.abs().max())


def test_row_kernels_write_bf16_operands(ops):
    """mml_gate_group.out_bf16 / mml_head_group.dh_bf16: the tensors only GEMMs read (mix; dE, dG; dH) leave the fast row
    kernels as bf16 -- exactly the fp32 results rounded to nearest even."""
    from mmlrec_amd import _lib as L
    g = torch.Generator().manual_seed(6)
    B, H, Gd, Ne, T = 1024, 256, 128, 4, 2
    E = [torch.randn(B, H, generator=g).relu().to(dev()) for _ in range(Ne)]   # (run() reads the current binding)
    Gs = [torch.randn(B, Gd, generator=g).relu().to(dev()) for _ in range(T)]
    Wg = [(torch.randn(Ne, Gd, generator=g) * 0.1).to(dev()) for _ in range(T)]

    def run(dtype):
        gates = [dict(G=Gs[t], Wg=Wg[t], P=torch.empty(B, Ne, device=dev()), mix=torch.empty(B, H, dtype=dtype, device=dev()),
                      expert=list(range(Ne))) for t in range(T)]
        ops.gate_mix_fwd(ops.make_gate_group(E, gates, B, H))
        dmix = [torch.randn(B, H, generator=torch.Generator().manual_seed(9 + t)).to(dev()) for t in range(T)]
        dE = [torch.empty(B, H, dtype=dtype, device=dev()) for _ in range(Ne)]
        for t in range(T):
            gates[t].update(dmix=dmix[t], dG=torch.empty(B, Gd, dtype=dtype, device=dev()),
                            dWg=torch.empty(Ne, Gd, device=dev()), g_relu=1, active=1)
        ops.gate_mix_bwd(ops.make_gate_group(E, gates, B, H, d_experts=dE, e_relu=True), dev())
        return [q["mix"] for q in gates], dE, [q["dG"] for q in gates], [q["dWg"] for q in gates]
    m32, e32, g32, w32 = run(torch.float32)
    m16, e16, g16, w16 = run(torch.bfloat16)
    # expert outputs ARRIVING as bf16 (out_bf16 bit 3) are widened exactly: the fp32 run on the same (rounded) values
    E_keep = E
    E = [e.to(torch.bfloat16) for e in E_keep]
    m16b, e16b, g16b, w16b = run(torch.bfloat16)
    E = [e.float() for e in E]
    m32b, e32b, g32b, w32b = run(torch.float32)
    # (round 6: 256-wide bf16 expert rows run on 32-lane groups with eight columns per lane -- two samples per wave and
    #  trip, 16-byte loads -- the fp32 rows of the comparison run on 64-lane groups: the same sums in another order, so
    #  the rounded results agree to one bf16 step, and exactly almost everywhere)
    for a, b in zip(m32b + e32b + g32b, m16b + e16b + g16b):
        ref = a.to(torch.bfloat16)
        same = float((ref == b).float().mean())
        dev_ = float(((ref.float() - b.float()).abs() / (a.abs() * 2.0 ** -7 + 1e-6 * float(a.abs().max()))).max())
        assert same > 0.995 and dev_ <= 1.0, (same, dev_)
    for a, b in zip(w32b, w16b):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
    E = E_keep
    for a, b in zip(m32 + e32 + g32, m16 + e16 + g16):
        assert b.dtype == torch.bfloat16 and torch.equal(a.to(torch.bfloat16), b)
    for a, b in zip(w32, w16):
        assert torch.equal(a, b)
    # heads
    Hin = [torch.randn(B, 128, generator=g).relu().to(dev()) for _ in range(T)]
    w = [(torch.randn(128, generator=g) * 0.1).to(dev()) for _ in range(T)]
    y = (torch.rand(B, T, generator=g) > 0.5).float().to(dev())
    outs = {}
    for dtype in (torch.float32, torch.bfloat16):
        heads = [dict(Hin=Hin[t], w=w[t], bias=torch.zeros(1, device=dev()), dH=torch.empty(B, 128, dtype=dtype, device=dev()),
                      dw=torch.empty(128, device=dev()), dbias=torch.empty(1, device=dev()), h_relu=1) for t in range(T)]
        prob, loss = torch.empty(B, T, device=dev()), torch.zeros(1, device=dev())
        ops.head_bce_fwd_bwd(ops.make_head_group(heads, prob, y=y, loss=loss), dev())
        outs[dtype] = ([q["dH"] for q in heads], prob, loss, [q["dw"] for q in heads])
    for a, b in zip(outs[torch.float32][0], outs[torch.bfloat16][0]):
        assert torch.equal(a.to(torch.bfloat16), b)
    assert torch.equal(outs[torch.float32][1], outs[torch.bfloat16][1])
    assert torch.equal(outs[torch.float32][2], outs[torch.bfloat16][2])
