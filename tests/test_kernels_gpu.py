"""Kernel-level parity: every C-ABI entry point on the MI355X against numpy / the oracle on seeded inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mmlrec_oracle as orc  # noqa: E402


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def ops():
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import ops as o
    return o


@pytest.mark.parametrize("E,nd,B", [(8, 0, 1000), (16, 5, 333), (6, 3, 257), (8, 63, 4096)])
def test_gather_bit_exact(ops, E, nd, B):
    rng = np.random.default_rng(0)
    vocab = [1, 2, 7, 100, 1000, 50000, 3]
    F = len(vocab)
    tabs = [rng.standard_normal((v, E)).astype(np.float32) for v in vocab]
    # formula-defined bit patterns in one table: any bit flip is visible
    tabs[5] = (np.arange(vocab[5] * E, dtype=np.uint32) * np.uint32(2654435761)).view(np.float32).reshape(vocab[5], E)
    tabs[5] = np.where(np.isfinite(tabs[5]), tabs[5], np.float32(1.0)).astype(np.float32)
    idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
    idx[0, :] = 0
    idx[1, :] = np.array(vocab) - 1
    X = np.concatenate([idx.astype(np.float32), rng.random((B, nd), dtype=np.float32)], 1)
    ref = np.concatenate([tabs[f][idx[:, f]] for f in range(F)] + [X[:, F:]], 1)
    status = ops.new_status(dev())
    out = ops.gather_fwd([T(t) for t in tabs], T(X), list(range(F)), F, nd, status=status)
    ops.check_status(status)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # native int32 index variant
    out2 = ops.gather_fwd_idx32([T(t) for t in tabs], T(idx.astype(np.int32)),
                                T(X[:, F:].copy()) if nd else None)
    assert np.array_equal(out2.cpu().numpy().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("nd", [63, 62, 61, 5, 1])
def test_gather_dense_pieces_into_padded_rows(ops, nd):
    """The vectorised kernel moves the dense features four at a time; a row whose dense part is not a multiple of four
    (AE: 63 columns, reference configs_msl/config_AE.json -> K0 = 303 in rows of 304) ends in scalar stores, and the
    columns behind the row are not touched (engine.Val.kpad keeps them zero for the GEMM that reads the padded row)."""
    import mmlrec_amd._lib as L_
    rng = np.random.default_rng(nd)
    E, B = 8, 3001
    vocab = [3, 100, 5000, 70000, 2, 17]
    F = len(vocab)
    K0 = F * E + nd
    ld = (K0 + 15) // 16 * 16
    tabs = [rng.standard_normal((v, E)).astype(np.float32) for v in vocab]
    idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
    X = np.concatenate([idx.astype(np.float32), (rng.random((B, nd), dtype=np.float32) - 0.5) * 100], 1)
    ref = np.concatenate([tabs[f][idx[:, f]] for f in range(F)] + [X[:, F:]], 1)
    dt, dX = [T(t) for t in tabs], T(X)
    buf = torch.full((B + 1, ld), 12345.0, device=dev())
    ops.gather_fwd(dt, dX, list(range(F)), F, nd, out=buf[:B, :K0])
    assert L_.load().mml_gather_last_kernel().decode() == "gather_vec4_kernel"
    got = buf.cpu().numpy()
    assert np.array_equal(got[:B, :K0].view(np.uint32), ref.view(np.uint32))
    assert (got[:B, K0:] == 12345.0).all() and (got[B] == 12345.0).all()
    buf2 = torch.full((B + 1, ld), 12345.0, device=dev())
    out2, wg = ops.gather_fwd_wgmax(dt, dX, list(range(F)), F, nd, out=buf2[:B, :K0])
    assert torch.equal(buf2, buf)
    assert not torch.isnan(wg).any() and float(wg.max()) == float(np.abs(ref).max())


@pytest.mark.parametrize("E,nd,B", [(8, 0, 65536), (16, 4, 333), (8, 64, 4096)])
def test_gather_workgroup_maxima(ops, E, nd, B):
    """mml_gather_fwd_wgmax: the same output bits, and per-workgroup maxima whose maximum IS the magnitude of the output
    (fed to mml_amax_batch as one row of floats: the slot the stand-alone pass over the output would give)."""
    rng = np.random.default_rng(1)
    vocab = [3, 100, 5000, 70000]
    F = len(vocab)
    tabs = [T((rng.standard_normal((v, E)) * (10.0 ** (f - 2))).astype(np.float32)) for f, v in enumerate(vocab)]
    idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
    X = T(np.concatenate([idx.astype(np.float32), -50.0 * rng.random((B, nd), dtype=np.float32)], 1))
    ref = ops.gather_fwd(tabs, X, list(range(F)), F, nd)
    out, wg = ops.gather_fwd_wgmax(tabs, X, list(range(F)), F, nd)
    assert torch.equal(out, ref)
    assert not torch.isnan(wg).any() and float(wg.min()) >= 0.0     # every workgroup wrote its value
    assert float(wg.max()) == float(ref.abs().max())
    slots = ops.amax_slots(2, dev())
    ops.amax_batch([(wg, slots[0]), (ref, slots[1])])
    torch.cuda.synchronize()
    assert int(slots[0].max()) == int(slots[1].max())
    # E not a multiple of 4: no partial maxima (the caller keeps the stand-alone pass)
    import mmlrec_amd._lib as L_
    assert L_.load().mml_gather_wgmax_len(F, 6, 0, B) == 0


@pytest.mark.parametrize("wg_per_cu", [1, 4])
def test_gather_lds_variant_bit_exact(ops, wg_per_cu, monkeypatch):
    """north_star's "LDS-staged index dedup" (gather_lds_kernel: tables of <= 128 rows served from a per-workgroup LDS
    copy; opt-in, MMLREC_GATHER_LDS = workgroups per CU): the same bits as gather_vec4_kernel and as numpy, on a field
    mix with LDS-resident tables, HBM tables, a table that would overflow the 48 KiB LDS budget, dense columns, a ragged
    batch -- and on the reference-made golden dnn_input of the AE-30 fixture.  The symbol that ran is asserted."""
    import mmlrec_amd._lib as L_
    from conftest import load_golden
    lib = L_.load()
    rng = np.random.default_rng(7)
    E, nd, B = 8, 4, 70001   # (F E + nd = 156: rows stay 16-byte multiples, the vectorised kernels' condition)
    vocab = [2, 100, 128, 129, 5000, 100, 1, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 120, 90000]
    F = len(vocab)
    tabs = [rng.standard_normal((v, E)).astype(np.float32) for v in vocab]
    tabs[2] = (np.arange(vocab[2] * E, dtype=np.uint32) * np.uint32(2654435761)).view(np.float32).reshape(vocab[2], E)
    tabs[2] = np.where(np.isfinite(tabs[2]), tabs[2], np.float32(-3.0)).astype(np.float32)
    idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
    idx[0, :] = 0
    idx[1, :] = np.array(vocab) - 1
    X = np.concatenate([idx.astype(np.float32), rng.random((B, nd), dtype=np.float32)], 1)
    ref = np.concatenate([tabs[f][idx[:, f]] for f in range(F)] + [X[:, F:]], 1)
    dt, dX = [T(t) for t in tabs], T(X)
    monkeypatch.delenv("MMLREC_GATHER_LDS", raising=False)
    plain = ops.gather_fwd(dt, dX, list(range(F)), F, nd)
    assert lib.mml_gather_last_kernel().decode() == "gather_vec4_kernel"
    monkeypatch.setenv("MMLREC_GATHER_LDS", str(wg_per_cu))
    status = ops.new_status(dev())
    out = ops.gather_fwd(dt, dX, list(range(F)), F, nd, status=status)
    assert lib.mml_gather_last_kernel().decode() == "gather_lds_kernel"
    ops.check_status(status)
    assert torch.equal(out, plain)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # out-of-range indices are flagged by this kernel too
    bad = dX.clone()
    bad[5, 1] = 100.0
    ops.gather_fwd(dt, bad, list(range(F)), F, nd, status=status)
    with pytest.raises(IndexError):
        ops.check_status(status)
    # the reference's own dnn_input (tests/golden/mmoe_ae30.npz: 30 fields of 2 .. 96 rows -- every table LDS-resident)
    g = load_golden("mmoe_ae30")
    names = [str(n) for n in g["sparse_names"]]
    gt = [T(g[f"state/embedding_dict.{n}.weight"]) for n in names]
    out = ops.gather_fwd(gt, T(g["X0"]), list(range(len(names))), len(names), 0)
    assert lib.mml_gather_last_kernel().decode() == "gather_lds_kernel"
    assert np.array_equal(out.cpu().numpy(), g["dnn_input"])
    # the variants that mark rows / leave workgroup maxima keep the plain kernel (they have no LDS form)
    ops.gather_fwd_wgmax(dt, dX, list(range(F)), F, nd)
    assert lib.mml_gather_last_kernel().decode() == "gather_vec4_kernel"


def test_gather_out_of_range_sets_status(ops):
    tabs = [torch.zeros(10, 8, device=dev())]
    X = torch.tensor([[3.0], [10.0], [-1.0]], device=dev())
    status = ops.new_status(dev())
    ops.gather_fwd(tabs, X, [0], status=status)
    with pytest.raises(IndexError):
        ops.check_status(status)


def test_gather_index_near_2pow24(ops):
    V = (1 << 24)
    tab = torch.arange(V, device=dev(), dtype=torch.float32).reshape(V, 1).repeat(1, 4)
    X = torch.tensor([[float(V - 1)], [float(V - 2)], [0.0]], device=dev())
    out = ops.gather_fwd([tab], X, [0])
    assert out[:, 0].tolist() == [float(V - 1), float(V - 2), 0.0]


@pytest.mark.parametrize("E,B", [(8, 5000), (16, 777), (5, 300)])
def test_scatter_matches_index_add(ops, E, B):
    rng = np.random.default_rng(1)
    vocab = [2, 30, 1000, 20000]
    F = len(vocab)
    idx = np.stack([np.minimum((v ** rng.random(B)).astype(np.int64), v - 1) for v in vocab], 1)
    X = idx.astype(np.float32)
    d_out = rng.standard_normal((B, F * E + 3)).astype(np.float32)
    ref = []
    for f, v in enumerate(vocab):
        g = np.zeros((v, E), np.float64)
        np.add.at(g, idx[:, f], d_out[:, f * E:(f + 1) * E].astype(np.float64))
        ref.append(g)
    gt = [torch.zeros(v, E, device=dev()) for v in vocab]
    seen = [torch.zeros((v + 31) // 32, dtype=torch.int32, device=dev()) for v in vocab]
    rowbase = np.concatenate([[0], np.cumsum(vocab)]).tolist()
    touched = torch.full((B * F,), -1, dtype=torch.int32, device=dev())
    count = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.scatter_bwd(gt, T(X), list(range(F)), T(d_out), seen=seen, rowbase=rowbase, touched=touched,
                    touched_count=count)
    for f in range(F):
        assert rel(gt[f].cpu().numpy(), ref[f]) < 1e-5
    n = int(count.item())
    got = np.sort(touched[:n].cpu().numpy())
    want = np.sort(np.concatenate([np.unique(idx[:, f]) + rowbase[f] for f in range(F)]))
    assert np.array_equal(got, want)


def _gemm_case(rng, M, N, K):
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    return A, W, b


@pytest.mark.parametrize("M,shapes", [
    (257, [(256, 240), (64, 240)]),          # AE-30 layer-1 experts + gates, ragged M
    (512, [(128, 256)] * 4),                 # expert layer 2
    (130, [(64, 128), (64, 128)]),           # towers
    (100, [(5, 56), (8, 56), (1, 33)]),      # skinny / odd sizes, unaligned K
    (4096, [(256, 512), (128, 512)]),        # KuaiRec E=16 first layer
])
def test_gemm_fwd(ops, M, shapes):
    from mmlrec_amd import _lib as L
    rng = np.random.default_rng(2)
    A0 = rng.standard_normal((M, shapes[0][1])).astype(np.float32)
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        _, W, b = _gemm_case(rng, M, N, K)
        A = A0 if K == A0.shape[1] else rng.standard_normal((M, K)).astype(np.float32)
        act = [L.ACT_RELU, L.ACT_NONE, L.ACT_SIGMOID, L.ACT_SIGMOID2][i % 4]
        z = A.astype(np.float64) @ W.T.astype(np.float64) + b
        ref = {L.ACT_RELU: np.maximum(z, 0), L.ACT_NONE: z, L.ACT_SIGMOID: 1 / (1 + np.exp(-z)),
               L.ACT_SIGMOID2: 2 / (1 + np.exp(-z))}[act]
        Cc = torch.empty(M, N, device=dev())
        probs.append(dict(A=T(A), W=T(W), bias=T(b), C=Cc, act=act))
        refs.append(ref)
    ops.gemm_fwd(probs)
    for p, r in zip(probs, refs):
        assert rel(p["C"].cpu().numpy(), r) < 2e-5


def test_gemm_fwd_kn_layout_and_strided_output(ops):
    from mmlrec_amd import _lib as L
    rng = np.random.default_rng(3)
    M, N, K = 300, 96, 64
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wkn = rng.standard_normal((K, N)).astype(np.float32)
    big = torch.zeros(M, 3 * N, device=dev())
    ops.gemm_fwd([dict(A=T(A), W=T(Wkn), bias=None, C=big[:, N:2 * N], act=L.ACT_NONE, w_kn=1)])
    ref = A.astype(np.float64) @ Wkn.astype(np.float64)
    assert rel(big[:, N:2 * N].cpu().numpy(), ref) < 2e-5
    assert float(big[:, :N].abs().max()) == 0.0 and float(big[:, 2 * N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,K,srcN,w_kn", [(257, 240, [256, 256, 64], 0), (512, 128, [64], 0), (100, 56, [5, 8], 0),
                                           (300, 64, [128], 1)])
def test_gemm_dgrad(ops, M, K, srcN, w_kn):
    from mmlrec_amd import _lib as L
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((M, K)).astype(np.float32)
    Y = np.maximum(Y, 0)
    srcs, ref = [], np.zeros((M, K))
    for N in srcN:
        dC = rng.standard_normal((M, N)).astype(np.float32)
        W = rng.standard_normal((N, K)).astype(np.float32)
        ref += dC.astype(np.float64) @ W.astype(np.float64)
        srcs.append((T(dC), T(W.T.copy()) if w_kn else T(W), w_kn))
    prev = rng.standard_normal((M, K)).astype(np.float32)
    dA = T(prev)
    ops.gemm_dgrad([dict(dA=dA, Y=T(Y), act=L.ACT_RELU, accumulate=1, srcs=srcs)])
    assert rel(dA.cpu().numpy(), prev + ref * (Y > 0)) < 2e-5
    dA2 = torch.empty(M, K, device=dev())
    ops.gemm_dgrad([dict(dA=dA2, Y=None, srcs=srcs)])
    assert rel(dA2.cpu().numpy(), ref) < 2e-5


@pytest.mark.parametrize("M,shapes", [(1000, [(256, 240), (64, 240)]), (4096, [(128, 256)] * 3), (333, [(5, 56), (1, 33)]),
                                      (70000, [(64, 128)])])
def test_gemm_wgrad(ops, M, shapes):
    rng = np.random.default_rng(5)
    probs, refs = [], []
    for N, K in shapes:
        dC = rng.standard_normal((M, N)).astype(np.float32)
        A = rng.standard_normal((M, K)).astype(np.float32)
        probs.append(dict(dC=T(dC), A=T(A), dW=torch.empty(N, K, device=dev()), dbias=torch.empty(N, device=dev())))
        refs.append((dC.astype(np.float64).T @ A.astype(np.float64), dC.astype(np.float64).sum(0)))
    ops.gemm_wgrad(probs)
    first = [p["dW"].clone() for p in probs]
    for p, (rw, rb) in zip(probs, refs):
        assert rel(p["dW"].cpu().numpy(), rw) < 2e-5
        assert rel(p["dbias"].cpu().numpy(), rb) < 2e-5
    ops.gemm_wgrad(probs)  # bitwise reproducible (fixed-order slab reduction)
    for p, f in zip(probs, first):
        assert torch.equal(p["dW"], f)


def test_gemm_wgrad_kn_layout_accumulate(ops):
    rng = np.random.default_rng(6)
    M, N, K = 900, 40, 72
    dC = rng.standard_normal((M, N)).astype(np.float32)
    A = rng.standard_normal((M, K)).astype(np.float32)
    prev = rng.standard_normal((K, N)).astype(np.float32)
    dW = T(prev)
    db = torch.zeros(N, device=dev())
    ops.gemm_wgrad([dict(dC=T(dC), A=T(A), dW=dW, dbias=db, accumulate=1, w_kn=1)])
    assert rel(dW.cpu().numpy(), prev + A.astype(np.float64).T @ dC.astype(np.float64)) < 2e-5
    assert rel(db.cpu().numpy(), dC.astype(np.float64).sum(0)) < 2e-5


def _gate_setup(rng, B, H, nexp, gate_sets, Gd):
    E = [np.maximum(rng.standard_normal((B, H)), 0).astype(np.float32) for _ in range(nexp)]
    gates = []
    for members in gate_sets:
        G = np.maximum(rng.standard_normal((B, Gd)), 0).astype(np.float32)
        Wg = (rng.standard_normal((len(members), Gd)) * 0.3).astype(np.float32)
        gates.append((G, Wg, members))
    return E, gates


@pytest.mark.parametrize("B,H,nexp,gate_sets,Gd", [
    (1000, 128, 4, [[0, 1, 2, 3], [0, 1, 2, 3]], 64),                        # MMoE
    (515, 32, 8, [[0, 1, 2, 6, 7], [3, 4, 5, 6, 7], [0, 1, 2, 3, 4, 5, 6, 7]], 16),  # PLE CGC
    (70, 300, 3, [[2, 0]], 100),                                             # odd sizes
    # round 6 (packed sums of the 4 experts x 2 gates kernels): KuaiRec's widths (one sample per wave), one gate only,
    # three experts in another order, a second gate over a subset, 16-lane groups
    (300, 256, 4, [[0, 1, 2, 3], [0, 1, 2, 3]], 128),
    (257, 64, 3, [[2, 0, 1]], 32),
    (129, 128, 4, [[0, 1, 2, 3], [3, 1]], 64),
    (1000, 64, 4, [[0, 1, 2, 3], [0, 1, 2, 3]], 16),
])
def test_gate_mix_fwd_bwd(ops, B, H, nexp, gate_sets, Gd):
    rng = np.random.default_rng(7)
    E, gates = _gate_setup(rng, B, H, nexp, gate_sets, Gd)
    Et = [T(e) for e in E]
    gd = []
    for G, Wg, members in gates:
        gd.append(dict(G=T(G), Wg=T(Wg), P=torch.empty(B, len(members), device=dev()),
                       mix=torch.empty(B, H, device=dev()), expert=members))
    ops.gate_mix_fwd(ops.make_gate_group(Et, gd, B, H))
    refs = []
    for (G, Wg, members), q in zip(gates, gd):
        p, mix = orc.gate_mix_fwd(orc.linear_fwd(G, Wg), np.stack([E[x] for x in members], 1))
        assert rel(q["P"].cpu().numpy(), p) < 1e-5
        assert rel(q["mix"].cpu().numpy(), mix) < 1e-5
        refs.append(p)
    # backward; the last gate of the PLE-like case is inactive (its mix is never consumed)
    dE_ref = [np.zeros((B, H), np.float64) for _ in range(nexp)]
    dEt = [torch.full((B, H), 7.0, device=dev()) for _ in range(nexp)]
    for i, ((G, Wg, members), q, p) in enumerate(zip(gates, gd, refs)):
        active = not (len(gate_sets) == 3 and i == 2)
        dmix = rng.standard_normal((B, H)).astype(np.float32)
        q.update(dmix=T(dmix), dG=torch.empty(B, Gd, device=dev()), dWg=torch.empty(len(members), Gd, device=dev()),
                 active=int(active), g_relu=1)
        if not active:
            continue
        ex = np.stack([E[x] for x in members], 1)
        dlog, de = orc.gate_mix_bwd(p, ex, dmix)
        for s, x in enumerate(members):
            dE_ref[x] += de[:, s]
        q["ref_dG"] = (dlog @ Wg) * (G > 0)
        q["ref_dWg"] = dlog.T.astype(np.float64) @ G.astype(np.float64)
    ops.gate_mix_bwd(ops.make_gate_group(Et, gd, B, H, d_experts=dEt), dev())
    for x in range(nexp):
        assert rel(dEt[x].cpu().numpy(), dE_ref[x] * (E[x] > 0)) < 2e-5
    for q in gd:
        if "ref_dG" in q:
            assert rel(q["dG"].cpu().numpy(), q["ref_dG"]) < 2e-5
            assert rel(q["dWg"].cpu().numpy(), q["ref_dWg"]) < 2e-5
    # the two-launch form (mml_gate_mix_bwd_phase: row kernel, then the reduction of its partial sums): the same bits
    act = [q for q in gd if "ref_dG" in q]
    one_w, one_g = [q["dWg"].clone() for q in act], [q["dG"].clone() for q in act]
    for q in act:
        q["dWg"].fill_(float("nan"))
    dE2 = [torch.full((B, H), 7.0, device=dev()) for _ in range(nexp)]
    ops.gate_mix_bwd(ops.make_gate_group(Et, gd, B, H, d_experts=dE2), dev(), phases=True)
    for q, w_, g_ in zip(act, one_w, one_g):
        # (the generic kernel -- H = 300 here -- sums dWg with LDS float atomics: its order, hence the last bit, varies run to run)
        assert rel(q["dWg"].cpu().numpy(), w_.cpu().numpy()) < 1e-6 and torch.equal(q["dG"], g_)
    for a_, b_ in zip(dEt, dE2):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("B,H,T_,masked", [(1000, 64, 2, False), (333, 16, 4, True), (64, 200, 1, False)])
def test_head_bce(ops, B, H, T_, masked):
    rng = np.random.default_rng(8)
    y = (rng.random((B, T_)) < 0.4).astype(np.float32)
    mask = (rng.random((B, 2)) < 0.5).astype(np.float32) if masked else None
    heads, ref_loss = [], 0.0
    prob = torch.empty(B, T_, device=dev())
    loss = torch.zeros(1, device=dev())
    for t in range(T_):
        Hin = np.maximum(rng.standard_normal((B, H)), 0).astype(np.float32)
        w = (rng.standard_normal(H) * 1.5 / np.sqrt(H)).astype(np.float32)
        bias = rng.standard_normal(1).astype(np.float32)
        logit = (Hin.astype(np.float64) @ w + bias).astype(np.float32)
        p = orc.sigmoid(logit)  # fp32 like the reference: saturates to exactly 0/1 for |logit| > ~17
        m = mask[:, t % 2] if masked else np.float32(1.0)
        pm = (p * m).astype(np.float32)
        ref_loss += orc.bce_sum(pm, y[:, t])
        pm64, p64 = pm.astype(np.float64), p.astype(np.float64)
        dlogit = (pm64 - y[:, t]) / np.maximum(pm64 * (1 - pm64), 1e-12) * m * p64 * (1 - p64)
        heads.append(dict(Hin=T(Hin), w=T(w), bias=T(bias), dH=torch.empty(B, H, device=dev()),
                          dw=torch.empty(H, device=dev()), dbias=torch.empty(1, device=dev()), h_relu=1,
                          mask_col=(t % 2 if masked else -1),
                          ref=(pm, dlogit[:, None] * w[None, :] * (Hin > 0), dlogit @ Hin.astype(np.float64),
                               dlogit.sum())))
    ops.head_bce_fwd_bwd(ops.make_head_group(heads, prob, y=T(y), mask=T(mask) if masked else None, loss=loss), dev())
    assert abs(float(loss.item()) - ref_loss) / ref_loss < 1e-4
    for t, h in enumerate(heads):
        pm, dH, dw, db = h["ref"]
        assert rel(prob[:, t].cpu().numpy(), pm) < 1e-5
        assert rel(h["dH"].cpu().numpy(), dH) < 2e-5
        assert rel(h["dw"].cpu().numpy(), dw) < 2e-5
        assert abs(float(h["dbias"].item()) - db) < 2e-5 * max(abs(db), 1.0)
    prob2 = torch.empty(B, T_, device=dev())
    ops.head_fwd(ops.make_head_group(heads, prob2, mask=T(mask) if masked else None))
    assert torch.equal(prob, prob2)
    # the two-launch form (mml_head_bce_fwd_bwd_phase): the same bits
    keep = [(h["dw"].clone(), h["dbias"].clone(), h["dH"].clone()) for h in heads]
    for h in heads:
        h["dw"].fill_(float("nan"))
        h["dbias"].fill_(float("nan"))
    loss2 = torch.full((1,), float("nan"), device=dev())
    ops.head_bce_fwd_bwd(ops.make_head_group(heads, prob, y=T(y), mask=T(mask) if masked else None, loss=loss2), dev(),
                         phases=True)
    assert float(loss2.item()) == float(loss.item())
    for h, (w_, b_, d_) in zip(heads, keep):
        assert torch.equal(h["dw"], w_) and torch.equal(h["dbias"], b_) and torch.equal(h["dH"], d_)


@pytest.mark.parametrize("B,H,T_,gate_act", [(1000, 128, 4, "sigmoid2"), (65536, 128, 4, "sigmoid2"), (333, 64, 2, "none"),
                                            (4099, 16, 3, "sigmoid")])
def test_gated_head_bce(ops, B, H, T_, gate_act):
    """Gated heads (round 6; PepNet's last PPNet layer, reference model/pepnet.py:72-78 + :139-140): the head's input is
    Hin (.) gate, formed inside the kernel -- probabilities, loss, dH = dlogit w gate relu'(Hin), dgate = dlogit w Hin
    act'(gate), dw = sum dlogit Hin gate, dbias against float64; the forward-only call and the two-launch form give the
    same bits; a mixed group (one plain head next to the gated ones); the magnitude slots bound what was stored."""
    from mmlrec_amd import _lib as L
    rng = np.random.default_rng(B + H)
    y = (rng.random((B, T_)) < 0.4).astype(np.float32)
    acts = {"none": L.ACT_NONE, "sigmoid": L.ACT_SIGMOID, "sigmoid2": L.ACT_SIGMOID2}
    heads, ref_loss = [], 0.0
    prob = torch.empty(B, T_, device=dev())
    loss = torch.zeros(1, device=dev())
    for t in range(T_):
        Hin = np.maximum(rng.standard_normal((B, H)), 0).astype(np.float32)
        z = rng.standard_normal((B, H))
        gated = not (t == T_ - 1 and T_ == 3)          # (the (4099, 16, 3) case: its last head is a plain one)
        if gate_act == "none":
            g = z.astype(np.float32)
            dact = np.ones_like(z)
        elif gate_act == "sigmoid":
            g = (1 / (1 + np.exp(-z))).astype(np.float32)
            dact = g.astype(np.float64) * (1 - g.astype(np.float64))
        else:
            g = (2 / (1 + np.exp(-z))).astype(np.float32)
            dact = g.astype(np.float64) * (1 - g.astype(np.float64) / 2)
        if not gated:
            g, dact = np.ones_like(g), np.ones_like(dact)
        # (logits of a few units: beyond ~16 the fp32 probability is one ulp from 1 and a last-bit difference of the logit
        # moves that sample's clamped-log loss by 0.7 -- 83 when it reaches exactly 1 --, which tests nothing but rounding)
        w = (rng.standard_normal(H) * 0.5 / np.sqrt(H)).astype(np.float32)
        bias = rng.standard_normal(1).astype(np.float32)
        he = Hin.astype(np.float64) * g.astype(np.float64)
        logit = (he @ w + bias).astype(np.float32)
        p = orc.sigmoid(logit)
        ref_loss += orc.bce_sum(p, y[:, t])
        p64 = p.astype(np.float64)
        dlogit = (p64 - y[:, t]) / np.maximum(p64 * (1 - p64), 1e-12) * p64 * (1 - p64)
        q = dict(Hin=T(Hin), w=T(w), bias=T(bias), dH=torch.full((B, H), float("nan"), device=dev()),
                 dw=torch.empty(H, device=dev()), dbias=torch.empty(1, device=dev()), h_relu=1, mask_col=-1,
                 ref=(p, dlogit[:, None] * w[None, :] * g * (Hin > 0), dlogit[:, None] * w[None, :] * Hin * dact,
                      dlogit @ he, dlogit.sum()))
        if gated:
            q.update(gate=T(g), dgate=torch.full((B, H), float("nan"), device=dev()), gate_act=acts[gate_act])
        heads.append(q)
    yt = T(y)   # (the group holds raw pointers: the label tensor must outlive the launch)
    grp = ops.make_head_group(heads, prob, y=yt, mask=None, loss=loss)
    slots = ops.amax_slots(2, dev())
    grp.amax_dH, grp.amax_dG = slots[0].data_ptr(), slots[1].data_ptr()
    ops.head_bce_fwd_bwd(grp, dev())
    assert abs(float(loss.item()) - ref_loss) / ref_loss < 1e-4
    for t, h in enumerate(heads):
        p, dH, dG, dw, db = h["ref"]
        assert rel(prob[:, t].cpu().numpy(), p) < 1e-5
        assert rel(h["dH"].cpu().numpy(), dH) < 2e-5
        assert rel(h["dw"].cpu().numpy(), dw) < 2e-5
        assert abs(float(h["dbias"].item()) - db) < 2e-5 * max(abs(db), 1.0)
        if "gate" in h:
            assert rel(h["dgate"].cpu().numpy(), dG) < 2e-5
    am_h = max(float(h["dH"].abs().max()) for h in heads)
    am_g = max(float(h["dgate"].abs().max()) for h in heads if "gate" in h)
    for s_, am in ((slots[0], am_h), (slots[1], am_g)):
        v = float(torch.max(s_.view(torch.float32)))
        assert v >= am and v <= am * (1 + 1e-6)
    prob2 = torch.empty(B, T_, device=dev())
    ops.head_fwd(ops.make_head_group(heads, prob2))
    assert torch.equal(prob, prob2)
    keep = [(h["dw"].clone(), h["dbias"].clone(), h["dH"].clone(), h["dgate"].clone() if "gate" in h else None) for h in heads]
    for h in heads:
        h["dw"].fill_(float("nan"))
        h["dbias"].fill_(float("nan"))
    loss2 = torch.full((1,), float("nan"), device=dev())
    ops.head_bce_fwd_bwd(ops.make_head_group(heads, prob, y=T(y), mask=None, loss=loss2), dev(), phases=True)
    assert float(loss2.item()) == float(loss.item())
    for h, (w_, b_, d_, g_) in zip(heads, keep):
        assert torch.equal(h["dw"], w_) and torch.equal(h["dbias"], b_) and torch.equal(h["dH"], d_)
        assert g_ is None or torch.equal(h["dgate"], g_)


def test_gated_head_needs_the_fast_row_kernel(ops):
    """A width the fast row kernel does not serve (H % 4 != 0) is refused, not silently computed without the gate."""
    from mmlrec_amd import _lib as L
    B, H = 64, 30
    rng = np.random.default_rng(1)
    h = dict(Hin=T(rng.standard_normal((B, H)).astype(np.float32)), w=T(rng.standard_normal(H).astype(np.float32)),
             bias=T(np.zeros(1, np.float32)), gate=T(rng.standard_normal((B, H)).astype(np.float32)), gate_act=L.ACT_NONE,
             mask_col=-1)
    with pytest.raises(L.MMLError):
        ops.head_fwd(ops.make_head_group([h], torch.empty(B, 1, device=dev())))


@pytest.mark.parametrize("H", [64, 200])
def test_head_bce_loss_propagates_nan(ops, H):
    """F.binary_cross_entropy clamps its log terms with torch.clamp(.., min=-100), which keeps a NaN (the oracle's
    np.maximum does too); fmaxf alone returned -100 and a diverged model kept reporting a finite loss.  Both head kernels
    (the row-fast one for H <= 128 multiples of 16, the generic one)."""
    B = 256
    rng = np.random.default_rng(3)
    Hin = np.maximum(rng.standard_normal((B, H)), 0).astype(np.float32)
    Hin[17, 3] = np.nan
    w = (rng.standard_normal(H) / np.sqrt(H)).astype(np.float32)
    y = (rng.random((B, 1)) < 0.4).astype(np.float32)
    prob = torch.empty(B, 1, device=dev())
    loss = torch.zeros(1, device=dev())
    heads = [dict(Hin=T(Hin), w=T(w), bias=T(np.zeros(1, np.float32)), dH=torch.empty(B, H, device=dev()),
                  dw=torch.empty(H, device=dev()), dbias=torch.empty(1, device=dev()), h_relu=1, mask_col=-1)]
    ops.head_bce_fwd_bwd(ops.make_head_group(heads, prob, y=T(y), mask=None, loss=loss), dev())
    assert np.isnan(float(loss.item()))
    p = prob.cpu().numpy()[:, 0]
    assert np.isnan(p[17]) and np.isfinite(np.delete(p, 17)).all()
    logit = (Hin.astype(np.float64) @ w).astype(np.float32)
    assert np.isnan(orc.bce_sum(orc.sigmoid(logit), y[:, 0]))  # the oracle agrees


@pytest.mark.parametrize("B,levels", [(65536, 1), (4096, 1), (3001, 2)])
def test_rows_reduce_batch_equals_the_phase_two_calls(ops, B, levels):
    """mml_rows_reduce_batch: the reductions of a head group and of `levels` gate groups (MMoE: one; a PLE: one per level)
    in ONE launch give what the groups' own phase-2 calls give -- dw / dbias / loss / dWg bit for bit at these sizes
    (the same reduction kernel either way), row outputs untouched."""
    rng = np.random.default_rng(B + levels)
    H, T_, Gd, nexp = 64, 2, 64, 4
    y = T((rng.random((B, T_)) < 0.4).astype(np.float32))

    def head_group(loss, prob):
        heads = []
        for t in range(T_):
            r = np.random.default_rng(100 + t)
            heads.append(dict(Hin=T(np.maximum(r.standard_normal((B, H)), 0).astype(np.float32)),
                              w=T((r.standard_normal(H) / np.sqrt(H)).astype(np.float32)),
                              bias=T(r.standard_normal(1).astype(np.float32)), dH=torch.empty(B, H, device=dev()),
                              dw=torch.full((H,), float("nan"), device=dev()),
                              dbias=torch.full((1,), float("nan"), device=dev()), h_relu=1, mask_col=-1))
        return heads, ops.make_head_group(heads, prob, y=y, mask=None, loss=loss)

    def gate_group(level):
        r = np.random.default_rng(200 + level)
        E, gates = _gate_setup(r, B, 128, nexp, [[0, 1, 2, 3], [0, 1, 2, 3]], Gd)
        Et = [T(e) for e in E]
        gd = []
        for G, Wg, members in gates:
            gd.append(dict(G=T(G), Wg=T(Wg), P=torch.empty(B, len(members), device=dev()),
                           mix=torch.empty(B, 128, device=dev()), expert=members))
        ops.gate_mix_fwd(ops.make_gate_group(Et, gd, B, 128))
        for q in gd:
            q.update(dmix=T(r.standard_normal((B, 128)).astype(np.float32)), dG=torch.empty(B, Gd, device=dev()),
                     dWg=torch.full((len(q["expert"]), Gd), float("nan"), device=dev()), active=1, g_relu=1)
        dE = [torch.empty(B, 128, device=dev()) for _ in range(nexp)]
        return gd, dE, ops.make_gate_group(Et, gd, B, 128, d_experts=dE), Et  # (Et: the group holds raw pointers)

    # one by one
    loss1, prob1 = torch.zeros(1, device=dev()), torch.empty(B, T_, device=dev())
    heads1, hg1 = head_group(loss1, prob1)
    ops.head_bce_fwd_bwd(hg1, dev(), phases=True)
    gates1 = [gate_group(lv) for lv in range(levels)]
    for gd, dE, gg, _ in gates1:
        ops.gate_mix_bwd(gg, dev(), phases=True)
    # phase 1 each, then ONE reduction launch
    loss2, prob2 = torch.zeros(1, device=dev()), torch.empty(B, T_, device=dev())
    heads2, hg2 = head_group(loss2, prob2)
    gates2 = [gate_group(lv) for lv in range(levels)]
    ops.rows_phase1_then_batched_reduce([hg2], [g[2] for g in gates2], dev())
    torch.cuda.synchronize()
    assert torch.equal(loss1, loss2) and float(loss1) > 0 and torch.equal(prob1, prob2)
    for a, b in zip(heads1, heads2):
        assert torch.equal(a["dw"], b["dw"]) and torch.equal(a["dbias"], b["dbias"]) and torch.equal(a["dH"], b["dH"])
        assert not torch.isnan(a["dw"]).any()
    for (gd1, dE1, _, _), (gd2, dE2, _, _) in zip(gates1, gates2):
        for a, b in zip(gd1, gd2):
            assert torch.equal(a["dWg"], b["dWg"]) and torch.equal(a["dG"], b["dG"]) and not torch.isnan(a["dWg"]).any()
        for a, b in zip(dE1, dE2):
            assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["sgd", "adam", "adagrad", "rmsprop"])
def test_optimizer_dense_matches_oracle(ops, kind):
    rng = np.random.default_rng(9)
    sizes = [7, 1024, 100003, 8 * 50000]
    params = {str(i): rng.standard_normal(n).astype(np.float32) for i, n in enumerate(sizes)}
    tp = {k: T(v) for k, v in params.items()}
    s1 = {k: torch.zeros_like(v) for k, v in tp.items()}
    s2 = {k: torch.zeros_like(v) for k, v in tp.items()}
    opt = orc.DenseOptimizer(kind, 0.01)
    step_dev = torch.zeros(1, dtype=torch.int32, device=dev())
    for step in range(1, 4):
        grads = {k: (rng.standard_normal(v.shape) * (rng.random(v.shape) < 0.3)).astype(np.float32)
                 for k, v in params.items()}
        opt.step(params, grads)
        tg = {k: T(v) for k, v in grads.items()}
        ops.counter_update(step_dev, 1)
        hyper = ops.make_hyper(kind, 0.01, step=step if step % 2 else 0, step_dev=None if step % 2 else step_dev,
                               zero_grad=True)
        ops.opt_step_dense([(tp[k], tg[k], s1[k] if kind != "sgd" else None, s2[k] if kind == "adam" else None)
                            for k in params], hyper)
        for k in params:
            assert rel(tp[k].cpu().numpy(), params[k]) < 2e-6, (kind, step, k)
            assert float(tg[k].abs().max()) == 0.0


@pytest.mark.parametrize("kind,E", [("adam", 8), ("adagrad", 16), ("adam", 4)])
def test_optimizer_dense_marked_gradients(ops, kind, E):
    """mml_opt_tensor.grad_marks: the scatter marks the rows it adds to (row_marks without a touched list), the
    streaming dense launch does not read the gradient of unmarked rows and clears the marks -- bitwise the plain dense
    step (an unmarked row's gradient IS zero), gradients re-zeroed, marks all-zero afterwards."""
    rng = np.random.default_rng(12)
    vocab = [(1 << 24) // E + 37, 1000]          # one streaming table (>= 2^24 parameters) and a small one
    F, B = len(vocab), 5000
    g = torch.Generator(device="cpu").manual_seed(0)
    tabs = [torch.randn(v, E, generator=g).to(dev()) for v in vocab]
    ref = [t.clone() for t in tabs]
    st = [[torch.rand(v, E, generator=g).to(dev()) for v in vocab] for _ in range(2)]
    st_ref = [[t.clone() for t in s_] for s_ in st]
    grads = [torch.zeros(v, E, device=dev()) for v in vocab]
    grads_ref = [torch.zeros(v, E, device=dev()) for v in vocab]
    marks = torch.zeros(ops.marks_bytes(vocab), dtype=torch.uint8, device=dev())
    base = [0, (vocab[0] + 31) // 32 * 32]
    for step in (1, 2):
        X = np.stack([np.r_[0, vocab[f] - 1, rng.integers(0, vocab[f], B - 2)] for f in range(F)], 1).astype(np.float32)
        d_out = T(rng.standard_normal((B, F * E)).astype(np.float32))
        ops.scatter_bwd(grads, T(X), list(range(F)), d_out, marks=marks)           # marks only: no touched list
        ops.scatter_bwd(grads_ref, T(X), list(range(F)), d_out)
        for f in range(F):
            m = marks[base[f]:base[f] + vocab[f]].cpu().numpy().astype(bool)
            want = np.zeros(vocab[f], bool)
            want[X[:, f].astype(np.int64)] = True
            assert np.array_equal(m, want)
        hyper = ops.make_hyper(kind, 0.01, step=step, zero_grad=True)
        s2 = (lambda s_, f: s_[1][f]) if kind == "adam" else (lambda s_, f: None)
        ops.opt_step_dense([(tabs[0], grads[0], st[0][0], s2(st, 0), None, None, marks[base[0]:base[0] + vocab[0]])],
                           hyper)
        ops.opt_step_dense([(ref[0], grads_ref[0], st_ref[0][0], s2(st_ref, 0))], hyper)
        # (scatter atomics land in a different order in the two accumulators: compare to atomic-order noise, and the
        # rows no sample touched bit for bit)
        hit = np.zeros(vocab[0], bool)
        hit[X[:, 0].astype(np.int64)] = True
        untouched = torch.from_numpy(~hit).to(dev())
        assert torch.equal(tabs[0][untouched], ref[0][untouched])
        assert rel(tabs[0].cpu().numpy(), ref[0].cpu().numpy()) < 1e-6
        assert rel(st[0][0].cpu().numpy(), st_ref[0][0].cpu().numpy()) < 1e-6
        assert float(grads[0].abs().max()) == 0.0
        assert int(marks[base[0]:base[0] + vocab[0]].max()) == 0      # cleared by the optimizer
        assert int(marks[base[1]:base[1] + vocab[1]].max()) == 1      # the small table's marks are not consumed here
        marks[base[1]:].zero_()
        grads[1].zero_()
        grads_ref[1].zero_()


@pytest.mark.parametrize("cap", [0, 1 << 20, 300])
def test_optimizer_dense_marked_gradients_many_tables_one_launch(ops, cap):
    """Round 6: every table of a model in ONE marked streaming launch (more than four tensors: a 1-D grid whose workgroups
    are dealt to the tensors in proportion to their sizes) -- bitwise the per-table plain dense steps, gradients re-zeroed,
    marks cleared; with and without a workgroup cap (the capped form keeps several chunks in flight per thread)."""
    rng = np.random.default_rng(21)
    E = 8
    vocab = [(1 << 24) // E + 5, 1, 37, 300000, 4097, 64, 250001]
    F, B = len(vocab), 3000
    g = torch.Generator(device="cpu").manual_seed(3)
    tabs = [torch.randn(v, E, generator=g).to(dev()) for v in vocab]
    ref = [t.clone() for t in tabs]
    st = [[torch.rand(v, E, generator=g).to(dev()) for v in vocab] for _ in range(2)]
    st_ref = [[t.clone() for t in s_] for s_ in st]
    grads = [torch.zeros(v, E, device=dev()) for v in vocab]
    marks = torch.zeros(ops.marks_bytes(vocab), dtype=torch.uint8, device=dev())
    base = np.concatenate([[0], np.cumsum([(v + 31) // 32 * 32 for v in vocab])]).tolist()
    for step in (1, 2, 3):
        X = np.stack([np.r_[0, vocab[f] - 1, rng.integers(0, vocab[f], B - 2)] for f in range(F)], 1).astype(np.float32)
        d_out = T(rng.standard_normal((B, F * E)).astype(np.float32))
        ops.scatter_bwd(grads, T(X), list(range(F)), d_out, marks=marks)
        grads_ref = [g_.clone() for g_ in grads]              # the SAME accumulators: the comparison is bitwise
        hyper = ops.make_hyper("adam", 0.01, step=step, zero_grad=True, max_blocks=cap)
        ops.opt_step_dense([(tabs[f], grads[f], st[0][f], st[1][f], None, None, marks[base[f]:base[f] + vocab[f]])
                            for f in range(F)], hyper)
        plain = ops.make_hyper("adam", 0.01, step=step, zero_grad=True)
        for f in range(F):
            ops.opt_step_dense([(ref[f], grads_ref[f], st_ref[0][f], st_ref[1][f])], plain)
        for f in range(F):
            assert torch.equal(tabs[f], ref[f]), (step, f)
            assert torch.equal(st[0][f], st_ref[0][f]) and torch.equal(st[1][f], st_ref[1][f]), (step, f)
            assert float(grads[f].abs().max()) == 0.0
        assert int(marks.max()) == 0


def test_marks_only_needs_the_fold_scatter_and_the_streaming_launch(ops):
    from mmlrec_amd import _lib as L
    vocab, E, B = [100], 6, 64                    # E = 6: not served by the LDS-fold kernel
    grads = [torch.zeros(vocab[0], E, device=dev())]
    marks = torch.zeros(ops.marks_bytes(vocab), dtype=torch.uint8, device=dev())
    X = torch.zeros(B, 1, device=dev())
    with pytest.raises(L.MMLError):
        ops.scatter_bwd(grads, X, [0], torch.zeros(B, E, device=dev()), marks=marks)
    p = torch.zeros(1000, 8, device=dev())        # far below 2^24 parameters: the flat kernel ignores marks -> rejected
    with pytest.raises(L.MMLError):
        ops.opt_step_dense([(p, torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p), None, None,
                             torch.zeros(1000, dtype=torch.uint8, device=dev()))], ops.make_hyper("adam", 0.01, step=1))


@pytest.mark.parametrize("kind", ["sgd", "adagrad", "adam"])
def test_optimizer_rows(ops, kind):
    """Sparse-row update equals the dense update exactly for SGD/Adagrad (rows with zero gradient do not move);
    for Adam it equals a dense Adam restricted to the touched rows."""
    rng = np.random.default_rng(10)
    vocab, E, B = [50, 3000], 8, 400
    F = len(vocab)
    rowbase = np.concatenate([[0], np.cumsum(vocab)]).tolist()
    tabs = [rng.standard_normal((v, E)).astype(np.float32) for v in vocab]
    tt = [T(t) for t in tabs]
    gt = [torch.zeros(v, E, device=dev()) for v in vocab]
    s1 = [torch.zeros(v, E, device=dev()) for v in vocab]
    s2 = [torch.zeros(v, E, device=dev()) for v in vocab]
    seen = [torch.zeros((v + 31) // 32, dtype=torch.int32, device=dev()) for v in vocab]
    touched = torch.zeros(B * F, dtype=torch.int32, device=dev())
    count = torch.zeros(1, dtype=torch.int32, device=dev())
    ref = {str(f): tabs[f].copy() for f in range(F)}
    opt = orc.DenseOptimizer(kind, 0.05)
    for step in range(1, 4):
        idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
        d_out = rng.standard_normal((B, F * E)).astype(np.float32)
        ops.counter_update(count, 0, reset=True)
        ops.scatter_bwd(gt, T(idx.astype(np.float32)), list(range(F)), T(d_out), seen=seen, rowbase=rowbase,
                        touched=touched, touched_count=count)
        grads = {}
        for f in range(F):
            g = np.zeros_like(tabs[f])
            np.add.at(g, idx[:, f], d_out[:, f * E:(f + 1) * E])
            grads[str(f)] = g
        if kind == "adam":  # lazy Adam: state of untouched rows is frozen -> emulate by masking the dense update
            before = {k: v.copy() for k, v in ref.items()}
            st_before = {k: {n: a.copy() for n, a in opt.state.get(k, {}).items()} for k in ref}
        opt.step(ref, grads)
        if kind == "adam":
            for f in range(F):
                k = str(f)
                un = np.ones(vocab[f], bool)
                un[np.unique(idx[:, f])] = False
                ref[k][un] = before[k][un]
                for n, a in st_before[k].items():
                    opt.state[k][n][un] = a[un]
        hyper = ops.make_hyper(kind, 0.05, step=step)
        ops.opt_step_rows(tt, gt, s1 if kind != "sgd" else None, s2 if kind == "adam" else None, seen, rowbase,
                          touched, count, hyper)
        for f in range(F):
            assert rel(tt[f].cpu().numpy(), ref[str(f)]) < 1e-5, (kind, step, f)
            assert float(gt[f].abs().max()) == 0.0
            assert int(seen[f].abs().max()) == 0


def test_elementwise(ops):
    from mmlrec_amd import _lib as L
    rng = np.random.default_rng(11)
    a = rng.standard_normal((300, 70)).astype(np.float32)
    b = rng.standard_normal((300, 70)).astype(np.float32)
    d = rng.standard_normal((300, 70)).astype(np.float32)
    out = torch.empty(300, 70, device=dev())
    ops.ew_mul(T(a), T(b), out)
    assert np.array_equal(out.cpu().numpy(), a * b)
    da, db = T(a.copy()), torch.empty(300, 70, device=dev())
    ops.ew_mul_bwd(T(d), T(a), T(b), da=da, db=db, acc_a=True)
    assert np.allclose(da.cpu().numpy(), a + d * b, rtol=1e-6, atol=1e-6)
    assert np.array_equal(db.cpu().numpy(), d * a)
    # folded activation derivatives: a is a relu output, b a 2*sigmoid output, each consumed only by this product
    lib = L.load()
    ar = np.maximum(a, 0).astype(np.float32)
    bs = (2 / (1 + np.exp(-b))).astype(np.float32)
    da2, db2 = torch.empty(300, 70, device=dev()), torch.empty(300, 70, device=dev())
    tar, tbs, td = T(ar), T(bs), T(d)
    rc = lib.mml_ew_mul_bwd_act(td.data_ptr(), tar.data_ptr(), tbs.data_ptr(), da2.data_ptr(), db2.data_ptr(), 0, 0,
                                ar.size, L.ACT_RELU, L.ACT_SIGMOID2, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert np.allclose(da2.cpu().numpy(), d * bs * (ar > 0), rtol=1e-6, atol=1e-7)
    assert np.allclose(db2.cpu().numpy(), d * ar * bs * (1 - bs / 2), rtol=1e-5, atol=1e-7)
    assert lib.mml_ew_mul_bwd_act(td.data_ptr(), tar.data_ptr(), tbs.data_ptr(), da2.data_ptr(), db2.data_ptr(), 1, 0,
                                  ar.size, L.ACT_RELU, 0, torch.cuda.current_stream().cuda_stream) != 0  # acc + fold
    s = torch.empty(300, 70, device=dev())
    ops.ew_add_n([T(a), T(b), T(d)], s)
    assert np.allclose(s.cpu().numpy(), a + b + d, rtol=1e-6, atol=1e-6)
    wide = torch.zeros(300, 100, device=dev())
    ops.copy2d(T(a), wide[:, 10:80])
    ops.copy2d(T(b), wide[:, 10:80], accumulate=True)
    assert np.allclose(wide[:, 10:80].cpu().numpy(), a + b, rtol=1e-6, atol=1e-6)
    assert float(wide[:, :10].abs().max()) == 0.0 and float(wide[:, 80:].abs().max()) == 0.0
    y = (2 / (1 + np.exp(-a))).astype(np.float32)
    dst = torch.empty(300, 70, device=dev())
    ops.act_bwd(T(y), T(d), dst, L.ACT_SIGMOID2)
    sg = y / 2
    assert np.allclose(dst.cpu().numpy(), d * 2 * sg * (1 - sg), rtol=1e-5, atol=1e-6)


def test_copy2d_batch(ops):
    """mml_copy2d_batch: independent [rows_i, cols_i] copies (different shapes and pitches) in one launch."""
    from mmlrec_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    shapes = [(256, 303), (64, 303), (4, 56), (1, 1), (128, 72)] * 8  # 40 items: more than one launch
    items, refs = [], []
    arr = (L.Copy2dDesc * len(shapes))()
    for k, (r, c) in enumerate(shapes):
        src = torch.randn(r, c + 3, generator=g).to(dev())
        dst = torch.randn(r, c + 13, generator=g).to(dev())
        acc = k % 2
        ref = dst.clone()
        ref[:, :c] = ref[:, :c] + src[:, :c] if acc else src[:, :c]
        d = arr[k]
        d.src, d.lds, d.dst, d.ldd, d.rows, d.cols, d.accumulate = src.data_ptr(), c + 3, dst.data_ptr(), c + 13, r, c, acc
        items.append((src, dst))
        refs.append(ref)
    assert lib.mml_copy2d_batch(arr, len(shapes), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    for (src, dst), ref in zip(items, refs):
        assert torch.equal(dst, ref)


def test_copy2d_batch_raises_the_magnitude_slot_of_what_it_stores(ops):
    """mml_copy2d_desc.amax_out (round 6): the copies that assemble a GEMM operand measure it on the way -- two column blocks
    into one buffer raise ONE slot to max |x| of both, an accumulating item registers the SUM it stores, an item without a
    slot leaves every slot alone, and the slot equals what mml_amax_batch measures on the assembled buffer."""
    from mmlrec_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    B = 5000
    a = (torch.randn(B, 64, generator=g) * 3.0).to(dev())
    b = (torch.randn(B, 8, generator=g) * 0.5).to(dev())
    b[17, 3] = -77.5                                   # the maximum sits in the narrow block, negative
    whole = torch.zeros(B, 80, device=dev())           # columns 72..79: padding, stays zero
    c = torch.randn(300, 40, generator=g).to(dev())
    acc_dst = torch.full((300, 40), 100.0, device=dev())
    plain_src, plain_dst = torch.randn(7, 5, generator=g).to(dev()), torch.zeros(7, 5, device=dev())
    slots = ops.amax_slots(3, dev())
    arr = (L.Copy2dDesc * 4)()
    for d, (src, dst, acc, slot) in zip(arr, ((a, whole[:, :64], 0, slots[0]), (b, whole[:, 64:72], 0, slots[0]),
                                               (c, acc_dst, 1, slots[1]), (plain_src, plain_dst, 0, None))):
        d.src, d.lds, d.dst, d.ldd = src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0)
        d.rows, d.cols, d.accumulate = src.shape[0], src.shape[1], acc
        if slot is not None:
            d.amax_out = slot.data_ptr()
    assert lib.mml_copy2d_batch(arr, 4, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(whole[:, :64], a) and torch.equal(whole[:, 64:72], b) and float(whole[:, 72:].abs().max()) == 0.0
    assert torch.equal(plain_dst, plain_src)
    assert ops.amax_value(slots[0]) == 77.5 == float(whole.abs().max())
    assert ops.amax_value(slots[1]) == float(acc_dst.abs().max()) and torch.equal(acc_dst, c + 100.0)
    assert ops.amax_value(slots[2]) == 0.0
    ref = ops.amax_slots(1, dev())
    ops.amax_batch([(whole, ref[0])])
    torch.cuda.synchronize()
    assert ops.amax_value(ref[0]) == ops.amax_value(slots[0])


@pytest.mark.parametrize("widths,rows,accumulate", [
    ([8, 8, 16, 8], 1000, 0),        # every segment a multiple of four floats: 16-byte kernel
    ([8] * 30, 4099, 1),             # the row-exchange shape (30 fields, E = 8), accumulate
    ([8, 3, 1, 12], 777, 0),         # odd widths: element-wise kernel
    ([1] * 30, 2048, 0),             # the index pack (one column per field)
    ([64] * 33, 300, 0),             # more than 1024 float4 columns would need the fallback: 33*16 = 528 stays vec
])
def test_copy_cols_segments(ops, widths, rows, accumulate):
    """mml_copy_cols: segment s copies src_s[:, :w_s] -> dst_s[:, :w_s] (column windows of wider matrices), bit-exact."""
    from mmlrec_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(sum(widths) + rows)
    n = len(widths)
    W = sum(widths)
    src_big = torch.randn(rows, W + 8, generator=g).to(dev())
    dst_big = torch.randn(rows, W + 12, generator=g).to(dev())
    ref = dst_big.clone()
    srcs, dsts, c = [], [], 0
    perm = list(range(n))[::-1]  # destination windows in reverse order of the source windows
    offs = np.concatenate([[0], np.cumsum(widths)])
    doff = np.concatenate([[0], np.cumsum([widths[p] for p in perm])])
    for s in range(n):
        srcs.append(src_big[:, offs[s]:offs[s] + widths[s]])
        k = perm.index(s)
        dsts.append(dst_big[:, 4 + doff[k]:4 + doff[k] + widths[s]])
        r = ref[:, 4 + doff[k]:4 + doff[k] + widths[s]]
        r.copy_(r + srcs[-1] if accumulate else srcs[-1])
    src = ops._ptr_array(srcs)
    dst = ops._ptr_array(dsts)
    lds = (L.i64 * n)(*[src_big.stride(0)] * n)
    ldd = (L.i64 * n)(*[dst_big.stride(0)] * n)
    wid = (L.i32 * n)(*widths)
    rc = lib.mml_copy_cols(src, lds, dst, ldd, wid, n, rows, accumulate, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(dst_big, ref)


@pytest.mark.parametrize("kind", ["adam", "rmsprop"])
def test_lazy_exact_optimizer_matches_dense_trajectory(ops, kind):
    """Touched-rows-only updates + catch-up of the skipped zero-gradient steps == the reference's dense optimizer
    (every row every step), over a trajectory with rows that stay untouched for 1..40 steps."""
    rng = np.random.default_rng(12)
    vocab, E, B, steps = [40, 600], 8, 48, 45
    F = len(vocab)
    rowbase = np.concatenate([[0], np.cumsum(vocab)]).tolist()
    tabs0 = [rng.standard_normal((v, E)).astype(np.float32) * 0.1 for v in vocab]
    lr = 0.01

    def fresh():
        return ([T(t.copy()) for t in tabs0], [torch.zeros(v, E, device=dev()) for v in vocab],
                [torch.zeros(v, E, device=dev()) for v in vocab])

    dt, dm, dv = fresh()      # dense path
    lt, lm, lv = fresh()      # lazy path
    dG = [torch.zeros(v, E, device=dev()) for v in vocab]
    lG = [torch.zeros(v, E, device=dev()) for v in vocab]
    seen = [torch.zeros((v + 31) // 32, dtype=torch.int32, device=dev()) for v in vocab]
    last = [torch.zeros(v, dtype=torch.int32, device=dev()) for v in vocab]
    touched = torch.zeros(B * F, dtype=torch.int32, device=dev())
    count = torch.zeros(1, dtype=torch.int32, device=dev())
    s2d = dv if kind == "adam" else None
    s2l = lv if kind == "adam" else None
    for step in range(1, steps + 1):
        # skewed indices: a few hot rows, most rows rare; some steps touch only a handful of rows
        idx = np.stack([np.minimum((v ** rng.random(B)).astype(np.int64), v - 1) for v in vocab], 1)
        if step % 7 == 0:
            idx[:] = idx[0]
        X = T(idx.astype(np.float32))
        g = T(rng.standard_normal((B, F * E)).astype(np.float32))
        hyper = ops.make_hyper(kind, lr, step=step)
        # dense reference: scatter into dense G, update every row
        ops.scatter_bwd(dG, X, list(range(F)), g)
        hz = ops.make_hyper(kind, lr, step=step, zero_grad=True)
        ops.opt_step_dense([(dt[f], dG[f], dm[f], s2d[f] if s2d else None) for f in range(F)], hz)
        # lazy: unique rows -> catch-up to step-1 -> (gather would read here) -> scatter -> rows update at `step`
        ops.counter_update(count, 0, reset=True)
        ops.index_unique(vocab, list(range(F)), E, X, seen, rowbase, touched, count)
        ops.opt_catchup_rows(lt, lm, s2l, last, rowbase, touched, count, hyper)
        ops.scatter_bwd(lG, X, list(range(F)), g)
        ops.opt_step_rows(lt, lG, lm, s2l, seen, rowbase, touched, count, hyper, last=last)
        if step in (1, 9, steps):
            # rows read by the NEXT step would be caught up first; emulate a full read by flushing a copy
            ft = [t.clone() for t in lt]
            fm = [t.clone() for t in lm]
            fv = [t.clone() for t in lv]
            fl = [t.clone() for t in last]
            hf = ops.make_hyper(kind, lr, step=step)
            for f in range(F):
                ops.opt_catchup_dense(ft[f], fm[f], fv[f] if kind == "adam" else None, fl[f], hf)
                assert int(fl[f].min()) == step
                a, b = ft[f].cpu().numpy(), dt[f].cpu().numpy()
                assert np.abs(a - b).max() <= 2e-6 * max(np.abs(b).max(), 1e-30) + 1e-9, (kind, step, f)
                assert rel(fm[f].cpu().numpy(), dm[f].cpu().numpy()) < 2e-5, (kind, step, f, "state1")
                if kind == "adam":
                    assert rel(fv[f].cpu().numpy(), dv[f].cpu().numpy()) < 2e-5, (kind, step, f, "state2")
    for f in range(F):
        assert float(lG[f].abs().max()) == 0.0 and int(seen[f].abs().max()) == 0


@pytest.mark.parametrize("n,cols,seg", [(1000, 2, 256), (4096 * 3 + 17, 1, 4096), (5000, 3, 1000), (300, 2, 300), (64, 1, 7)])
def test_auc_segments_match_sklearn(ops, n, cols, seg):
    """mml_auc_segments == sklearn.metrics.roc_auc_score per (batch, column) -- what the reference evaluates on the
    host after every step (model/basemodel.py:316-331) -- including heavy ties, exact 0 / 1 predictions, -0.0, and
    single-class batches (sklearn raises; the kernel reports NaN)."""
    from sklearn.metrics import roc_auc_score
    rng = np.random.default_rng(n + seg)
    pred = rng.random((n, cols)).astype(np.float32)
    pred[:, 0] = np.round(pred[:, 0] * 8) / 8           # nine distinct values: large tie groups incl. 0.0 and 1.0
    if cols > 1:
        pred[::7, 1] = -0.0
        pred[1::7, 1] = 0.0
    y = (rng.random((n, cols)) < 0.3).astype(np.float32)
    y[:min(seg, n), -1] = 1.0                             # first segment of the last column: one class only
    got = ops.auc_segments(T(pred), T(y), seg).cpu().numpy()
    nseg = (n + seg - 1) // seg
    assert got.shape == (nseg, cols)
    for s in range(nseg):
        sl = slice(s * seg, min(n, (s + 1) * seg))
        for c in range(cols):
            yt = y[sl, c]
            if yt.min() == yt.max():
                assert np.isnan(got[s, c])
                continue
            want = roc_auc_score(yt, pred[sl, c].astype(np.float64))
            assert abs(got[s, c] - want) < 1e-12, (s, c, got[s, c], want)


def test_auc_segments_rejects_long_segments(ops):
    from mmlrec_amd import _lib as L
    p = torch.rand(10000, 1, device=dev())
    with pytest.raises(L.MMLError):
        ops.auc_segments(p, (p > 0.5).float(), 5000)


@pytest.mark.parametrize("B,H", [(257, 24), (1000, 128), (3, 7)])
def test_attn2_fwd_bwd(ops, B, H):
    """mml_attn2_fwd / _bwd == the two-token attention of AITM (reference model/aitm.py:84-93) in float64."""
    import ctypes as C
    from mmlrec_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(B + H)
    V, K, Q = (rng.standard_normal((2, B, H)).astype(np.float32) for _ in range(3))
    dout = rng.standard_normal((B, H)).astype(np.float32)
    tV, tK, tQ = [T(V[t]) for t in range(2)], [T(K[t]) for t in range(2)], [T(Q[t]) for t in range(2)]
    out, A, tdo = torch.empty(B, H, device=dev()), torch.empty(B, 2, device=dev()), T(dout)
    g = [[torch.empty(B, H, device=dev()) for _ in range(2)] for _ in range(3)]
    d = L.Attn2Desc()
    for t in range(2):
        d.V[t], d.K[t], d.Q[t] = tV[t].data_ptr(), tK[t].data_ptr(), tQ[t].data_ptr()
        d.ldv[t] = d.ldk[t] = d.ldq[t] = H
        d.dV[t], d.dK[t], d.dQ[t] = g[0][t].data_ptr(), g[1][t].data_ptr(), g[2][t].data_ptr()
        d.lddv[t] = d.lddk[t] = d.lddq[t] = H
    d.out, d.ldo, d.A, d.dout, d.lddo = out.data_ptr(), H, A.data_ptr(), tdo.data_ptr(), H
    d.B, d.H, d.sqrt_h = B, H, float(np.float32(np.sqrt(H)))
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mml_attn2_fwd(C.byref(d), st) == 0
    assert lib.mml_attn2_bwd(C.byref(d), st) == 0
    torch.cuda.synchronize()
    V64, K64, Q64, do64 = V.astype(np.float64), K.astype(np.float64), Q.astype(np.float64), dout.astype(np.float64)
    s = (K64 * Q64).sum(2) / np.sqrt(H)                      # [2,B]
    a = np.exp(s - s.max(0)) / np.exp(s - s.max(0)).sum(0)   # softmax over the two tokens
    ref = (a[:, :, None] * V64).sum(0)
    assert rel(out.cpu().numpy(), ref) < 1e-5
    assert rel(A.cpu().numpy(), a.T) < 1e-5
    da = (do64[None] * V64).sum(2)
    ds = a * (da - (a * da).sum(0)) / np.sqrt(H)
    for t in range(2):
        assert rel(g[0][t].cpu().numpy(), a[t][:, None] * do64) < 1e-5
        assert rel(g[1][t].cpu().numpy(), ds[t][:, None] * Q64[t]) < 2e-5
        assert rel(g[2][t].cpu().numpy(), ds[t][:, None] * K64[t]) < 2e-5


@pytest.mark.parametrize("B,n,act", [(1000, 48, "relu"), (4099, 130, "none"), (300, 7, "relu")])
def test_batchnorm_fwd_bwd(ops, B, n, act):
    """mml_bn_fwd / mml_bn_bwd == torch.nn.BatchNorm1d semantics (reference model/utils.py:132-134, :153-157) in float64:
    several row chunks, odd widths, strided rows, running statistics and the batch counter, eval mode."""
    from mmlrec_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(B + n)
    z = (rng.standard_normal((B, n)) * 2 + 0.5).astype(np.float32)
    gamma, beta = (1 + 0.2 * rng.standard_normal(n)).astype(np.float32), (0.1 * rng.standard_normal(n)).astype(np.float32)
    rm0, rv0 = rng.standard_normal(n).astype(np.float32), (1 + rng.random(n)).astype(np.float32)
    dy = rng.standard_normal((B, n)).astype(np.float32)
    acode = L.ACT_RELU if act == "relu" else L.ACT_NONE
    zb = torch.zeros(B, n + 5, device=dev())
    zb[:, :n] = T(z)
    tz = zb[:, :n]
    tg, tb, rm, rv = T(gamma), T(beta), T(rm0.copy()), T(rv0.copy())
    nbt = torch.tensor([3], dtype=torch.int64, device=dev())
    mean, rstd = torch.empty(n, device=dev()), torch.empty(n, device=dev())
    y = torch.empty(B, n, device=dev())
    nbytes = int(lib.mml_bn_workspace_bytes(B, n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.mml_bn_fwd(tz.data_ptr(), tz.stride(0), tg.data_ptr(), tb.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                        nbt.data_ptr(), mean.data_ptr(), rstd.data_ptr(), y.data_ptr(), n, B, n, acode, 1, 1e-5, 0.1,
                        ws.data_ptr(), nbytes, st)
    assert rc == 0
    z64 = z.astype(np.float64)
    mu, var = z64.mean(0), z64.var(0)
    xhat = (z64 - mu) / np.sqrt(var + 1e-5)
    pre = xhat * gamma + beta
    ref = np.maximum(pre, 0) if act == "relu" else pre
    assert rel(y.cpu().numpy(), ref) < 1e-5
    assert rel(rm.cpu().numpy(), 0.9 * rm0 + 0.1 * mu) < 1e-5
    assert rel(rv.cpu().numpy(), 0.9 * rv0 + 0.1 * var * B / (B - 1)) < 1e-5
    assert int(nbt.item()) == 4
    # backward (dy = gradient w.r.t. the BatchNorm output, activation derivative already applied)
    dyr = dy * (pre > 0) if act == "relu" else dy
    tdy = T(dyr.astype(np.float32))
    dz, dg, db = torch.empty(B, n, device=dev()), torch.empty(n, device=dev()), torch.empty(n, device=dev())
    rc = lib.mml_bn_bwd(tdy.data_ptr(), n, tz.data_ptr(), tz.stride(0), tg.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                        dz.data_ptr(), n, dg.data_ptr(), db.data_ptr(), 0, B, n, ws.data_ptr(), nbytes, st)
    assert rc == 0
    d64 = dyr.astype(np.float64)
    dbeta, dgamma = d64.sum(0), (d64 * xhat).sum(0)
    dz_ref = gamma / np.sqrt(var + 1e-5) * (d64 - (dbeta + xhat * dgamma) / B)
    assert rel(db.cpu().numpy(), dbeta) < 1e-5 and rel(dg.cpu().numpy(), dgamma) < 1e-5
    assert rel(dz.cpu().numpy(), dz_ref) < 2e-5
    # eval mode: running statistics, nothing is updated
    rm1, rv1 = rm.clone(), rv.clone()
    rc = lib.mml_bn_fwd(tz.data_ptr(), tz.stride(0), tg.data_ptr(), tb.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                        nbt.data_ptr(), mean.data_ptr(), rstd.data_ptr(), y.data_ptr(), n, B, n, acode, 0, 1e-5, 0.1,
                        ws.data_ptr(), nbytes, st)
    assert rc == 0
    pre = (z64 - rm1.cpu().numpy()) / np.sqrt(rv1.cpu().numpy().astype(np.float64) + 1e-5) * gamma + beta
    assert rel(y.cpu().numpy(), np.maximum(pre, 0) if act == "relu" else pre) < 1e-5
    assert torch.equal(rm, rm1) and torch.equal(rv, rv1) and int(nbt.item()) == 4


# ---------------------------------------------------------------------------------------------- row-sharded tables
@pytest.mark.parametrize("E,B", [(8, 4000), (16, 555), (5, 300)])
def test_scatter_idx32_and_unique_idx32(ops, E, B):
    """Native-index variants (owner side of row-sharded tables, vocabularies past 2^24)."""
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib as L
    rng = np.random.default_rng(4)
    vocab = [3, 500, 30000]
    F = len(vocab)
    idx = np.stack([np.minimum((v ** rng.random(B)).astype(np.int64), v - 1) for v in vocab], 1).astype(np.int32)
    d_out = rng.standard_normal((B, F * E)).astype(np.float32)
    gt = [torch.zeros(v, E, device=dev()) for v in vocab]
    seen = [torch.zeros((v + 31) // 32, dtype=torch.int32, device=dev()) for v in vocab]
    rowbase = np.concatenate([[0], np.cumsum(vocab)]).tolist()
    touched = torch.full((B * F,), -1, dtype=torch.int32, device=dev())
    count = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.scatter_bwd_idx32(gt, T(idx), T(d_out), seen=seen, rowbase=rowbase, touched=touched, touched_count=count)
    for f, v in enumerate(vocab):
        g = np.zeros((v, E), np.float64)
        np.add.at(g, idx[:, f], d_out[:, f * E:(f + 1) * E].astype(np.float64))
        assert rel(gt[f].cpu().numpy(), g) < 1e-5
    want = np.sort(np.concatenate([np.unique(idx[:, f]) + rowbase[f] for f in range(F)]))
    assert np.array_equal(np.sort(touched[:int(count.item())].cpu().numpy()), want)
    # index-only pass
    seen2 = [torch.zeros_like(s) for s in seen]
    t2 = torch.full((B * F,), -1, dtype=torch.int32, device=dev())
    c2 = torch.zeros(1, dtype=torch.int32, device=dev())
    lib = L.load()
    ii = T(idx)
    L.check(lib.mml_index_unique_idx32((L.i64 * F)(*vocab), F, E if E <= 16 else 16, ii.data_ptr(), ii.stride(0), B,
                                       ops._ptr_array(seen2), (L.i64 * (F + 1))(*rowbase), t2.data_ptr(), c2.data_ptr(),
                                       t2.numel(), None, None, ops._stream()), "mml_index_unique_idx32")
    assert np.array_equal(np.sort(t2[:int(c2.item())].cpu().numpy()), want)
    # ... and with the byte-mark scratch map (plain stores + compaction instead of atomics): same set, marks left zero
    seen3 = [torch.zeros_like(s) for s in seen]
    t3 = torch.full((B * F,), -1, dtype=torch.int32, device=dev())
    c3 = torch.zeros(1, dtype=torch.int32, device=dev())
    marks = torch.zeros(ops.marks_bytes(vocab), dtype=torch.uint8, device=dev())
    L.check(lib.mml_index_unique_idx32((L.i64 * F)(*vocab), F, E if E <= 16 else 16, ii.data_ptr(), ii.stride(0), B,
                                       ops._ptr_array(seen3), (L.i64 * (F + 1))(*rowbase), t3.data_ptr(), c3.data_ptr(),
                                       t3.numel(), marks.data_ptr(), None, ops._stream()), "mml_index_unique_idx32")
    assert np.array_equal(np.sort(t3[:int(c3.item())].cpu().numpy()), want)
    assert int(marks.max().item()) == 0
    for a_, b_ in zip(seen2, seen3):
        assert torch.equal(a_, b_)
    # scatter with marks: gradients and list as before
    gt3 = [torch.zeros(v, E, device=dev()) for v in vocab]
    seen4 = [torch.zeros_like(s) for s in seen]
    t4 = torch.full((B * F,), -1, dtype=torch.int32, device=dev())
    c4 = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.scatter_bwd_idx32(gt3, T(idx), T(d_out), seen=seen4, rowbase=rowbase, touched=t4, touched_count=c4, marks=marks)
    assert np.array_equal(np.sort(t4[:int(c4.item())].cpu().numpy()), want)
    assert int(marks.max().item()) == 0
    for f in range(F):
        assert rel(gt3[f].cpu().numpy(), gt[f].cpu().numpy()) < 1e-5


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("E,nd,B", [(8, 0, 3000), (4, 3, 257), (6, 0, 100)])
def test_route_expand_permute(ops, world, E, nd, B):
    """Routing of a batch to row owners (csrc/shard.hip) against the host arithmetic of parallel.RowSharding: counts,
    owner-grouped keys, positions; expansion of the returned rows == the plain gather (bit-exact); mml_rows_permute is
    its inverse; shard <-> table conversion round-trips."""
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd.parallel import RowSharding
    rng = np.random.default_rng(5)
    vocab = [1, 2, 7, 100, 1000, 50000, 3]
    F = len(vocab)
    idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
    idx[0, :] = 0
    idx[1, :] = np.array(vocab) - 1
    X = np.concatenate([idx.astype(np.float32), rng.random((B, nd), dtype=np.float32)], 1)
    tabs = [rng.standard_normal((v, E)).astype(np.float32) for v in vocab]
    shs = [RowSharding(vocab, E, world, r) for r in range(world)]
    sh = shs[0]
    counts, keys, pos = ops.route(T(X), list(range(F)), vocab, sh.keybase, world)
    counts, keys, pos = counts.cpu().numpy(), keys.cpu().numpy(), pos.cpu().numpy()
    own = (idx + np.arange(F)[None, :]) % world
    key = np.array(sh.keybase[:F])[None, :] + idx // world
    assert np.array_equal(counts, np.bincount(own.ravel(), minlength=world))
    off = np.concatenate([[0], np.cumsum(counts)])
    assert np.array_equal(np.sort(pos.ravel()), np.arange(B * F))  # a permutation of the send layout
    assert np.array_equal(keys[pos], key)
    seg_of_pos = np.searchsorted(off, pos, side="right") - 1
    assert np.array_equal(seg_of_pos, own)
    # int32 index input gives the same routing
    c2, k2, p2 = ops.route(T(idx.astype(np.int32)), list(range(F)), vocab, sh.keybase, world)
    assert np.array_equal(c2.cpu().numpy(), counts)
    assert np.array_equal(k2.cpu().numpy()[p2.cpu().numpy()], key)
    # owners: shards of the tables, gather of the received keys, rows back in the same order
    dt = [T(t) for t in tabs]
    shards = [s.tables_to_shard(dt) for s in shs]
    for r, s in enumerate(shs):  # shard layout == host arithmetic
        ref = np.zeros((s.R, E), np.float32)
        for f, v in enumerate(vocab):
            rows = np.arange(s.first(f), v, world)
            assert len(rows) == s.owned_rows(f)
            ref[s.keybase[f]:s.keybase[f] + len(rows)] = tabs[f][rows]
        assert np.array_equal(shards[r].cpu().numpy(), ref)
    rows_recv = torch.empty(B * F, E, device=dev())
    for r in range(world):
        seg = torch.from_numpy(keys[off[r]:off[r + 1]].astype(np.int32)).to(dev()).view(-1, 1)
        if seg.numel():
            rows_recv[off[r]:off[r + 1]] = ops.gather_fwd_idx32([shards[r]], seg)
    out = ops.gather_fwd_idx32([rows_recv] * F, T(pos.astype(np.int32)), T(X[:, F:].copy()) if nd else None)
    ref = np.concatenate([tabs[f][idx[:, f]] for f in range(F)] + [X[:, F:]], 1)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # ---- requester-side de-duplication: the distinct rows are routed, every lookup finds its row's slot
    from mmlrec_amd import _lib as L
    lib = L.load()
    rowbase = np.concatenate([[0], np.cumsum(vocab)]).tolist()
    seen = [torch.zeros((v + 31) // 32, dtype=torch.int32, device=dev()) for v in vocab]
    marks = torch.zeros(ops.marks_bytes(vocab), dtype=torch.uint8, device=dev())
    tl = torch.full((B * F,), -1, dtype=torch.int32, device=dev())
    tc = torch.zeros(1, dtype=torch.int32, device=dev())
    Xd = T(X)
    ops.index_unique(vocab, list(range(F)), min(E, 16), Xd, seen, rowbase, tl, tc, marks=marks)
    uniq = np.unique(np.concatenate([idx[:, f] + rowbase[f] for f in range(F)]))
    assert int(tc.item()) == len(uniq)
    counters = torch.zeros(2 * world, dtype=torch.int32, device=dev())
    voc, rb, kb = (L.i64 * F)(*vocab), (L.i64 * (F + 1))(*rowbase), (L.i64 * F)(*sh.keybase[:F])
    st = ops._stream()
    L.check(lib.mml_route_list_count(tl.data_ptr(), tc.data_ptr(), tl.numel(), voc, rb, F, world, counters.data_ptr(), st),
            "mml_route_list_count")
    cnt_d = counters[:world].cpu().numpy().copy()
    keys_d = torch.full((B * F,), -1, dtype=torch.int32, device=dev())
    slot_of = torch.full((rowbase[-1],), -1, dtype=torch.int32, device=dev())
    L.check(lib.mml_route_list_place(tl.data_ptr(), tc.data_ptr(), tl.numel(), voc, rb, kb, F, world,
                                     counters.data_ptr(), keys_d.data_ptr(), slot_of.data_ptr(), st), "mml_route_list_place")
    pos_d = torch.empty(B, F, dtype=torch.int32, device=dev())
    colv = (L.i32 * F)(*range(F))
    L.check(lib.mml_lookup_slots(Xd.data_ptr(), Xd.stride(0), colv, voc, rb, F, B, slot_of.data_ptr(), pos_d.data_ptr(),
                                 None, st), "mml_lookup_slots")
    uf = np.searchsorted(np.array(rowbase), uniq, side="right") - 1       # field of every distinct row
    ur = uniq - np.array(rowbase)[uf]
    assert np.array_equal(cnt_d, np.bincount((ur + uf) % world, minlength=world))
    u = len(uniq)
    kd, pd = keys_d.cpu().numpy(), pos_d.cpu().numpy()
    assert pd.min() >= 0 and pd.max() < u
    assert np.array_equal(kd[pd], key)                                     # every lookup lands on its row's key
    off_d = np.concatenate([[0], np.cumsum(cnt_d)])
    assert np.array_equal(np.searchsorted(off_d, pd, side="right") - 1, own)
    assert len(np.unique(pd)) == u                                         # one slot per distinct row
    rows_d = torch.empty(u, E, device=dev())
    for r in range(world):
        seg = torch.from_numpy(kd[off_d[r]:off_d[r + 1]].astype(np.int32)).to(dev()).view(-1, 1)
        if seg.numel():
            rows_d[off_d[r]:off_d[r + 1]] = ops.gather_fwd_idx32([shards[r]], seg)
    out_d = ops.gather_fwd_idx32([rows_d] * F, pos_d, T(X[:, F:].copy()) if nd else None)
    assert np.array_equal(out_d.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    L.check(lib.mml_rows_clear(tl.data_ptr(), tc.data_ptr(), tl.numel(), rb, ops._ptr_array(seen), F, st), "mml_rows_clear")
    assert all(int(sn.abs().max().item()) == 0 for sn in seen) and int(marks.max().item()) == 0
    # gradients of duplicates summed into the slot of their row (what travels in the de-duplicated exchange)
    dg = rng.standard_normal((B, F * E)).astype(np.float32)
    gsend = torch.zeros(u, E, device=dev())
    ops.scatter_bwd_idx32([gsend] * F, pos_d, T(dg))
    want_g = np.zeros((u, E), np.float64)
    np.add.at(want_g, pd.ravel(), dg.reshape(B * F, E).astype(np.float64))
    assert rel(gsend.cpu().numpy(), want_g) < 1e-5
    # inverse: pack per-sample gradient pieces into the send order
    d_out = rng.standard_normal((B, F * E + nd)).astype(np.float32)
    packed = torch.zeros(B * F, E, device=dev())
    ops.rows_permute(T(d_out), T(pos.astype(np.int32)), E, packed)
    want = np.zeros((B * F, E), np.float32)
    want[pos.ravel()] = d_out[:, :F * E].reshape(B * F, E)
    assert np.array_equal(packed.cpu().numpy(), want)
    # shards -> tables round trip
    back = [torch.full_like(t, float("nan")) for t in dt]
    for r, s in enumerate(shs):
        s.shard_to_tables(shards[r], back, rank=r)
    for f in range(F):
        assert torch.equal(back[f], dt[f])


def test_route_out_of_range_sets_status(ops):
    status = ops.new_status(dev())
    X = torch.tensor([[3.0], [10.0], [-1.0]], device=dev())
    counts, keys, pos = ops.route(X, [0], [10], [0, 5], 2, status=status)
    assert int(counts.sum().item()) == 3
    with pytest.raises(IndexError):
        ops.check_status(status)


@pytest.mark.parametrize("E", [4, 8, 16])
def test_scatter_det_is_bitwise_repeatable_and_order_independent(E):
    """mml_scatter_bwd_det (SURVEY 8(b) "mode: sorted", 5 "deterministic variant for bit-stable debugging"): integer
    fixed-point row totals.  Two runs agree BIT FOR BIT; so does a run on the same batch in another sample order (a
    sorted fp32 sum would not); the result is the float64 index_add to 1e-6; the 64-bit totals are zero again afterwards
    and the marks cleared.  Tables: a hot tiny one (every sample hits 2 rows), direct-mapped, hashed and a large one."""
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import ops
    dev = torch.device("cuda:0")
    vocab = [2, 100, 1000, 5000, 300000]
    F, B = len(vocab), 20000
    g = torch.Generator().manual_seed(E)
    X = torch.stack([(torch.rand(B, generator=g) ** 3 * v).floor().clamp_(0, v - 1) for v in vocab], 1).contiguous()
    d = torch.randn(B, F * E, generator=g) * torch.logspace(-6, 2, B).unsqueeze(1)[torch.randperm(B, generator=g)]
    marks_n = ops.marks_bytes(vocab)

    def run(Xh, dh):
        gt = [torch.zeros(v, E, device=dev) for v in vocab]
        acc = [torch.zeros(v, E, dtype=torch.int64, device=dev) for v in vocab]
        marks = torch.zeros(marks_n, dtype=torch.uint8, device=dev)
        ops.scatter_bwd_det(gt, Xh.to(dev), list(range(F)), dh.to(dev), acc, marks)
        torch.cuda.synchronize()
        assert all(int(a.abs().max()) == 0 for a in acc) and int(marks.max()) == 0
        return gt

    g1, g2 = run(X, d), run(X, d)
    perm = torch.randperm(B, generator=g)
    g3 = run(X[perm].contiguous(), d[perm].contiguous())
    for f in range(F):
        assert torch.equal(g1[f].view(torch.int32), g2[f].view(torch.int32)), f
        assert torch.equal(g1[f].view(torch.int32), g3[f].view(torch.int32)), f
        ref = torch.zeros(vocab[f], E, dtype=torch.float64, device=dev)
        ref.index_add_(0, X[:, f].long().to(dev), d[:, f * E:(f + 1) * E].double().to(dev))
        err = (g1[f].double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-6, (f, err)
    # second call ADDS to the accumulators (like mml_scatter_bwd): 2 x the first result, exactly (powers of two)
    gt = [t.clone() for t in g1]
    acc = [torch.zeros(v, E, dtype=torch.int64, device=dev) for v in vocab]
    marks = torch.zeros(marks_n, dtype=torch.uint8, device=dev)
    ops.scatter_bwd_det(gt, X.to(dev), list(range(F)), d.to(dev), acc, marks)
    for f in range(F):
        assert torch.allclose(gt[f], 2 * g1[f], rtol=1e-6, atol=0)


def test_concurrent_stream_runs_beside_the_current_stream(ops):
    """ops.concurrent_stream: the stream it returns executes a kernel WHILE the current stream executes another (the
    trainer's forked tail and the routing prefetch depend on it; HIP's round-robin hardware-queue assignment does not
    guarantee it for an arbitrary new stream)."""
    if not hasattr(torch.cuda, "_sleep"):
        pytest.skip("no spin kernel in this torch build")
    d = dev()
    main = torch.cuda.current_stream(d)
    side = ops.concurrent_stream(d)
    assert side != main
    spin = 400_000

    def timed(fn):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        fn()
        b.record(main)
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    def pair():
        j = torch.cuda.Event()
        with torch.cuda.stream(side):
            torch.cuda._sleep(spin)
            j.record(side)
        torch.cuda._sleep(spin)
        main.wait_event(j)

    torch.cuda._sleep(spin)
    one = min(timed(lambda: torch.cuda._sleep(spin)) for _ in range(3))
    both = min(timed(pair) for _ in range(3))
    assert both < 1.5 * one, (one, both)
