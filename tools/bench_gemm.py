#!/usr/bin/env python3
"""Micro-benchmark of the grouped GEMM entry points on random data (HIP-event timed, median of interleaved rounds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import _lib as L, ops  # noqa: E402


def timeit(fn, rounds=7, inner=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    return ts[len(ts) // 2]


AMAX = os.environ.get("GEMM_AMAX", "1") != "0"  # measure the operand magnitudes first -> two-plane fp16 arithmetic in auto mode


def accuracy(dev):
    """max-norm relative error of fwd / dgrad / wgrad against float64 torch on one L1-like problem."""
    torch.manual_seed(0)
    M, N, K = 8192, 256, 240
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    ops.gemm_fwd([dict(A=A, W=W, bias=b, C=C, act=L.ACT_NONE)], amax=AMAX)
    ref = A.double() @ W.double().t() + b.double()
    e_f = float((C.double() - ref).abs().max() / ref.abs().max())
    dC = torch.randn(M, N, device=dev)
    dA = torch.empty(M, K, device=dev)
    ops.gemm_dgrad([dict(dA=dA, Y=None, act=L.ACT_NONE, srcs=[(dC, W, 0)])], amax=AMAX)
    ref = dC.double() @ W.double()
    e_d = float((dA.double() - ref).abs().max() / ref.abs().max())
    dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    ops.gemm_wgrad([dict(dC=dC, A=A, dW=dW, dbias=db)], amax=AMAX)
    ref = dC.double().t() @ A.double()
    e_w = float((dW.double() - ref).abs().max() / ref.abs().max())
    print(f"accuracy vs float64 (max |err| / max |ref|): fwd {e_f:.2e}  dgrad {e_d:.2e}  wgrad {e_w:.2e}")


def main():
    dev = torch.device("cuda:0")
    print("gemm mode", L.load().mml_gemm_get_mode(), " MMLREC_GEMM_BN =", os.environ.get("MMLREC_GEMM_BN"),
          " lib =", os.environ.get("MMLREC_LIB"))
    if not os.environ.get("MMLREC_LIB"):
        accuracy(dev)
    cases = [("square 4096^3", 4096, [(4096, 4096)]),
             ("L1 experts+gates", 65536, [(256, 240)] * 4 + [(64, 240)] * 2),
             ("L1 experts only", 65536, [(256, 240)] * 4),
             ("L2 experts", 65536, [(128, 256)] * 4),
             ("towers", 65536, [(64, 128)] * 2)]
    only = os.environ.get("CASES")
    if only:
        cases = [c for c in cases if any(c[0].startswith(o) for o in only.split(","))]
    for name, M, shapes in cases:
        A = {}
        probs_f, probs_w = [], []
        flops = 0
        for N, K in shapes:
            if K not in A:
                A[K] = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev) / K ** 0.5
            Cc = torch.empty(M, N, device=dev)
            probs_f.append(dict(A=A[K], W=W, bias=torch.zeros(N, device=dev), C=Cc, act=L.ACT_RELU))
            if os.environ.get("GEMM_MASK", "0") == "1":  # the training step's forward also writes the relu sign masks
                probs_f[-1]["mask"] = torch.zeros(M, (N + 31) // 32, dtype=torch.int32, device=dev)
            probs_w.append(dict(dC=Cc, A=A[K], dW=torch.empty(N, K, device=dev), dbias=torch.empty(N, device=dev)))
            flops += 2.0 * M * N * K
        key = lambda t_: (t_.data_ptr(), tuple(t_.shape), t_.stride(0))  # noqa: E731
        cache = {}
        if AMAX:  # magnitudes measured ONCE, outside the timed launches (the step gets them from the producers)
            ops.gemm_fwd(probs_f)  # (C as the timed launches will leave it: the dgrad / wgrad read it as dC)
            ops._measured([p["A"] for p in probs_f] + [p["W"] for p in probs_f] + [p["C"] for p in probs_f], cache)
            for p, q in zip(probs_f, probs_w):
                p.update(amax_a=cache[key(p["A"])], amax_w=cache[key(p["W"])])
                q.update(amax_dc=cache[key(q["dC"])], amax_a=cache[key(q["A"])])
        amx = lambda p: ((cache[key(p["C"])], cache[key(p["W"])]) if AMAX else ())  # noqa: E731
        # pre-cut weight planes (as the step has them): own exponent for the forward, one exponent per input-gradient
        # problem (GEMM_PLANES=0: the kernels cut the weight fragments themselves)
        PLANES = AMAX and os.environ.get("GEMM_PLANES", "1") != "0" and all(k % 16 == 0 and n % 16 == 0 for n, k in shapes)
        pl_f, pl_d = {}, {}
        if PLANES:
            items = []
            one_group = not name.startswith("L2") and all(k == shapes[0][1] for _, k in shapes)
            kx_grp = torch.zeros(1, dtype=torch.int32, device=dev)
            for p in probs_f:
                Wt, sl = p["W"], cache[key(p["W"])]
                pf, kf = torch.zeros(Wt.shape, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
                items.append((Wt, pf, ops.PLANES_ROWS, [sl], kf))
                p.update(w_planes=pf, w_kexp=kf)
                pd_, kd = torch.zeros(Wt.shape, dtype=torch.int32, device=dev), (kx_grp if one_group else torch.zeros(1, dtype=torch.int32, device=dev))
                grp = [cache[key(q["W"])] for q in probs_f[:8]] if one_group else [sl]
                items.append((Wt, pd_, ops.PLANES_COLS, grp, kd))
                pl_d[id(p)] = (pd_, kd)
            ops.planes_cut(items)
        amx_p = lambda p: (amx(p) + pl_d[id(p)]) if PLANES else amx(p)  # noqa: E731
        AOUT = os.environ.get("GEMM_AMAX_OUT", "0") == "1"  # the launches also produce the magnitude of their output
        if AOUT:
            for p in probs_f:
                p["amax_out"] = ops.amax_slots(1, dev)[0]
        t = timeit(lambda: ops.gemm_fwd(probs_f))
        print(f"{name:22s} fwd   {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TFLOP/s   [{L.load().mml_gemm_last_kernel().decode()}]")
        if os.environ.get("FWD_ONLY") == "1":
            continue
        t = timeit(lambda: ops.gemm_wgrad(probs_w))
        print(f"{name:22s} wgrad {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TFLOP/s   [{L.load().mml_gemm_last_kernel().decode()}]")
        K0 = shapes[0][1]
        if name.startswith("L2"):  # one dgrad problem per expert (single source each), as the step runs them
            pd = [dict(dA=torch.empty(M, K0, device=dev), Y=A[K0], act=L.ACT_RELU, srcs=[(p["C"], p["W"], 0) + amx_p(p)],
                       amax_out=ops.amax_slots(1, dev)[0] if AOUT else None) for p in probs_f]
            t = timeit(lambda: ops.gemm_dgrad(pd))
            print(f"{name:22s} dgrad {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TFLOP/s  (one problem per expert)")
        elif all(k == K0 for _, k in shapes):
            dA = torch.empty(M, K0, device=dev)
            # (the first layers' input is dnn_input: no activation derivative in its gradient, as in the step)
            first = name.startswith("L1")
            pd = [dict(dA=dA, Y=None if first else A[K0], act=L.ACT_NONE if first else L.ACT_RELU,
                       srcs=[(p["C"], p["W"], 0) + amx_p(p) for p in probs_f[:8]],
                       amax_out=ops.amax_slots(1, dev)[0] if AOUT else None)]
            fl = sum(2.0 * M * n * k for n, k in shapes[:8])
            t = timeit(lambda: ops.gemm_dgrad(pd))
            print(f"{name:22s} dgrad {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s   [{L.load().mml_gemm_last_kernel().decode()}]")


if __name__ == "__main__":
    main()
