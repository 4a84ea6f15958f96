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


def main():
    dev = torch.device("cuda:0")
    cases = [("square 4096^3", 4096, [(4096, 4096)]),
             ("L1 experts+gates", 65536, [(256, 240)] * 4 + [(64, 240)] * 2),
             ("L1 experts only", 65536, [(256, 240)] * 4),
             ("L2 experts", 65536, [(128, 256)] * 4),
             ("towers", 65536, [(64, 128)] * 2)]
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
            probs_w.append(dict(dC=Cc, A=A[K], dW=torch.empty(N, K, device=dev), dbias=torch.empty(N, device=dev)))
            flops += 2.0 * M * N * K
        t = timeit(lambda: ops.gemm_fwd(probs_f))
        print(f"{name:22s} fwd   {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TFLOP/s")
        t = timeit(lambda: ops.gemm_wgrad(probs_w))
        print(f"{name:22s} wgrad {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TFLOP/s")
        K0 = shapes[0][1]
        if all(k == K0 for _, k in shapes):
            dA = torch.empty(M, K0, device=dev)
            pd = [dict(dA=dA, Y=A[K0], act=L.ACT_RELU, srcs=[(p["C"], p["W"], 0) for p in probs_f[:8]])]
            fl = sum(2.0 * M * n * k for n, k in shapes[:8])
            t = timeit(lambda: ops.gemm_dgrad(pd))
            print(f"{name:22s} dgrad {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
