#!/usr/bin/env python3
"""Runs one grouped GEMM case a few times (for rocprofv3 --pmc passes). usage: prof_gemm.py [fwd|dgrad|wgrad] [case]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import _lib as L, ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
case = sys.argv[2] if len(sys.argv) > 2 else "l1"
dev = torch.device("cuda:0")
M = 65536
shapes = {"l1": [(256, 240)] * 4 + [(64, 240)] * 2, "l2": [(128, 256)] * 4, "sq": [(4096, 4096)]}[case]
if case == "sq":
    M = 4096
A = torch.randn(M, shapes[0][1], device=dev)
pf, pw = [], []
for N, K in shapes:
    W = torch.randn(N, K, device=dev) / K ** 0.5
    Cc = torch.empty(M, N, device=dev)
    pf.append(dict(A=A, W=W, bias=torch.zeros(N, device=dev), C=Cc, act=L.ACT_RELU))
    pw.append(dict(dC=Cc, A=A, dW=torch.empty(N, K, device=dev), dbias=torch.empty(N, device=dev)))
dA = torch.empty(M, shapes[0][1], device=dev)
pd = [dict(dA=dA, Y=A, act=L.ACT_RELU, srcs=[(p["C"], p["W"], 0) for p in pf[:8]])]
ops.gemm_fwd(pf)
torch.cuda.synchronize()
for _ in range(3):
    if what == "fwd":
        ops.gemm_fwd(pf)
    elif what == "dgrad":
        ops.gemm_dgrad(pd)
    else:
        ops.gemm_wgrad(pw)
torch.cuda.synchronize()
