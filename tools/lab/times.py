#!/usr/bin/env python3
"""Per-phase cycle breakdown of the pipelined GEMM (lab build with -DMML_LAB_TIMES): wave 0 of workgroup 0."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MMLREC_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_times.so"))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import _lib as L, ops  # noqa: E402

lib = L.load()
raw = C.CDLL(os.environ["MMLREC_LIB"])
buf = (C.c_ulonglong * 16)()
dev = torch.device("cuda:0")
M = 65536


def report(name, fn):
    fn(); fn()
    torch.cuda.synchronize()
    raw.mml_lab_times(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    torch.cuda.synchronize()
    raw.mml_lab_times(buf, 1)
    t = list(buf)
    steps, tiles = max(t[4], 1), max(t[6], 1)
    us = a.elapsed_time(b) * 1e3
    print(f"{name:10s} {us:7.1f} us  steps {steps} tiles {tiles} | per step (counter ticks): wait+barrier {t[0]/steps:7.1f}  "
          f"issue+scalar {t[1]/steps:7.1f}  reads+blocks12 {t[2]/steps:7.1f}  blocks34 {t[3]/max(steps-tiles,1):7.1f}  "
          f"blocks34+epilogue (last steps) {t[5]/tiles:7.1f} | sum/step {(t[0]+t[1]+t[2]+t[3]+t[5])/steps:7.1f}  "
          f"=> ticks/us {(t[0]+t[1]+t[2]+t[3]+t[5])/us:7.1f}  [per tile: drain {t[7]/tiles:6.0f} epilogue {t[8]/tiles:6.0f} zero {t[9]/tiles:6.0f} advance(cur) {t[10]/tiles:6.0f} | pf switch {t[11]/max(t[12],1):6.0f} x{t[12]}]")


shapes = [(256, 240)] * 4 + [(64, 240)] * 2
A = torch.randn(M, 240, device=dev)
pf, pw = [], []
for N, K in shapes:
    W = torch.randn(N, K, device=dev) / K ** 0.5
    Cc = torch.empty(M, N, device=dev)
    pf.append(dict(A=A, W=W, bias=torch.zeros(N, device=dev), C=Cc, act=L.ACT_RELU))
    pw.append(dict(dC=Cc, A=A, dW=torch.empty(N, K, device=dev), dbias=torch.empty(N, device=dev)))
dA = torch.empty(M, 240, device=dev)
pd = [dict(dA=dA, Y=A, act=L.ACT_RELU, srcs=[(p["C"], p["W"], 0) for p in pf[:8]])]
report("L1 fwd", lambda: ops.gemm_fwd(pf))
report("L1 dgrad", lambda: ops.gemm_dgrad(pd))
report("L1 wgrad", lambda: ops.gemm_wgrad(pw))
A2 = torch.randn(M, 256, device=dev)
p2 = []
for _ in range(4):
    W = torch.randn(128, 256, device=dev) / 16
    p2.append(dict(A=A2, W=W, bias=torch.zeros(128, device=dev), C=torch.empty(M, 128, device=dev), act=L.ACT_RELU))
pd2 = [dict(dA=torch.empty(M, 256, device=dev), Y=A2, act=L.ACT_RELU, srcs=[(p["C"], p["W"], 0)]) for p in p2]
report("L2 fwd", lambda: ops.gemm_fwd(p2))
report("L2 dgrad", lambda: ops.gemm_dgrad(pd2))
