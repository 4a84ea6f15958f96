"""The weight-gradient launches of the AE-30 step (B = 65 536) on the cut-once kernel (csrc/gemm_nt.hip) and on the tile
kernel: cold-cache device time of the partial-product launch and of the reduction.  usage: python tools/lab/nt_time.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import _lib as L, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = L.load()
lib.mml_gemm_set_mode(4)
M = 65536
dev = torch.device("cuda:0")
junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def timeit(f):
    ts = []
    for _ in range(reps):
        junk.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        f()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


cases = [("L1 4x(256<-240) + 2x(64<-240)", [(256, 240)] * 4 + [(64, 240)] * 2, True),
         ("L2 4x(128<-256)", [(128, 256)] * 4, False), ("towers 2x(64<-128)", [(64, 128)] * 2, False),
         ("KuaiRec L1 4x(512<-512)+2x(128<-512)", [(512, 512)] * 4 + [(128, 512)] * 2, True)]
g = torch.Generator(device="cpu").manual_seed(1)
s = torch.cuda.current_stream().cuda_stream
for name, shapes, shared in cases:
    probs, A0 = [], None
    for N, K in shapes:
        if not shared or A0 is None:
            A0 = torch.randn(M, K, generator=g).to(dev)
        dC = (torch.randn(M, N, generator=g) * 0.01).to(dev)
        probs.append(dict(dC=dC, A=A0, dW=torch.empty(N, K, device=dev), dbias=torch.empty(N, device=dev)))
    cache = ops._measured([p["dC"] for p in probs] + [p["A"] for p in probs], {})
    key = lambda t: (t.data_ptr(), tuple(t.shape), t.stride(0))  # noqa: E731
    for p in probs:
        p.update(amax_dc=cache[key(p["dC"])], amax_a=cache[key(p["A"])])
    arr = ops.make_wgrad_descs(probs)
    n = len(probs)
    ws = torch.empty(int(lib.mml_gemm_grouped_wgrad_workspace_bytes(arr, n)), dtype=torch.uint8, device=dev)
    flops = sum(2.0 * M * N * K for N, K in shapes)
    out = []
    for nt in (1, 0):
        lib.mml_gemm_set_nt(nt)
        t1 = timeit(lambda: L.check(lib.mml_gemm_grouped_wgrad_phase(arr, n, ws.data_ptr(), ws.numel(), 1, s)))
        t2 = timeit(lambda: L.check(lib.mml_gemm_grouped_wgrad_phase(arr, n, ws.data_ptr(), ws.numel(), 2, s)))
        out.append("%s %.1f us (%.0f TFLOP/s fp32-equivalent) + reduce %.1f us" % (lib.mml_gemm_last_kernel().decode()[:24] if nt == 0 else "gemm_nt_kernel", t1, flops / t1 / 1e6, t2))
    print("%-40s %s | %s" % (name, out[0], out[1]))
lib.mml_gemm_set_nt(1)
