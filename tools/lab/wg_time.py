"""The cut-once weight-gradient kernel (csrc/gemm_wgws.hip) at the AE-30 shapes (B = 65 536): float64 error and time beside
the tile kernel.  usage: MMLREC_LIB=<lib with mml_gemm_wgws_wgrad> python tools/lab/wg_time.py [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import _lib as L, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = L.load()
from mmlrec_amd import build as _b  # noqa: E402
raw = C.CDLL(os.environ.get("MMLREC_LIB") or _b.LIBPATH)
fn = raw.mml_gemm_wgws_wgrad
fn.restype = C.c_int
fn.argtypes = [C.POINTER(L.GemmWgradDesc), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
wsq = raw.mml_gemm_wgws_wgrad_workspace_bytes
wsq.restype = C.c_int64
wsq.argtypes = [C.POINTER(L.GemmWgradDesc), C.c_int32]
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


cases = [("L2 4 x (128 <- 256)", [(128, 256)] * 4, False), ("towers 2 x (64 <- 128)", [(64, 128)] * 2, False),
         ("L1 gates 2 x (64 <- 240)", [(64, 240)] * 2, True), ("128 <- 128 x 4", [(128, 128)] * 4, False)]
g = torch.Generator(device="cpu").manual_seed(1)
for name, shapes, shared in cases:
    probs, A0 = [], None
    for N, K in shapes:
        if not shared or A0 is None:
            A0 = torch.randn(M, K, generator=g).to(dev)
        dC = (torch.randn(M, N, generator=g) * 0.01).to(dev)
        probs.append(dict(dC=dC, A=A0, dW=torch.full((N, K), float("nan"), device=dev),
                          dbias=torch.full((N,), float("nan"), device=dev), accumulate=0, w_kn=0))
    cache = ops._measured([p["dC"] for p in probs] + [p["A"] for p in probs], {})
    key = lambda t: (t.data_ptr(), tuple(t.shape), t.stride(0))  # noqa: E731
    for p in probs:
        p.update(amax_dc=cache[key(p["dC"])], amax_a=cache[key(p["A"])])
    arr = ops.make_wgrad_descs(probs)
    n = len(probs)
    nb = int(wsq(arr, n))
    assert nb > 0, "not served"
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = ops._stream()
    L.check(fn(arr, n, ws.data_ptr(), nb, st), "mml_gemm_wgws_wgrad")
    torch.cuda.synchronize()
    err = 0.0
    for p in probs:
        ref = p["dC"].double().t() @ p["A"].double()
        err = max(err, float((p["dW"].double() - ref).abs().max() / ref.abs().max()))
        rb = p["dC"].double().sum(0)
        err = max(err, float((p["dbias"].double() - rb).abs().max() / rb.abs().max()))
    t_new = timeit(lambda: fn(arr, n, ws.data_ptr(), nb, st))
    nb2 = int(lib.mml_gemm_grouped_wgrad_workspace_bytes(arr, n))
    ws2 = torch.empty(nb2, dtype=torch.uint8, device=dev)
    t_old = timeit(lambda: lib.mml_gemm_grouped_wgrad(arr, n, ws2.data_ptr(), nb2, st))
    print("%-40s err %.2e   cut-once %7.1f us | tile %7.1f us" % (name, err, t_new, t_old), flush=True)
