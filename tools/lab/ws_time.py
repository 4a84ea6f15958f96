"""Times the weight-stationary kernel's launches at the benchmark shapes (AE-30, B = 65 536) beside the tile kernel.
usage: python tools/lab/ws_time.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import _lib as L, ops  # noqa: E402
import test_gemm_ws_gpu as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = L.load()
lib.mml_gemm_set_mode(4)
M = 65536
# something that evicts the caches between launches (the step streams 2.3 GB of tables between two uses of a tensor)
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")


def timeit(fn, arr, n):
    ts = []
    for _ in range(reps):
        junk.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(arr, n, ops._stream())
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


cases = [("L2 fwd  4 x 256->128", "f", 256, 128, 4, True), ("tower fwd 2 x 128->64", "f", 128, 64, 2, True),
         ("L2 dgrad 4 x 128->256", "d", 128, 256, 4, True), ("tower dgrad 2 x 64->128", "d", 64, 128, 2, False)]
for name, kind, K, N, n, flag in cases:
    if kind == "f":
        probs = T.fwd_launch(torch, L, ops, M, K, N, n, seed=1)
        for p in probs:
            p["C"] = torch.empty(M, N, device="cuda")
            p["mask"] = torch.zeros(M, (N + 31) // 32, dtype=torch.int32, device="cuda")
        arr = ops.make_fwd_descs(probs)
        fn = lib.mml_gemm_grouped_fwd
        nbytes = n * M * (K + N) * 4
    else:
        probs = T.dgrad_launch(torch, L, ops, M, K, N, n, relu=flag, seed=1)
        for p in probs:
            p["dA"] = torch.empty(M, N, device="cuda")
        arr = ops.make_dgrad_descs(probs)
        fn = lib.mml_gemm_grouped_dgrad
        nbytes = n * M * (K + N) * 4
    out = []
    for ws in (1, 0):
        lib.mml_gemm_set_ws(ws)
        fn(arr, n, ops._stream())
        torch.cuda.synchronize()
        t = timeit(fn, arr, n)
        out.append("%s %6.1f us (%.2f TB/s)" % ("ws  " if ws else "tile", t, nbytes / t / 1e6))
    print("%-26s %s | %s" % (name, out[0], out[1]), flush=True)
