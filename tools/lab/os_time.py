#!/usr/bin/env python3
"""Stand-alone device time of the d(dnn_input) launch: gemm_os_kernel against the tile kernel (MMLREC_GEMM_OS = 1 / 0),
each replayed 20x from a HIP graph, cold-ish caches (a 1 GB fill between replays is NOT done: both see the same state).
usage: os_time.py [M] [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: E402


def main():
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib as L, ops
    import test_gemm_os_gpu as T
    lib = L.load()
    lib.mml_gemm_set_mode(4)
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 240
    for Ns in ([256, 256, 256, 256, 64, 64], [512, 512, 512, 512, 128, 128]):
        prob, _ = T.launch(torch, L, ops, M, K, Ns, seed=1)
        dev = torch.device("cuda:0")
        prob = dict(prob, dA=torch.empty(M, K, device=dev), accumulate=0)
        for on in ("1", "0", "1", "0"):
            os.environ["MMLREC_GEMM_OS"] = on
            ops.gemm_dgrad([prob])
            torch.cuda.synchronize()
            name = lib.mml_gemm_last_kernel().decode()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    for _ in range(20):
                        ops.gemm_dgrad([prob])
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / 20
            fl = 2.0 * M * K * sum(Ns) * 3
            print(f"M={M} K={K} N={Ns}: {name[:48]:48s} {us:7.1f} us  {fl / us / 1e6:6.0f} TFLOP/s (16-bit products)")


main()
