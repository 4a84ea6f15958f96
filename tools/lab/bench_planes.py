#!/usr/bin/env python3
"""Forward GEMM on pre-cut planes vs the in-register-cut kernel, AE-30 shapes at M = 65 536 (HIP-event timed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import ops, _lib as L
    dev = torch.device("cuda:0")
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    for name, K, Ns in (("L1 4x(240->256)+2x(240->64)", 240, [256] * 4 + [64] * 2), ("L2 4x(256->128)", 256, [128] * 4),
                        ("KuaiRec L1 4x(512->512)", 512, [512] * 4)):
        g = torch.Generator(device="cpu").manual_seed(0)
        A = torch.randn(M, K, generator=g).to(dev)
        pa = ops.Planes(M, K, dev)
        Ws = [(torch.randn(N, K, generator=g) / K ** 0.5).to(dev) for N in Ns]
        bs = [torch.randn(N, generator=g).to(dev) for N in Ns]
        pws = [ops.Planes(N, K, dev) for N in Ns]
        Cs = [torch.empty(M, N, device=dev) for N in Ns]
        Cs2 = [torch.empty(M, N, device=dev) for N in Ns]
        masks = [torch.zeros(M, (N + 31) // 32, dtype=torch.int32, device=dev) for N in Ns]
        t_cutA = timed(lambda: ops.planes_cut([(A, pa, False)]))
        t_cutW = timed(lambda: ops.planes_cut([(W, p, False) for W, p in zip(Ws, pws)]))
        newp = [dict(A=pa, W=p, bias=b, C=c, act=L.ACT_RELU, mask=m) for p, b, c, m in zip(pws, bs, Cs, masks)]
        oldp = [dict(A=A, W=W, bias=b, C=c, act=L.ACT_RELU, mask=m) for W, b, c, m in zip(Ws, bs, Cs2, masks)]
        t_new = timed(lambda: ops.gemm_planes_fwd(newp))
        import ctypes
        lib = L.load()
        if hasattr(lib, "mml_lab_planes_times"):
            buf = (ctypes.c_ulonglong * 8)()
            lib.mml_lab_planes_times(buf, 1)
            ops.gemm_planes_fwd(newp)
            torch.cuda.synchronize()
            lib.mml_lab_planes_times(buf, 1)
            t = list(buf)
            n = max(t[5], 1)
            print(f"   cycles per step (wave 0 of workgroup 0, {t[5]} steps, {t[6]} tiles): vmcnt wait {t[0] / n:.0f}, barrier "
                  f"{t[1] / n:.0f}, body {t[2] / n:.0f}, lgkm wait {t[3] / n:.0f}, epilogue+rest {t[4] / n:.0f}")
        t_old = timed(lambda: ops.gemm_fwd(oldp))
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(Cs, Cs2))
        fl = 2.0 * M * K * sum(Ns)
        print(f"{name}: planes {t_new:.1f} us ({fl / t_new / 1e6:.0f} TFLOP/s)  in-register cuts {t_old:.1f} us "
              f"({fl / t_old / 1e6:.0f} TFLOP/s)  cut A {t_cutA:.1f} us  cut W {t_cutW:.1f} us  max rel diff {err:.2e}",
              flush=True)


if __name__ == "__main__":
    main()
