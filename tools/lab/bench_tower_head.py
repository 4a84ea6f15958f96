"""K5' against the three launches it replaces (AE-30's towers: 2 x (128 -> 64) + heads), B = 65 536, device time from a replayed
HIP graph of 20 repetitions each."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import mmlrec_amd
from mmlrec_amd import _lib as L, ops
from test_tower_head_gpu import build
lib = L.load(); lib.mml_gemm_set_mode(4)
dev = torch.device("cuda:0")
for M in (65536, 4096):
    y, mask, tasks = build(torch, L, ops, M, 128, 64, 2, True, seed=3)
    prob, loss = torch.empty(M, 2, device=dev), torch.zeros(1, device=dev)
    grp = ops.make_tower_head_group(tasks, prob, y, mask=mask, loss=loss)
    nws = int(lib.mml_tower_head_workspace_bytes(grp)); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    def fused():
        L.check(lib.mml_tower_head_fwd_bwd(grp, ws.data_ptr(), ws.numel(), 0, ops._stream()), "th")
    hs = [torch.empty(M, 64, device=dev) for _ in tasks]
    masks = [torch.zeros(M, 2, dtype=torch.int32, device=dev) for _ in tasks]
    fw = ops.make_fwd_descs([dict(A=q["A"], W=q["W"], bias=q["bias1"], C=h_, act=L.ACT_RELU, amax_a=q["amax_a"], amax_w=q["amax_w"],
                                  w_planes=q["planes_fwd"], w_kexp=q["kexp_fwd"], mask=m_) for q, h_, m_ in zip(tasks, hs, masks)])
    heads = [dict(Hin=h_, w=q["w"], bias=q["hbias"], dH=torch.empty(M, 64, device=dev), dw=torch.empty(64, device=dev),
                  dbias=torch.empty(1, device=dev), h_relu=1, mask_col=q["mask_col"]) for q, h_ in zip(tasks, hs)]
    hg = ops.make_head_group(heads, prob, y=y, mask=mask, loss=loss)
    hws = torch.empty(int(lib.mml_head_workspace_bytes(hg)), dtype=torch.uint8, device=dev)
    slots = ops.amax_slots(2, dev)
    for hd, s_ in zip(heads, slots):
        ops.amax_batch([(hd["dH"], s_)])
    dAs = [torch.empty(M, 128, device=dev) for _ in tasks]   # (the descriptors hold raw pointers: keep the tensors)
    dg = ops.make_dgrad_descs([dict(dA=dA_, Y=None, act=L.ACT_NONE,
                                    srcs=[(hd["dH"], q["W"], 0, s_, q["amax_w"], q["planes_bwd"], q["kexp_bwd"])])
                               for q, hd, s_, dA_ in zip(tasks, heads, slots, dAs)])
    def unfused():
        L.check(lib.mml_gemm_grouped_fwd(fw, 2, ops._stream()), "fwd")
        L.check(lib.mml_head_bce_fwd_bwd(hg, hws.data_ptr(), hws.numel(), ops._stream()), "head")
        L.check(lib.mml_gemm_grouped_dgrad(dg, 2, ops._stream()), "dgrad")
    for name, fn in (("fused", fused), ("three launches", unfused)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph(); st_ = torch.cuda.Stream()
        with torch.cuda.stream(st_):
            with torch.cuda.graph(g_, stream=st_):
                for _ in range(20):
                    fn()
        g_.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g_.replay(); b.record(); torch.cuda.synchronize()
        print("M = %d  %-16s %.1f us per call   (%s)" % (M, name, a.elapsed_time(b) * 1e3 / 20, lib.mml_gemm_last_kernel().decode()))
