"""Prints the GEMM problems (M, K, N, activation, layout, planes?) of every grouped forward / input-gradient launch of a
workload's training plan at a batch size.  usage: python tools/lab/list_gemm_shapes.py <workload> [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import _lib as L, workloads as W  # noqa: E402

wl = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
model, cfg, vocab, dense = W.build_model(wl, dev)
model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], ["auc"])
model.train()
st = model.train_step_runner(B)
lib = L.load()
p = st.plan
for name, calls in (("fwd", p.fwd), ("bwd", p.bwd)):
    for c in calls:
        if c[0] is lib.mml_gemm_grouped_fwd or c[0] is lib.mml_pep_gate_fwd:
            arr, n = c[1][0], c[1][1]
            print(name, "FWD ", [(arr[i].M, arr[i].K, arr[i].N, arr[i].act, arr[i].w_kn, bool(arr[i].w_planes)) for i in range(n)])
        elif c[0] is lib.mml_gemm_grouped_dgrad or c[0] is lib.mml_pep_gate_bwd:
            arr, n = c[1][0], c[1][1]
            print(name, "DGRD", [(arr[i].M, arr[i].K, [arr[i].N[s] for s in range(arr[i].n_src)], arr[i].act,
                                  arr[i].w_kn[0], bool(arr[i].w_planes[0]), bool(arr[i].relu_mask), arr[i].accumulate)
                                 for i in range(n)])
