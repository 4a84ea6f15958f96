"""Diagnosis: per-gradient relative rms between GEMM mode 4 (fp32-equivalent), mode 1 on fp32 buffers and mode 1 with
bf16 storage, KuaiRec-32 MMoE at a small batch.  usage: python tools/lab/diag_bf16_modes.py [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mmlrec_amd  # noqa
from mmlrec_amd import _lib, workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
dev = torch.device("cuda:0")
res = {}
for tag, mode, storage in (("fp32", 4, "0"), ("m1_f32buf", 1, "0"), ("m1_bf16", 1, "1")):
    lib.mml_gemm_set_mode(mode)
    os.environ["MMLREC_BF16_STORAGE"] = storage
    model, cfg, vocab, dense = W.build_model("mmoe_kuairec", dev, vocab_scale=0.05, seed=0, table_update="dense_exact")
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 2:
                sc = 0.1 if n.startswith("embedding") else (2.0 / p.shape[1]) ** 0.5
                p.copy_((torch.randn(p.shape, generator=g) * sc).to(dev))
    T = W.num_tasks(cfg)
    X, y = W.synth_batch(vocab, len(dense), B, T, seed=5)
    model.compile("adam", cfg["optim_config"]["loss"], ["auc"])
    model.train()
    step = model.train_step_runner(B, use_graph=False)
    step.plan.X.copy_(X.to(dev))
    step.plan.y.copy_(y.to(dev))
    step.plan.run_train_fwd_bwd()
    torch.cuda.synchronize()
    st = model._store()
    res[tag] = {k: pv.grad.detach().float().cpu().numpy().astype(np.float64) for k, pv in st.pvals.items()
                if pv.grad is not None and not pv.is_table}
    names = []
    for c in list(step.plan.fwd) + list(step.plan.bwd) + list(step.plan.bwd_side):
        names.append(getattr(c[0], "__name__", "?"))
    print(tag, "loss", float(step.plan.loss.item()), "calls:", {n: names.count(n) for n in sorted(set(names))})
    del model, step


def rr(a, b):
    return np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30)


for k in res["fp32"]:
    print(f"{k:36s} f32buf-vs-fp32 {rr(res['m1_f32buf'][k], res['fp32'][k]):.5f}  bf16-vs-fp32 {rr(res['m1_bf16'][k], res['fp32'][k]):.5f}"
          f"  bf16-vs-f32buf {rr(res['m1_bf16'][k], res['m1_f32buf'][k]):.5f}")
