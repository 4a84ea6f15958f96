"""Diagnosis: element-wise update errors of the MLP tensors after n fused steps of a secondary configuration against the
oracle (tests/test_fullsize_gpu.py::test_bench_secondary_configurations_steps_match_oracle).
usage: python tools/lab/diag_secondary_steps.py workload B nsteps [graph]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mmlrec_amd  # noqa
from mmlrec_amd import workloads as W
from oracle import mmlrec_oracle as orc
from conftest import table_update_report
from test_fullsize_gpu import _randomize

wl, B, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
graph = len(sys.argv) > 4 and sys.argv[4] == "graph"
dev = torch.device("cuda:0")
orc.use_fast(True)
model, cfg, vocab, dense = W.build_model(wl, dev, table_update="auto", use_hip_graph=graph)
frozen = _randomize(model, 11)
names = [f.name for f in model._sparse_cols()]
spec = orc.Spec(cfg, names, vocab, dense)
params = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
before = {k: v.copy() for k, v in params.items()}
T = W.num_tasks(cfg)
kind, lr = cfg["optim_config"]["optimizer"], cfg["optim_config"]["lr"]
model.compile(kind, cfg["optim_config"]["loss"], cfg["optim_config"]["metrics"])
model.train()
runner = model.train_step_runner(B, use_graph=graph)
opt = orc.DenseOptimizer(kind, lr)
for i in range(nsteps):
    X, y = W.synth_batch(vocab, len(dense), B, T, seed=1 + i)
    runner.load(X.to(dev), y.to(dev))
    runner.run()
    lg = float(runner.plan.loss.item())
    if i == nsteps - 1:  # gradients of the last step, both sides
        _, grads_ref, _ = orc.loss_and_grads(spec, params, X.numpy(), y.numpy(), frozen or None)
        st = runner.store
        ggpu = {n: pv.grad.detach().cpu().numpy().copy() for n, pv in st.pvals.items() if pv.grad is not None and not pv.is_table}
    lr_ = orc.train_step(spec, params, opt, X.numpy(), y.numpy(), frozen or None)
    print("step", i, "loss", lg, lr_, abs(lg - lr_) / lr_)
sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
for k, ref in params.items():
    if k.startswith("embedding_dict."):
        continue
    b = before[k] if before[k].ndim else before[k].reshape(1)
    rows = np.arange(b.shape[0])
    share, rel = table_update_report(b, sd[k].reshape(b.shape), ref.reshape(b.shape), rows)
    line = f"{k:44s} share {share:.5f} rel {rel:.3f}"
    if k in ggpu and k in grads_ref:
        # NOTE: ggpu is the gradient of the LAST gpu step at the gpu's parameters; grads_ref at the oracle's
        g, r = ggpu[k].astype(np.float64), grads_ref[k].astype(np.float64)
        line += f" | grad maxrel {np.abs(g - r).max() / max(np.abs(r).max(), 1e-30):.2e} |g|max {np.abs(r).max():.3e} frac|g|<1e-4max {(np.abs(r) < 1e-4 * np.abs(r).max()).mean():.4f}"
    print(line)
    if share > 1e-3 and b.ndim == 2:
        d_ref = ref.astype(np.float64) - b
        d_got = sd[k].astype(np.float64) - b
        tol = np.maximum(np.maximum(0.05 * np.abs(d_ref), 1e-3 * np.abs(d_ref).max()), 2.0 * np.spacing(np.abs(b)))
        bad = np.abs(d_got - d_ref) > tol
        print("    bad per row (top):", np.sort(bad.sum(1))[-8:], "rows with any bad:", int((bad.sum(1) > 0).sum()), "of", b.shape[0],
              "| per col (top):", np.sort(bad.sum(0))[-8:], "cols with any bad:", int((bad.sum(0) > 0).sum()), "of", b.shape[1])
        if k in grads_ref:
            r = np.abs(grads_ref[k])
            print("    |g_ref| of bad elements: median %.3e, all elements median %.3e, max %.3e" % (np.median(r[bad]), np.median(r), r.max()))
