#!/usr/bin/env python3
"""Why does scatter_fold_kernel take 4 ms on snr_trans_ae30 (0.1 ms on every other AE-30 workload)?  Statistics of
d(dnn_input) after one eager step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import workloads as W  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "snr_trans_ae30"
B = 65536
dev = torch.device("cuda:0")
model, cfg, vocab, dense = W.build_model(name, dev, table_update="dense_exact", use_hip_graph=False)
model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], ["auc"])
model.train()
step = model.train_step_runner(B, use_graph=False)
X, y = W.synth_batch(vocab, len(dense), B, W.num_tasks(cfg), seed=1)
X, y = X.to(dev), y.to(dev)
step.plan.X.copy_(X)
step.plan.y.copy_(y)
step.run()
torch.cuda.synchronize()
x0 = step.plan.layer_outputs["dnn_input"]
gr = x0.grad
print(name, "loss/sample", float(step.plan.loss) / B)
print("d(dnn_input): shape", tuple(gr.shape), "zeros", float((gr == 0).float().mean()), "nonfinite", int((~torch.isfinite(gr)).sum()),
      "max |g|", float(gr.abs().max()), "min nonzero |g|", float(gr.abs()[gr != 0].min()) if (gr != 0).any() else None)
a = gr.abs()
e = torch.frexp(a[a > 0])[1] if (a > 0).any() else None
if e is not None:
    print("exponent range of nonzero |g|:", int(e.min()), "..", int(e.max()))
for it in range(int(os.environ.get('STEPS', '40'))):
    Xb, yb = W.synth_batch(vocab, len(dense), B, W.num_tasks(cfg), seed=(1 + it % 2) if os.environ.get('REPEAT') else 2 + it)
    step.plan.X.copy_(Xb.to(dev))
    step.plan.y.copy_(yb.to(dev))
    step.run()
    if it % 10 == 9:
        torch.cuda.synchronize()
        gr = x0.grad
        a = gr.abs()
        nz = a[a > 0]
        den = int(((a > 0) & (a < 1.1754944e-38)).sum())
        print("step", it + 2, "loss/sample %.5f" % (float(step.plan.loss) / B), "zeros %.4f" % float((gr == 0).float().mean()),
              "nonfinite", int((~torch.isfinite(gr)).sum()), "denormal", den, "max %.3e" % float(a.max()),
              "min nz %.3e" % (float(nz.min()) if nz.numel() else 0.0), "rows with max<1e-30: %d" % int((a.max(1).values < 1e-30).sum()),
              "col-block (field) max range: %.2e .. %.2e" % (float(a.view(B, -1, 8).amax((0, 2)).min()), float(a.view(B, -1, 8).amax((0, 2)).max())))
import time
from mmlrec_amd import engine as E
torch.cuda.synchronize()
print("bwd_tail:", [(c[0].__name__ if c[0] is not E.PY else "PY", (c[2] if len(c) > 2 and isinstance(c[2], dict) else {}).get("kernel")) for c in step.plan.bwd_tail])
for rep in range(3):
    c = step.plan.bwd_tail[0]
    s = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record(s)
    rc = c[0](*c[1], s.cuda_stream)
    b.record(s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("scatter call: host %.3f ms, events %.3f ms, rc %d" % ((t1 - t0) * 1e3, a.elapsed_time(b), rc))
acc = {}
E.Plan.run_timed(step.plan.bwd_tail, acc)
print({k: v["ms"] for k, v in acc.items()})
acc = {}
E.Plan.run_timed(step.plan.bwd, acc)
E.Plan.run_timed(step.plan.bwd_tail, acc)
print({k: round(v["ms"], 3) for k, v in acc.items()})
