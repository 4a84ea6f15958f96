#!/usr/bin/env python3
"""snr_trans_ae30 with HIP graphs and two streams: where do the 4 ms the instrumented pass charges to the scatter go?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import engine as E, workloads as W  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "snr_trans_ae30"
overlap = (sys.argv[2] != "serial") if len(sys.argv) > 2 else True
B = 65536
dev = torch.device("cuda:0")
model, cfg, vocab, dense = W.build_model(name, dev, table_update="dense_exact", use_hip_graph=True)
model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], ["auc"])
model.train()
step = model.train_step_runner(B, use_graph=True, overlap=overlap)
bs = [tuple(t.to(dev) for t in W.synth_batch(vocab, len(dense), B, W.num_tasks(cfg), seed=1 + i)) for i in range(2)]
for it in range(12):
    step.plan.X.copy_(bs[it % 2][0]); step.plan.y.copy_(bs[it % 2][1])
    step.run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for it in range(10):
    step.plan.X.copy_(bs[it % 2][0]); step.plan.y.copy_(bs[it % 2][1])
    step.run()
b.record(); torch.cuda.synchronize()
print(name, "overlap", overlap, "step %.3f ms" % (a.elapsed_time(b) / 10))
p = step.plan
for rep in range(2):
    acc = {}
    E.Plan.run_timed(p.bwd_tail, acc)
    print("bwd_tail alone:", {k: round(v["ms"], 3) for k, v in acc.items()})
lists = dict(fwd=p.fwd, head=p.head_train, bwd=p.bwd, tail=p.bwd_tail, side=p.bwd_side, opt=step.opt_calls)
acc = {}
E.Plan.run_timed([c for l in lists.values() for c in l], acc)
print("one list:", {k[:40]: round(v["ms"], 3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["ms"])[:6]})
# which predecessor makes the scatter slow?  time it right after each phase
for ph in ("fwd", "head", "bwd", "side", "opt"):
    acc = {}
    E.Plan.run_timed(list(lists[ph]) + list(p.bwd_tail), acc)
    print("after", ph, ":", round(acc.get("scatter_fold_kernel", {"ms": -1})["ms"], 3))
acc = {}
bw = list(p.bwd)
for cut in (len(bw) // 4, len(bw) // 2, 3 * len(bw) // 4, len(bw) - 1):
    acc = {}
    E.Plan.run_timed(bw[cut:] + list(p.bwd_tail), acc)
    print("after bwd[%d:]" % cut, [(c[2] if len(c) > 2 and isinstance(c[2], dict) else {}).get("kernel", c[0].__name__ if c[0] is not E.PY else "PY") for c in bw[cut:cut + 1]], round(acc["scatter_fold_kernel"]["ms"], 3))
