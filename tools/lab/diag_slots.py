"""Diagnostic: after one eager forward (+backward) of a golden model, compare every magnitude slot with the true max |x|
of the tensor it stands for."""
import sys, os, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from test_models_gpu import build, load_state
from mmlrec_amd import engine as E, ops
name = sys.argv[1] if len(sys.argv) > 1 else "mmoe_ae30"
init = len(sys.argv) > 2 and sys.argv[2] == "init"
g = load_golden(name)
model, cfg = build(g)
if not init:
    load_state(model, g)
model.compile("adam", cfg["optim_config"]["loss"], ["auc"]); model.train()
X = torch.from_numpy(g["X0"]).cuda(); y = torch.from_numpy(g["y0"]).cuda()
step = model.train_step_runner(X.shape[0], use_graph=False)
plan = step.plan
plan.X.copy_(X); plan.y.copy_(y)
plan.run_train_fwd_bwd(); torch.cuda.synchronize()
seen = set()
def show(tag, slot, t):
    if slot is None or t is None: return
    v = ops.amax_value(slot); true = float(t.abs().max())
    flag = "" if v >= true else "   <<<<<< LOW"
    print(f"{tag:40s} slot {v:.4e} true {true:.4e}{flag}")
for op in plan.ops:
    for v in list(op.inputs()) + list(op.outputs()):
        if isinstance(v, E.Val) and id(v) not in seen:
            seen.add(id(v))
            show("val " + v.name, v.amax, v.buf)
            if v.grad is not None: show("grad " + v.name, v.gamax, v.grad)
for (t, slot) in plan.amax_wlist:
    show("weight " + str(tuple(t.shape)), slot, t)
print("loss", float(plan.loss.item()), "golden", float(g["loss"]) if not init else None)
