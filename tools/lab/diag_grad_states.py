"""Diagnostic: gradient accuracy (autograd path AND fused-plan path) at the states of an oracle Adam trajectory."""
import sys, os, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from test_models_gpu import build, load_state
from oracle import mmlrec_oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else "mmoe_ae30"
g = load_golden(name)
spec = orc.Spec.from_golden(g)
params = orc.params_from_golden(g)
cfg = json.loads(str(g["cfg"]))
opt = orc.DenseOptimizer("adam", cfg["optim_config"]["lr"])
for i in range(3):
    X, y = g[f"X{i}"], g[f"y{i}"]
    loss, grads, _ = orc.loss_and_grads(spec, params, X, y)
    model, _ = build(g)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    model.train()
    # fused plan path without the optimizer
    model.compile("adam", cfg["optim_config"]["loss"], ["auc"])
    step = model.train_step_runner(X.shape[0], use_graph=False)
    step.plan.X.copy_(torch.from_numpy(X).cuda()); step.plan.y.copy_(torch.from_numpy(y).cuda())
    step.plan.run_train_fwd_bwd()
    torch.cuda.synchronize()
    st = model._store()
    rows = []
    for k, gr in grads.items():
        pv = st.pvals[k]
        got = pv.grad.cpu().numpy().astype(np.float64)
        d = np.abs(got - gr)
        rows.append((float(d.max() / max(np.abs(gr).max(), 1e-30)), k, float(np.abs(gr).max())))
    rows.sort(reverse=True)
    print(f"state {i}: loss plan {float(step.plan.loss.item()):.5f} oracle {loss:.5f}; worst grad rel errs:", [(f"{a:.1e}", b, f"{c:.1e}") for a, b, c in rows[:5]])
    # magnitudes the plan holds
    pool = step.plan.amax_pool
    if pool is not None:
        vals = pool[:step.plan.amax_next].max(dim=1).values.view(torch.float32).cpu().numpy()
        print("   slots:", " ".join(f"{v:.2e}" for v in vals))
    orc_opt_grads = grads
    opt.step(params, grads)
