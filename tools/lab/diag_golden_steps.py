"""Diagnostic: outlier share per tensor of the golden Adam trajectory, and gradient errors, for MMLREC_AMAX=0/1."""
import sys, os, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from test_models_gpu import build, load_state
name = sys.argv[1] if len(sys.argv) > 1 else "mmoe_ae30"
g = load_golden(name)
for tu in ("dense_exact",):
    model, cfg = build(g, table_update=tu)
    load_state(model, g)
    model.compile("adam", cfg["optim_config"]["loss"], ["auc"]); model.train()
    lr = cfg["optim_config"]["lr"]
    for i in range(3):
        X = torch.from_numpy(g[f"X{i}"]).cuda(); y = torch.from_numpy(g[f"y{i}"]).cuda()
        step = model.train_step_runner(X.shape[0], use_graph=False)
        step.plan.X.copy_(X); step.plan.y.copy_(y); step.run()
        print("loss", i, float(step.plan.loss.item()), float(g["adam_losses"][i]))
        if i + 1 in (1, 3):
            sd = model.state_dict()
            for k in sd:
                ref = g[f"adam{i+1}/{k}"].astype(np.float64)
                dv = np.abs(sd[k].cpu().numpy().astype(np.float64) - ref)
                sh = (dv > 1e-4 * max(np.abs(ref).max(), 1e-30)).mean()
                if sh > 2e-4:
                    print(f"  step {i+1} {k:40s} share {sh:.2e} n_bad {(dv > 1e-4 * np.abs(ref).max()).sum()} max {dv.max():.2e} maxref {np.abs(ref).max():.2e}")
# gradient accuracy through autograd
model, cfg = build(g); load_state(model, g); model.train()
X = torch.from_numpy(g["X0"]).cuda(); y = torch.from_numpy(g["y0"]).cuda()
yp = model(X)
loss = sum(torch.nn.functional.binary_cross_entropy(yp[:, i], y[:, i], reduction="sum") for i in range(yp.shape[1]))
loss.backward()
worst = []
for n, p in model.named_parameters():
    if "grad/" + n in g.files and p.grad is not None:
        r = g["grad/" + n].astype(np.float64); d = np.abs(p.grad.cpu().numpy() - r)
        worst.append((float(d.max() / max(np.abs(r).max(), 1e-30)), n))
worst.sort(reverse=True)
print("grad rel errors (max-norm), worst 6:", worst[:6])
