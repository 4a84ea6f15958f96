"""Diagnostic: same runner over 3 steps; after every step compare the MLP gradients left in the arena with the oracle's
gradients at the oracle trajectory's state; overlap on/off."""
import sys, os, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from test_models_gpu import build, load_state
from oracle import mmlrec_oracle as orc
name, overlap = sys.argv[1], sys.argv[2] == "1"
g = load_golden(name)
spec = orc.Spec.from_golden(g); params = orc.params_from_golden(g)
cfg = json.loads(str(g["cfg"]))
opt = orc.DenseOptimizer("adam", cfg["optim_config"]["lr"])
model, cfg = build(g, table_update="dense_exact"); load_state(model, g)
model.compile("adam", cfg["optim_config"]["loss"], ["auc"]); model.train()
for i in range(3):
    X, y = g[f"X{i}"], g[f"y{i}"]
    loss, grads, _ = orc.loss_and_grads(spec, params, X, y)
    step = model.train_step_runner(X.shape[0], use_graph=False, overlap=overlap)
    step.plan.X.copy_(torch.from_numpy(X).cuda()); step.plan.y.copy_(torch.from_numpy(y).cuda()); step.run()
    torch.cuda.synchronize()
    st = model._store()
    rows = []
    for k, gr in grads.items():
        pv = st.pvals[k]
        if pv.is_table: continue
        d = np.abs(pv.grad.cpu().numpy().astype(np.float64) - gr)
        rows.append((float(d.max() / max(np.abs(gr).max(), 1e-30)), k))
    rows.sort(reverse=True)
    print(f"step {i} overlap={overlap}: worst MLP grad rel errs:", [(f"{a:.1e}", b) for a, b in rows[:4]])
    pool = step.plan.amax_pool
    if pool is not None:
        vals = pool[:step.plan.amax_next].max(dim=1).values.view(torch.float32).cpu().numpy()
        print("   slots:", " ".join(f"{v:.2e}" for v in vals))
    opt.step(params, grads)
    sd = model.state_dict()
    bad = []
    for k in sd:
        ref = params[k].astype(np.float64); dv = np.abs(sd[k].cpu().numpy() - ref)
        sh = (dv > 1e-4 * max(np.abs(ref).max(), 1e-30)).mean()
        if sh > 2e-3: bad.append((k, f"{sh:.1e}"))
    print("   params vs oracle trajectory, share > 2e-3:", bad[:6])
    top = sorted(((float(np.abs(sd[k].cpu().numpy() - params[k]).max()), k) for k in sd), reverse=True)[:4]
    print("   largest |param - oracle|:", [(f"{a:.2e}", b) for a, b in top])
    for a, k in top[:2]:
        d = np.abs(sd[k].cpu().numpy() - params[k]); idx = np.unravel_index(d.argmax(), d.shape)
        print("     at", k, idx, "oracle grad there", grads[k][idx], "grad max", np.abs(grads[k]).max())
