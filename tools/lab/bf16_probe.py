import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import mmlrec_amd
from mmlrec_amd import _lib
from conftest import load_golden
from test_models_gpu import build, load_state
lib = _lib.load()
g = load_golden("mmoe_kuairec")
for mode in (4, 1):
    lib.mml_gemm_set_mode(mode)
    model, cfg = build(g)
    load_state(model, g)
    model.train()
    X = torch.from_numpy(g["X0"]).cuda(); y = torch.from_numpy(g["y0"]).cuda()
    yp = model(X)
    loss = sum(torch.nn.functional.binary_cross_entropy(yp[:, i], y[:, i], reduction="sum") for i in range(yp.shape[1]))
    loss.backward()
    worst = 0
    for n, p in model.named_parameters():
        if "grad/" + n in g.files and p.grad is not None:
            ref = g["grad/" + n].astype(np.float64); got = p.grad.cpu().numpy().astype(np.float64)
            r = np.sqrt(np.mean((got - ref) ** 2)) / max(np.sqrt(np.mean(ref ** 2)), 1e-30)
            worst = max(worst, r)
    print("mode", mode, "loss rel", abs(float(loss) - float(g["loss"])) / float(g["loss"]), "worst grad rms rel", worst)
    model, cfg = build(g, table_update="dense_exact")
    load_state(model, g)
    model.compile("adam", cfg["optim_config"]["loss"], ["auc"])
    model.train()
    lr = cfg["optim_config"]["lr"]
    losses = []
    for i in range(3):
        step = model.train_step_runner(64, use_graph=False)
        step.plan.X.copy_(torch.from_numpy(g[f"X{i}"]).cuda()); step.plan.y.copy_(torch.from_numpy(g[f"y{i}"]).cuda())
        step.run(); losses.append(float(step.plan.loss.item()))
    print(" losses", losses, list(g["adam_losses"]))
    sd = model.state_dict()
    stats = []
    for k in sd:
        ref = g[f"adam3/{k}"].astype(np.float64); got = sd[k].cpu().numpy().astype(np.float64)
        dv = np.abs(got - ref)
        stats.append((k, dv.mean() / (lr * 3), (dv > lr).mean(), dv.max() / (lr * 3)))
    stats.sort(key=lambda t: -t[1])
    for s in stats[:6]:
        print("  ", s)
lib.mml_gemm_set_mode(4)
