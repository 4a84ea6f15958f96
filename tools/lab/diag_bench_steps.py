"""Diagnostic: error statistics of the bench-configuration steps against the oracle (per tensor)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import mmlrec_amd
from mmlrec_amd import workloads as W
from oracle import mmlrec_oracle as orc
from conftest import table_update_report
orc.use_fast(True)
dev = torch.device("cuda:0")
tu = sys.argv[1] if len(sys.argv) > 1 else "dense_exact"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model, cfg, vocab, dense = W.build_model("mmoe_ae30", dev, table_update=tu, use_hip_graph=True)
model.compile("adam", cfg["optim_config"]["loss"], ["auc"]); model.train()
names = [f.name for f in model._sparse_cols()]
spec = orc.Spec(cfg, names, vocab, dense)
params = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
before = {k: v.copy() for k, v in params.items()}
B, T, lr = 65536, 2, 0.005
runner = model.train_step_runner(B, use_graph=True, overlap=True, split_dense=False)
opt = orc.DenseOptimizer("adam", lr)
Xs = []
for i in range(nsteps):
    X, y = W.synth_batch(vocab, 0, B, T, seed=1 + i)
    Xs.append(X.numpy())
    runner.plan.X.copy_(X.to(dev)); runner.plan.y.copy_(y.to(dev)); runner.run()
    lg = float(runner.plan.loss.item()); lr_ = orc.train_step(spec, params, opt, X.numpy(), y.numpy())
    print("step", i, lg, lr_, abs(lg - lr_) / lr_)
sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
Xall = np.concatenate(Xs)
for k, ref in params.items():
    got = sd[k].astype(np.float64)
    d_ref = ref.astype(np.float64) - before[k]
    err = np.abs(got - ref)
    if k.startswith("embedding_dict."):
        f = names.index(k.split(".")[1])
        rows = np.unique(Xall[:, f].astype(np.int64))
        share, rel = table_update_report(before[k], sd[k], ref, rows)
        print(f"{k:40s} rows {len(rows):8d} share {share:.2e} rel {rel:.2e} maxupd {np.abs(d_ref).max():.2e}")
    else:
        m = np.abs(ref).max(); u = np.abs(d_ref).max()
        q = [float((err > t * m).mean()) for t in (1e-4, 1e-3, 1e-2)]
        r = err / np.maximum(np.abs(d_ref), 1e-3 * u)
        qr = [float((r > t).mean()) for t in (1e-3, 1e-2, 5e-2)]
        print(f"{k:40s} max|ref| {m:.2e} maxupd {u:.2e} maxerr {err.max():.2e} share>1e-4/1e-3/1e-2 of max: {q} ; err/upd >1e-3/1e-2/5e-2: {qr}")
