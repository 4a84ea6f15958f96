"""Dropout on the MI355X (reference model/utils.py:121, :159).  The kernel's mask is a pure function of (seed, step,
layer, row, column) -- Philox4x32-10, restated by the oracle and pinned to the generator's published vectors in
tests/test_dropout_cpu.py -- so kernel and model results are compared with the oracle under the SAME mask: bit-exact for
the kernel, the 1e-4 contract for losses / gradients / fused steps.  (torch's own mask stream for a seed cannot be
reproduced: what is shared with the reference is the arithmetic and the distribution.)"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_models_gpu import RTOL, build, elem_rel, gpu_state, load_state, rel

pytestmark = pytest.mark.gpu
P = 0.3


@pytest.mark.parametrize("rows,cols,pitched", [(333, 37, False), (1024, 64, False), (257, 48, True), (1, 5, False),
                                               (4096, 256, True)])
def test_dropout_kernel_is_the_oracle_mask(rows, cols, pitched):
    from mmlrec_amd import ops
    from oracle import mmlrec_oracle as orc
    rng = np.random.default_rng(rows + cols)
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    seed, site, step = 0xfeedfacecafebeef, 0x9e3779b9, 5
    want = x * orc.dropout_scale(rows, cols, P, seed, step, site)
    if pitched:
        xb = torch.zeros(rows, cols + 16, device="cuda")
        ob = torch.full((rows, cols + 32), 7.0, device="cuda")
        xd, od = xb[:, 8:8 + cols], ob[:, 16:16 + cols]
        xd.copy_(torch.from_numpy(x))
    else:
        xd, od = torch.from_numpy(x).cuda(), torch.empty(rows, cols, device="cuda")
    ops.dropout(xd, od, P, seed, site, step)
    assert np.array_equal(od.cpu().numpy(), want)
    if pitched:  # nothing outside the view was written
        assert float(ob[:, :16].min()) == 7.0 and float(ob[:, 16 + cols:].min()) == 7.0
    # the step word from device memory (what a replayed HIP graph reads), accumulation, in place
    sd = torch.tensor([step], dtype=torch.int32, device="cuda")
    acc = torch.ones(rows, cols, device="cuda")
    ops.dropout(xd, acc, P, seed, site, step=99, step_dev=sd, accumulate=True)
    assert np.array_equal(acc.cpu().numpy(), (want + np.float32(1.0)).astype(np.float32))
    inp = xd.clone()
    ops.dropout(inp, inp, P, seed, site, step)
    assert np.array_equal(inp.cpu().numpy(), want)
    sd += 1  # another step: another mask, with the same rate
    ops.dropout(xd, od, P, seed, site, step=0, step_dev=sd)
    nxt = od.cpu().numpy()
    assert np.array_equal(nxt, x * orc.dropout_scale(rows, cols, P, seed, step + 1, site))
    if rows * cols > 10000:
        assert not np.array_equal(nxt, want) and abs(float((nxt == 0).mean()) - P) < 0.02
    # a data-parallel rank's slice of the global batch draws the global mask
    if rows > 100:
        ops.dropout(xd[40:90], od[40:90], P, seed, site, step, row0=40)
        assert np.array_equal(od[40:90].cpu().numpy(), want[40:90])
    # p = 0 is the identity, p outside [0, 1) is refused
    ops.dropout(xd, od, 0.0, seed, site, step)
    assert np.array_equal(od.cpu().numpy(), x)
    from mmlrec_amd import _lib
    with pytest.raises(_lib.MMLError):
        ops.dropout(xd, od, 1.0, seed, site, step)


def _spec(g, p):
    from oracle import mmlrec_oracle as orc
    cfg = json.loads(str(g["cfg"]))
    cfg["model_config"]["dnn_dropout"] = p
    return orc.Spec(cfg, [str(s) for s in g["sparse_names"]], g["vocab"], [str(s) for s in g["dense_names"]])


CASES = ["sharedbottom_ml", "mmoe_ae30", "ple_ijcai", "cross_stitch_ae", "hmoe_ml", "aitm_ml", "esmm_ml", "mssm_ml",
         "snr_trans_ae",
         # Linear -> BatchNorm -> ReLU -> Dropout (model/utils.py:153-159), stacked and as blocks writing into a
         # concatenation buffer
         "sharedbottom_bn", "mmoe_bn", "cross_stitch_bn", "mssm_bn"]


@pytest.mark.parametrize("name", CASES)
def test_model_gradients_with_dropout_match_the_oracle_under_the_same_mask(name):
    from oracle import mmlrec_oracle as orc
    g = load_golden(name)
    model, cfg = build(g, dnn_dropout=P)
    load_state(model, g)
    X, y = torch.from_numpy(g["X0"]).cuda(), torch.from_numpy(g["y0"]).cuda()
    # evaluation mode: dropout is the identity (the golden forward of the dropout-free fixture)
    model.eval()
    with torch.no_grad():
        assert rel(model(X).cpu().numpy(), g["y_pred"]) < RTOL
    model.train()
    yp = model(X)
    bce = torch.nn.functional.binary_cross_entropy
    loss = sum(bce(yp[:, i], y[:, i], reduction="sum") for i in range(yp.shape[1]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) / float(g["loss"]) > 1e-6  # (the mask did something)
    spec = _spec(g, P)
    params = orc.params_from_golden(g)
    frozen = orc.params_from_golden(g, "frozen/")
    # (no optimizer: the plan owns its step counter and every training-mode forward bumps it first -- this was forward 1)
    orc.set_dropout(P, seed=model.dropout_seed, step=1)
    try:
        ref_loss, ref_grads, _ = orc.loss_and_grads(spec, params, g["X0"], g["y0"], frozen or None)
    finally:
        orc.set_dropout(0)
    assert abs(float(loss.detach()) - ref_loss) / ref_loss < RTOL
    checked = 0
    from conftest import bn_noise_keys
    noise_bias, _ = bn_noise_keys(model.state_dict().keys())
    for n, p in model.named_parameters():
        if n in noise_bias:  # bias in front of a BatchNorm: structurally zero gradient, rounding noise on both sides
            continue
        if n in ref_grads and p.grad is not None:
            assert rel(p.grad.cpu().numpy(), ref_grads[n]) < RTOL, n
            if n.startswith("embedding_dict."):
                assert elem_rel(p.grad.cpu().numpy(), ref_grads[n]) <= 1.0, n
            checked += 1
    assert checked >= 4
    # a second forward of the uncompiled model draws ANOTHER mask (the counter used to stay at 0), the
    # oracle's for step word 2
    with torch.no_grad():
        yp2 = model(X).cpu().numpy()
    assert np.abs(yp2 - yp.detach().cpu().numpy()).max() > 1e-6
    orc.set_dropout(P, seed=model.dropout_seed, step=2)
    try:
        _, _, cache2 = orc.loss_and_grads(spec, params, g["X0"], g["y0"], frozen or None)
    finally:
        orc.set_dropout(0)
    assert rel(yp2, cache2["p"]) < RTOL


@pytest.mark.parametrize("name,graph", [("sharedbottom_ml", False), ("mmoe_ae30", True), ("ple_ijcai", True)])
def test_fused_steps_with_dropout_follow_the_oracle(name, graph):
    """Three fused Adam steps with dropout 0.3: every step is the oracle's step from the MI355X's own state under the
    mask of that step (the device step counter the optimizer bumps: 1, 2, 3), HIP graph replay included."""
    from conftest import table_update_report
    from oracle import mmlrec_oracle as orc
    g = load_golden(name)
    model, cfg = build(g, dnn_dropout=P, table_update="dense_exact")
    load_state(model, g)
    model.compile("adam", cfg["optim_config"]["loss"], ["auc"])
    model.train()
    lr = cfg["optim_config"]["lr"]
    spec = _spec(g, P)
    losses = []
    for i in range(3):
        X, y = torch.from_numpy(g[f"X{i}"]).cuda(), torch.from_numpy(g[f"y{i}"]).cuda()
        step = model.train_step_runner(X.shape[0], use_graph=graph)
        sd, mom, t = gpu_state(model)
        step.plan.X.copy_(X)
        step.plan.y.copy_(y)
        step.run()
        losses.append(float(step.plan.loss.item()))
        after = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        params = {k: v.copy() for k, v in sd.items()}
        opt = orc.DenseOptimizer("adam", lr)
        opt.t = t
        for k, (m1, m2) in mom.items():
            if m1 is not None:
                opt.state[k] = {"m": m1.copy(), "v": m2.copy()}
        orc.set_dropout(P, seed=model.dropout_seed, step=t + 1)
        try:
            ref_loss = orc.train_step(spec, params, opt, g[f"X{i}"], g[f"y{i}"])
        finally:
            orc.set_dropout(0)
        assert abs(losses[-1] - ref_loss) / ref_loss < RTOL, (i, losses[-1], ref_loss)
        for k, r in params.items():
            b, a = sd[k], after[k]
            if np.abs(r - b).max() == 0 and np.abs(a - b).max() == 0:
                continue
            b2 = b.reshape(b.shape[0], -1) if b.ndim > 1 else b.reshape(1, -1)
            rows = np.nonzero(np.abs(r.reshape(b2.shape) - b2).max(1) + np.abs(a.reshape(b2.shape) - b2).max(1))[0]
            share, _ = table_update_report(b2, a.reshape(b2.shape), r.reshape(b2.shape), rows)
            assert share < 2e-3, (i, k, share)
    assert len(set(losses)) == 3
