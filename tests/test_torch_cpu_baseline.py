"""Pins oracle/torch_cpu.py -- the torch-CPU restatement bench.py times as `cpu_baseline.torch_cpu` -- against fixtures the
unmodified reference produced (tests/golden/make_golden.py): same ATen kernels in the same order, so the forward is
bit-identical and the rest agrees to fp32 rounding."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import torch_cpu as tc

CASES = ["mmoe_ae30", "mmoe_ae30_s4", "mmoe_kuairec", "sharedbottom_ml"]


def _state(g, prefix="state/"):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("name", CASES)
def test_forward_loss_gradients_match_reference_fixture(name):
    g = load_golden(name)
    spec = tc.Spec.from_golden(g)
    p = tc.params_from_numpy(_state(g))
    X, y = torch.from_numpy(g["X0"]), torch.from_numpy(g["y0"])
    yp = tc.forward(spec, p, X)
    assert np.array_equal(yp.detach().numpy(), g["y_pred"])  # bit for bit
    assert np.array_equal(tc.dnn_input(spec, p, X).detach().numpy(), g["dnn_input"])
    if "y_pred_masked" in g.files:
        ym = tc.forward(spec, p, X, torch.from_numpy(g["mask0"]))
        assert np.array_equal(ym.detach().numpy(), g["y_pred_masked"])
    loss = tc.loss_sum(yp, y)
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    loss.backward()
    for k, v in p.items():
        ref = g["grad/" + k]
        assert np.abs(v.grad.numpy() - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-30), k


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("kind", ["adam", "adagrad"])
def test_three_optimizer_steps_match_reference_fixture(name, kind):
    g = load_golden(name)
    spec = tc.Spec.from_golden(g)
    p = tc.params_from_numpy(_state(g))
    opt = tc.make_optimizer(kind, p, spec.cfg["optim_config"]["lr"])
    losses = [tc.train_step(spec, p, opt, torch.from_numpy(g[f"X{i}"]), torch.from_numpy(g[f"y{i}"])) for i in range(3)]
    assert np.allclose(losses, g[f"{kind}_losses"], rtol=1e-6)
    for k, v in p.items():
        ref = g[f"{kind}3/{k}"]
        assert np.abs(v.detach().numpy() - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-30), (kind, k)
