"""The optional C/OpenMP loops of the oracle (oracle/fast.c) against its numpy definition."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import build_fast, mmlrec_oracle as orc


@pytest.fixture()
def fast():
    build_fast.build()
    assert orc.use_fast(True)
    yield
    orc.use_fast(False)


def test_fast_paths_match_numpy(fast):
    g = load_golden("mmoe_ae30d")
    spec = orc.Spec.from_golden(g)
    outs = []
    for enable in (True, False):
        orc.use_fast(enable)
        params = orc.params_from_golden(g)
        opt = orc.DenseOptimizer("adam", 0.005)
        losses = [orc.train_step(spec, params, opt, g[f"X{i}"], g[f"y{i}"]) for i in range(3)]
        x, _ = orc.gather_dnn_input(spec, params, g["X0"])
        outs.append((losses, params, x))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-6)
    assert np.array_equal(outs[0][2], outs[1][2])  # gather stays bit-exact
    for k in outs[0][1]:
        assert np.allclose(outs[0][1][k], outs[1][1][k], rtol=1e-5, atol=1e-7), k


def test_fast_adam_on_a_large_tensor(fast):
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(200_000).astype(np.float32)
    # same gradients for both runs
    gs = [rng.standard_normal(p0.shape).astype(np.float32) * (rng.random(p0.shape) < 0.3) for _ in range(3)]
    finals = []
    for enable in (True, False):
        orc.use_fast(enable)
        params = {"w": p0.copy()}
        opt = orc.DenseOptimizer("adam", 0.01)
        for gsi in gs:
            opt.step(params, {"w": gsi.astype(np.float32)})
        finals.append(params["w"])
    assert np.allclose(finals[0], finals[1], rtol=2e-6, atol=1e-7)
