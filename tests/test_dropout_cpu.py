"""Dropout (reference model/utils.py:121, :159): the CPU side.  The mask stream of this build is Philox4x32-10 keyed by
(seed, step, layer, row, column / 4) (include/mmlrec.h: mml_dropout); the oracle restates it in numpy.  Here: the
restatement against the published known-answer vectors of the generator, and the distribution / arithmetic nn.Dropout
has (kept elements scaled by 1 / (1 - p), dropped ones zero, rate p, masks independent across steps and layers)."""
import numpy as np
import pytest

from oracle import mmlrec_oracle as orc

# known-answer vectors of Philox4x32-10 (Random123 distribution, kat_vectors: counter words, key words, output words)
KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = orc.philox4x32(*[np.array([c]) for c in ctr], key[0], key[1])
        assert tuple(int(w[0]) for w in got) == want
    # vectorised over counters: the same words as one call per counter
    c = np.arange(1000, dtype=np.uint64)
    many = orc.philox4x32(c, c * 7, c * 0 + 3, c * 0 + 9, 123, 456)
    for i in (0, 1, 17, 999):
        one = orc.philox4x32(np.array([i]), np.array([i * 7]), np.array([3]), np.array([9]), 123, 456)
        assert [int(w[i]) for w in many] == [int(w[0]) for w in one]


@pytest.mark.parametrize("p", [0.1, 0.3, 0.5, 0.9])
def test_dropout_scale_distribution_and_arithmetic(p):
    rows, cols = 4096, 130  # (cols not a multiple of 4: the last Philox block of a row is cut)
    s = orc.dropout_scale(rows, cols, p, seed=0x1234567890abcdef, step=1, site=42)
    keep = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    assert s.dtype == np.float32 and set(np.unique(s)) <= {np.float32(0), keep}
    rate = float((s == 0).mean())
    n = rows * cols
    assert abs(rate - p) < 5 * np.sqrt(p * (1 - p) / n)  # five sigma of a Bernoulli(p) mean
    assert abs(float(s.mean()) - 1.0) < 5 * np.sqrt(p / (1 - p) / n)  # E[scale] = 1: activations keep their mean
    # columns and rows are not correlated (lag-1 agreement of the drop indicator is p^2 + (1-p)^2)
    d = (s == 0)
    agree = p * p + (1 - p) * (1 - p)
    assert abs(float((d[:, 1:] == d[:, :-1]).mean()) - agree) < 0.01
    assert abs(float((d[1:] == d[:-1]).mean()) - agree) < 0.01
    # another step / layer / seed = another mask, the same call = the same mask
    assert np.array_equal(s, orc.dropout_scale(rows, cols, p, 0x1234567890abcdef, 1, 42))
    for other in (dict(seed=0x1234567890abcdee, step=1, site=42), dict(seed=0x1234567890abcdef, step=2, site=42),
                  dict(seed=0x1234567890abcdef, step=1, site=43)):
        o = orc.dropout_scale(rows, cols, p, **other)
        assert abs(float(((o == 0) == d).mean()) - agree) < 0.01


def test_oracle_dnn_dropout_forward_backward_consistent():
    """dnn_fwd / dnn_bwd with dropout armed: the backward is the gradient of the forward (finite differences in
    float64 on a small stack), dropout off in eval mode, and the MLP model's blocks stay dropout-free."""
    rng = np.random.default_rng(0)
    params = {"tower.linears.0.weight": rng.standard_normal((8, 6)).astype(np.float32),
              "tower.linears.0.bias": rng.standard_normal(8).astype(np.float32),
              "tower.linears.1.weight": rng.standard_normal((4, 8)).astype(np.float32),
              "tower.linears.1.bias": rng.standard_normal(4).astype(np.float32)}
    x = rng.standard_normal((16, 6)).astype(np.float32)
    orc.set_dropout(0.4, seed=7, step=3)
    try:
        orc.set_training(False)
        y_eval, _ = orc.dnn_fwd(params, "tower", x)
        orc.set_training(True)
        y, acts = orc.dnn_fwd(params, "tower", x)
        assert len(acts[0]) == 4 and (acts[0][3] == 0).any()
        assert not np.array_equal(y, y_eval)
        w = rng.standard_normal(y.shape).astype(np.float32)
        grads = {}
        dx = orc.dnn_bwd(params, "tower", acts, w.copy(), grads)
        k, idx, eps = "tower.linears.0.weight", (3, 2), 1e-3
        vals = []
        for sgn in (+1, -1):
            q = {a: b.copy() for a, b in params.items()}
            q[k][idx] += sgn * eps
            vals.append(float((orc.dnn_fwd(q, "tower", x)[0].astype(np.float64) * w).sum()))
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - float(grads[k][idx])) < 2e-2 * max(abs(fd), 1.0)
        assert dx.shape == x.shape
        q = {a.replace("tower", "mlp_layers.0"): b for a, b in params.items()}
        y_mlp, a_mlp = orc.dnn_fwd(q, "mlp_layers.0", x)  # (reference model/mlp.py: DNN blocks built without dropout)
        assert len(a_mlp[0]) == 3 and np.array_equal(y_mlp, y_eval)
    finally:
        orc.set_dropout(0)
        orc.set_training(False)
