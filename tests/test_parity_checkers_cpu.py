"""The parity checkers of tests/conftest.py, tested themselves (no GPU): a criterion that accepts a wrong result is
worse than none (the round-2 full-size table check was vacuous)."""
import numpy as np

from conftest import table_update_report


def test_table_update_checker_rejects_a_sign_flipped_scatter():
    """Negative control of the checker itself: the update of field 0's touched rows with the wrong sign (what a
    sign-flipped scatter produces under first-step Adam: +lr instead of -lr) must fail, the correct one must pass --
    with 4 514 touched rows of 10 000 000 (B = 8 192 on the 1e7-row table)."""
    rng = np.random.default_rng(0)
    V, E, lr = 10_000_000, 8, 0.005
    before = (rng.standard_normal((V, E)) * 0.05).astype(np.float32)
    rows = np.unique(rng.integers(0, V, 4514))
    upd = (lr * np.sign(rng.standard_normal((len(rows), E)))).astype(np.float32)
    ref = before.copy()
    ref[rows] -= upd
    good = ref.copy()
    good[rows[:5], 0] = before[rows[:5], 0] + upd[:5, 0]  # five noise-level sign flips: inside the allowance
    share, _ = table_update_report(before, good, ref, rows)
    assert 0 < share < 2e-3
    flipped = before.copy()
    flipped[rows] += upd
    share, rel = table_update_report(before, flipped, ref, rows)
    assert share > 0.99 and rel > 1.9
    # the round-2 criterion (outlier share over ALL rows, max <= 2.5 lr) accepted exactly this
    dv = np.abs(flipped.astype(np.float64) - ref)
    assert dv.max() <= 2.5 * lr and (dv > 1e-4 * np.abs(ref).max()).mean() < 2e-3
