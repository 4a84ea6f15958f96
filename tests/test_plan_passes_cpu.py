"""Call-list passes of engine.Plan that need no GPU: Plan._merge_copies (runs of neighbouring strided copies as one
launch).  The pass only looks at the functions' identity and the descriptors, so stand-in functions serve; the merged
lists are EXECUTED here on numpy buffers, with every launch's items applied in a scrambled order (one launch has no
order), and must give what the original list gives call by call."""
import ctypes as C
import types

import numpy as np
import pytest


@pytest.fixture()
def env():
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import _lib as L, engine as E
    lib = types.SimpleNamespace(mml_copy2d=lambda *a: 0, mml_copy2d_batch=lambda *a: 0, other=lambda *a: 0)
    fake = types.SimpleNamespace(keep=[])
    return L, E, lib, fake


def addr(a):
    return a.ctypes.data


def run(lib, calls, scramble=None):
    """Executes copy calls on the numpy buffers their pointers name (float32)."""
    def one(src, lds, dst, ldd, rows, cols, acc):
        s = np.ctypeslib.as_array(C.cast(src, C.POINTER(C.c_float)), shape=((rows - 1) * lds + cols,))
        d = np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_float)), shape=((rows - 1) * ldd + cols,))
        for r in range(rows):
            if acc:
                d[r * ldd:r * ldd + cols] += s[r * lds:r * lds + cols].copy()
            else:
                d[r * ldd:r * ldd + cols] = s[r * lds:r * lds + cols].copy()
    for c in calls:
        if c[0] is lib.mml_copy2d:
            one(*c[1])
        elif c[0] is lib.mml_copy2d_batch:
            arr, n = c[1]
            order = list(range(n))
            if scramble is not None:
                scramble.shuffle(order)
            for k in order:
                d = arr[k]
                one(d.src, d.lds, d.dst, d.ldd, d.rows, d.cols, d.accumulate)


def test_independent_neighbours_become_one_launch(env):
    L, E, lib, fake = env
    a, b, c, d = (np.arange(12, dtype=np.float32).reshape(3, 4) + 100 * i for i in range(4))
    out1, out2 = np.zeros((3, 8), np.float32), np.zeros((3, 4), np.float32)
    calls = [(lib.mml_copy2d, (addr(a), 4, addr(out1), 8, 3, 4, 0)),                 # concat: a | b into out1
             (lib.mml_copy2d, (addr(b), 4, addr(out1) + 16, 8, 3, 4, 0)),
             (lib.mml_copy2d, (addr(c), 4, addr(out2), 4, 3, 4, 0), dict(kernel="copy2d_kernel")),
             (lib.other, (1, 2)),
             (lib.mml_copy2d, (addr(d), 4, addr(out2), 4, 3, 4, 1))]
    merged = E.Plan._merge_copies(fake, calls, lib=lib)
    assert [c[0] for c in merged] == [lib.mml_copy2d_batch, lib.other, lib.mml_copy2d]
    assert merged[0][1][1] == 3 and merged[0][2]["kernel"] == "copy2d_batch_kernel"
    run(lib, merged, scramble=np.random.default_rng(0))
    assert np.array_equal(out1, np.concatenate([a, b], 1)) and np.array_equal(out2, c + d)


def test_dependent_copies_keep_their_order(env):
    L, E, lib, fake = env
    rng = np.random.default_rng(1)
    x, y, z = (rng.standard_normal((4, 6)).astype(np.float32) for _ in range(3))
    t = np.zeros((4, 6), np.float32)
    g = np.zeros((4, 6), np.float32)
    def calls_on(t_, g_, z_):
        return [(lib.mml_copy2d, (addr(x), 6, addr(t_), 6, 4, 6, 0)),       # t = x
                (lib.mml_copy2d, (addr(t_), 6, addr(g_), 6, 4, 6, 1)),      # g += t        (reads what the first wrote)
                (lib.mml_copy2d, (addr(y), 6, addr(g_), 6, 4, 6, 1)),       # g += y        (writes what the second wrote)
                (lib.mml_copy2d, (addr(z_), 6, addr(t_), 6, 4, 6, 0)),      # t = z         (overwrites what the second read)
                (lib.mml_copy2d, (addr(y), 6, addr(z_), 6, 4, 6, 0))]       # z = y         (overwrites what the fourth read)
    t0, g0, z0 = t.copy(), g.copy(), z.copy()
    run(lib, calls_on(t0, g0, z0))
    merged = E.Plan._merge_copies(fake, calls_on(t, g, z), lib=lib)
    # 1 | 2 (reads what 1 wrote) | 3 (adds to what 2 wrote) + 4 (independent of 3; 2 has run by then) | 5 (overwrites what
    # 4 reads): four launches, the third a batch of two
    assert [c[0] is lib.mml_copy2d_batch for c in merged] == [False, False, True, False] and merged[2][1][1] == 2
    for trial in range(5):
        t1, g1, z1 = np.zeros_like(t), np.zeros_like(g), z.copy()
        m = E.Plan._merge_copies(fake, calls_on(t1, g1, z1), lib=lib)
        run(lib, m, scramble=np.random.default_rng(trial))
        assert np.array_equal(t1, t0) and np.array_equal(g1, g0) and np.array_equal(z1, z0)


def test_existing_batches_join_and_strided_ranges_count(env):
    L, E, lib, fake = env
    buf = np.zeros((5, 16), np.float32)
    src = np.arange(20, dtype=np.float32).reshape(5, 4)
    arr = (L.Copy2dDesc * 2)()
    for k, col in enumerate((0, 4)):
        d = arr[k]
        d.src, d.lds, d.dst, d.ldd, d.rows, d.cols, d.accumulate = addr(src), 4, addr(buf) + 4 * col, 16, 5, 4, 0
    calls = [(lib.mml_copy2d_batch, (arr, 2), dict(kernel="copy2d_batch_kernel")),
             (lib.mml_copy2d, (addr(src), 4, addr(buf) + 4 * 8, 16, 5, 4, 0)),      # a third column block: independent
             # the column block 2..5 of the same rows: its strided range interleaves with the blocks above -> the
             # conservative interval test refuses it (a new launch), which is always safe
             (lib.mml_copy2d, (addr(src), 4, addr(buf) + 4 * 2, 16, 5, 4, 1))]
    merged = E.Plan._merge_copies(fake, calls, lib=lib)
    assert [c[0] for c in merged] == [lib.mml_copy2d_batch, lib.mml_copy2d] and merged[0][1][1] == 3
    run(lib, merged, scramble=np.random.default_rng(3))
    want = np.zeros((5, 16), np.float32)
    for col in (0, 4, 8):
        want[:, col:col + 4] = src
    want[:, 2:6] += src
    assert np.array_equal(buf, want)
    # more than 32 items never share a launch
    many = [(lib.mml_copy2d, (addr(src), 4, addr(np.zeros((5, 4), np.float32)), 4, 5, 4, 0)) for _ in range(40)]
    keep = [c[1][2] for c in many]  # (the destination buffers above are temporaries: only the split is checked)
    m = E.Plan._merge_copies(fake, many, lib=lib)
    assert sum(c[1][1] if c[0] is lib.mml_copy2d_batch else 1 for c in m) == 40 and len(m) >= 2 and keep


# ---------------------------------------------------------------------------------------------------------------
# Plan.merge_row_reduces: the deferred head / gate reductions as few launches as the C side's segment capacity allows
# (a PLE with 7 tasks and 4 levels needs 15 + 8 + 8 + 8 + 7 = 46 segments; csrc/reduce.hpp takes 40 per launch)
# ---------------------------------------------------------------------------------------------------------------
def _reduce_plan(env, n_heads, gate_groups):
    """A stand-in plan whose side lists hold the phase-2 calls of one head group (n_heads heads + the loss) and of gate
    groups with the given numbers of active gates (one inactive gate each on top)."""
    L, E, _, _ = env
    lib = types.SimpleNamespace(mml_head_bce_fwd_bwd_phase=lambda *a: 0, mml_gate_mix_bwd_phase=lambda *a: 0,
                                mml_rows_reduce_batch=lambda *a: 0, other=lambda *a: 0)
    hg = L.HeadGroup()
    hg.n_heads, hg.loss = n_heads, 12345
    plan = types.SimpleNamespace(keep=[hg])
    plan.head_side = [(lib.mml_head_bce_fwd_bwd_phase, (C.byref(hg), 1000, 64, 2),
                       dict(kernel="slab_reduce", bytes=64.0, side=True, rank=1, ready=0))]
    plan.bwd_side = []
    for i, n_active in enumerate(gate_groups):
        gg = L.GateGroup()
        gg.n_gates = min(n_active + 1, L.MAX_GATES)
        for k in range(n_active):
            gg.gate[k].active = 1
        plan.keep.append(gg)
        plan.bwd_side.append((lib.other, (i,), dict(kernel="gemm", side=True, ready=i + 1)))
        plan.bwd_side.append((lib.mml_gate_mix_bwd_phase, (C.byref(gg), 2000 + i, 32, 2),
                              dict(kernel="slab_reduce", bytes=32.0, side=True, rank=1, ready=i + 1)))
    return lib, plan


def _segments_of(L, lib, call):
    """Segments a reduction launch of the merged list carries (what csrc/gate_head.hip makes of its items)."""
    def of_group(fn, g):
        if fn is lib.mml_head_bce_fwd_bwd_phase:
            return 2 * g.n_heads + (1 if g.loss else 0)
        return sum(1 for k in range(g.n_gates) if g.gate[k].active)
    if call[0] is lib.mml_rows_reduce_batch:
        items, n = call[1]
        tot = 0
        for k in range(n):
            it = items[k]
            if it.kind == L.ROWS_REDUCE_HEAD:
                tot += of_group(lib.mml_head_bce_fwd_bwd_phase, C.cast(it.group, C.POINTER(L.HeadGroup)).contents)
            else:
                tot += of_group(lib.mml_gate_mix_bwd_phase, C.cast(it.group, C.POINTER(L.GateGroup)).contents)
        return tot, n
    return of_group(call[0], call[1][0]._obj), 1


def test_merge_row_reduces_mmoe_is_one_launch(env):
    L, E, _, _ = env
    lib, plan = _reduce_plan(env, 2, [2])
    assert E.Plan.merge_row_reduces(plan, lib=lib)
    assert plan.head_side == []
    assert [c[0] for c in plan.bwd_side] == [lib.mml_rows_reduce_batch, lib.other]
    assert _segments_of(L, lib, plan.bwd_side[0]) == (2 * 2 + 1 + 2, 2)
    assert plan.bwd_side[0][2]["ready"] == 1


def test_merge_row_reduces_deep_ple_is_chunked(env):
    L, E, _, _ = env
    lib, plan = _reduce_plan(env, 7, [8, 8, 8, 7])           # 15 + 8 + 8 + 8 + 7 = 46 segments
    assert E.Plan.merge_row_reduces(plan, lib=lib)
    reds = [c for c in plan.bwd_side if c[0] is not lib.other]
    assert plan.head_side == [] and len(reds) == 2 and plan.bwd_side[:2] == reds
    segs = [_segments_of(L, lib, c) for c in reds]
    assert all(s <= L.MAX_REDUCE_SEGS for s, _ in segs)
    assert sum(s for s, _ in segs) == 46 and sum(n for _, n in segs) == 5   # every group exactly once
    assert segs[0] == (39, 4) and segs[1] == (7, 1)
    assert reds[1][0] is lib.mml_gate_mix_bwd_phase   # (a chunk of one keeps its own phase-2 call)
    assert sum(1 for c in plan.bwd_side if c[0] is lib.other) == 4


def test_merge_row_reduces_single_group_is_left_alone(env):
    L, E, _, _ = env
    lib, plan = _reduce_plan(env, 2, [])
    before = list(plan.head_side)
    assert not E.Plan.merge_row_reduces(plan, lib=lib)
    assert plan.head_side == before and plan.bwd_side == []


def test_merge_copies_reports_where_every_call_went(env):
    """`where` maps old indices to merged ones: Plan.finish turns the side calls' `ready` tags (counted on the unmerged
    backward chain) into the merged list's index space with it."""
    L, E, lib, fake = env
    bufs = [np.zeros((3, 4), np.float32) for _ in range(8)]
    calls = [(lib.other, (0,)),
             (lib.mml_copy2d, (addr(bufs[0]), 4, addr(bufs[1]), 4, 3, 4, 0)),
             (lib.mml_copy2d, (addr(bufs[2]), 4, addr(bufs[3]), 4, 3, 4, 0)),
             (lib.other, (1,)),
             (lib.mml_copy2d, (addr(bufs[4]), 4, addr(bufs[5]), 4, 3, 4, 0)),
             (lib.mml_copy2d, (addr(bufs[5]), 4, addr(bufs[6]), 4, 3, 4, 0)),   # reads what the previous wrote: its own launch
             (lib.other, (2,))]
    where = []
    merged = E.Plan._merge_copies(fake, calls, lib=lib, where=where)
    assert len(merged) == 6 and where == [0, 1, 1, 2, 3, 4, 5]
    # ready = k (first k old entries issued) -> where[k - 1] + 1 merged entries
    assert [where[k - 1] + 1 for k in (1, 2, 3, 4, 7)] == [1, 2, 2, 3, 6]


def test_fork_conflicts_sees_shared_scratch_and_common_pointers():
    """trainer.fork_conflicts: a fork inside the step's graph is refused when either branch names the shared
    scratch buffer, or both name one pointer."""
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import engine as E, trainer
    f = lambda *a: 0  # noqa: E731
    side = [(f, (0x7f0000100000, 64, 1)), (E.INLINE, f, ()), (f, (0x7f0000200000, 3), dict(kernel="k"))]
    mid = [(f, (0x7f0000300000, 0x7f0000400000, 128)), (E.PY, f, (0x7f0000100000,))]
    assert trainer.fork_conflicts(side, mid) == []
    assert trainer.fork_conflicts(side, mid, shared_scratch=[0x7f0000900000]) == []
    assert trainer.fork_conflicts(side, mid, shared_scratch=[0x7f0000200000]) == [0x7f0000200000]     # side uses the scratch
    assert trainer.fork_conflicts(side, mid, shared_scratch=[0x7f0000400000]) == [0x7f0000400000]     # mid uses it
    mid2 = mid + [(f, (0x7f0000100000, 8))]
    assert trainer.fork_conflicts(side, mid2) == [0x7f0000100000]                                     # a common pointer
    assert trainer.fork_conflicts([(f, (64, 1, True))], [(f, (64, 1))]) == []                         # small integers are sizes
    assert trainer.fork_conflicts([(f, (0x10000, 1 << 24))], [(f, (0x10000, 1 << 24))]) == []         # ... a batch of 65 536 rows too
