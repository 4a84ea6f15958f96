"""Property tests (hypothesis) of the host-side arithmetic the HIP kernels share a layout contract with: the row-wise
shard layout (parallel.RowSharding <-> csrc/shard.hip), the table-wise placement, the mark-map layout of the row
bookkeeping (ops.marks_bytes <-> csrc/gather_scatter.hip) and the step segmenter (trainer.Segments)."""
import numpy as np
from hypothesis import given, settings, strategies as st

import mmlrec_amd  # noqa: F401
from mmlrec_amd import engine as E
from mmlrec_amd import ops
from mmlrec_amd.parallel import FieldSharding, RowSharding
from mmlrec_amd.trainer import Segments

vocabs = st.lists(st.integers(min_value=1, max_value=5000), min_size=1, max_size=12)


@settings(max_examples=60, deadline=None)
@given(vocab=vocabs, world=st.integers(min_value=1, max_value=9), emb=st.sampled_from([4, 8, 16]))
def test_row_sharding_is_a_bijection(vocab, world, emb):
    shs = [RowSharding(vocab, emb, world, r) for r in range(world)]
    sh = shs[0]
    assert all(s.R == sh.R and s.keybase == sh.keybase for s in shs)
    assert sh.R == sum(-(-v // world) for v in vocab)
    seen = [set() for _ in range(world)]
    for f, v in enumerate(vocab):
        rows = np.arange(v)
        own = (rows + f) % world
        key = sh.keybase[f] + rows // world
        assert key.min() >= sh.keybase[f] and key.max() < sh.keybase[f + 1]    # inside field f's slice of the flat space
        for k in range(world):
            mine = rows[own == k]
            assert len(mine) == shs[k].owned_rows(f)
            if len(mine):
                assert mine[0] == shs[k].first(f)                                # smallest owned row
                assert np.array_equal(mine, shs[k].first(f) + world * np.arange(len(mine)))  # local row l <-> l*world+first
            ks = set(key[own == k].tolist())
            assert len(ks) == len(mine) and not (ks & seen[k])                   # no two rows share a key on a rank
            seen[k] |= ks
        assert sum(s.owned_rows(f) for s in shs) == v                            # every row owned exactly once
    real = [sum(s.owned_rows(f) for f in range(len(vocab))) for s in shs]
    assert max(real) - min(real) <= len(vocab)                                   # balanced to one row per field


@settings(max_examples=40, deadline=None)
@given(vocab=vocabs, world=st.integers(min_value=1, max_value=8), B=st.integers(min_value=1, max_value=512))
def test_field_sharding_covers_every_field_once(vocab, world, B):
    shs = [FieldSharding(vocab, 8, world, r, batch_per_rank=B) for r in range(world)]
    assert all(s.owner == shs[0].owner for s in shs)
    owned = sorted(f for s in shs for f in s.mine)
    assert owned == list(range(len(vocab)))
    isend, irecv = shs[0].idx_splits(B)
    assert sum(isend) == B * len(vocab) and all(x == B * shs[0].nf[0] for x in irecv)
    rsend, rrecv = shs[0].row_splits(B)
    assert sum(rrecv) == B * len(vocab) * 8


@settings(max_examples=60, deadline=None)
@given(vocab=vocabs)
def test_mark_map_layout(vocab):
    """Field f owns bytes [32 * wordbase_f, 32 * wordbase_f + V_f): fields never overlap and every field starts on a
    32-byte boundary (the compaction reads a bitmap word's 32 mark bytes as two 16-byte loads)."""
    words = [(v + 31) // 32 for v in vocab]
    assert ops.marks_bytes(vocab) == 32 * sum(words)
    base = 0
    for v, w in zip(vocab, words):
        assert base % 32 == 0 and v <= 32 * w
        base += 32 * w


@settings(max_examples=80, deadline=None)
@given(kinds=st.lists(st.booleans(), min_size=0, max_size=20), min_calls=st.integers(min_value=1, max_value=4))
def test_segments_partition_the_call_list(kinds, min_calls):
    """Python-issued entries cut the list; every C call stays in order inside exactly one run."""
    calls, log = [], []
    for i, is_py in enumerate(kinds):
        if is_py:
            calls.append((E.PY, (lambda j: lambda: log.append(("py", j)))(i), (), {}))
        else:
            calls.append(((lambda j: lambda *a: log.append(("c", j)) or 0)(i), ()))
    seg = Segments(calls, use_graph=False, min_calls=min_calls)
    flat = []
    for kind, item, graph in seg.parts:
        assert graph is None
        if kind == "py":
            flat.append(item)
        else:
            assert len(item) >= 1 and all(c[0] is not E.PY for c in item)
            flat.extend(item)
    assert len(flat) == len(calls) and all(a is b for a, b in zip(flat, calls))
    for a, b in zip(seg.parts, seg.parts[1:]):
        assert not (a[0] == "c" and b[0] == "c")                                  # runs are maximal
