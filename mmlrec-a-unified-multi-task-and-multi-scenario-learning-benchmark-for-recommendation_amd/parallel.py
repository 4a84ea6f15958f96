"""Multi-GPU execution: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl"), no reference
behaviour to match (the reference's --is_parallel path is a broken stub).

Scheme ("table-wise sharding"): samples are data-parallel (each rank trains on its own batch shard, MLP gradients are
summed with ONE all-reduce of the flat gradient arena -- the loss is a plain sum, so no averaging); every embedding
TABLE lives on exactly one rank, chosen by a longest-processing-time greedy over (rows to update + lookups to serve).
Per step and per direction there is ONE all-to-all with sizes that are static functions of (B, fields per rank):

  forward : pack X[:, field] columns per owner -> all_to_all(indices) -> owner runs the fused gather kernel on the
            world*B samples it received -> all_to_all(rows) -> unpack into dnn_input [B, F*E + Nd]
  backward: pack d(dnn_input) per owner -> all_to_all(row gradients) -> owner runs the scatter kernel and updates ITS
            tables only (no table all-reduce at all; dense Adam touches 1/world of the rows per rank)

Static split sizes mean no host synchronisation and no index sorting in the step.  Row-wise sharding of a single table
(needed only when one table outgrows a GPU's 288 GB) is a later extension of the same exchange.
"""
import torch

from . import _lib as L
from . import engine as E
from . import ops


class FieldSharding:
    """Which rank owns which sparse field, and where each field sits inside its owner's blocks."""

    def __init__(self, vocab, emb, world, rank, batch_per_rank=65536, lookup_weight=17.0):
        self.world, self.rank, self.emb = int(world), int(rank), int(emb)
        F = len(vocab)
        # cost model in "row updates": updating a row streams ~32 B/float-row-slot; serving one lookup (gather +
        # scatter) moves ~170 B at a lower effective rate -> one lookup ~ `lookup_weight` row updates
        cost = [float(v) + lookup_weight * world * batch_per_rank for v in vocab]
        load = [0.0] * world
        self.owner = [0] * F
        for f in sorted(range(F), key=lambda i: -cost[i]):
            r = min(range(world), key=lambda k: (load[k], k))
            self.owner[f] = r
            load[r] += cost[f]
        self.fields_of = [[f for f in range(F) if self.owner[f] == r] for r in range(world)]
        self.slot = {f: self.fields_of[self.owner[f]].index(f) for f in range(F)}
        self.nf = [len(x) for x in self.fields_of]
        self.mine = self.fields_of[self.rank]

    # split sizes (elements) of the three exchanges for a per-rank batch of B samples
    def idx_splits(self, B):
        send = [B * n for n in self.nf]
        recv = [B * self.nf[self.rank]] * self.world
        return send, recv

    def row_splits(self, B):
        send = [B * self.nf[self.rank] * self.emb] * self.world   # owner -> requester
        recv = [B * n * self.emb for n in self.nf]
        return send, recv

    @staticmethod
    def offsets(splits):
        off, o = [], 0
        for s in splits:
            off.append(o)
            o += s
        return off

    # ---- segment tables: (source view [B,w], destination view [B,w]) pairs --------------------------------
    def pack_index_segments(self, X, cols, send_idx, B):
        """send_idx (flat) = concat over owners j of a [B, nf_j] block holding X[:, col(field)] per slot."""
        send, _ = self.idx_splits(B)
        off = self.offsets(send)
        segs = []
        for j in range(self.world):
            if self.nf[j] == 0:
                continue
            blk = send_idx[off[j]:off[j] + send[j]].view(B, self.nf[j])
            for s, f in enumerate(self.fields_of[j]):
                segs.append((X[:, cols[f]:cols[f] + 1], blk[:, s:s + 1]))
        return segs

    def unpack_row_segments(self, rows_recv, out, B):
        """rows_recv (flat) = concat over owners j of [B, nf_j*E]; out[:, f*E:(f+1)*E] takes slot(f) of owner(f)."""
        _, recv = self.row_splits(B)
        off = self.offsets(recv)
        Em = self.emb
        segs = []
        for j in range(self.world):
            if self.nf[j] == 0:
                continue
            blk = rows_recv[off[j]:off[j] + recv[j]].view(B, self.nf[j] * Em)
            for s, f in enumerate(self.fields_of[j]):
                segs.append((blk[:, s * Em:(s + 1) * Em], out[:, f * Em:(f + 1) * Em]))
        return segs

    def pack_grad_segments(self, d_out, grad_send, B):
        """Inverse of unpack_row_segments: grad_send block j [B, nf_j*E] <- d_out[:, f*E:(f+1)*E]."""
        return [(dst, src) for src, dst in self.unpack_row_segments(grad_send, d_out, B)]


def copy_cols_call(plan, segs, rows, accumulate=0):
    """Turns (src view, dst view) pairs into one mml_copy_cols call entry."""
    n = len(segs)
    src = ops._ptr_array([s for s, _ in segs])
    dst = ops._ptr_array([d for _, d in segs])
    lds = (L.i64 * n)(*[ops._ld(s) for s, _ in segs])
    ldd = (L.i64 * n)(*[ops._ld(d) for _, d in segs])
    wid = (L.i32 * n)(*[s.shape[1] for s, _ in segs])
    plan.keep += [src, dst, lds, ldd, wid]
    return (L.load().mml_copy_cols, (src, lds, dst, ldd, wid, n, rows, accumulate),
            dict(kernel="copy_cols_kernel", bytes=8.0 * rows * sum(s.shape[1] for s, _ in segs)))


class ShardedGatherOp(E.Op):
    """K1/K2 across ranks: index / row / gradient exchange around the local fused gather and scatter kernels."""

    def __init__(self, sharding, dist, tables, X, cols, dense_col0, nd, out, sparse_rows=None, group=None):
        self.sh, self.dist, self.group = sharding, dist, group
        self.tables, self.X, self.cols, self.dense_col0, self.nd, self.out = tables, X, cols, dense_col0, nd, out
        self.sparse_rows = sparse_rows

    def outputs(self):
        return [self.out]

    def _a2a(self, out, inp, out_splits, in_splits):
        self.dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def fwd_calls(self, plan):
        sh, lib = self.sh, L.load()
        B, W, Em = plan.B, sh.world, sh.emb
        F = len(self.tables)
        nfm = sh.nf[sh.rank]
        isend, irecv = sh.idx_splits(B)
        rsend, rrecv = sh.row_splits(B)
        self.send_idx = plan.empty(max(sum(isend), 1))
        self.recv_idx = plan.empty(max(sum(irecv), 1))
        self.rows_send = plan.empty(max(sum(rsend), 1))
        self.rows_recv = plan.empty(max(sum(rrecv), 1))
        calls = [copy_cols_call(plan, sh.pack_index_segments(self.X, self.cols, self.send_idx, B), B)]
        calls.append((E.PY, self._a2a, (self.recv_idx[:sum(irecv)], self.send_idx[:sum(isend)], irecv, isend),
                      dict(kernel="all_to_all(indices)")))
        if nfm:
            mine = [self.tables[f] for f in sh.mine]
            tabs = ops._ptr_array([t.data for t in mine])
            vocab = (L.i64 * nfm)(*[t.data.shape[0] for t in mine])
            col = (L.i32 * nfm)(*range(nfm))
            plan.keep += [tabs, vocab, col]
            calls.append((lib.mml_gather_fwd, (tabs, vocab, col, nfm, Em, self.recv_idx.data_ptr(), nfm, 0, 0, W * B,
                                               self.rows_send.data_ptr(), nfm * Em, plan.status.data_ptr()),
                          dict(kernel="gather_vec4_kernel", bytes=float(W * B) * nfm * (4 + 8 * Em))))
        calls.append((E.PY, self._a2a, (self.rows_recv[:sum(rrecv)], self.rows_send[:sum(rsend)], rrecv, rsend),
                      dict(kernel="all_to_all(rows)")))
        segs = sh.unpack_row_segments(self.rows_recv, self.out.buf, B)
        if self.nd:
            segs.append((self.X[:, self.dense_col0:self.dense_col0 + self.nd], self.out.buf[:, F * Em:F * Em + self.nd]))
        calls.append(copy_cols_call(plan, segs, B))
        return calls

    def bwd_calls(self, plan):
        sh, lib = self.sh, L.load()
        if self.out.grad is None:
            return []
        B, W, Em = plan.B, sh.world, sh.emb
        nfm = sh.nf[sh.rank]
        rsend, rrecv = sh.row_splits(B)  # gradient traffic runs the row exchange backwards
        self.grad_send = plan.empty(max(sum(rrecv), 1))
        self.grad_recv = plan.empty(max(sum(rsend), 1))
        pk = copy_cols_call(plan, sh.pack_grad_segments(self.out.grad, self.grad_send, B), B)
        pk[2]["tail"] = True
        calls = [pk]
        calls.append((E.PY, self._a2a, (self.grad_recv[:sum(rsend)], self.grad_send[:sum(rrecv)], rsend, rrecv),
                      dict(kernel="all_to_all(row grads)", tail=True)))
        if nfm:
            mine = [self.tables[f] for f in sh.mine]
            if not all(t.needs_grad for t in mine):
                raise L.MMLError("sharded tables need gradient accumulators (ParamStore.ensure_table_grads)")
            gt = ops._ptr_array([t.grad for t in mine])
            vocab = (L.i64 * nfm)(*[t.data.shape[0] for t in mine])
            col = (L.i32 * nfm)(*range(nfm))
            plan.keep += [gt, vocab, col]
            for t in mine:
                E._claim(t)
            sr = self.sparse_rows
            if sr is not None:
                seen = ops._ptr_array(sr.seen)
                rb = (L.i64 * (nfm + 1))(*sr.rowbase)
                plan.keep += [seen, rb]
                extra = (seen, rb, sr.touched.data_ptr(), sr.count.data_ptr(), sr.touched.numel())
            else:
                extra = (None, None, None, None, 0)
            calls.append((lib.mml_scatter_bwd, (gt, vocab, col, nfm, Em, self.recv_idx.data_ptr(), nfm, W * B,
                                                self.grad_recv.data_ptr(), nfm * Em) + extra +
                          (plan.status.data_ptr(),),
                          dict(kernel="scatter_hash_kernel", bytes=float(W * B) * nfm * (4 + 12 * Em), tail=True)))
        return calls


def make_allreduce(dist, group=None):
    """Sum of the flat MLP-gradient arena over ranks (loss is reduction='sum', model/basemodel.py:295)."""
    def allreduce(arena):
        dist.all_reduce(arena, op=dist.ReduceOp.SUM, group=group)
    return allreduce


def shard_model(model, dist, batch_per_rank, group=None):
    """Switch a model to table-wise sharded execution on the current process group."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    vocab = [f.vocabulary_size for f in model._sparse_cols()]
    model._sharding = FieldSharding(vocab, model.embedding_size, world, rank, batch_per_rank)
    model._dist, model._dist_group = dist, group
    model._caches = {"store": model._caches.get("store"), "plans": {}, "steps": {}}
    return model._sharding


def sync_tables(model):
    """Broadcast every table from its owner so all ranks hold the trained rows (before eval / state_dict)."""
    sh = getattr(model, "_sharding", None)
    if sh is None:
        return
    for f, feat in enumerate(model._sparse_cols()):
        w = model.embedding_dict[feat.embedding_name].weight.data
        model._dist.broadcast(w, src=sh.owner[f], group=model._dist_group)
