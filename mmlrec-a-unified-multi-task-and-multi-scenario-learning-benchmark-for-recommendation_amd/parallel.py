"""Multi-GPU execution: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl"), no reference
behaviour to match (the reference's --is_parallel path is a broken stub, main.py:81-83, model/basemodel.py:235-238,
SURVEY 2.1 / D6).  Contract: an N-rank step on a batch split N ways == a 1-rank step on the whole batch.

Samples are data-parallel in every mode: each rank trains on its own batch shard and the MLP gradients are summed with
ONE all-reduce of the flat gradient arena (the loss is a plain sum, so no averaging).  The tables come in three modes:

  row_sharded (default; the one BASELINE.json's configs 4 / 5 name): row r of field f lives on rank (r + f) mod N at
      local row r // N; a rank keeps its F shards back to back in ONE flat [R, E] buffer (`embedding_shard`), so a
      lookup travels as a single int32 key into the owner's flat row space and the owner-side gather / scatter /
      optimizer are single-table launches.  Per step and direction ONE all-to-all over all fields:
        forward : route (count, place) -> all_to_all(keys) -> owner gather -> all_to_all(rows) -> expand to dnn_input
        backward: pack d(dnn_input) -> all_to_all(row gradients) -> owner scatter + optimizer on its shard only
      Per-owner counts depend on the data: they are exchanged first (one tiny all-to-all + ONE host read per step).
      Every rank updates 1/N of every table (dense Adam streams 1/N of the rows), so the load is balanced by
      construction and `lazy_exact` works unchanged on the flat shard.
  replicated: every rank holds every table and gathers locally; the (index, row-gradient) pairs of all ranks are
      all-gathered (fixed sizes, no host read) and every rank applies the same update to its copy.  Float atomics make
      the copies differ in the last bits; `sync_tables` re-broadcasts rank 0's copy.
  table_wise: every TABLE lives on one rank (longest-processing-time greedy); static all-to-all split sizes.  Kept
      from round 1; unbalanced for AE-30 (the owner of the 1e7-row table carries 80 % of the dense-Adam rows).

The full `embedding_dict.<name>.weight` parameters stay on every rank (state_dict / predict contract, 400 MB for
AE-30); in row_sharded mode they are stale while training and `sync_tables` (all-gather of the shards) refreshes them --
BaseModel.flush_tables does that before anything outside the fused step reads a table.  All of these are collective
calls: every rank must reach them.
"""
import torch

from . import _lib as L
from . import engine as E
from . import ops


class CollectiveError(RuntimeError):
    """A collective failed on this rank: the ranks are out of step and the job has to end."""


def exit_on_collective_error(fn, *args, **kwargs):
    """Run fn(*args, **kwargs); a CollectiveError ends the PROCESS at once with exit code 70 (traceback printed, streams
    flushed) instead of unwinding through interpreter / ProcessGroup teardown with a broken communicator -- that can block
    while the other ranks sit in the collective until the RCCL timeout.  The launcher (torch.distributed.run) then reaps
    the other ranks.  Used by main.run (--is_parallel) and bench.py; a library caller of fit() under its own launcher
    wraps its entry point the same way (or sets MMLREC_COMM_EXIT=1: exit inside the failing collective)."""
    try:
        return fn(*args, **kwargs)
    except CollectiveError:
        import os
        import sys
        import traceback
        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(70)


class Comm:
    """The few collectives the step needs, on one process group.  RCCL ("nccl") moves device buffers directly; under
    "gloo" (the world-2 tests: two processes on one GPU or on CPU) device buffers are staged through the host."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.staged = dist.get_backend(group) == "gloo"
        # what was issued, by kind: calls / bytes this rank sent to OTHER ranks / bytes it received from them (host-side
        # bookkeeping only; bench.py prints the per-step figures so that a multi-GPU run can be checked line by line
        # against the per-N table of DESIGN section 5)
        self.stats = {}

    def __deepcopy__(self, memo):
        return self

    def _count(self, kind, sent, received):
        e = self.stats.setdefault(kind, {"calls": 0, "bytes_sent": 0, "bytes_received": 0})
        e["calls"] += 1
        e["bytes_sent"] += int(sent)
        e["bytes_received"] += int(received)

    def stats_snapshot(self):
        return {k: dict(v) for k, v in self.stats.items()}

    @staticmethod
    def stats_delta(a, b, per=1):
        """(b - a) / per for two stats_snapshot() results."""
        out = {}
        for k, v in b.items():
            w = a.get(k, {})
            d = {f: (v[f] - w.get(f, 0)) / per for f in v}
            if d["calls"]:
                out[k] = d
        return out

    def _stage(self, t):
        return t.detach().cpu() if (self.staged and t.is_cuda) else t

    def _call(self, what, fn, *a, **k):
        """A failed collective leaves the ranks out of step: every later exchange would hang or mix batches.  Say which
        rank failed in what and re-raise as CollectiveError; launcher scripts (bench.py, main.run: exit_on_collective_error
        above) turn that into exit code 70 of this rank at once (failure detection; the launcher reaps the others).  Errors that never
        left this rank (a bad split list, a dtype mismatch: TypeError / ValueError) pass through unchanged.  With
        MMLREC_COMM_EXIT=1 a backend failure ends the process at once (for jobs without such a launcher)."""
        try:
            return fn(*a, **k)
        except (TypeError, ValueError):
            raise
        except Exception as e:  # RCCL / gloo errors surface as RuntimeError / DistBackendError
            import os
            import sys
            import traceback
            sys.stderr.write(f"[mmlrec rank {self.rank}/{self.world}] collective {what} failed: {e!r}\n")
            traceback.print_exc()
            sys.stdout.flush()
            sys.stderr.flush()
            if os.environ.get("MMLREC_COMM_EXIT") == "1":
                os._exit(70)
            raise CollectiveError(f"rank {self.rank}/{self.world}: collective {what} failed: {e!r}") from e

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        o, i = self._stage(out), self._stage(inp)
        row_i = inp.element_size() * (inp.numel() // max(inp.shape[0], 1)) if inp.dim() else inp.element_size()
        row_o = out.element_size() * (out.numel() // max(out.shape[0], 1)) if out.dim() else out.element_size()
        if in_splits is not None:
            sent = sum(int(n) for r, n in enumerate(in_splits) if r != self.rank) * row_i
        else:
            sent = inp.numel() * inp.element_size() * (self.world - 1) // self.world
        if out_splits is not None:
            recv = sum(int(n) for r, n in enumerate(out_splits) if r != self.rank) * row_o
        else:
            recv = out.numel() * out.element_size() * (self.world - 1) // self.world
        self._count("all_to_all", sent, recv)
        self._call("all_to_all_single", self.dist.all_to_all_single, o, i, out_splits, in_splits, group=self.group)
        if o is not out:
            out.copy_(o)

    def all_reduce(self, t):
        s = self._stage(t)
        # (a ring moves 2 (N - 1) / N of the buffer per rank and direction)
        ring = 2 * t.numel() * t.element_size() * (self.world - 1) // self.world
        self._count("all_reduce", ring, ring)
        self._call("all_reduce(sum)", self.dist.all_reduce, s, op=self.dist.ReduceOp.SUM, group=self.group)
        if s is not t:
            t.copy_(s)

    def all_reduce_max(self, t):
        s = self._stage(t)
        ring = 2 * t.numel() * t.element_size() * (self.world - 1) // self.world
        self._count("all_reduce", ring, ring)
        self._call("all_reduce(max)", self.dist.all_reduce, s, op=self.dist.ReduceOp.MAX, group=self.group)
        if s is not t:
            t.copy_(s)

    def all_gather_into_tensor(self, out, inp):
        o, i = self._stage(out), self._stage(inp)
        nb = inp.numel() * inp.element_size() * (self.world - 1)
        self._count("all_gather", nb, nb)
        if self.staged:
            parts = list(o.view(self.world, -1).unbind(0))
            self._call("all_gather", self.dist.all_gather, parts, i.reshape(-1).contiguous(), group=self.group)
        else:
            self._call("all_gather_into_tensor", self.dist.all_gather_into_tensor, o, i, group=self.group)
        if o is not out:
            out.copy_(o)

    def broadcast(self, t, src):
        s = self._stage(t)
        nb = t.numel() * t.element_size()
        self._count("broadcast", nb * (self.world - 1) if self.rank == src else 0, 0 if self.rank == src else nb)
        self._call("broadcast", self.dist.broadcast, s, src=src, group=self.group)
        if s is not t:
            t.copy_(s)

    def barrier(self):
        self._call("barrier", self.dist.barrier, group=self.group)


# ======================================================================================================
# row-wise sharding
# ======================================================================================================
class RowSharding:
    """Host-side arithmetic of the row-wise layout (the device side is csrc/shard.hip)."""

    def __init__(self, vocab, emb, world, rank):
        self.vocab = [int(v) for v in vocab]
        self.emb, self.world, self.rank = int(emb), int(world), int(rank)
        self.rows_local = [(v + self.world - 1) // self.world for v in self.vocab]  # same on every rank
        self.keybase = [0]
        for n in self.rows_local:
            self.keybase.append(self.keybase[-1] + n)
        self.R = self.keybase[-1]
        if self.R >= 2 ** 31:
            raise L.MMLError("row-sharded tables: a rank's flat row space must fit int32 keys")

    def owner(self, f, r):
        return (r + f) % self.world

    def key(self, f, r):
        return self.keybase[f] + r // self.world

    def first(self, f, rank=None):
        """Smallest row of field f that rank owns."""
        rank = self.rank if rank is None else rank
        return (rank - f) % self.world

    def owned_rows(self, f, rank=None):
        """Number of REAL rows of field f on a rank (the shard may end in padding rows)."""
        v, first = self.vocab[f], self.first(f, rank)
        return 0 if first >= v else (v - 1 - first) // self.world + 1

    # ---- device helpers ---------------------------------------------------------------------------
    def tables_to_shard(self, tables, shard=None, rank=None):
        """flat[keybase[f] + l] = tables[f][l * world + first(f)] (padding rows zero)."""
        dev = tables[0].device
        if shard is None:
            shard = torch.empty(self.R, self.emb, dtype=torch.float32, device=dev)
        lib, s = L.load(), ops._stream()
        for f, t in enumerate(tables):
            L.check(lib.mml_shard_rows(t.data_ptr(), t.shape[0], shard[self.keybase[f]:].data_ptr(), self.rows_local[f],
                                       self.emb, self.world, self.first(f, rank), 0, s), "mml_shard_rows")
        return shard

    def shard_to_tables(self, shard, tables, rank=None):
        lib, s = L.load(), ops._stream()
        for f, t in enumerate(tables):
            L.check(lib.mml_shard_rows(t.data_ptr(), t.shape[0], shard[self.keybase[f]:].data_ptr(), self.rows_local[f],
                                       self.emb, self.world, self.first(f, rank), 1, s), "mml_shard_rows")


class _RouteSet:
    """Index-only state of one batch on the row-sharded path (see RowShardedGatherOp.fwd_calls)."""

    def __init__(self, op, plan, k):
        sh, dev = op.sh, plan.device
        F, W, B = len(sh.vocab), sh.world, plan.B
        self.counters = torch.zeros(2 * W, dtype=torch.int32, device=dev)
        self.cnt_pair = torch.zeros(2, W, dtype=torch.int32, device=dev)
        self.host = torch.zeros(2, W, dtype=torch.int32)
        if dev.type == "cuda":
            self.host = self.host.pin_memory()
        self.ready = torch.cuda.Event() if dev.type == "cuda" else None
        self.send_keys = torch.empty(B * F, dtype=torch.int32, device=dev)
        self.pos = torch.empty(B, F, dtype=torch.int32, device=dev)
        if op.dedup:
            self.req = E.TableRows(sh.vocab, dev, B * F)      # requester-side row set of the batch (full vocabulary)
            self.req_seen = ops._ptr_array(self.req.seen)
            self.req_rb = (L.i64 * (F + 1))(*self.req.rowbase)
            self.slot_of = torch.empty(self.req.rowbase[-1], dtype=torch.int32, device=dev)


class RowShardedGatherOp(E.Op):
    """K1/K2 over row-sharded tables.  The exchange sizes are only known at run time, so forward and backward are one
    Python-issued entry each (engine.PY): kernels + collectives launched eagerly; everything between them has static
    shapes and is replayed from HIP graphs by the trainer."""
    owns_lazy = True  # the lazy-exact catch-up runs on the owner side, inside the forward exchange

    def __init__(self, par, shard_pv, X, cols, dense_col0, nd, out, sparse_rows=None):
        self.par, self.sh, self.comm = par, par.sharding, par.comm
        self.shard, self.X, self.cols, self.dense_col0, self.nd, self.out = shard_pv, X, cols, dense_col0, nd, out
        self.sparse_rows = sparse_rows
        self.lazy_launch = None
        self.tables = [shard_pv]   # (engine.Optimizer: the one table this op's scatter feeds)
        self.grad_marks = None     # byte per shard row: marked-gradient dense update (engine.GatherOp.grad_marks)
        # requester-side de-duplication: the exchange carries every DISTINCT row of the local batch once (under Zipf a
        # 65 536-sample AE-30 batch holds 209 k distinct rows against 1.97 M lookups), gradients of duplicates are
        # summed locally before they travel
        self.dedup = bool(getattr(par, "dedup", True))
        self.stats = {"n_recv": 0, "n_sent": 0, "lookups": 0, "steps": 0}

    def outputs(self):
        return [self.out]

    # ---- build time --------------------------------------------------------------------------------
    def fwd_calls(self, plan):
        sh, dev = self.sh, plan.device
        F, W, B = len(sh.vocab), sh.world, plan.B
        self.F, self.B = F, B
        self.col = (L.i32 * F)(*self.cols)
        self.vocab = (L.i64 * F)(*sh.vocab)
        self.keybase = (L.i64 * F)(*sh.keybase[:F])
        # Everything that depends only on the batch's INDICES -- the distinct rows, their owners, the send order, the
        # per-owner counts (exchanged and read by the host: the split sizes of the all-to-alls) and the position of
        # every lookup in the returned row block -- lives in a route set.  There are two: while step k runs, the trainer
        # routes batch k + 1 on a side stream (prefetch_route), so the step itself starts at all_to_all(keys) and never
        # waits for a device -> host read.
        self.routes = [_RouteSet(self, plan, k) for k in range(2)]
        self.cur = self.routes[0]     # the set the forward / backward exchange of the current step uses
        self.staged = None            # a set routed ahead of time for the NEXT step, or None
        # (a stream that really runs beside the step's: ops.concurrent_stream probes the hardware-queue assignment)
        self.route_stream = ops.concurrent_stream(dev) if dev.type == "cuda" else None
        self.rows_recv = torch.empty(B * F, sh.emb, dtype=torch.float32, device=dev)
        self.recv_keys = self.rows_send = self.grad_recv = None
        self.n_recv = 0
        # owner side: ONE table = the flat shard
        self.o_tab = ops._ptr_array([self.shard.data])
        self.o_vocab = (L.i64 * 1)(sh.R)
        # expansion: F "fields" that all read the received row block
        self.x_tab = ops._ptr_array([self.rows_recv] * F)
        self.x_vocab = (L.i64 * F)(*([B * F] * F))
        self.status = plan.status
        plan.keep.append(self)
        return [(E.PY, self._forward, (), dict(kernel="row_sharded_forward_exchange"))]

    def bwd_calls(self, plan):
        if self.out.grad is None:
            return []
        if not self.shard.needs_grad:
            raise L.MMLError("row-sharded tables need the shard's gradient accumulator (ParamStore.ensure_table_grads)")
        E._claim(self.shard)
        self.grad_send = torch.empty(self.B * self.F, self.sh.emb, dtype=torch.float32, device=plan.device)
        self.d_tab = ops._ptr_array([self.grad_send] * self.F)
        self.o_grad = ops._ptr_array([self.shard.grad])
        sr = self.sparse_rows
        if sr is not None:
            self.o_seen = ops._ptr_array(sr.seen)
            self.o_rb = (L.i64 * 2)(*sr.rowbase)
            self.o_extra = (self.o_seen, self.o_rb, sr.touched.data_ptr(), sr.count.data_ptr(), sr.touched.numel(),
                            sr.marks.data_ptr())
        else:
            self.o_extra = (None, None, None, None, 0, L.ptr(self.grad_marks))
        return [(E.PY, self._backward, (), dict(kernel="row_sharded_backward_exchange", tail=True))]

    # ---- run time ----------------------------------------------------------------------------------
    def _grow(self, n):
        if self.recv_keys is None or self.recv_keys.numel() < n:
            cap = max(int(n * 1.25) + 1024, 1)
            dev, Em = self.rows_recv.device, self.sh.emb
            self.recv_keys = torch.empty(cap, dtype=torch.int32, device=dev)
            self.rows_send = torch.empty(cap, Em, dtype=torch.float32, device=dev)
            self.grad_recv = torch.empty(cap, Em, dtype=torch.float32, device=dev)

    def _route(self, rs, X):
        """Index-only half of the exchange for the batch in X ([B, ldX] fp32-encoded indices) into route set rs, on the
        CURRENT stream: distinct rows -> owners -> send order + counts -> all_to_all(counts) -> pinned host copy (an
        event marks it), position of every lookup in the row block that will come back."""
        lib, sh, comm = L.load(), self.sh, self.comm
        s = ops._stream()
        B, F, W, Em = self.B, self.F, sh.world, sh.emb
        Xp, ldX = X.data_ptr(), ops._ld(X)
        st = self.status.data_ptr()
        cp = rs.counters.data_ptr()
        if self.dedup:
            rq = rs.req
            tl, tc, cap = rq.touched.data_ptr(), rq.count.data_ptr(), rq.touched.numel()
            L.check(lib.mml_index_unique(self.vocab, self.col, F, min(Em, 16), Xp, ldX, B, rs.req_seen, rs.req_rb, tl,
                                         tc, cap, rq.marks.data_ptr(), st, s), "mml_index_unique(requester)")
            L.check(lib.mml_route_list_count(tl, tc, cap, self.vocab, rs.req_rb, F, W, cp, s), "mml_route_list_count")
            rs.cnt_pair[0].copy_(rs.counters[:W])
            L.check(lib.mml_route_list_place(tl, tc, cap, self.vocab, rs.req_rb, self.keybase, F, W, cp,
                                             rs.send_keys.data_ptr(), rs.slot_of.data_ptr(), s),
                    "mml_route_list_place")
            # position of every lookup's row in the returned block, then the requester's bitmaps are reset
            L.check(lib.mml_lookup_slots(Xp, ldX, self.col, self.vocab, rs.req_rb, F, B, rs.slot_of.data_ptr(),
                                         rs.pos.data_ptr(), st, s), "mml_lookup_slots")
            L.check(lib.mml_rows_clear(rq.touched.data_ptr(), rq.count.data_ptr(), rq.touched.numel(), rs.req_rb,
                                       rs.req_seen, F, s), "mml_rows_clear")
            L.check(lib.mml_counter_update(rq.count.data_ptr(), 0, 1, s), "mml_counter_update")
        else:
            L.check(lib.mml_route_count(Xp, ldX, None, 0, self.col, self.vocab, F, B, W, cp, st, s), "mml_route_count")
            rs.cnt_pair[0].copy_(rs.counters[:W])  # (mml_route_place moves the cursors, not the counts)
            L.check(lib.mml_route_place(Xp, ldX, None, 0, self.col, self.vocab, self.keybase, F, B, W, cp,
                                        rs.send_keys.data_ptr(), rs.pos.data_ptr(), st, s), "mml_route_place")
        comm.all_to_all_single(rs.cnt_pair[1], rs.cnt_pair[0])
        rs.host.copy_(rs.cnt_pair, non_blocking=True)
        rs.ready.record(torch.cuda.current_stream())
        rs.splits = None

    def prefetch_route(self, X_next):
        """Route the NEXT step's batch now, on the side stream, beside whatever the current step still has queued
        (collective: every rank calls it at the same point of its program).  The next forward exchange then finds
        its counts on the host already.  X_next: a buffer the caller keeps unchanged until that step has started
        (trainer.TrainStep.prefetch owns one), holding the batch that step will find in plan.X."""
        if self.route_stream is None:
            return
        rs = self.routes[1] if self.cur is self.routes[0] else self.routes[0]
        with torch.cuda.stream(self.route_stream):  # (TrainStep.prefetch already runs on it: X_next is written there)
            self._route(rs, X_next)
        self.staged = rs

    def _forward(self):
        lib, sh, comm = L.load(), self.sh, self.comm
        s = ops._stream()
        B, F, W, Em = self.B, self.F, sh.world, sh.emb
        st = self.status.data_ptr()
        if self.staged is not None:   # routed while the previous step ran: the counts are (about to be) on the host
            rs, self.staged = self.staged, None
            torch.cuda.current_stream().wait_event(rs.ready)  # (the kernels below read rs.send_keys / rs.pos)
        else:                         # first step / no prefetch: route here; the host read below waits for the device
            rs = self.cur
            self._route(rs, self.X)
        self.cur = rs
        rs.ready.synchronize()
        pair = rs.host
        self.send_splits, self.recv_splits = pair[0].tolist(), pair[1].tolist()
        n = self.n_recv = int(sum(self.recv_splits))
        u = self.n_send = int(sum(self.send_splits))  # == B * F without de-duplication
        self.stats["n_recv"] += n
        self.stats["n_sent"] += u
        self.stats["lookups"] += B * F
        self.stats["steps"] += 1
        self._grow(n)
        comm.all_to_all_single(self.recv_keys[:n], rs.send_keys[:u], self.recv_splits, self.send_splits)
        if self.lazy_launch is not None:  # bring exactly the rows about to be read up to date (lazy-exact Adam)
            self.lazy_launch(self.recv_keys.data_ptr(), n, s)
        if n:
            L.check(lib.mml_gather_fwd_idx32(self.o_tab, self.o_vocab, 1, Em, self.recv_keys.data_ptr(), 1, None, 0, 0,
                                             n, self.rows_send.data_ptr(), Em, st, s), "mml_gather_fwd_idx32(owner)")
        comm.all_to_all_single(self.rows_recv.view(-1)[:u * Em], self.rows_send.view(-1)[:n * Em],
                               [c * Em for c in self.send_splits], [c * Em for c in self.recv_splits])
        ldX = ops._ld(self.X)
        dense = self.X[:, self.dense_col0:].data_ptr() if self.nd else None
        L.check(lib.mml_gather_fwd_idx32(self.x_tab, self.x_vocab, F, Em, rs.pos.data_ptr(), F, dense, ldX, self.nd, B,
                                         self.out.buf.data_ptr(), ops._ld(self.out.buf), st, s),
                "mml_gather_fwd_idx32(expand)")

    def _backward(self):
        lib, sh, comm = L.load(), self.sh, self.comm
        s = ops._stream()
        B, F, Em, n, u = self.B, self.F, sh.emb, self.n_recv, self.n_send
        g = self.out.grad
        if self.dedup:  # gradients of duplicate lookups are summed HERE, one row per distinct key travels
            self.grad_send[:u].zero_()
            L.check(lib.mml_scatter_bwd_idx32(self.d_tab, self.x_vocab, F, Em, self.cur.pos.data_ptr(), F, B, g.data_ptr(),
                                              ops._ld(g), None, None, None, None, 0, None, self.status.data_ptr(), s),
                    "mml_scatter_bwd_idx32(requester)")
        else:
            L.check(lib.mml_rows_permute(g.data_ptr(), ops._ld(g), self.cur.pos.data_ptr(), F, Em, B,
                                         self.grad_send.data_ptr(), s), "mml_rows_permute")
        comm.all_to_all_single(self.grad_recv.view(-1)[:n * Em], self.grad_send.view(-1)[:u * Em],
                               [c * Em for c in self.recv_splits], [c * Em for c in self.send_splits])
        # (n == 0 included: the entry point then resets the touched-row list, so the previous step's rows are not updated
        # a second time)
        L.check(lib.mml_scatter_bwd_idx32(self.o_grad, self.o_vocab, 1, Em, self.recv_keys.data_ptr(), 1, n,
                                          self.grad_recv.data_ptr(), Em, *self.o_extra, self.status.data_ptr(), s),
                "mml_scatter_bwd_idx32(owner)")


class ReplicatedGatherOp(E.GatherOp):
    """Tables replicated: local gather; backward = all-gather of (indices, row gradients), then every rank scatters
    the WHOLE global batch into its own copy (identical updates, fixed sizes, no host read)."""

    def __init__(self, par, *a, **k):
        super().__init__(*a, **k)
        self.par, self.comm = par, par.comm

    def fwd_calls(self, plan):
        W = self.comm.world
        self.X_all = plan.empty(W * plan.B, self.X.shape[1])
        return super().fwd_calls(plan)

    def index_view(self, plan):
        return self.X_all, self.comm.world * plan.B

    def pre_index_calls(self, plan):
        """Indices of the global batch (needed before the lazy-exact catch-up and by the scatter)."""
        return [(E.PY, self.comm.all_gather_into_tensor, (self.X_all, self.X), dict(kernel="all_gather(indices)"))]

    def bwd_calls(self, plan):
        if self.out.grad is None or not any(t.needs_grad for t in self.tables):
            return []
        W = self.comm.world
        g = self.out.grad
        ld = g.stride(0)
        g_full = g.as_strided((plan.B, ld), (ld, 1), g.storage_offset())
        self.g_all = plan.empty(W * plan.B, ld)
        local_X, local_g = self.X, self.out.grad
        calls = []
        if not getattr(self, "x_in_pre", False):
            calls += self.pre_index_calls(plan)
        calls.append((E.PY, self.comm.all_gather_into_tensor, (self.g_all, g_full), dict(kernel="all_gather(row grads)")))
        # the inherited scatter over the gathered batch
        self.X, B0 = self.X_all, plan.B
        self.out.grad = self.g_all[:, :g.shape[1]]
        plan.B = W * B0
        try:
            sc = super().bwd_calls(plan)
        finally:
            self.X, self.out.grad, plan.B = local_X, local_g, B0
        for c in calls:
            c[3]["tail"] = True
        return calls + sc


# ======================================================================================================
# table-wise sharding (round 1)
# ======================================================================================================
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
    """K1/K2 with whole tables on owner ranks: index / row / gradient exchange around the local fused gather and
    scatter kernels (static split sizes)."""

    def __init__(self, par, tables, X, cols, dense_col0, nd, out, sparse_rows=None):
        self.sh, self.comm = par.sharding, par.comm
        self.tables, self.X, self.cols, self.dense_col0, self.nd, self.out = tables, X, cols, dense_col0, nd, out
        self.sparse_rows = sparse_rows

    def outputs(self):
        return [self.out]

    def _a2a(self, out, inp, out_splits, in_splits):
        self.comm.all_to_all_single(out, inp, out_splits, in_splits)

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
                extra = (seen, rb, sr.touched.data_ptr(), sr.count.data_ptr(), sr.touched.numel(), sr.marks.data_ptr())
            else:
                extra = (None, None, None, None, 0, None)
            calls.append((lib.mml_scatter_bwd, (gt, vocab, col, nfm, Em, self.recv_idx.data_ptr(), nfm, W * B,
                                                self.grad_recv.data_ptr(), nfm * Em) + extra +
                          (plan.status.data_ptr(),),
                          dict(kernel=E.scatter_symbol(Em), bytes=float(W * B) * nfm * (4 + 12 * Em), tail=True)))
        return calls


# ======================================================================================================
# model-level switch
# ======================================================================================================
MODES = ("row_sharded", "replicated", "table_wise")


class ParallelState:
    """What a model needs to know about its process group.  Never deep-copied: fit()'s best-model snapshot is a plain
    single-GPU model holding the synchronised tables."""

    def __init__(self, comm, mode, sharding):
        self.comm, self.mode, self.sharding = comm, mode, sharding
        self.shard = None      # row_sharded: this rank's flat [R, E] rows
        self.dirty = False     # the full embedding_dict tables are behind the trained state

    def __deepcopy__(self, memo):
        return None

    @property
    def world(self):
        return self.comm.world

    @property
    def rank(self):
        return self.comm.rank


def make_allreduce(par):
    """Sum of the flat MLP-gradient arena over ranks (loss is reduction='sum', model/basemodel.py:295)."""
    return par.comm.all_reduce


def _full_tables(model):
    return [model.embedding_dict[f.embedding_name].weight.data for f in model._sparse_cols()]


def shard_model(model, dist, batch_per_rank=4096, group=None, mode="row_sharded", dedup=True):
    """Switch a model to multi-GPU execution on the current process group (collective: every rank calls it on an
    identically initialised model).  Returns the ParallelState (also stored as model._parallel)."""
    if mode not in MODES:
        raise ValueError(f"mode must be one of {MODES}")
    # The N-rank contract (an N-rank step on a batch split N ways == a 1-rank step on the whole batch) holds for steps
    # that are per-sample sums.  Two model features are not: ESCM's IPW loss normalises by batch-wide sums (N = sum y0,
    # L1, S: mml_escm_combine sees the local shard only) and BatchNorm / DomainBatchNorm use batch statistics (per-rank
    # statistics would silently train a different model, and the running statistics would diverge across ranks).
    if type(model).__name__ == "ESCM":
        raise NotImplementedError("shard_model: ESCM's IPW loss is normalised over the whole batch (not a per-sample "
                                  "sum): the multi-GPU step would optimise a different objective")
    bn = (bool((getattr(model, "config", None) or {}).get("model_config", {}).get("dnn_use_bn", False)) or
          any(type(m).__name__ in ("BatchNorm1d", "DomainBatchNorm") for m in model.modules()))
    if bn:
        raise NotImplementedError("shard_model: BatchNorm / DomainBatchNorm use batch statistics (per-rank statistics "
                                  "differ from the whole-batch ones): not supported on the multi-GPU path")
    comm = Comm(dist, group)
    vocab = [f.vocabulary_size for f in model._sparse_cols()]
    Em = model.embedding_size
    if mode == "row_sharded":
        sharding = RowSharding(vocab, Em, comm.world, comm.rank)
    elif mode == "table_wise":
        sharding = FieldSharding(vocab, Em, comm.world, comm.rank, batch_per_rank)
    else:
        sharding = None
    par = ParallelState(comm, mode, sharding)
    par.dedup = bool(dedup)  # row_sharded: exchange the batch's distinct rows instead of its lookups
    if mode == "row_sharded":
        tabs = _full_tables(model)
        if not tabs[0].is_cuda:
            raise L.MMLError("shard_model: move the model to its MI355X first (no CPU path)")
        par.shard = sharding.tables_to_shard(tabs)
    model._parallel = par
    model._caches = {"store": None, "plans": {}, "steps": {}}
    if getattr(model, "_optimizer", None) is not None:
        model._optimizer = None  # optimizer state follows the parameter store (moments restart with the new layout)
    return par


def _scatter_mode(model):
    return getattr(model, "scatter_mode", None) or (getattr(model, "config", None) or {}).get("model_config", {}).get(
        "scatter_mode", "atomic")


def sync_tables(model):
    """Make every rank's full embedding_dict tables equal to the trained state (collective).
    row_sharded: all-gather of the flat shards; table_wise: broadcast from each owner; replicated: rank 0's copy."""
    par = getattr(model, "_parallel", None)
    if par is None:
        return
    comm = par.comm
    tabs = _full_tables(model)
    if par.mode == "row_sharded":
        sh = par.sharding
        allsh = torch.empty(comm.world, sh.R, sh.emb, dtype=torch.float32, device=par.shard.device)
        comm.all_gather_into_tensor(allsh, par.shard)
        for k in range(comm.world):
            sh.shard_to_tables(allsh[k], tabs, rank=k)
    elif par.mode == "table_wise":
        for f, w in enumerate(tabs):
            comm.broadcast(w, src=par.sharding.owner[f])
    elif _scatter_mode(model) == "deterministic":
        # replicated tables + deterministic scatter: every rank applied bitwise the same update to its copy (identical
        # all-gathered inputs, order-independent integer sums, the all-reduced MLP gradients): nothing to reconcile
        pass
    else:
        for w in tabs:
            comm.broadcast(w, src=0)
    par.dirty = False
