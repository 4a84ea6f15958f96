"""One fused training step = forward launches + head/BCE launch + backward launches + optimizer launches, recorded
once per batch size and replayed (as HIP graphs when enabled).  Counterpart of the loop body of BaseModel.fit in the
reference (model/basemodel.py:261-313) minus logging.

Default (round 5): ONE HIP stream, the whole step one HIP graph (resolve_overlap below holds the A/B).  With two streams
(overlap=True / MMLREC_STREAMS=2), after the backward chain has produced every dL/d(pre-activation), the step forks --
  main stream : table scatter -> table optimizer            (HBM / atomics bound)
  side stream : all weight-gradient GEMMs -> [all-reduce] -> MLP optimizer   (MFMA bound)
-- and joins at the end: the table stream of the reference-exact dense Adam beside the wgrad GEMMs.

Multi-GPU steps (parallel.py) contain Python-issued entries (collectives, the row-sharded exchange with its run-time
sizes).  Those run eagerly; the runs of C-ABI calls between them -- static shapes, static pointers -- are still captured
and replayed as HIP graphs (`Segments`).
"""
import os

import torch

from . import engine as E
from .ops import concurrent_stream as ops_concurrent_stream
from . import profiling


class Segments:
    """A call list cut at its Python-issued entries: [graphable run, PY entry, graphable run, ...]."""

    def __init__(self, calls, use_graph, min_calls=2):
        self.parts = []  # ("c", [calls]) | ("py", call)
        run = []
        for c in calls:
            if c[0] is E.PY:
                if run:
                    self.parts.append(["c", run, None])
                    run = []
                self.parts.append(["py", c, None])
            else:
                run.append(c)
        if run:
            self.parts.append(["c", run, None])
        self.use_graph = bool(use_graph)
        self.min_calls = min_calls
        self.captured = False

    def capture(self):
        """Record every long-enough run as its own single-stream HIP graph (nothing executes)."""
        if not self.use_graph or self.captured:
            return
        for p in self.parts:
            if p[0] == "c" and len(p[1]) >= self.min_calls:
                g = torch.cuda.CUDAGraph()
                # thread_local: only THIS thread's calls are checked during capture.  In the default global mode an
                # event query of RCCL's watchdog thread (it polls while a process group exists) is an illegal call
                # during capture: it invalidates the capture or aborts the process at teardown (seen as a sporadic
                # "Fatal Python error: Aborted" in destroy_process_group).
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    E.Plan._run(p[1])
                p[2] = g
        self.captured = True

    def run(self, skip_py=None):
        for kind, item, graph in self.parts:
            if kind == "py":
                if skip_py is not None and skip_py(item):
                    continue
                item[1](*item[2])
            elif graph is not None:
                if profiling.enabled:
                    with profiling.range("hip_graph[%d calls: %s ...]" % (len(item), (item[0][2] if len(item[0]) > 2 and
                                         isinstance(item[0][2], dict) else {}).get("kernel", getattr(item[0][0], "__name__", "inline")))):
                        graph.replay()
                else:
                    graph.replay()
            else:
                E.Plan._run(item)

    @property
    def n_graphs(self):
        return sum(1 for p in self.parts if p[2] is not None)


def fork_conflicts(side_calls, mid_calls, shared_scratch=()):
    """Raw device pointers that make a fork of `side_calls` beside `mid_calls` unsafe, as far as the call lists show it:
    (a) a shared scratch buffer (ops.workspace: every non-deferred row kernel writes its partial sums there) named by
    EITHER branch -- the other branch, or the chain that follows, may overwrite it -- and (b) a pointer argument both
    branches carry (each branch must own what it names; descriptors hold further pointers the lists do not show: the
    operands of the weight-gradient GEMMs are written by the chain BEFORE the fork and only read after it).  Returns the
    offending pointers (empty = no conflict seen); TrainStep refuses to fork on any (tests/test_plan_passes_cpu.py)."""
    def ptrs(calls):
        out = set()
        for c in calls:
            if c[0] is E.PY or c[0] is E.INLINE:
                continue
            for a in c[1]:
                # (device addresses on this platform are 47-bit values far above 2^32; sizes, pitches and counts -- a
                # batch of 65 536 rows is 0x10000 in BOTH lists -- stay below)
                if isinstance(a, int) and not isinstance(a, bool) and a >= (1 << 32):
                    out.add(a)
        return out
    a, b = ptrs(side_calls), ptrs(mid_calls)
    scratch = {int(x) for x in shared_scratch if x}
    return sorted(((a | b) & scratch) | (a & b))


class InnerFork:
    """A fork / join INSIDE one call list (one HIP graph): `fork` sends `calls` to a second stream behind everything
    issued so far, `join` makes the current stream wait for them.  Knob MMLREC_INNER_FORK (TrainStep): no graph seam,
    unlike the two-stream schedule of overlap=True -- but a multi-branch graph (see TrainStep.run's note on
    hip::Graph::UpdateStreams and the soak of round 6)."""

    def __init__(self, device, calls):
        self.side = torch.cuda.Stream(device=device)
        self.calls = calls
        self.ev_fork, self.ev_join = torch.cuda.Event(), torch.cuda.Event()

    def fork(self):
        self.ev_fork.record(torch.cuda.current_stream())
        self.side.wait_event(self.ev_fork)
        with torch.cuda.stream(self.side):
            E.Plan._run(self.calls)
            self.ev_join.record(self.side)

    def join(self):
        torch.cuda.current_stream().wait_event(self.ev_join)


def _quiesce_collective_watchdog():
    """Called after a device-wide synchronize and before a HIP-graph capture.  While an RCCL process group exists, its
    watchdog thread polls the end events of the collectives it has not seen complete yet (every ~100 ms).  A poll that
    lands inside a capture is an event query during stream capture: with capture_error_mode="thread_local" (Segments)
    it no longer invalidates the capture, but it was still seen -- 1 run in 24 of the suite's graph tests, round 6 -- as an
    exception in the watchdog that surfaces as "Fatal Python error: Aborted" in the NEXT destroy_process_group.  Every
    collective issued so far HAS completed (the synchronize above); giving the watchdog two of its periods to notice
    leaves it nothing to query while the capture runs.  Once per TrainStep (captures happen on the second call of run())."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            import time
            time.sleep(0.25)
    except Exception:  # (never cost a step for this)
        pass


def resolve_overlap(overlap):
    """Stream schedule of a fused step: None = the default.  Round 5: ONE stream -- the whole step as one HIP graph.
    Same-box interleaved A/B of bench.py (three pairs each, tools/lab/ab_streams.sh, profiles/r05_ab_streams.txt): the
    forked tail (table scatter + table optimizer | weight-gradient GEMMs + MLP optimizer on a second stream) returned
    +0.5 % on AE-30 at B = 65 536 (1.7158 against 1.7265 ms) and LOST 3.4 % at B = 4 096 (0.693 against 0.670), 1.8 % on
    PepNet / Amazon-8 (2.238 against 2.200), level on KuaiRec-32: everything it co-schedules is HBM-bound together, and
    every graph seam costs ~16 us of idle stream.  Two streams stay available: overlap=True, or MMLREC_STREAMS=2."""
    if overlap is None:
        return os.environ.get("MMLREC_STREAMS", "1") == "2"
    return bool(overlap)


class TrainStep:
    def __init__(self, model, B, use_graph=True, allreduce=None, overlap=None, split_dense=True):
        overlap = resolve_overlap(overlap)
        self.model = model
        self.store = model._store()
        self.opt = model.optimizer()
        self.want_split = bool(split_dense)
        rows = None
        par = getattr(model, "_parallel", None)
        self.par = par
        if allreduce is None and par is not None:
            from .parallel import make_allreduce
            allreduce = make_allreduce(par)
        if (self.opt.table_update == "lazy_exact" and getattr(self.opt, "auto", False) and par is not None and
                par.mode == "table_wise" and self.opt.steps_done == 0 and self.opt.last is None):
            self.opt.table_update = "dense_exact"  # ('auto' picked lazy_exact before the tables were sharded table-wise)
        lazy = self.opt.table_update == "lazy_exact"
        if lazy and par is not None and par.mode == "table_wise":
            raise NotImplementedError("lazy_exact table updates on the table-wise sharded path (use row_sharded)")
        # Split dense table update (engine.Optimizer.can_split_dense): the reference-exact dense optimizer as
        #   early : every row the batch does NOT touch (zero gradient), streamed beside the forward / backward,
        #   tables: the touched rows, with their gradients, after the scatter (mml_opt_step_rows)
        # -- the same arithmetic on every row as one dense launch.  Same-box A/B on AE-30: 0.70-0.76 against 0.77 ms at
        # B = 4 096, 1.95-2.01 against 1.92 ms at B = 65 536 (nothing co-resides with the GEMMs: the early pass only
        # adds its bookkeeping).  Since the single launch learnt to skip the gradient read of the rows the scatter did
        # not mark (grad_marks below: 24 instead of 28 bytes per Adam parameter, no extra launch or stream) the two
        # are level at B = 4 096 too (0.767 / 0.769 ms on one box), so the split form only runs on request
        # (split_dense="force").
        split = (split_dense == "force" and self.opt.table_update == "dense_exact" and
                 (par is None or par.mode == "replicated") and
                 not self.opt._table_reg(self.opt._reg_map()) and model.embedding_size <= 16 and
                 model.embedding_size % 4 == 0)
        if self.opt.table_update in ("sparse_rows", "lazy_exact") or split:
            if par is None:
                rows = self.store.ensure_rows(int(B) * max(len(model._sparse_cols()), 1))
            elif par.mode == "row_sharded":  # one flat table: at most every local row is touched
                rows = self.store.ensure_rows(par.sharding.R)
            elif par.mode == "replicated":
                rows = self.store.ensure_rows(par.world * int(B) * max(len(model._sparse_cols()), 1))
            else:  # this rank serves world*B lookups for each of ITS fields
                sp = model._sparse_cols()
                names = [f"embedding_dict.{sp[f].embedding_name}.weight" for f in par.sharding.mine]
                rows = self.store.ensure_rows(par.world * int(B) * max(len(names), 1), names)
        # lazy_exact lists the batch's rows in a pre-pass (before the gather), so the scatter only accumulates
        # single GPU: the gather marks the rows it reads (no separate pass over X); replicated tables need the rows of
        # the GLOBAL batch, listed by the index pre-pass
        # Single-launch dense update: the scatter marks the rows it adds to, the optimizer reads the gradient of those
        # rows only (mml_opt_tensor.grad_marks): 24 instead of 28 bytes per Adam parameter, no extra launch.
        marked = (self.opt.table_update == "dense_exact" and not split and
                  (par is None or par.mode in ("row_sharded", "replicated")) and
                  os.environ.get("MMLREC_GRAD_MARKS", "1") != "0" and not os.environ.get("MMLREC_SCATTER_OLD"))
        from . import engine as _E
        # one stream (and no split table update, whose early pass forks anyway): the reductions of the head / gate kernels'
        # partial sums are deferred behind the backward chain and merged into ONE launch (Plan.merge_row_reduces)
        one_list = not overlap and not split and os.environ.get("MMLREC_MERGE_REDUCES", "1") != "0"
        _E._DEFER = one_list
        try:
            self.plan = model._record(B, True, False, self.store, sparse_rows=None if (lazy or split) else rows,
                                      lazy=lazy or split, mark_rows=rows if (split and par is None) else None,
                                      grad_marks=marked)
        finally:
            _E._DEFER = None
        if one_list:
            # the top of the network -- last tower layer, heads + BCE, the towers' input gradient -- as one launch where
            # the recorded lists hold that pattern (csrc/tower_head.hip; before the reductions are merged: it brings its own)
            self.tower_head_fused = self.plan.fuse_tower_head()
            self.plan.merge_row_reduces()
        self.grad_marks = getattr(self.plan.ops[0], "grad_marks", None) is not None
        # every weight-gradient GEMM of the step in one launch: at small batches (a layer's launch does not fill the chip)
        # and whenever no table stream runs beside them (same-box A/B at B = 65 536: lazy_exact 1.677 -> 1.628 ms, but
        # dense_exact 1.94 -> 1.98: next to the dense table update the per-layer order shares the chip better)
        # Round 5, ONE stream (no table update beside the weight gradients): merged at every batch -- the per-layer
        # launches of a large batch do not all fill the chip either (AE-30's tower layers: 2 tiles x 64 slabs = 128
        # workgroups for 512 slots), and one reduction over 17 slabs replaces three over 25 / 64 / 64.
        self.wgrad_merged = (int(B) <= 8192 or self.opt.table_update != "dense_exact" or not overlap) and \
            os.environ.get("MMLREC_MERGE_WGRAD", "1") != "0" and self.plan.merge_wgrad()
        if not overlap and os.environ.get("MMLREC_MERGE_WGRAD", "1") != "0":
            self.plan.merge_wgrad16()  # (the bf16-storage path's launches: csrc/gemm16.hip)
        self.opt_split = self.opt.calls_split(self.plan, split_dense=split)
        self.split_dense = bool(self.opt_split["early"])
        self.opt_calls = (self.opt_split["pre"] + self.opt_split["early"] + self.opt_split["mlp"] +
                          self.opt_split["tables"])
        self.allreduce = allreduce  # callable(flat dense-gradient arena) or None
        self.use_graph = bool(use_graph)
        # Two streams pay when there is a long table stream to put beside the weight-gradient GEMMs.  The row-wise table
        # updates have none, and at small batches the fork / join (two more graph seams, ~16 us each) costs more than
        # the overlap returns -- same-box A/B, lazy_exact on AE-30: 0.374 forked vs 0.338 ms serial at B = 4 096, level
        # at 16 384, 1.655 vs 1.682 at 65 536 (dense_exact: forked wins at every batch).
        if overlap and self.opt.table_update != "dense_exact" and int(B) <= 8192 and par is None and allreduce is None:
            overlap = False
        self.overlap = bool(overlap)
        self.side = ops_concurrent_stream(self.store.device) if self.overlap else None
        # CU partition of the forked tail (lab knob MMLREC_CU_TAIL = n: table scatter + table optimizer on compute
        # units [0, n), weight-gradient GEMMs + MLP optimizer on [n, all))
        self.tail_stream = None
        n_tail = int(os.environ.get("MMLREC_CU_TAIL", "0"))
        if self.overlap and n_tail > 0 and par is None:
            from . import ops
            ncu = torch.cuda.get_device_properties(self.store.device).multi_processor_count
            self.tail_stream = ops.cu_range_stream(self.store.device, 0, n_tail)
            self.side = ops.cu_range_stream(self.store.device, n_tail, ncu)
            self.ev_tail = torch.cuda.Event()
        # (a high-priority stream, or more hardware queues (GPU_MAX_HW_QUEUES=8), for the early pass made a B = 4 096
        # step twice as slow: 0.82 -> 1.6 ms; the default priority it is)
        self.side2 = torch.cuda.Stream(device=self.store.device) if (self.overlap and self.split_dense) else None
        n_early = int(os.environ.get("MMLREC_CU_EARLY", "0"))
        if self.side2 is not None and n_early > 0:
            from . import ops
            self.side2 = ops.cu_range_stream(self.store.device, 0, n_early)
        # fork / join events live as long as the step
        self.ev_fork = torch.cuda.Event() if self.overlap else None
        self.ev_join = torch.cuda.Event() if self.overlap else None
        self.ev_pre = torch.cuda.Event() if self.side2 is not None else None
        self.ev_early = torch.cuda.Event() if self.side2 is not None else None
        p = self.plan
        ar = [(E.PY, self._allreduce, (), dict(kernel="all_reduce(mlp grads)"))] if allreduce is not None else []
        # the touched-row update clears the `seen` bits the early pass is still reading: it waits for that pass
        wait = ([(E.PY, self._wait_early, (), dict(kernel="wait(early table pass)"))] if self.side2 is not None else [])
        # (without an early pass the counter bump / lazy pre-pass simply lead the front graph; with one, `pre` is
        # everything the early pass waits for: the counter, and the row list -- from the index pre-pass, or from the
        # marking gather + compaction that open the forward)
        n_lead = 2 if (self.split_dense and getattr(p.ops[0], "mark_rows", None) is not None) else 0
        n_lead += getattr(p, "n_pre", 0) if self.split_dense else 0  # (the magnitude reset / weight pass open `fwd`)
        self.pre = Segments((self.opt_split["pre"] + p.fwd[:n_lead]) if self.split_dense else [], self.use_graph)
        self.early = Segments(self.opt_split["early"], self.use_graph, min_calls=1)
        # Early fork of the side stream: when no long table stream waits in the tail to hide the weight-gradient GEMMs
        # behind (row-wise table updates, small tables), the GEMMs whose operands the backward chain has already produced
        # start beside the REST of the chain instead -- worth it where that rest holds HBM-bound launches (PepNet's gate
        # products, the gate / head row kernels) that leave the matrix pipe idle.  One more graph seam on each stream.
        k = self._early_fork(p, int(B)) if self.overlap and not self.split_dense else 0
        self.front = Segments(([] if self.split_dense else self.opt_split["pre"]) + p.fwd[n_lead:] + p.head_train +
                              (p.bwd[:k] if k else p.bwd), self.use_graph)
        self.front_b = Segments(p.bwd[k:], self.use_graph, min_calls=1) if k else None
        side_a = [c for c in p.bwd_side if c[-1].get("ready", 1 << 30) <= k] if k else []
        self.side_a = Segments(side_a, self.use_graph, min_calls=1) if k else None
        self.ev_fork_a = torch.cuda.Event() if k else None
        self.early_fork = k
        head_side = list(getattr(p, "head_side", []))
        self.sideq = Segments(head_side + p.bwd_side[len(side_a):] + ar + self.opt_split["mlp"], self.use_graph)
        self.tail = Segments(p.bwd_tail + wait + self.opt_split["tables"], self.use_graph)
        # one stream: the whole step is ONE call list (one HIP graph when it holds no Python-issued entry) -- every graph
        # seam is ~16 us of idle stream, a tenth of a small-batch step
        self.whole = None
        if not self.overlap and not self.split_dense:
            # The weight gradients on a second stream INSIDE the step's one graph (a fork / join of graph nodes: no seam):
            # beside the table scatter and the table optimizer.  Same-box interleaved pairs (tools/lab/ab_env.sh
            # MMLREC_INNER_FORK=0 / 2, B = 65 536): AE-30 1.4773 / 1.4818 / 1.4860 / 1.4952 / 1.4772 -> 1.4635 / 1.4683 /
            # 1.4851 / 1.4623 / 1.4731 ms, AE-30d 1.613 -> 1.582, PLE 1.535 -> 1.509, PepNet 2.028 -> 2.006; at B = 4 096
            # it loses (0.670 -> 0.679 ms), so large batches only.  MMLREC_INNER_FORK: 0 off, 1 joined in front of the table
            # optimizer (beside the scatter only: level), 2 joined behind it (the default form).
            # Round 6: the fork makes the step's graph a multi-branch graph, the kind whose launch segfaulted sporadically
            # in this runtime (hip::Graph::UpdateStreams, see run() below: dependent on how many streams the process
            # created).  Soaked (tools/lab/fork_soak.sh, profiles/r06_fork_soak.txt): 0 crashes in 32 fresh processes with
            # the fork forced on for EVERY batch size (22 runs of the suite's graph tests -- every model of the zoo, its
            # own streams, the original repro's shape -- and 10 of bench.py), so it stays the default for large batches;
            # it never applies to a step that holds a collective (Python-issued entries between fork and join), i.e. to no
            # multi-GPU step.  MMLREC_INNER_FORK=0 turns it off.
            env_fork = os.environ.get("MMLREC_INNER_FORK")
            inner = int(env_fork) if env_fork is not None else (2 if int(B) >= 16384 else 0)
            side_calls = head_side + p.bwd_side
            mid = (p.bwd_tail + self.opt_split["tables"])
            self.fork_refused = []
            mlp_side = os.environ.get("MMLREC_FORK_MLP", "0") == "1" and not ar
            if mlp_side:  # the MLP optimizer behind the weight gradients on their branch (it needs nothing of the other one)
                side_calls = side_calls + self.opt_split["mlp"]
            if (inner in (1, 2, 3) and side_calls and not any(c[0] is E.PY for c in side_calls) and
                    not any(c[0] is E.PY for c in mid)):  # (fork and join must land in ONE graph)
                from . import ops as _ops
                bad = fork_conflicts(side_calls, mid, [w.data_ptr() for w in _ops._workspaces.values()])
                self.fork_refused = bad  # (a step whose branches share a buffer simply runs unforked)
            if (inner in (1, 2, 3) and side_calls and not any(c[0] is E.PY for c in side_calls) and
                    not any(c[0] is E.PY for c in mid) and not self.fork_refused):
                self.inner_fork = InnerFork(self.store.device, side_calls)
                fk = [(E.INLINE, self.inner_fork.fork, ())]
                jn = [(E.INLINE, self.inner_fork.join, ())]
                mid = (fk + p.bwd_tail + jn + self.opt_split["tables"]) if inner == 1 else \
                    (fk + p.bwd_tail + self.opt_split["tables"] + jn) if inner == 2 else \
                    (p.bwd_tail + fk + self.opt_split["tables"] + jn)   # (3: forked behind the scatter)
                self.whole = Segments(self.opt_split["pre"] + p.fwd + p.head_train + p.bwd + mid + ar +
                                      ([] if mlp_side else self.opt_split["mlp"]), self.use_graph)
            else:
                self.whole = Segments(self.opt_split["pre"] + p.fwd + p.head_train + p.bwd + p.bwd_tail +
                                      self.opt_split["tables"] + head_side + p.bwd_side + ar + self.opt_split["mlp"],
                                      self.use_graph)
        self.calls = 0
        self._nX = self._ny = None  # staging buffers of a prefetched batch
        self._has_next = False

    @staticmethod
    def _cost(c):
        meta = c[-1] if isinstance(c[-1], dict) else {}
        return 4e-6 + meta.get("flops", 0.0) / 5e14 + meta.get("bytes", 0.0) / 4e12

    def _early_fork(self, p, B):
        """Index into plan.bwd at which the side stream forks early (0 = only at the end of the chain).
        MMLREC_EARLY_WGRAD: unset or 0 = off (the default: measured a loss or level on every workload),
        n > 0 = that index, "auto" = where the side calls that are ready by then take about as long as the rest of the
        chain -- but only when the tail has no dense table stream of the same length to put them beside."""
        env = os.environ.get("MMLREC_EARLY_WGRAD", "0")
        ready = [c[-1].get("ready") for c in p.bwd_side]
        if env == "0" or not p.bwd_side or any(r is None for r in ready) or any(c[0] is E.PY for c in p.bwd):
            return 0
        if ready != sorted(ready):  # (program order: a later side call is never ready before an earlier one)
            return 0
        if env != "auto":
            return max(0, min(int(env), len(p.bwd) - 1))
        side_t = sum(self._cost(c) for c in p.bwd_side)
        table_t = sum(self._cost(c) for c in self.opt_split["tables"])
        if os.environ.get("MMLREC_EARLY_WGRAD_DEBUG"):
            import sys
            print("early fork: side %.0f us, tables %.0f us, chain %.0f us, ready %s of %d" % (
                side_t * 1e6, table_t * 1e6, sum(self._cost(c) for c in p.bwd) * 1e6, ready, len(p.bwd)), file=sys.stderr)
        if B < 16384 or table_t > 0.5 * side_t:
            return 0
        best, best_k = 0.0, 0
        for k in sorted(set(ready)):
            if k <= 0 or k >= len(p.bwd):
                continue
            a = sum(self._cost(c) for c, r in zip(p.bwd_side, ready) if r <= k)
            rest = sum(self._cost(c) for c in p.bwd[k:])
            if min(a, rest) > best:
                best, best_k = min(a, rest), k
        return best_k if best > 40e-6 else 0

    def prefetch(self, X=None, y=None, fence=None, fill=None):
        """Hand over the NEXT step's batch while this one is still in flight: it is copied into staging buffers (the
        next run() moves it into plan.X / plan.y itself -- do not copy it there as well) and, on row-sharded tables, its
        index-only routing work (distinct rows, owners, per-owner counts incl. their exchange and the host read of the
        split sizes) starts at once on a side stream.  Collective on the multi-GPU paths: every rank calls it at the
        same point.  Optional: a step whose batch was simply copied into plan.X / plan.y routes it itself.
        fill(X_buf, y_buf): instead of X / y, a callable that writes the staging buffers (it runs on the side stream;
        whatever it reads must be complete and must stay alive until the next run())."""
        p = self.plan
        if self._nX is None:
            self._nX, self._ny = torch.empty_like(p.X), torch.empty_like(p.y)
            self._stage_stream = ops_concurrent_stream(p.device)
            self._ev_staged, self._ev_consumed = torch.cuda.Event(), torch.cuda.Event()
            self._ev_consumed.record(torch.cuda.current_stream())
        op = p.ops[0] if p.ops else None
        side = getattr(op, "route_stream", None) or self._stage_stream
        # X / y must be complete when this is called (resident batches, or `fence` = an event recorded after the work
        # that produces them): the copies are NOT queued behind the step that is in flight on the caller's stream --
        # that is the point -- only behind the previous staged batch having been moved into the plan
        if fence is not None:
            side.wait_event(fence)
        side.wait_event(self._ev_consumed)
        with torch.cuda.stream(side):
            if fill is not None:   # the caller writes the staging buffers itself (e.g. index_select from a resident set)
                fill(self._nX, self._ny)
            else:
                self._nX.copy_(X, non_blocking=True)
                self._ny.copy_(y, non_blocking=True)
            if hasattr(op, "prefetch_route"):
                op.prefetch_route(self._nX)
            self._ev_staged.record(side)
        self._has_next = True

    def load(self, X, y):
        """plan.X / plan.y <- the batch (device tensors of the plan's shapes) in ONE launch on the current stream: two
        tensor copies are two launches in front of every step (6 us each at the head of a 1.8 ms step)."""
        from . import _lib as L
        p = self.plan
        if (X.shape != p.X.shape or y.shape != p.y.shape or X.dtype != torch.float32 or y.dtype != torch.float32 or
                X.stride(1) != 1 or y.stride(1) != 1 or X.device != p.X.device or y.device != p.y.device):
            p.X.copy_(X)
            p.y.copy_(y)
            return
        arr = (L.Copy2dDesc * 2)()
        for d, (src, dst) in zip(arr, ((X, p.X), (y, p.y))):
            d.src, d.lds, d.dst, d.ldd = src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0)
            d.rows, d.cols, d.accumulate = src.shape[0], src.shape[1], 0
        L.check(L.load().mml_copy2d_batch(arr, 2, torch.cuda.current_stream().cuda_stream), "mml_copy2d_batch")

    def drop_prefetch(self):
        self._has_next = False
        op = self.plan.ops[0] if self.plan.ops else None
        if getattr(op, "staged", None) is not None:
            op.staged = None

    def _allreduce(self):
        self.allreduce(self.store.arena)

    def _wait_early(self):
        torch.cuda.current_stream().wait_event(self.ev_early)

    def _forked(self, side, tail):
        main = torch.cuda.current_stream()
        self.ev_fork.record(main)
        self.side.wait_event(self.ev_fork)
        with torch.cuda.stream(self.side):
            side()
            self.ev_join.record(self.side)
        if self.tail_stream is not None:
            self.tail_stream.wait_event(self.ev_fork)
            with torch.cuda.stream(self.tail_stream):
                tail()
                self.ev_tail.record(self.tail_stream)
            main.wait_event(self.ev_tail)
        else:
            tail()
        main.wait_event(self.ev_join)

    def run(self):
        """plan.X / plan.y must hold the batch. After the call plan.prob / plan.loss hold this step's outputs.
        The first call runs eagerly (HIP graph capture needs warmed-up state and does not execute what it records);
        the second call captures, then every call replays.

        With two streams the step is THREE single-stream graph sequences (front, side, tail) forked and joined with
        events at replay time, not one graph with two branches: hipGraphLaunch of a multi-branch graph walks past the
        end of the exec's parallel-stream vector when one of those streams shares a hardware queue with the launch
        stream (hip::Graph::UpdateStreams, ROCm 7.0 runtime bundled with torch 2.10) -- a sporadic segfault that
        depends on how many streams the process has created.  Single-branch graphs never enter that loop."""
        profiling.push("train_step")
        if self._has_next:  # the batch handed over by prefetch()
            main = torch.cuda.current_stream()
            main.wait_event(self._ev_staged)
            self.load(self._nX, self._ny)
            self._ev_consumed.record(main)
            self._has_next = False
        if self.use_graph and self.calls == 1:
            torch.cuda.synchronize()
            _quiesce_collective_watchdog()
            for seg in ((self.whole,) if self.whole is not None else
                        (self.pre, self.early, self.front, self.front_b, self.side_a, self.sideq, self.tail)):
                if seg is not None:
                    seg.capture()
            torch.cuda.synchronize()
        if self.whole is not None:
            self.whole.run()
            self._done()
            return
        self.pre.run()
        if self.side2 is not None:  # untouched table rows: their own stream, beside everything up to the row update
            main = torch.cuda.current_stream()
            self.ev_pre.record(main)
            self.side2.wait_event(self.ev_pre)
            with torch.cuda.stream(self.side2):
                self.early.run()
                self.ev_early.record(self.side2)
        else:
            self.early.run()
        self.front.run()
        if self.front_b is not None:  # early fork: ready weight-gradient GEMMs beside the rest of the backward chain
            main = torch.cuda.current_stream()
            self.ev_fork_a.record(main)
            self.side.wait_event(self.ev_fork_a)
            with torch.cuda.stream(self.side):
                self.side_a.run()
            self.front_b.run()
        if not self.overlap:
            self.tail.run()
            self.sideq.run()
        else:
            self._forked(self.sideq.run, self.tail.run)
        self._done()

    def _done(self):
        profiling.pop()
        self.calls += 1
        self.opt.steps_done += 1
        self.opt.dirty = True
        if self.par is not None:
            self.par.dirty = True
