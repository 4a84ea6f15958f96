"""One fused training step = forward launches + head/BCE launch + backward launches + optimizer launches, recorded
once per batch size and replayed (as HIP graphs when enabled).  Counterpart of the loop body of BaseModel.fit in the
reference (model/basemodel.py:261-313) minus logging.

Two HIP streams: after the backward chain has produced every dL/d(pre-activation), the step forks --
  main stream : table scatter -> table optimizer            (HBM / atomics bound)
  side stream : all weight-gradient GEMMs -> [all-reduce] -> MLP optimizer   (MFMA bound)
-- and joins at the end, so the 3.2 GB table stream of the reference-exact dense Adam hides behind the wgrad GEMMs.
"""
import torch

from . import engine as E


class TrainStep:
    def __init__(self, model, B, use_graph=True, allreduce=None, overlap=True):
        self.model = model
        self.store = model._store()
        self.opt = model.optimizer()
        rows = None
        sharding = getattr(model, "_sharding", None)
        if allreduce is None and sharding is not None:
            from .parallel import make_allreduce
            allreduce = make_allreduce(model._dist, model._dist_group)
        lazy = self.opt.table_update == "lazy_exact"
        if lazy and sharding is not None:
            raise NotImplementedError("lazy_exact table updates on the table-sharded path")
        if self.opt.table_update in ("sparse_rows", "lazy_exact"):
            if sharding is None:
                rows = self.store.ensure_rows(int(B) * max(len(model._sparse_cols()), 1))
            else:  # this rank serves world*B lookups for each of ITS fields
                sp = model._sparse_cols()
                names = [f"embedding_dict.{sp[f].embedding_name}.weight" for f in sharding.mine]
                rows = self.store.ensure_rows(sharding.world * int(B) * max(len(names), 1), names)
        # lazy_exact lists the batch's rows in a pre-pass (before the gather), so the scatter only accumulates
        self.plan = model._record(B, True, False, self.store, sparse_rows=None if lazy else rows)
        self.opt_split = self.opt.calls_split(self.plan)
        self.opt_calls = self.opt_split["pre"] + self.opt_split["mlp"] + self.opt_split["tables"]
        self.allreduce = allreduce  # callable(flat dense-gradient arena) or None
        # collectives are issued from Python between kernel launches: keep them out of HIP graph capture
        self.use_graph = bool(use_graph) and sharding is None
        self.overlap = bool(overlap)
        self.side = torch.cuda.Stream(device=self.store.device) if self.overlap else None
        # fork / join events live as long as the step
        self.ev_fork = torch.cuda.Event() if self.overlap else None
        self.ev_join = torch.cuda.Event() if self.overlap else None
        self.g_fb = self.g_side = self.g_tail = None
        self.calls = 0

    def _front(self):
        p, run = self.plan, E.Plan._run
        run(self.opt_split["pre"])
        run(p.fwd)
        run(p.head_train)
        run(p.bwd)

    def _side(self, with_allreduce=True):
        E.Plan._run(self.plan.bwd_side)
        if self.allreduce is not None and with_allreduce:
            self.allreduce(self.store.arena)
        E.Plan._run(self.opt_split["mlp"])

    def _tail(self):
        E.Plan._run(self.plan.bwd_tail)
        E.Plan._run(self.opt_split["tables"])

    def _forked(self, side, tail):
        main = torch.cuda.current_stream()
        self.ev_fork.record(main)
        self.side.wait_event(self.ev_fork)
        with torch.cuda.stream(self.side):
            side()
            self.ev_join.record(self.side)
        tail()
        main.wait_event(self.ev_join)

    def _body(self, with_allreduce=True):
        # (mml_gemm_set_wgrad_lds_pad can cap the wgrad GEMMs' residency so that the table optimizer co-resides; with
        # the direct-to-LDS GEMMs the standalone speed of wgrad at 4 workgroups/CU wins, so the pad stays 0)
        self._front()
        if not self.overlap:
            self._tail()
            self._side(with_allreduce)
            return
        self._forked(lambda: self._side(with_allreduce), self._tail)

    def _capture(self, fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    def run(self):
        """plan.X / plan.y must hold the batch. After the call plan.prob / plan.loss hold this step's outputs.
        The first call runs eagerly (HIP graph capture needs warmed-up state and does not execute what it records);
        the second call captures, then every call replays.

        With two streams the step is THREE single-stream graphs (front, side, tail) forked and joined with events at
        replay time, not one graph with two branches: hipGraphLaunch of a multi-branch graph walks past the end of the
        exec's parallel-stream vector when one of those streams shares a hardware queue with the launch stream
        (hip::Graph::UpdateStreams, ROCm 7.0 runtime bundled with torch 2.10) -- a sporadic segfault that depends on
        how many streams the process has created.  Single-branch graphs never enter that loop."""
        if not self.use_graph or self.calls == 0:
            self._body()
        elif not self.overlap:
            if self.g_fb is None:
                torch.cuda.synchronize()
                self.g_fb = self._capture(self._body)
            self.g_fb.replay()
        else:
            if self.g_fb is None:
                torch.cuda.synchronize()
                self.g_fb = self._capture(self._front)
                self.g_side = self._capture(self._side)
                self.g_tail = self._capture(self._tail)
                torch.cuda.synchronize()
            self.g_fb.replay()
            self._forked(self.g_side.replay, self.g_tail.replay)
        self.calls += 1
        self.opt.steps_done += 1
        self.opt.dirty = True
