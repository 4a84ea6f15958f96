"""One fused training step = forward launches + head/BCE launch + backward launches + optimizer launches, recorded
once per batch size and replayed (as HIP graphs when enabled).  Counterpart of the loop body of BaseModel.fit in the
reference (model/basemodel.py:261-313) minus logging."""
import torch

from . import engine as E


class TrainStep:
    def __init__(self, model, B, use_graph=True, allreduce=None):
        self.model = model
        self.store = model._store()
        self.opt = model.optimizer()
        rows = None
        sharding = getattr(model, "_sharding", None)
        if allreduce is None and sharding is not None:
            from .parallel import make_allreduce
            allreduce = make_allreduce(model._dist, model._dist_group)
        if self.opt.table_update == "sparse_rows":
            if sharding is None:
                rows = self.store.ensure_rows(int(B) * max(len(model._sparse_cols()), 1))
            else:  # this rank serves world*B lookups for each of ITS fields
                sp = model._sparse_cols()
                names = [f"embedding_dict.{sp[f].embedding_name}.weight" for f in sharding.mine]
                rows = self.store.ensure_rows(sharding.world * int(B) * max(len(names), 1), names)
        self.plan = model._record(B, True, False, self.store, sparse_rows=rows)
        self.opt_calls = self.opt.calls(self.plan)
        self.allreduce = allreduce  # callable(flat dense-gradient arena) or None
        # collectives are issued from Python between kernel launches: keep them out of HIP graph capture
        self.use_graph = bool(use_graph) and sharding is None
        self.g_fb = self.g_opt = None
        self.calls = 0

    def _eager(self):
        self.plan.run_train_fwd_bwd()
        if self.allreduce is not None:
            self.allreduce(self.store.arena)
        E.Plan._run(self.opt_calls)

    def _capture(self):
        self.g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fb):
            self.plan.run_train_fwd_bwd()
            if self.allreduce is None:
                E.Plan._run(self.opt_calls)
        if self.allreduce is not None:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt):
                E.Plan._run(self.opt_calls)

    def run(self):
        """plan.X / plan.y must hold the batch. After the call plan.prob / plan.loss hold this step's outputs.
        The first call runs eagerly (HIP graph capture needs a warmed-up allocator/module state and does not execute
        what it records); the second call captures, then every call replays."""
        if not self.use_graph or self.calls == 0:
            self._eager()
        else:
            if self.g_fb is None:
                torch.cuda.synchronize()
                self._capture()
            self.g_fb.replay()
            if self.allreduce is not None:
                self.allreduce(self.store.arena)
                self.g_opt.replay()
        self.calls += 1
        self.opt.steps_done += 1
