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
        if self.opt.table_update == "sparse_rows":
            rows = self.store.ensure_rows(int(B) * max(len(model._sparse_cols()), 1))
        self.plan = model._record(B, True, False, self.store, sparse_rows=rows)
        self.opt_calls = self.opt.calls(self.plan)
        self.allreduce = allreduce  # callable(flat dense-gradient arena) or None
        self.use_graph = bool(use_graph)
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
