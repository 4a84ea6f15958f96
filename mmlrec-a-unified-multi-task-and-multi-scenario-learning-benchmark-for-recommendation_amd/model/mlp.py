"""MLP (reference model/mlp.py:8-66): a chain of one-layer DNN blocks, ONE final Linear(H -> 1, no bias) shared by all
tasks, and a PredictionLayer per task -- first member of the remaining model zoo on the same kernels.

Reference behaviour kept: every head applies its PredictionLayer to the SAME logit tensor and PredictionLayer adds its
bias in place (`output = X; output += self.bias`, model/utils.py:242-245), so head t sees z + b_0 + ... + b_t.
Here: head t gets the derived bias sum(b_0..b_t) (PAddOp, gradients flow back to every b_j, j <= t) and all heads
share the final weight (HeadOp sums their weight gradients)."""
import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .utils import DNN, PredictionLayer, emit_dnn_stacks, l2_on_weights


class MLP(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        if not dnn_feature_columns:
            raise ValueError("dnn_feature_columns is null!")
        self.dnn_use_bn = mc.get("dnn_use_bn", False)
        self.dnn_hidden_units = mc.get("dnn_hidden_units", [256, 128])
        widths = [self.compute_input_dim(dnn_feature_columns), *self.dnn_hidden_units]
        print(f"hidden_units:{widths}")
        l2 = mc.get("l2_reg_dnn", 0)
        # one single-layer DNN block per width step; the reference passes neither init_std nor dropout / bn here, so
        # the blocks are built with the DNN defaults (model/mlp.py:24-26)
        self.mlp_layers = nn.ModuleList(DNN(inputs_dim=k, hidden_units=[n], activation="relu", l2_reg=l2, device=device)
                                        for k, n in zip(widths[:-1], widths[1:]))
        self.final_layer = nn.Linear(widths[-1], 1, bias=False)
        self.out = nn.ModuleList(PredictionLayer(task) for task in self.task_types)
        l2_on_weights(self, (self.mlp_layers,), l2)
        self.to(device)

    def _head_mask_cols(self):
        # model/mlp.py:53-54: heads are masked in the msl mode only (column i), never in mtmsl
        return list(range(self.num_tasks)) if self.task_name == "msl" else None

    def _build_graph(self, plan, store, x0):
        h = x0
        for i, blk in enumerate(self.mlp_layers):
            h = emit_dnn_stacks(plan, [blk.layer_problems(plan, store, f"mlp_layers.{i}", h)])[0]
            plan.layer_outputs[f"mlp_output_{i}"] = h
        w = store.pvals["final_layer.weight"]
        heads = []
        for t in range(self.num_tasks):
            if t == 0:
                bias = store.pvals["out.0.bias"]
            else:  # the in-place bias adds of the heads before this one are part of its logit
                bias = E.PVal(plan.empty(1), plan.zeros(1), f"out.cumbias.{t}")
                plan.add(E.PAddOp([store.pvals[f"out.{j}.bias"] for j in range(t + 1)], bias))
            heads.append(dict(Hin=h, w=w, bias=bias))
        return E.HeadOp(heads)
