"""Cross-Stitch networks (reference model/cross_stitch.py:7-121): one shared layer, then per level one Linear+ReLU per
task followed by a cross-stitch unit -- the concatenated task activations times a learned [T*d, T*d] matrix (stored
[in, out], initialised to the identity) -- then the usual towers and heads.  Everything is a GEMM: the task layers write
straight into column slices of one [B, T*d] buffer (no torch.cat pass), the stitch is one `[K,N]`-layout GEMM on it,
and the next level reads column slices of its output.  Third member of the wider model zoo."""
import torch
import torch.nn as nn

from .. import _lib as L
from .. import engine as E
from .basemodel import BaseModel
from .towers import build_tower_modules, emit_towers
from .utils import DNN, blocks_out_act, emit_blocks_into, emit_dnn_stacks


class CrossStitchLayer(nn.Module):
    def __init__(self, input_dims, device="cpu"):
        super().__init__()
        self.last_dims = list(input_dims)
        self.total_last_dim = sum(self.last_dims)
        self.cross_stitch_weight = nn.Parameter(torch.eye(self.total_last_dim))


class CrossStitch(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.shared_hidden_unit = mc.get("shared_hidden_unit", 256)
        self.dnn_hidden_units = mc.get("dnn_hidden_units", [256, 128])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        l2 = mc.get("l2_reg_dnn", 0)
        drop, act, bn = mc.get("dnn_dropout", 0), mc.get("dnn_activation", "relu"), mc.get("dnn_use_bn", False)
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        T = self.num_tasks

        def block(k, n):
            return DNN(k, [n], activation=act, l2_reg=l2, dropout_rate=drop, use_bn=bn, init_std=init_std,
                       device=device)

        self.shared_layer = block(self.input_dim, self.shared_hidden_unit)
        self.cross_stitch = nn.ModuleDict()
        for i, d in enumerate(self.dnn_hidden_units):  # same registration order as the reference (:49-62)
            k = self.shared_hidden_unit if i == 0 else self.dnn_hidden_units[i - 1]
            self.cross_stitch[f"task_layer_{i}"] = nn.ModuleList([block(k, d) for _ in range(T)])
            self.cross_stitch[f"gate_{i}"] = CrossStitchLayer([d] * T, device=device)
        build_tower_modules(self, self.dnn_hidden_units[-1], self.tower_dnn_hidden_units, act, l2, drop, bn, init_std,
                            device)
        self.to(device)

    def _build_graph(self, plan, store, x0):
        T = self.num_tasks
        sh = emit_dnn_stacks(plan, [self.shared_layer.layer_problems(plan, store, "shared_layer", x0)])[0]
        ins = [sh] * T
        for i, d in enumerate(self.dnn_hidden_units):
            if d % 4:
                raise NotImplementedError("cross-stitch widths must be multiples of 4 (16-byte column slices)")
            act = blocks_out_act(plan, self.cross_stitch[f"task_layer_{i}"])
            cat = plan.val(T * d, act=act, name=f"cross_stitch.{i}.cat")
            parts = [E.Val(cat.buf[:, j * d:(j + 1) * d], act, name=f"cross_stitch.{i}.task.{j}")
                     for j in range(T)]
            pfx = f"cross_stitch.task_layer_{i}"
            emit_blocks_into(plan, store, self.cross_stitch[f"task_layer_{i}"], [f"{pfx}.{j}" for j in range(T)], ins,
                             parts)
            plan.add(E.JoinOp(parts, cat))
            mix = plan.val(T * d, name=f"cross_stitch.{i}.mix")
            plan.add(E.LinearGroupOp([dict(x=cat, W=store.pvals[f"cross_stitch.gate_{i}.cross_stitch_weight"], b=None,
                                           out=mix, w_kn=1)]))
            ins = plan.add(E.SplitOp(plan, mix, [d] * T)).parts
        plan.layer_outputs["cross_stitch_outputs"] = ins
        return emit_towers(self, plan, store, ins)
