"""AITM (reference model/aitm.py:8-111): one bottom DNN per task; task i > 0 transfers information from task i-1
through a two-token attention over (g(feat[i-1]), feat[i]) with shared value / key / query projections h1, h2, h3
(model/aitm.py:84-93), then towers + heads.  The projections are grouped GEMMs, the attention itself is the
`mml_attn2` row kernel.  Fifth member of the wider zoo."""
import numpy as np
import torch.nn as nn

from .. import _lib as L
from .. import engine as E
from .basemodel import BaseModel
from .towers import emit_towers
from .utils import DNN, PredictionLayer, emit_dnn_stacks


class AITM(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.task_names = mc.get("task_names", ["ctr", "ctcvr"])
        self.task_types = mc.get("task_types", ["binary", "binary"])
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.bottom_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        l2 = mc.get("l2_reg_dnn", 0)
        drop, act, bn = mc.get("dnn_dropout", 0), mc.get("dnn_activation", "relu"), mc.get("dnn_use_bn", False)
        self.num_tasks = len(self.task_names)
        if self.num_tasks != 2:
            raise ValueError("the length of task_names must be equal to 2")
        if len(dnn_feature_columns) == 0:
            raise ValueError("dnn_feature_columns is null!")
        if len(self.task_types) != self.num_tasks:
            raise ValueError("num_tasks must be equal to the length of task_types")
        for task_type in self.task_types:
            if task_type != "binary":
                raise ValueError("task must be binary in ESMM, {} is illegal".format(task_type))
        H, T = self.bottom_dnn_hidden_units[-1], self.num_tasks
        # registration (and random-draw) order of the reference: g, h1, h2, h3, bottom, tower_dnn, final layers, out
        self.g = nn.ModuleList([nn.Linear(H, H) for _ in range(T - 1)])
        self.h1, self.h2, self.h3 = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        kw = dict(activation=act, dropout_rate=drop, use_bn=bn, init_std=init_std, device=device)
        self.bottom = nn.ModuleList([DNN(self.input_dim, self.bottom_dnn_hidden_units, l2_reg=l2, **kw)
                                     for _ in range(T)])
        if len(self.tower_dnn_hidden_units) > 0:
            self.tower_dnn = nn.ModuleList([DNN(H, self.tower_dnn_hidden_units, **kw) for _ in range(T)])
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], self.tower_dnn.named_parameters()), l2=l2)
        Ht = self.tower_dnn_hidden_units[-1] if len(self.tower_dnn_hidden_units) > 0 else H
        self.tower_dnn_final_layer = nn.ModuleList([nn.Linear(Ht, 1, bias=False) for _ in range(T)])
        self.out = nn.ModuleList([PredictionLayer(task) for task in self.task_types])
        for mods in (self.bottom, self.tower_dnn_final_layer):
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], mods.named_parameters()), l2=l2)
        self.to(device)

    def _head_mask_cols(self):
        # model/aitm.py:104-105: msl only (`output * domain_mask[:, i]`)
        return list(range(self.num_tasks)) if self.task_name == "msl" else None

    def _build_graph(self, plan, store, x0):
        T, H = self.num_tasks, self.bottom_dnn_hidden_units[-1]
        feat = emit_dnn_stacks(plan, [self.bottom[i].layer_problems(plan, store, f"bottom.{i}", x0)
                                      for i in range(T)])
        pv = store.pvals
        for i in range(1, T):
            p = plan.val(H, name=f"aitm.{i}.p")
            plan.add(E.LinearGroupOp([dict(x=feat[i - 1], W=pv[f"g.{i - 1}.weight"], b=pv[f"g.{i - 1}.bias"], out=p)]))
            toks = []
            # h1 / h2 / h3 are shared by the two tokens: one launch per token, so that the second use of a weight
            # ACCUMULATES its gradient after the first has written it
            for name, x in (("p", p), ("q", feat[i])):
                vkq = [plan.val(H, name=f"aitm.{i}.{name}.{h}") for h in ("v", "k", "q")]
                plan.add(E.LinearGroupOp([dict(x=x, W=pv[f"{h}.weight"], b=pv[f"{h}.bias"], out=o)
                                          for h, o in zip(("h1", "h2", "h3"), vkq)]))
                toks.append(tuple(vkq))
            out = plan.val(H, name=f"aitm.{i}.out")
            plan.add(E.Attn2Op(toks, out, np.float32(np.sqrt(H))))
            feat[i] = out
        return emit_towers(self, plan, store, feat)
