"""AITM (reference model/aitm.py:8-111): one bottom DNN per task; task i > 0 transfers information from task i-1
through a two-token attention over (g(feat[i-1]), feat[i]) with shared value / key / query projections h1, h2, h3
(model/aitm.py:84-93), then towers + heads.  The projections are grouped GEMMs, the attention itself is the
`mml_attn2` row kernel.  Fifth member of the wider zoo."""
import numpy as np
import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .towers import emit_towers
from .utils import DNN, PredictionLayer, dnn_options, emit_dnn_stacks, l2_on_weights


class AITM(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.task_names = mc.get("task_names", ["ctr", "ctcvr"])
        self.task_types = mc.get("task_types", ["binary", "binary"])
        self.num_tasks = T = len(self.task_names)
        # the reference's argument checks (model/aitm.py:31-41)
        if T != 2:
            raise ValueError("the length of task_names must be equal to 2")
        if not dnn_feature_columns:
            raise ValueError("dnn_feature_columns is null!")
        if len(self.task_types) != T:
            raise ValueError("num_tasks must be equal to the length of task_types")
        bad = [t for t in self.task_types if t != "binary"]
        if bad:
            raise ValueError("task must be binary in ESMM, {} is illegal".format(bad[0]))
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.bottom_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        H, l2 = self.bottom_dnn_hidden_units[-1], mc.get("l2_reg_dnn", 0)
        opts = dnn_options(mc, init_std, device)
        # registration (= random-draw) order of the reference: g, h1, h2, h3, bottom, tower_dnn, final layers, out
        self.g = nn.ModuleList(nn.Linear(H, H) for _ in range(T - 1))
        self.h1, self.h2, self.h3 = (nn.Linear(H, H) for _ in range(3))
        self.bottom = nn.ModuleList(DNN(self.input_dim, self.bottom_dnn_hidden_units, l2_reg=l2, **opts)
                                    for _ in range(T))
        Ht = H
        if self.tower_dnn_hidden_units:
            self.tower_dnn = nn.ModuleList(DNN(H, self.tower_dnn_hidden_units, **opts) for _ in range(T))
            Ht = self.tower_dnn_hidden_units[-1]
        self.tower_dnn_final_layer = nn.ModuleList(nn.Linear(Ht, 1, bias=False) for _ in range(T))
        self.out = nn.ModuleList(PredictionLayer(task) for task in self.task_types)
        l2_on_weights(self, (getattr(self, "tower_dnn", None), self.bottom, self.tower_dnn_final_layer), l2)
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
