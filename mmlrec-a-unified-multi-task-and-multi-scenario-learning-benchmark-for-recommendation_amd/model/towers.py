"""Task towers + prediction heads shared by SharedBottom / MMoE / PLE (reference model/mmoe.py:43-57, :91-108)."""
import torch.nn as nn

from .. import engine as E
from .utils import DNN, PredictionLayer, emit_dnn_stacks


def build_tower_modules(model, in_dim, tower_units, activation, l2_reg, dropout, use_bn, init_std, device):
    """Registers tower_dnn (if any), tower_dnn_final_layer and out on `model` in the reference's order."""
    T = model.num_tasks
    if len(tower_units) > 0:
        model.tower_dnn = nn.ModuleList([DNN(in_dim, tower_units, activation=activation, l2_reg=l2_reg,
                                             dropout_rate=dropout, use_bn=use_bn, init_std=init_std, device=device)
                                         for _ in range(T)])
        model.add_regularization_weight(
            filter(lambda x: "weight" in x[0] and "bn" not in x[0], model.tower_dnn.named_parameters()), l2=l2_reg)
    model.tower_dnn_final_layer = nn.ModuleList(
        [nn.Linear(tower_units[-1] if len(tower_units) > 0 else in_dim, 1, bias=False) for _ in range(T)])
    model.out = nn.ModuleList([PredictionLayer(task) for task in model.task_types])


def emit_towers(model, plan, store, streams):
    """streams[t] -> tower DNN -> Linear(H->1, no bias) -> +bias -> sigmoid.  Returns the HeadOp."""
    T = model.num_tasks
    if hasattr(model, "tower_dnn"):
        stacks = [model.tower_dnn[t].layer_problems(plan, store, f"tower_dnn.{t}", streams[t]) for t in range(T)]
        tops = emit_dnn_stacks(plan, stacks)
        plan.layer_outputs["tower_outputs"] = tops
    else:
        tops = streams
    heads = [dict(Hin=tops[t], w=store.pvals[f"tower_dnn_final_layer.{t}.weight"], bias=store.pvals[f"out.{t}.bias"])
             for t in range(T)]
    return E.HeadOp(heads)
