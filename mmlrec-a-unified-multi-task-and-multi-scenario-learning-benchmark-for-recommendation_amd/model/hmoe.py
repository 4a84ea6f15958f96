"""HMoE (reference model/hmoe.py:10-153): MMoE (experts, gates, tower DNNs) plus a second, task-level mixture: every
task i has a softmax "task weight" gate over the T tower outputs and its head reads sum_j w_i[j] * tower_j -- with
tower_j DETACHED for j != i (model/hmoe.py:124-129), so a tower only learns from its own task while the task-weight
networks see all of them.  Both mixtures run on the gate kernels; the detach is expressed by giving gate i the real
tower i and gradient-free aliases of the other towers as its experts.  Fourth member of the wider zoo."""
import torch.nn as nn

from .. import engine as E
from .mmoe import MMOE
from .utils import DNN, PredictionLayer, emit_dnn_stacks


class HMOE(MMOE):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        # MMOE.__init__ registers expert_dnn, gate_dnn, gate_dnn_final_layer, tower_dnn, tower_dnn_final_layer, out in
        # the reference's order; HMoE creates task_weight / task_weight_final_layer BEFORE tower_dnn_final_layer and
        # out (model/hmoe.py:51-69), so those two are re-created after them (same registration and random-draw order)
        nn.Module.__init__(self)
        from .basemodel import BaseModel
        BaseModel.__init__(self, linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns,
                           init_std=init_std, device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.num_experts = mc.get("num_experts", 4)
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.expert_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        self.gate_dnn_hidden_units = mc.get("gate_dnn_hidden_units", [64])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        self.task_weight_hidden_units = mc.get("task_weight_hidden_units", [64])
        l2 = mc.get("l2_reg_dnn", 0)
        drop, act, bn = mc.get("dnn_dropout", 0), mc.get("dnn_activation", "relu"), mc.get("dnn_use_bn", False)
        kw = dict(activation=act, l2_reg=l2, dropout_rate=drop, use_bn=bn, init_std=init_std, device=device)
        T = self.num_tasks
        self.expert_dnn = nn.ModuleList([DNN(self.input_dim, self.expert_dnn_hidden_units, **kw)
                                         for _ in range(self.num_experts)])
        if len(self.gate_dnn_hidden_units) > 0:
            self.gate_dnn = nn.ModuleList([DNN(self.input_dim, self.gate_dnn_hidden_units, **kw) for _ in range(T)])
        gate_in = self.gate_dnn_hidden_units[-1] if len(self.gate_dnn_hidden_units) > 0 else self.input_dim
        self.gate_dnn_final_layer = nn.ModuleList([nn.Linear(gate_in, self.num_experts, bias=False) for _ in range(T)])
        H = self.expert_dnn_hidden_units[-1]
        if len(self.tower_dnn_hidden_units) > 0:
            self.tower_dnn = nn.ModuleList([DNN(H, self.tower_dnn_hidden_units, **kw) for _ in range(T)])
        if len(self.task_weight_hidden_units) > 0:
            self.task_weight = nn.ModuleList([DNN(self.input_dim, self.task_weight_hidden_units, **kw)
                                              for _ in range(T)])
        tw_in = self.task_weight_hidden_units[-1] if len(self.task_weight_hidden_units) > 0 else self.input_dim
        self.task_weight_final_layer = nn.ModuleList([nn.Linear(tw_in, T, bias=False) for _ in range(T)])
        Ht = self.tower_dnn_hidden_units[-1] if len(self.tower_dnn_hidden_units) > 0 else H
        self.tower_dnn_final_layer = nn.ModuleList([nn.Linear(Ht, 1, bias=False) for _ in range(T)])
        self.out = nn.ModuleList([PredictionLayer(task) for task in self.task_types])
        for mods in [m for m in (getattr(self, "gate_dnn", None), getattr(self, "tower_dnn", None),
                                 getattr(self, "task_weight", None), self.expert_dnn, self.gate_dnn_final_layer,
                                 self.task_weight_final_layer, self.tower_dnn_final_layer) if m is not None]:
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], mods.named_parameters()), l2=l2)
        self.to(device)

    def _build_graph(self, plan, store, x0):
        Ne, T = self.num_experts, self.num_tasks
        stacks = [self.expert_dnn[e].layer_problems(plan, store, f"expert_dnn.{e}", x0) for e in range(Ne)]
        ng = nt = 0
        if hasattr(self, "gate_dnn"):
            stacks += [self.gate_dnn[t].layer_problems(plan, store, f"gate_dnn.{t}", x0) for t in range(T)]
            ng = T
        if hasattr(self, "task_weight"):
            stacks += [self.task_weight[t].layer_problems(plan, store, f"task_weight.{t}", x0) for t in range(T)]
            nt = T
        tops = emit_dnn_stacks(plan, stacks)
        experts = tops[:Ne]
        gate_in = tops[Ne:Ne + ng] if ng else [x0] * T
        tw_in = tops[Ne + ng:Ne + ng + nt] if nt else [x0] * T
        H = self.expert_dnn_hidden_units[-1]
        gates = [dict(G=gate_in[t], Wg=store.pvals[f"gate_dnn_final_layer.{t}.weight"],
                      mix=plan.val(H, name=f"mmoe_out.{t}"), expert=list(range(Ne))) for t in range(T)]
        plan.add(E.GateGroupOp(experts, gates, H))
        plan.layer_outputs["expert_outputs"] = experts
        plan.layer_outputs["mmoe_outputs"] = [g["mix"] for g in gates]
        plan.layer_outputs["gate_outputs"] = [g["P"] for g in gates]
        if hasattr(self, "tower_dnn"):
            towers = emit_dnn_stacks(plan, [self.tower_dnn[t].layer_problems(plan, store, f"tower_dnn.{t}",
                                                                            gates[t]["mix"]) for t in range(T)])
            plan.layer_outputs["tower_outputs"] = towers
        else:
            towers = [g["mix"] for g in gates]
        Ht = towers[0].n
        # task-level mixture: gate i mixes (alias_0, ..., tower_i, ..., alias_{T-1}); the aliases share the tower
        # buffers but take no gradient (`.detach()`, model/hmoe.py:128)
        aliases = [E.Val(v.buf, v.act, needs_grad=False, name=v.name + ".detached") for v in towers]
        tgates = [dict(G=tw_in[i], Wg=store.pvals[f"task_weight_final_layer.{i}.weight"],
                       mix=plan.val(Ht, name=f"task_out.{i}"),
                       expert=[j if j == i else T + j for j in range(T)]) for i in range(T)]
        plan.add(E.GateGroupOp(towers + aliases, tgates, Ht))
        heads = [dict(Hin=tgates[i]["mix"], w=store.pvals[f"tower_dnn_final_layer.{i}.weight"],
                      bias=store.pvals[f"out.{i}.bias"]) for i in range(T)]
        return E.HeadOp(heads)
