"""PLE (reference model/ple.py:10-198): `num_levels` CGC layers -- per task S specific experts on the task stream,
Sh shared experts on the shared stream, one softmax gate per stream mixing its visible experts -- then towers.

Reference quirks kept: `specific_expert_num` shared-expert modules are BUILT per level but only
`shared_expert_num` are used (ple.py:47 vs :120-121), and the last level's shared gate is never consumed
(ple.py:146-152), so its parameters never receive a gradient; training plans do not even compute it."""
import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .towers import build_tower_modules, emit_towers
from .utils import DNN, emit_dnn_stacks


class PLE(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.num_experts = mc.get("num_experts", 4)
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.shared_expert_num = mc.get("shared_expert_num", 1)
        self.specific_expert_num = mc.get("specific_expert_num", 3)
        self.num_levels = mc.get("num_levels", 1)
        self.expert_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        self.gate_dnn_hidden_units = mc.get("gate_dnn_hidden_units", [64])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        l2 = mc.get("l2_reg_dnn", 0)
        drop, act, bn = mc.get("dnn_dropout", 0), mc.get("dnn_activation", "relu"), mc.get("dnn_use_bn", False)
        kw = dict(activation=act, l2_reg=l2, dropout_rate=drop, use_bn=bn, init_std=init_std, device=device)
        T, S, Sh, Lv = self.num_tasks, self.specific_expert_num, self.shared_expert_num, self.num_levels
        H = self.expert_dnn_hidden_units[-1]

        def nested(num_tasks, expert_num, units):
            return nn.ModuleList([nn.ModuleList([nn.ModuleList(
                [DNN(self.input_dim if lv == 0 else H, units, **kw) for _ in range(expert_num)])
                for _ in range(num_tasks)]) for lv in range(Lv)])

        self.specific_experts = nested(T, S, self.expert_dnn_hidden_units)
        self.shared_experts = nested(1, S, self.expert_dnn_hidden_units)
        has_gate_dnn = len(self.gate_dnn_hidden_units) > 0
        if has_gate_dnn:
            self.specific_gate_dnn = nested(T, 1, self.gate_dnn_hidden_units)
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], self.specific_gate_dnn.named_parameters()),
                l2=l2)

        def gate_in(lv):
            return self.gate_dnn_hidden_units[-1] if has_gate_dnn else (self.input_dim if lv == 0 else H)

        self.specific_gate_dnn_final_layer = nn.ModuleList([nn.ModuleList(
            [nn.Linear(gate_in(lv), S + Sh, bias=False) for _ in range(T)]) for lv in range(Lv)])
        if has_gate_dnn:
            self.shared_gate_dnn = nn.ModuleList([DNN(self.input_dim if lv == 0 else H, self.gate_dnn_hidden_units,
                                                      **kw) for lv in range(Lv)])
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], self.shared_gate_dnn.named_parameters()),
                l2=l2)
        self.shared_gate_dnn_final_layer = nn.ModuleList([nn.Linear(gate_in(lv), T * S + Sh, bias=False)
                                                          for lv in range(Lv)])
        build_tower_modules(self, H, self.tower_dnn_hidden_units, act, l2, drop, bn, init_std, device)
        for module in (self.specific_experts, self.shared_experts, self.specific_gate_dnn_final_layer,
                       self.shared_gate_dnn_final_layer, self.tower_dnn_final_layer):
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], module.named_parameters()), l2=l2)
        self.to(device)

    def _build_graph(self, plan, store, x0):
        T, S, Sh = self.num_tasks, self.specific_expert_num, self.shared_expert_num
        H = self.expert_dnn_hidden_units[-1]
        has_gate_dnn = hasattr(self, "specific_gate_dnn")
        if not has_gate_dnn:
            raise NotImplementedError("PLE with gate_dnn_hidden_units == [] is not planned yet")
        streams = [x0] * (T + 1)
        for lv in range(self.num_levels):
            last = lv == self.num_levels - 1
            want_shared_gate = not (last and plan.training)  # dead compute in the reference (ple.py:146-152)
            stacks = []
            for i in range(T):
                for j in range(S):
                    stacks.append(self.specific_experts[lv][i][j].layer_problems(
                        plan, store, f"specific_experts.{lv}.{i}.{j}", streams[i]))
            for k in range(Sh):
                stacks.append(self.shared_experts[lv][0][k].layer_problems(
                    plan, store, f"shared_experts.{lv}.0.{k}", streams[T]))
            n_exp = len(stacks)
            for i in range(T):
                stacks.append(self.specific_gate_dnn[lv][i][0].layer_problems(
                    plan, store, f"specific_gate_dnn.{lv}.{i}.0", streams[i]))
            if want_shared_gate:
                stacks.append(self.shared_gate_dnn[lv].layer_problems(plan, store, f"shared_gate_dnn.{lv}", streams[T]))
            tops = emit_dnn_stacks(plan, stacks)
            experts, gins = tops[:n_exp], tops[n_exp:]
            gates = []
            for i in range(T):
                members = list(range(i * S, (i + 1) * S)) + [T * S + k for k in range(Sh)]
                gates.append(dict(G=gins[i], Wg=store.pvals[f"specific_gate_dnn_final_layer.{lv}.{i}.weight"],
                                  mix=plan.val(H, name=f"cgc.{lv}.{i}"), expert=members))
            if want_shared_gate:
                gates.append(dict(G=gins[T], Wg=store.pvals[f"shared_gate_dnn_final_layer.{lv}.weight"],
                                  mix=plan.val(H, name=f"cgc.{lv}.shared"), expert=list(range(T * S + Sh))))
            plan.add(E.GateGroupOp(experts, gates, H))
            outs = [g["mix"] for g in gates]
            if want_shared_gate:
                plan.layer_outputs[f"ple_output_{lv}"] = outs
            streams = outs
        return emit_towers(self, plan, store, streams[:T])
