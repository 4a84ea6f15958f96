"""MMoE (reference model/mmoe.py:8-119): Ne expert DNNs and T gate DNNs on the shared input, softmax-gated
expert mix per task, T towers + heads.  Layer l of every expert AND gate DNN is one grouped MFMA launch."""
import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .towers import build_tower_modules, emit_towers
from .utils import DNN, emit_dnn_stacks


class MMOE(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.num_experts = mc.get("num_experts", 4)
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.expert_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        self.gate_dnn_hidden_units = mc.get("gate_dnn_hidden_units", [64])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        l2 = mc.get("l2_reg_dnn", 0)
        drop, act, bn = mc.get("dnn_dropout", 0), mc.get("dnn_activation", "relu"), mc.get("dnn_use_bn", False)
        kw = dict(activation=act, l2_reg=l2, dropout_rate=drop, use_bn=bn, init_std=init_std, device=device)
        self.expert_dnn = nn.ModuleList([DNN(self.input_dim, self.expert_dnn_hidden_units, **kw)
                                         for _ in range(self.num_experts)])
        if len(self.gate_dnn_hidden_units) > 0:
            self.gate_dnn = nn.ModuleList([DNN(self.input_dim, self.gate_dnn_hidden_units, **kw)
                                           for _ in range(self.num_tasks)])
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], self.gate_dnn.named_parameters()), l2=l2)
        gate_in = self.gate_dnn_hidden_units[-1] if len(self.gate_dnn_hidden_units) > 0 else self.input_dim
        self.gate_dnn_final_layer = nn.ModuleList([nn.Linear(gate_in, self.num_experts, bias=False)
                                                   for _ in range(self.num_tasks)])
        build_tower_modules(self, self.expert_dnn_hidden_units[-1], self.tower_dnn_hidden_units, act, l2, drop, bn,
                            init_std, device)
        for module in (self.expert_dnn, self.gate_dnn_final_layer, self.tower_dnn_final_layer):
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], module.named_parameters()), l2=l2)
        self.to(device)

    def _dnn_input_store16(self, plan):
        """bf16-storage path: dnn_input is read by the first layers of the expert and gate networks only (no gate DNN:
        the gate kernel reads it too) -- a bf16 buffer when every one of those layers runs on the bf16-storage kernels."""
        if not hasattr(self, "gate_dnn"):
            return False
        plain = not self.model_config.get("dnn_use_bn", False)
        firsts = [d.linears[0] for d in list(self.expert_dnn) + list(self.gate_dnn)]
        return plain and all(E.g16_layer_ok(plan.B, l.in_features, l.out_features, plan.training) for l in firsts)

    def _build_graph(self, plan, store, x0):
        Ne, T = self.num_experts, self.num_tasks
        # bf16-storage path: the expert outputs are read by the gate kernels only, which widen bf16 exactly
        # (mml_gate_group.out_bf16 bit 3) -- half the bytes of the largest tensor three kernels move
        H_, G_ = self.expert_dnn_hidden_units[-1], (self.gate_dnn_hidden_units[-1] if hasattr(self, "gate_dnn") else 0)
        # Round 5 (KuaiRec-32, B = 65 536, same box): the second layers' forward 136 -> 121 us, but the gate kernels -- one
        # 16-byte load per lane and expert -- fell to 8-byte loads: backward 161 -> 191, forward 90 -> 98; a net loss.
        # Round 6: the MMoE gate kernels have a form with EIGHT row columns per lane on 32-lane groups (csrc/rows_fast.hip,
        # HV = 2: 16-byte loads of bf16 rows, two samples per wave and trip) for 129..256-wide experts under gate inputs
        # of at most 128 columns, 4 experts x 2 tasks: there the experts are bf16 by default (MMLREC_BF16_EXPERTS=0 / 1
        # forces either way).
        import os
        hv_form = 128 < H_ <= 256 and H_ % 8 == 0 and 0 < G_ <= 128 and Ne <= 4 and T <= 2
        env16 = os.environ.get("MMLREC_BF16_EXPERTS")
        e16 = (plan.bf16 and (env16 == "1" or (env16 is None and hv_form)) and hasattr(self, "gate_dnn") and
               E._fast_row_width_ok(H_) and E._fast_row_width_ok(G_) and H_ % 8 == 0 and Ne * max(T, 2) <= 32)
        stacks = [self.expert_dnn[e].layer_problems(plan, store, f"expert_dnn.{e}", x0, last16=e16) for e in range(Ne)]
        if hasattr(self, "gate_dnn"):
            stacks += [self.gate_dnn[t].layer_problems(plan, store, f"gate_dnn.{t}", x0) for t in range(T)]
        tops = emit_dnn_stacks(plan, stacks)
        experts = tops[:Ne]
        gate_in = tops[Ne:] if hasattr(self, "gate_dnn") else [x0] * T
        H = self.expert_dnn_hidden_units[-1]
        # bf16-storage path: a task's mixture is read by its tower's first layer only -- the gate kernel writes it as bf16
        # when that layer runs on the bf16-storage kernels (mml_gate_group.out_bf16)
        mc = self.model_config
        mix16 = (plan.bf16 and hasattr(self, "tower_dnn") and not mc.get("dnn_use_bn", False) and
                 E._fast_row_width_ok(H) and all(E._fast_row_width_ok(g.n) for g in gate_in) and
                 E.g16_layer_ok(plan.B, H, self.tower_dnn_hidden_units[0], plan.training))
        gates = [dict(G=gate_in[t], Wg=store.pvals[f"gate_dnn_final_layer.{t}.weight"],
                      mix=plan.val(H, name=f"mmoe_out.{t}", store16=mix16), expert=list(range(Ne))) for t in range(T)]
        plan.add(E.GateGroupOp(experts, gates, H))
        plan.layer_outputs["expert_outputs"] = experts
        plan.layer_outputs["mmoe_outputs"] = [g["mix"] for g in gates]
        plan.layer_outputs["gate_outputs"] = [g["P"] for g in gates]
        return emit_towers(self, plan, store, [g["mix"] for g in gates])
