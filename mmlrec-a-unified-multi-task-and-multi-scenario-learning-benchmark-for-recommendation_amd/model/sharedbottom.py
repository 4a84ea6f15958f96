"""SharedBottom (reference model/sharedbottom.py:9-86): one bottom DNN, T towers + heads."""
from .basemodel import BaseModel
from .towers import build_tower_modules, emit_towers
from .utils import DNN, emit_dnn_stacks


class SharedBottom(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.num_experts = mc.get("num_experts", 4)
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.bottom_dnn_hidden_units = mc.get("bottom_dnn_hidden_units", [256, 128])
        self.gate_dnn_hidden_units = mc.get("gate_dnn_hidden_units", [64])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        l2 = mc.get("l2_reg_dnn", 0)
        drop, act, bn = mc.get("dnn_dropout", 0), mc.get("dnn_activation", "relu"), mc.get("dnn_use_bn", False)
        self.bottom_dnn = DNN(self.input_dim, self.bottom_dnn_hidden_units, activation=act, dropout_rate=drop,
                              use_bn=bn, init_std=init_std, device=device)
        build_tower_modules(self, self.bottom_dnn_hidden_units[-1], self.tower_dnn_hidden_units, act, l2, drop, bn,
                            init_std, device)
        self.add_regularization_weight(
            filter(lambda x: "weight" in x[0] and "bn" not in x[0], self.bottom_dnn.named_parameters()), l2=l2)
        self.add_regularization_weight(
            filter(lambda x: "weight" in x[0] and "bn" not in x[0], self.tower_dnn_final_layer.named_parameters()),
            l2=l2)
        self.to(device)

    def _build_graph(self, plan, store, x0):
        bottom = emit_dnn_stacks(plan, [self.bottom_dnn.layer_problems(plan, store, "bottom_dnn", x0)])[0]
        plan.layer_outputs["shared_bottom_outputs"] = bottom
        return emit_towers(self, plan, store, [bottom] * self.num_tasks)
