"""ESMM (reference model/esmm.py:8-71): a CTR tower and a CVR tower on the shared embedding input; outputs
[ctr, ctr * cvr].  Both heads go through the ONE PredictionLayer BaseModel creates (`self.out`, state_dict key
`out.bias`).  Second member of the wider model zoo on the same kernels."""
import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .utils import DNN, emit_dnn_stacks


class ESMM(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        if self.num_tasks != 2:
            raise ValueError("ESMM has exactly two outputs (ctr, ctcvr)")
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.expert_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        l2 = mc.get("l2_reg_dnn", 0)
        drop, act, bn = mc.get("dnn_dropout", 0), mc.get("dnn_activation", "relu"), mc.get("dnn_use_bn", False)
        self.ctr_dnn = DNN(self.input_dim, self.expert_dnn_hidden_units, activation=act, dropout_rate=drop,
                           use_bn=bn, init_std=init_std, device=device)
        self.cvr_dnn = DNN(self.input_dim, self.expert_dnn_hidden_units, activation=act, dropout_rate=drop,
                           use_bn=bn, init_std=init_std, device=device)
        self.ctr_dnn_final_layer = nn.Linear(self.expert_dnn_hidden_units[-1], 1, bias=False)
        self.cvr_dnn_final_layer = nn.Linear(self.expert_dnn_hidden_units[-1], 1, bias=False)
        for dnn in (self.ctr_dnn, self.cvr_dnn):
            self.add_regularization_weight(
                filter(lambda x: "weight" in x[0] and "bn" not in x[0], dnn.named_parameters()), l2=l2)
        self.add_regularization_weight(self.ctr_dnn_final_layer.weight, l2=l2)
        self.add_regularization_weight(self.cvr_dnn_final_layer.weight, l2=l2)
        self.to(device)

    def _head_mask_cols(self):
        return None  # model/esmm.py:46-71 never looks at domain_mask

    def _build_graph(self, plan, store, x0):
        tops = emit_dnn_stacks(plan, [self.ctr_dnn.layer_problems(plan, store, "ctr_dnn", x0),
                                      self.cvr_dnn.layer_problems(plan, store, "cvr_dnn", x0)])
        plan.layer_outputs["target0_output"], plan.layer_outputs["target1_output"] = tops
        bias = store.pvals["out.bias"]
        return E.EsmmHeadOp([dict(Hin=tops[0], w=store.pvals["ctr_dnn_final_layer.weight"], bias=bias),
                             dict(Hin=tops[1], w=store.pvals["cvr_dnn_final_layer.weight"], bias=bias)])
