"""ESMM (reference model/esmm.py:8-71): a CTR tower and a CVR tower on the shared embedding input; outputs
[ctr, ctr * cvr].  Both heads go through the ONE PredictionLayer BaseModel creates (`self.out`, state_dict key
`out.bias`).  Second member of the wider model zoo on the same kernels."""
import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .utils import DNN, dnn_options, emit_dnn_stacks, l2_on_weights


class ESMM(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        if self.num_tasks != 2:
            raise ValueError("ESMM has exactly two outputs (ctr, ctcvr)")
        units = self.expert_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        opts = dnn_options(mc, init_std, device)
        # registration (= random-draw) order of the reference: ctr tower, cvr tower, then the two final layers
        self.ctr_dnn, self.cvr_dnn = DNN(self.input_dim, units, **opts), DNN(self.input_dim, units, **opts)
        self.ctr_dnn_final_layer, self.cvr_dnn_final_layer = (nn.Linear(units[-1], 1, bias=False) for _ in range(2))
        l2_on_weights(self, (self.ctr_dnn, self.cvr_dnn, self.ctr_dnn_final_layer.weight,
                             self.cvr_dnn_final_layer.weight), mc.get("l2_reg_dnn", 0))
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
