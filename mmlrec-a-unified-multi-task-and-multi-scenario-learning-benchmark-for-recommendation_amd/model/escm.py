"""ESCM (reference model/escm.py:10-112): ESMM's two towers with THREE outputs [ctr, cvr, ctr * cvr] for two tasks,
trained with the loss branch BaseModel.fit keeps for it (model/basemodel.py:284-292):
    loss = BCE(ctr, y0) + counterfactual_w * IPW(BCE_sum(cvr, y1)) + global_w * BCE(ctr * cvr, y1).
Both heads go through the ONE PredictionLayer BaseModel creates (`self.out`, state_dict key `out.bias`).

Reference behaviour kept: train metrics use output columns [0, 2] (basemodel.py:326-327), predict() returns all three
columns and the driver's per-task rows read columns 0 and 1 (main.py:128-172).  Reference defect NOT kept:
`evaluate()` there feeds sklearn a [N,3] prediction against a [N,2] label matrix and raises ValueError at the first
validation; here it scores columns [0, 2] like the training loop does.  `model_name: escm_dr` (a fourth "imputation"
tower whose output no loss term reads) is rejected."""
import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .utils import DNN, dnn_options, emit_dnn_stacks, l2_on_weights


class ESCM(BaseModel):
    num_outputs = 3
    metric_columns = (0, 2)

    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.model_name = mc.get("model_name", "escm")
        if self.model_name == "escm_dr":
            raise NotImplementedError("escm_dr: the imputation tower's output is read by no loss term of the reference")
        self.counterfactual_w, self.global_w = 0.1, 1
        if self.num_tasks != 2:
            raise ValueError("the length of task_names must be equal to 2")
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
        return None  # model/escm.py:74-96 never looks at domain_mask

    def _build_graph(self, plan, store, x0):
        tops = emit_dnn_stacks(plan, [self.ctr_dnn.layer_problems(plan, store, "ctr_dnn", x0),
                                      self.cvr_dnn.layer_problems(plan, store, "cvr_dnn", x0)])
        bias = store.pvals["out.bias"]
        return E.EscmHeadOp([dict(Hin=tops[0], w=store.pvals["ctr_dnn_final_layer.weight"], bias=bias),
                             dict(Hin=tops[1], w=store.pvals["cvr_dnn_final_layer.weight"], bias=bias)],
                            self.counterfactual_w, self.global_w)
