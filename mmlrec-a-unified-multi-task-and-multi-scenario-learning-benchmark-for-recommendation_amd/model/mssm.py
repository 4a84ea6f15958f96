"""MSSM (reference model/mssm.py:9-179; the model of the shipped configs_mtl/config_census.json and
configs_mtmsl/config_movielens.json): SNR-trans's layout -- levels of one-layer experts connected by routing gates --
with a coefficient PER OUTPUT COLUMN of every (output, input) transformation: out_o = sum_j (x_j @ M_oj) * z_oj[None, :].

Reference behaviour kept: both the coefficient seeds `u` (one [units] vector per pair) and the transformations live in
plain Python lists (model/mssm.py:26-36), so neither is in state_dict nor optimised; a gate learns its scalar `alpha`
only.  Same lowering as SNR-trans (one [K,N] GEMM per output on column-scaled frozen blocks).
BatchNorm inside the expert blocks and towers (`dnn_use_bn`, config_census) runs through BNOp."""
import torch
import torch.nn as nn

from .basemodel import BaseModel
from .snr_trans import SNR_trans
from .towers import build_tower_modules
from .utils import DNN


class gate(nn.Module):
    def __init__(self, input_dim, output_dim, units, device="cpu", **_unused):
        super().__init__()
        self.input_dim, self.output_dim, self.units = input_dim, output_dim, units
        e = 1e-8
        # draws in the reference's order: alpha, then every u vector (output-major), then every transformation
        self.alpha = nn.Parameter(torch.rand((1,)))
        self.u = torch.stack([torch.stack([nn.init.uniform_(torch.empty(units), e, 1 - e) for _ in range(input_dim)])
                              for _ in range(output_dim)])
        self.trans_matrix = torch.stack([torch.stack([nn.init.xavier_normal_(torch.empty(units, units))
                                                      for _ in range(input_dim)]) for _ in range(output_dim)])

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.u, self.trans_matrix = fn(self.u), fn(self.trans_matrix)  # unregistered, as in the reference
        return self


class MSSM(SNR_trans):
    _DICT, _EXPERT = "mssm", "expert"

    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        BaseModel.__init__(self, linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns,
                           init_std=init_std, device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.num_experts = mc.get("num_experts", 4)
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.expert_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        if self.num_tasks <= 1:
            raise ValueError("num_tasks must be greater than 1")
        if self.num_experts <= 1:
            raise ValueError("num_experts must be greater than 1")
        act, bn = mc.get("dnn_activation", "relu"), mc.get("dnn_use_bn", False)
        drop = mc.get("dnn_dropout", 0)  # experts and towers (reference model/mssm.py:76, :89, :127)
        units, Ne, T = self.expert_dnn_hidden_units, self.num_experts, self.num_tasks
        self.mssm = nn.ModuleDict()
        for i, d in enumerate(units):
            k = self.input_dim if i == 0 else units[i - 1]
            self.mssm[f"expert{i + 1}"] = nn.ModuleList(
                DNN(k, [d], activation=act, dropout_rate=drop, use_bn=bn, init_std=init_std, device=device)
                for _ in range(Ne))
            self.mssm[f"gate{i + 1}"] = gate(Ne, T if i == len(units) - 1 else Ne, d, device=device)
        build_tower_modules(self, units[-1], self.tower_dnn_hidden_units, act, mc.get("l2_reg_dnn", 0), drop, bn,
                            init_std, device)
        self.to(device)
