"""SNR-trans (reference model/snr_trans.py:8-163; the model of the shipped configs_msl/config_IAAC.json): levels of
Ne one-layer expert DNNs connected by sub-network-routing gates -- every (output, input) pair has a [units, units]
transformation scaled by a learned hard-concrete coefficient z(u, alpha) -- then towers and heads.

Reference behaviour kept: the transformations live in plain Python lists (`trans_matrix`, :30-34), so they are not in
state_dict and never optimised: only `alpha` and `u` of a gate learn.  Here a gate is ONE grouped GEMM: the experts
write into column slices of a [B, Ne*units] buffer, `mml_snr_gate_weights_fwd` materialises the scaled blocks as the
[K,N] weight of each output, and the weight gradients of that GEMM give du / dalpha."""
import torch
import torch.nn as nn

from .. import _lib as L
from .. import engine as E
from .basemodel import BaseModel
from .towers import build_tower_modules, emit_towers
from .utils import DNN, blocks_out_act, dnn_options, emit_blocks_into


class gate(nn.Module):  # (lower-case class name of the reference: it shows in nothing but repr)
    def __init__(self, input_dim, output_dim, units, device="cpu", **_unused):
        super().__init__()
        self.input_dim, self.output_dim, self.units = input_dim, output_dim, units
        e = 1e-8
        # same draws in the same order as the reference: alpha ~ U(0,1), u ~ U(e, 1-e), then one xavier-normal block per
        # (output, input) pair
        self.alpha = nn.Parameter(torch.rand((1,)))
        self.u = nn.Parameter(nn.init.uniform_(torch.empty(output_dim, input_dim), e, 1 - e))
        self.trans_matrix = torch.stack([torch.stack([nn.init.xavier_normal_(torch.empty(units, units))
                                                      for _ in range(input_dim)]) for _ in range(output_dim)])

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.trans_matrix = fn(self.trans_matrix)  # invisible to nn.Module (as in the reference): follow the device
        return self


class SNR_trans(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.num_experts = mc.get("num_experts", 4)
        self.input_dim = self.compute_input_dim(dnn_feature_columns)
        self.expert_dnn_hidden_units = mc.get("expert_dnn_hidden_units", [256, 128])
        self.tower_dnn_hidden_units = mc.get("tower_dnn_hidden_units", [64])
        if self.num_tasks <= 1:
            raise ValueError("num_tasks must be greater than 1")
        if self.num_experts <= 1:
            raise ValueError("num_experts must be greater than 1")
        l2 = mc.get("l2_reg_dnn", 0)
        opts = dnn_options(mc, init_std, device)
        units, Ne, T = self.expert_dnn_hidden_units, self.num_experts, self.num_tasks
        self.trans = nn.ModuleDict()
        for i, d in enumerate(units):
            k = self.input_dim if i == 0 else units[i - 1]
            self.trans[f"trans{i + 1}"] = nn.ModuleList(DNN(k, [d], l2_reg=l2, **opts) for _ in range(Ne))
            self.trans[f"gate{i + 1}"] = gate(Ne, T if i == len(units) - 1 else Ne, d, device=device)
        build_tower_modules(self, units[-1], self.tower_dnn_hidden_units, opts["activation"], l2, opts["dropout_rate"],
                            opts["use_bn"], init_std, device)
        self.to(device)

    _DICT, _EXPERT = "trans", "trans"  # ModuleDict attribute and the key stem of its expert lists

    def _build_graph(self, plan, store, x0):
        Ne = self.num_experts
        ins = [x0] * Ne
        mods, stem = getattr(self, self._DICT), f"{self._DICT}.{self._EXPERT}"
        for i, d in enumerate(self.expert_dnn_hidden_units):
            if d % 4:
                raise NotImplementedError("routing widths must be multiples of 4 (16-byte column slices)")
            g = mods[f"gate{i + 1}"]
            No = g.output_dim
            act = blocks_out_act(plan, mods[f"{self._EXPERT}{i + 1}"])
            cat = plan.val(Ne * d, act=act, name=f"snr.{i}.cat")
            parts = [E.Val(cat.buf[:, j * d:(j + 1) * d], act, name=f"snr.{i}.expert.{j}") for j in range(Ne)]
            pfx = f"{stem}{i + 1}"
            emit_blocks_into(plan, store, mods[f"{self._EXPERT}{i + 1}"], [f"{pfx}.{j}" for j in range(Ne)], ins, parts)
            plan.add(E.JoinOp(parts, cat))
            W = plan.empty(No, Ne * d, d)
            dW = plan.zeros(No, Ne * d, d) if plan.training else None
            views = [E.PVal(W[o], dW[o] if dW is not None else None, f"snr.{i}.w.{o}") for o in range(No)]
            u = store.pvals.get(f"{self._DICT}.gate{i + 1}.u", None)  # learned (SNR-trans) or frozen tensor (MSSM)
            plan.add(E.SnrWeightsOp(u if u is not None else g.u, store.pvals[f"{self._DICT}.gate{i + 1}.alpha"],
                                    g.trans_matrix, W, dW, views))
            outs = [plan.val(d, name=f"snr.{i}.out.{o}") for o in range(No)]
            plan.add(E.LinearGroupOp([dict(x=cat, W=views[o], b=None, out=outs[o], w_kn=1) for o in range(No)]))
            ins = outs
        return emit_towers(self, plan, store, ins)
