"""STAR (reference model/star.py:8-80): per head i a stack of star layers
h = relu(h @ (W_spec[i] * W_shared) + b_spec[i] + b_shared) and a final [H -> 1] star layer.  The Hadamard product of
the weights is a tiny elementwise kernel per step; the GEMMs then run on the plain [K,N]-layout path.

Reference quirk kept: only the LAST domain's specific tensors are registered parameters, so heads
i < T-1 run on frozen specific weights times the trained shared ones."""
import os

import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .utils import DomainBatchNorm, PredictionLayer, SharedSpecificLinear, activation_code


class STAR(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc = self.model_config
        self.dnn_use_bn = mc.get("dnn_use_bn", False)
        # dnn_use_bn (the shipped configs_msl/config_amazon.json): the reference builds a DomainBatchNorm whose tensors
        # are unregistered lists (no state_dict keys, ones / zeros: no random draws) and applies it only when forward()
        # is given a domain mask (model/star.py:50-51) -- which fit() and predict() never do.  The model
        # therefore trains and predicts exactly like the one without the flag; a direct forward(X, mask) runs it.
        self.dnn_hidden_units = mc.get("dnn_hidden_units", [256, 128])
        self.act_code = activation_code(mc.get("dnn_activation", "relu"))
        use_shared = mc.get("use_shared", True)
        input_dim = self.compute_input_dim(dnn_feature_columns)
        hidden_units = [input_dim] + list(self.dnn_hidden_units)
        print(f"hidden_units:{hidden_units}")
        T = self.num_tasks
        self.linears = nn.ModuleList([SharedSpecificLinear(hidden_units[i], hidden_units[i + 1], T,
                                                           use_shared=use_shared, device=device)
                                      for i in range(len(hidden_units) - 1)])
        if self.dnn_use_bn:
            self.domain_bn = DomainBatchNorm(num_features=hidden_units[1], num_domains=T, device=device)
        self.final_layers = nn.ModuleList([SharedSpecificLinear(hidden_units[-1], 1, T, use_shared=use_shared,
                                                                device=device) for _ in range(T)])
        self.out = nn.ModuleList([PredictionLayer(task) for task in self.task_types])
        self.to(device)

    def _star_params(self, plan, store, prefix, module, d):
        """(effective weight, effective bias) PVals of domain d of one SharedSpecificLinear, plus the ops producing
        them (elementwise product / sum, re-run every step because the factors are trained)."""
        T = self.num_tasks
        if d == T - 1:
            ws, bs = store.pvals[f"{prefix}.specific_weight"], store.pvals[f"{prefix}.specific_bias"]
        else:
            ws = E.PVal(module.specific_weights[d].data, None, f"{prefix}.frozen_w.{d}", needs_grad=False)
            bs = E.PVal(module.specific_biases[d].data, None, f"{prefix}.frozen_b.{d}", needs_grad=False)
            ws.stable = bs.stable = True  # (constants: their magnitude is taken once, at the start of a step)
        wsh, bsh = store.pvals[f"{prefix}.shared_weight"], store.pvals[f"{prefix}.shared_bias"]
        weff = E.PVal(plan.empty(*wsh.data.shape), plan.zeros(*wsh.data.shape), f"{prefix}.weff.{d}")
        # K6: the GEMMs take the derived weight as planes cut straight from its two factors (mml_gemm_planes_cut with W2,
        # mml_star_linear_fwd / _bwd); the fp32 product below is still formed for the heads and as the shape the weight
        # gradient is chained through.  Amazon-8 at 65 536 (round 4, same box, three interleaved pairs): with the tile kernel
        # alone this was level (0.687 vs 0.676 ms: GEMMs -5 us, magnitudes -9 us, the start-of-step cut +16 us); with the
        # weight-stationary kernel, which needs pre-cut planes and now serves these [K, N] layers, 0.6435 vs 0.6604 ms
        # (101.8 vs 99.2 M samples/s).  MMLREC_STAR_PLANES=0: the in-kernel cut of the stored product.
        if os.environ.get("MMLREC_STAR_PLANES", "1") != "0":
            weff.factors = (ws, wsh)
        beff = E.PVal(plan.empty(*bsh.data.shape), plan.zeros(*bsh.data.shape), f"{prefix}.beff.{d}")
        # (collected: ALL derived parameters of the model are produced by one batched launch, see _build_graph)
        self._derived.append((weff, [(ws, wsh)]))
        self._derived.append((beff, [(bs, None), (bsh, None)]))
        return weff, beff

    def _build_graph(self, plan, store, x0):
        T, nl = self.num_tasks, len(self.dnn_hidden_units)
        use_dbn = self.dnn_use_bn and plan.mask is not None
        if use_dbn and plan.mask.shape[1] != T:
            raise NotImplementedError("STAR's DomainBatchNorm indexes the mask with the HEAD number (model/utils.py:617-"
                                      f"622): {T} heads against {plan.mask.shape[1]} mask columns fail in the reference")
        hs = [x0] * T
        per_layer = []
        # K6: the Hadamard products / bias sums of every head and layer (and, on the way back, the sums of
        # their gradients over the heads) are ONE batched launch each way, issued before the first GEMM
        self._derived = []
        eff = {(j, i): self._star_params(plan, store, f"linears.{j}", self.linears[j], i)
               for j in range(nl) for i in range(T)}
        eff_final = [self._star_params(plan, store, f"final_layers.{i}", self.final_layers[i], i) for i in range(T)]
        plan.add(E.SumProdBatchOp(self._derived))
        for j in range(nl):
            probs = []
            for i in range(T):
                weff, beff = eff[(j, i)]
                o = plan.val(self.dnn_hidden_units[j], act=self.act_code, name=f"star.{j}.{i}")
                probs.append(dict(x=hs[i], W=weff, b=beff, out=o, w_kn=1))
            plan.add(E.LinearGroupOp(probs))
            hs = [q["out"] for q in probs]
            if j == 0 and use_dbn:  # model/star.py:50-51: after the activation of the first layer, head by head
                ys = []
                for i in range(T):
                    y = plan.val(self.dnn_hidden_units[j], name=f"star.{j}.{i}.dbn")
                    plan.add(E.DomainBNOp(hs[i], y, self.domain_bn, plan.mask))
                    ys.append(y)
                hs = ys
            per_layer.append(hs)
            plan.layer_outputs[f"star_output_{j}"] = hs
        heads = []
        for i in range(T):
            weff, beff = eff_final[i]
            heads.append(dict(Hin=hs[i], w=weff, bias=store.pvals[f"out.{i}.bias"], bias2=beff))
        return E.HeadOp(heads)
