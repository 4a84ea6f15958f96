"""APG (reference model/apg.py:9-193, as main.py builds it: use_uv_shared=True, use_mf_p=False, mf_k=4): every layer is
low-rank, in -> k -> k -> out with k = ceil(min(in, out) / 4); the outer maps are shared parameters, the middle [k,k]
weight and its bias are GENERATED per sample from the (detached) scene embedding by two Linear layers.

The reference materialises B x k x k weights and a batched matmul.  Here the generated layer is one ordinary GEMM on
the feature row [o1 (x) s | o1 | s] against a re-laid-out copy of the generator parameters (csrc/apg.hip,
engine.ApgFeatOp / ApgWeightsOp), so the whole model runs on the grouped-GEMM kernels.

Reference quirk kept: `scene_index` is a column offset used as a list position (apg.py:135, :152), valid because every
sparse feature occupies one column of X."""
import math

import torch
import torch.nn as nn

from .. import engine as E
from .basemodel import BaseModel
from .utils import DNN, PredictionLayer, activation_code


class APGLayer(nn.Module):
    """Parameter container with the reference's creation order (apg.py:32-59): the two generator DNNs first, then the
    shared n->k and k->m maps (xavier-uniform weights, zero biases)."""

    def __init__(self, input_dim, output_dim, scene_emb_dim, activation="relu", generate_activation=None,
                 inner_activation=None, use_uv_shared=True, mf_k=16, use_mf_p=True, mf_p=4, device="cpu"):
        super().__init__()
        if not use_uv_shared or use_mf_p or inner_activation is not None or generate_activation is not None:
            raise NotImplementedError("APGLayer options outside the configuration the reference's APG model builds")
        self.input_dim, self.output_dim = input_dim, output_dim
        self.act_code = activation_code(activation)
        min_dim = min(int(input_dim), int(output_dim))
        self.p_dim = math.ceil(float(min_dim) / float(mf_p))
        self.k_dim = k = math.ceil(float(min_dim) / float(mf_k))
        self.specific_weight_kk = DNN(inputs_dim=scene_emb_dim, hidden_units=[k * k], activation=None, device="cpu")
        self.specific_bias_kk = DNN(inputs_dim=scene_emb_dim, hidden_units=[k], activation=None, device="cpu")
        # drawn on the HOST generator (like a CPU construction of the reference), then moved
        self.shared_weight_nk = nn.Parameter(nn.init.xavier_uniform_(torch.empty((input_dim, k))))
        self.shared_bias_nk = nn.Parameter(torch.zeros((k,)))
        self.shared_weight_km = nn.Parameter(nn.init.xavier_uniform_(torch.empty((k, output_dim))))
        self.shared_bias_km = nn.Parameter(torch.zeros((output_dim,)))
        self.to(device)


class APG(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc, dc = self.model_config, self.data_config
        self.dnn_use_bn = mc.get("dnn_use_bn", False)
        self.dnn_hidden_units = mc.get("dnn_hidden_units", [256, 128])
        scene_emb_dim = mc.get("emb", 8)
        scene_feature = dc.get("scene_feature", "")
        if scene_feature == "":
            raise NotImplementedError("APG needs data_config.scene_feature (the reference fails in forward without it)")
        self.scene_index = self.feature_index[scene_feature]
        input_dim = self.compute_input_dim(dnn_feature_columns)
        dims = [input_dim] + list(self.dnn_hidden_units)
        self.apg_layers = nn.ModuleList([APGLayer(dims[i], dims[i + 1], scene_emb_dim,
                                                  activation=mc.get("dnn_activation", "relu"), use_uv_shared=True,
                                                  use_mf_p=False, mf_k=4, mf_p=4, device=device)
                                         for i in range(len(self.dnn_hidden_units))])
        self.final_layer = nn.ModuleList([nn.Linear(self.dnn_hidden_units[-1], 1, bias=False)
                                          for _ in range(self.num_tasks)])
        self.out = nn.ModuleList([PredictionLayer(task) for task in self.task_types])
        self.to(device)

    def _build_graph(self, plan, store, x0):
        Edim = self.embedding_size
        p = self.scene_index[0]  # column offset used as list position (reference apg.py:135, :152)
        scene = x0.buf[:, p * Edim:(p + 1) * Edim]  # .detach(): nothing below writes a gradient into these columns
        h = x0
        outs = []
        for i, layer in enumerate(self.apg_layers):
            pfx, k = f"apg_layers.{i}", layer.k_dim
            o1 = plan.val(k, name=f"{pfx}.nk", pad_k=True)
            plan.add(E.LinearGroupOp([dict(x=h, W=store.pvals[f"{pfx}.shared_weight_nk"],
                                           b=store.pvals[f"{pfx}.shared_bias_nk"], out=o1, w_kn=1)]))
            kf = (k * Edim + k + Edim + 15) // 16 * 16
            z = plan.val(kf, name=f"{pfx}.z")
            plan.add(E.ApgFeatOp(o1, scene, z, k, Edim))
            wcat = E.PVal(plan.zeros(kf, k), plan.zeros(kf, k), f"{pfx}.wcat")
            plan.add(E.ApgWeightsOp(store.pvals[f"{pfx}.specific_weight_kk.linears.0.weight"],
                                    store.pvals[f"{pfx}.specific_weight_kk.linears.0.bias"],
                                    store.pvals[f"{pfx}.specific_bias_kk.linears.0.weight"], wcat, k, Edim))
            o2 = plan.val(k, name=f"{pfx}.kk", pad_k=True)
            plan.add(E.LinearGroupOp([dict(x=z, W=wcat, b=store.pvals[f"{pfx}.specific_bias_kk.linears.0.bias"], out=o2,
                                           w_kn=1)]))
            o3 = plan.val(layer.output_dim, act=layer.act_code, name=f"{pfx}.km", pad_k=True)
            plan.add(E.LinearGroupOp([dict(x=o2, W=store.pvals[f"{pfx}.shared_weight_km"],
                                           b=store.pvals[f"{pfx}.shared_bias_km"], out=o3, w_kn=1)]))
            h = o3
            outs.append(o3)
            plan.layer_outputs[f"apg_output_{i}"] = o3
        heads = [dict(Hin=h, w=store.pvals[f"final_layer.{t}.weight"], bias=store.pvals[f"out.{t}.bias"])
                 for t in range(self.num_tasks)]
        return E.HeadOp(heads)
