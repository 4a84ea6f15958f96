"""torch-CPU restatement of the MMoE / SharedBottom training step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

What the reference runs on a CPU host is PyTorch (model/basemodel.py:268-313 drives ATen: `embedding`, `addmm`,
`softmax`, `bmm`, `sigmoid`, `binary_cross_entropy(reduction='sum')`, `embedding_dense_backward` through autograd and
`torch.optim.Adam` over DENSE [V, E] table gradients).  The reference tree cannot travel to the GPU box, so this file
restates that step with the same ATen calls in functional form -- parameters in a plain dict under the reference's
`state_dict` names -- and bench.py times it next to the numpy port (`cpu_baseline.torch_cpu`, SURVEY 8(d): "the build's
own torch-CPU restatement that is golden-checked").  Only tests/ and bench.py's cpu_baseline leg import it.

Parity status: PINNED -- tests/test_torch_cpu_baseline.py checks forward, loss, every gradient and three Adam steps
against the reference-made fixtures tests/golden/{mmoe_ae30, mmoe_kuairec, sharedbottom_ml}.npz (bit for bit on the
forward, 1e-6 on the rest: the same ATen kernels in the same order).

Covers `model_name` in {mmoe, sharedbottom} without BatchNorm / dropout (the BASELINE configs 0, 1 and 3).
"""
import json

import torch
import torch.nn.functional as F


class Spec:
    """Feature schema + config (model/utils.py:328-431, model/basemodel.py:93-102)."""

    def __init__(self, cfg, sparse_names, vocab, dense_names=()):
        self.cfg, self.mc, self.dc = cfg, cfg["model_config"], cfg["data_config"]
        self.sparse_names, self.vocab, self.dense_names = list(sparse_names), [int(v) for v in vocab], list(dense_names)
        self.emb = int(self.mc.get("emb", 8))
        self.F, self.Nd = len(self.sparse_names), len(self.dense_names)
        self.model_name = self.mc.get("model_name", "sharedbottom").lower()
        if self.model_name not in ("mmoe", "pcg", "sharedbottom"):
            raise NotImplementedError("torch_cpu covers mmoe / sharedbottom")
        if self.mc.get("dnn_use_bn") or float(self.mc.get("dnn_dropout", 0)):
            raise NotImplementedError("torch_cpu: no BatchNorm / dropout")
        task = self.mc.get("task_name", "mtl")
        self.T = (int(self.dc.get("num_domains", 1)) if task == "msl" else
                  len(self.dc["label_columns"]) if task == "mtmsl" else len(self.mc.get("task_names", ["ctr", "ctcvr"])))

    @staticmethod
    def from_golden(g):
        return Spec(json.loads(str(g["cfg"])), [str(s) for s in g["sparse_names"]], g["vocab"],
                    [str(s) for s in g["dense_names"]])


def params_from_numpy(d, requires_grad=True):
    return {k: torch.from_numpy(v.copy()).requires_grad_(requires_grad) for k, v in d.items()}


def _dnn(p, prefix, x):
    """DNN.forward (model/utils.py:146-161): Linear -> ReLU per layer."""
    i = 0
    while f"{prefix}.linears.{i}.weight" in p:
        x = torch.relu(F.linear(x, p[f"{prefix}.linears.{i}.weight"], p[f"{prefix}.linears.{i}.bias"]))
        i += 1
    return x


def dnn_input(spec, p, X):
    """input_from_feature_columns + combined_dnn_input (model/basemodel.py:461-487, model/utils.py:434-446)."""
    rows = [F.embedding(X[:, f].long(), p[f"embedding_dict.{n}.weight"]) for f, n in enumerate(spec.sparse_names)]
    x = torch.cat(rows, dim=-1)
    if spec.Nd:
        x = torch.cat([x, X[:, spec.F:spec.F + spec.Nd]], dim=-1)
    return x


def forward(spec, p, X, mask=None):
    """MMOE.forward (model/mmoe.py:65-119) / SharedBottom.forward (model/sharedbottom.py:52-86) -> [B, T] probabilities."""
    x = dnn_input(spec, p, X)
    if spec.model_name == "sharedbottom":
        h = _dnn(p, "bottom_dnn", x)
        streams = [h] * spec.T
    else:
        ne = int(spec.mc.get("num_experts", 4))
        experts = torch.stack([_dnn(p, f"expert_dnn.{e}", x) for e in range(ne)], dim=1)  # [B, Ne, H]
        streams = []
        for t in range(spec.T):
            g = _dnn(p, f"gate_dnn.{t}", x) if f"gate_dnn.{t}.linears.0.weight" in p else x
            gate = torch.softmax(F.linear(g, p[f"gate_dnn_final_layer.{t}.weight"]), dim=-1)
            streams.append(torch.matmul(gate.unsqueeze(1), experts).squeeze(1))
    outs = []
    for t in range(spec.T):
        h = _dnn(p, f"tower_dnn.{t}", streams[t]) if f"tower_dnn.{t}.linears.0.weight" in p else streams[t]
        logit = F.linear(h, p[f"tower_dnn_final_layer.{t}.weight"])
        y = torch.sigmoid(logit + p[f"out.{t}.bias"])  # PredictionLayer (model/utils.py:242-248)
        if mask is not None:                            # model/mmoe.py:101-106
            y = y * mask[:, t % mask.shape[1]].view(-1, 1)
        outs.append(y)
    return torch.cat(outs, dim=-1)


def loss_sum(y_pred, y):
    """model/basemodel.py:294-296: sum over tasks of binary_cross_entropy(reduction='sum')."""
    return sum(F.binary_cross_entropy(y_pred[:, t], y[:, t], reduction="sum") for t in range(y.shape[1]))


def make_optimizer(kind, params, lr):
    """model/basemodel.py:569-584: torch.optim with its defaults over EVERY parameter, tables included (dense)."""
    cls = {"adam": torch.optim.Adam, "adagrad": torch.optim.Adagrad, "sgd": torch.optim.SGD,
           "rmsprop": torch.optim.RMSprop}[kind]
    return cls(list(params.values()), lr=lr)


def train_step(spec, p, opt, X, y):
    """The body of the reference's batch loop (model/basemodel.py:268-313) minus logging / metrics."""
    yp = forward(spec, p, X)
    opt.zero_grad()
    loss = loss_sum(yp, y)
    loss.backward()
    opt.step()
    return float(loss.detach())
