#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the unmodified reference (alipay/MMLRec @ /root/reference) on CPU, instantiates the five
hot-path models on shrunken vocabularies / hidden widths, and dumps inputs + expected outputs as
compressed .npz fixtures next to this file.  The fixtures are data (inputs, weights, expected
outputs); no reference source travels.  Everything under tests/ and oracle/ is checked against
these files; the GPU box never sees /root/reference.

What is captured per case:
  cfg (json string), vocab, X0..X2 [B,Ftot] f32 (indices carried as floats, basemodel.py:262),
  mask0 [B,D], y0..y2 [B,T], state/<key> (state_dict after re-drawing weights N(0,0.1)),
  frozen/<key> (STAR's unregistered per-domain weights, utils.py:181-191),
  dnn_input, layer/<name> (reference save_layer_output hooks, e.g. mmoe.py:110-118),
  y_pred, y_pred_masked, loss, grad/<key> (dense [V,E] table grads: sparse=False, basemodel.py:122),
  adam1/, adam3/, adagrad3/ (parameters after 1 and 3 reference train steps over batches 0..2,
  basemodel.py:268-313), init_y_pred for the as-constructed (std=1e-4) weights;
  rmsprop1/, rmsprop3/, sgd1/, sgd3/ + their losses for the cases in EXTRA_OPTIMIZER_CASES (the other two optimizers
  basemodel.py:569-584 builds: torch.optim.RMSprop / SGD with the config's lr and torch's defaults).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import copy
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
if not os.path.isdir(REF):
    raise SystemExit("reference tree not present; goldens are generated in the build container only")
sys.path.insert(0, REF)

import numpy as np
import torch

from model.utils import SparseFeat, DenseFeat  # noqa: E402  (reference)
from model.sharedbottom import SharedBottom  # noqa: E402
from model.mmoe import MMOE  # noqa: E402
from model.ple import PLE  # noqa: E402
from model.star import STAR  # noqa: E402
from model.pepnet import PepNet  # noqa: E402
from model.mlp import MLP  # noqa: E402
from model.esmm import ESMM  # noqa: E402
from model.escm import ESCM  # noqa: E402
from model.apg import APG  # noqa: E402
from model.snr_trans import SNR_trans  # noqa: E402
from model.mssm import MSSM  # noqa: E402
from model.aitm import AITM  # noqa: E402
from model.hmoe import HMOE  # noqa: E402
from model.cross_stitch import CrossStitch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
B = 64
# cases that also carry 3-step RMSprop and SGD trajectories (the two optimizers of basemodel.py:569-584 that
# no fixture pinned)
EXTRA_OPTIMIZER_CASES = ("sharedbottom_ml", "mmoe_kuairec", "ple_ijcai", "mmoe_ae30", "mmoe_ae30d", "star_amazon",
                         "pepnet_amazon", "mmoe_ae30_s4", "mmoe_ae30_sat")


def base_config(task_name, model_name, label_columns, emb, optimizer, lr, **model_kw):
    cfg = {
        "data_config": {"data_name": "golden", "label_columns": label_columns, "dense_columns": []},
        "model_config": {
            "task_name": task_name, "model_name": model_name, "task": "binary", "emb": emb,
            "num_experts": 4, "shared_expert_num": 2, "specific_expert_num": 3, "num_levels": 2,
            "expert_dnn_hidden_units": [32, 16], "dnn_hidden_units": [32, 32],
            "bottom_dnn_hidden_units": [32, 16], "gate_dnn_hidden_units": [16],
            "tower_dnn_hidden_units": [16], "l2_reg_linear": 0, "l2_reg_embedding": 0, "l2_reg_dnn": 0,
            "dnn_use_bn": False, "dnn_dropout": 0.0, "dnn_activation": "relu", "use_cka_loss": False,
        },
        "optim_config": {"lr": lr, "optimizer": optimizer,
                         "loss": ["binary_crossentropy"] * len(label_columns), "metrics": ["auc", "acc"],
                         "early_stop": 3},
        "training_config": {"train_batch_size": B, "test_batch_size": B, "epochs": 1},
        "save_config": {"save_layer_output": False},
    }
    cfg["model_config"].update(model_kw)
    return cfg


def make_cases():
    cases = []
    # cfg1: SharedBottom / MovieLens shape (configs_mtl/config_movielens.json, model_name->sharedbottom)
    c = base_config("mtl", "sharedbottom", ["label2", "label3"], 8, "adam", 0.01,
                    task_names=["ctr", "ctcvr"], task_types=["binary", "binary"])
    cases.append(dict(name="sharedbottom_ml", cls=SharedBottom, cfg=c,
                      vocab=[96, 64, 2, 7, 21, 64, 48], nd=0))
    # cfg2: MMoE / KuaiRec shape, E=16 (configs_mtl/config_kuairec.json shape, model_name->mmoe)
    c = base_config("mtl", "mmoe", ["l1", "l2"], 16, "adam", 0.001,
                    task_names=["ctr", "ctcvr"], task_types=["binary", "binary"],
                    expert_dnn_hidden_units=[32, 24], gate_dnn_hidden_units=[16], tower_dnn_hidden_units=[16])
    v = [63, 4, 2, 2, 2, 48, 8, 48, 7, 40, 7, 64, 7, 2, 7, 50, 64, 15, 34, 3, 40, 48, 7, 5, 3, 2, 2, 2,
         2, 2, 64, 80]
    cases.append(dict(name="mmoe_kuairec", cls=MMOE, cfg=c, vocab=v, nd=0))
    # cfg3: PLE / Ijcai shape (configs_mtl/config_ijcai.json, model_name->ple)
    c = base_config("mtl", "ple", ["l1", "l2"], 8, "adam", 0.005,
                    task_names=["ctr", "ctcvr"], task_types=["binary", "binary"],
                    expert_dnn_hidden_units=[32], gate_dnn_hidden_units=[16], tower_dnn_hidden_units=[16])
    cases.append(dict(name="ple_ijcai", cls=PLE, cfg=c, vocab=[9, 3, 96, 64, 64, 80, 48], nd=0))
    # cfg4: MMoE msl / AE-30 shape (configs_msl/config_AE.json, model_name->mmoe), scene = last sparse field
    c = base_config("msl", "mmoe", ["label", "label"], 8, "adam", 0.005,
                    task_types=["binary", "binary"])
    c["data_config"].update({"num_domains": 2, "mask_values": [0, 1], "mask_column": "scene",
                             "scene_feature": "scene"})
    v = [96, 64, 64] + [48] * 4 + [32] * 8 + [24] * 8 + [10] * 6 + [2]
    cases.append(dict(name="mmoe_ae30", cls=MMOE, cfg=c, vocab=v, nd=0, scene_last=True))
    # cfg4 with dense columns (real-AE layout: sparse first then dense, data_utils.py:73-77)
    c = copy.deepcopy(c)
    cases.append(dict(name="mmoe_ae30d", cls=MMOE, cfg=c, vocab=v, nd=7, scene_last=True))
    # every fixture's probabilities sat in 0.45-0.55.  The same model with the last tower layer scaled up:
    # logits of +-4 (probabilities 0.02 .. 0.98), and a SATURATED variant (|logit| beyond 17 for many samples:
    # sigmoid gives exactly 0 / 1 in fp32, the BCE's log is clamped at -100, the gradient through the head is 0 there)
    for nm, sd in (("mmoe_ae30_s4", 2.5), ("mmoe_ae30_sat", 30.0)):
        c2 = copy.deepcopy(c)
        cases.append(dict(name=nm, cls=MMOE, cfg=c2, vocab=v, nd=0, scene_last=True, logit_std=sd))
    # cfg5: STAR + PepNet mtmsl / Amazon shape (configs_mtmsl/config_amazon.json)
    for nm, cls in (("star", STAR), ("pepnet", PepNet)):
        c = base_config("mtmsl", nm, ["label", "label", "label2", "label2"], 8, "adagrad", 0.01,
                        task_types=["binary"] * 4)
        c["data_config"].update({"num_domains": 2, "mask_values": [0, 1], "mask_column": "scene",
                                 "scene_feature": "scene"})
        cases.append(dict(name=f"{nm}_amazon", cls=cls, cfg=c,
                          vocab=[2, 12, 23, 96, 64, 64, 48, 2], nd=0, scene_last=True))
    # SURVEY 8(f) rank 3, first of the remaining model zoo: MLP (model/mlp.py) on the MovieLens shape
    c = base_config("mtl", "mlp", ["label2", "label3"], 8, "adam", 0.01,
                    task_names=["ctr", "ctcvr"], task_types=["binary", "binary"], dnn_hidden_units=[32, 16])
    cases.append(dict(name="mlp_ml", cls=MLP, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=0))
    # ... and in the msl mode (heads masked by domain, model/mlp.py:53-54) with dense columns
    c = base_config("msl", "mlp", ["label", "label"], 8, "adam", 0.005, task_types=["binary", "binary"],
                    dnn_hidden_units=[32, 16])
    c["data_config"].update({"num_domains": 2, "mask_values": [0, 1], "mask_column": "scene",
                             "scene_feature": "scene"})
    cases.append(dict(name="mlp_ae", cls=MLP, cfg=c, vocab=[96, 64, 48, 32, 24, 10, 2], nd=5, scene_last=True))
    # ESMM (model/esmm.py): outputs [ctr, ctr * cvr], one PredictionLayer for both heads
    c = base_config("mtl", "esmm", ["label2", "label3"], 8, "adam", 0.01,
                    task_names=["ctr", "ctcvr"], task_types=["binary", "binary"])
    cases.append(dict(name="esmm_ml", cls=ESMM, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=3))
    # Cross-Stitch (model/cross_stitch.py): msl mode so that the masked heads are covered too
    c = base_config("msl", "cross_stitch", ["label", "label"], 8, "adam", 0.005, task_types=["binary", "binary"],
                    shared_hidden_unit=32, dnn_hidden_units=[32, 16], tower_dnn_hidden_units=[16])
    c["data_config"].update({"num_domains": 2, "mask_values": [0, 1], "mask_column": "scene",
                             "scene_feature": "scene"})
    cases.append(dict(name="cross_stitch_ae", cls=CrossStitch, cfg=c, vocab=[96, 64, 48, 32, 24, 10, 2], nd=4,
                      scene_last=True))
    # HMoE (model/hmoe.py): MMoE + task-level mixture of the tower outputs (others detached)
    c = base_config("mtl", "hmoe", ["l1", "l2"], 8, "adam", 0.005, task_names=["ctr", "ctcvr"],
                    task_types=["binary", "binary"], expert_dnn_hidden_units=[32, 16], gate_dnn_hidden_units=[16],
                    tower_dnn_hidden_units=[16], task_weight_hidden_units=[16])
    cases.append(dict(name="hmoe_ml", cls=HMOE, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=0))
    # AITM (model/aitm.py): two bottoms, two-token attention transfer from task 0 to task 1
    c = base_config("mtl", "aitm", ["l1", "l2"], 8, "adam", 0.005, task_names=["ctr", "ctcvr"],
                    task_types=["binary", "binary"], expert_dnn_hidden_units=[32, 24], tower_dnn_hidden_units=[16])
    cases.append(dict(name="aitm_ml", cls=AITM, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=2))
    # SNR-trans (model/snr_trans.py; the model of the shipped configs_msl/config_IAAC.json), msl mode
    c = base_config("msl", "snr_trans", ["label", "label"], 8, "adam", 0.005, task_types=["binary", "binary"],
                    num_experts=3, expert_dnn_hidden_units=[32, 16], tower_dnn_hidden_units=[16])
    c["data_config"].update({"num_domains": 2, "mask_values": [0, 1], "mask_column": "scene",
                             "scene_feature": "scene"})
    cases.append(dict(name="snr_trans_ae", cls=SNR_trans, cfg=c, vocab=[96, 64, 48, 32, 24, 10, 2], nd=3,
                      scene_last=True))
    # MSSM (model/mssm.py; shipped configs_mtmsl/config_movielens.json and, with BN, configs_mtl/config_census.json)
    c = base_config("mtmsl", "mssm", ["label", "label", "label2", "label2"], 8, "adagrad", 0.01,
                    task_types=["binary"] * 4, num_experts=3, expert_dnn_hidden_units=[32, 16],
                    tower_dnn_hidden_units=[16])
    c["data_config"].update({"num_domains": 2, "mask_values": [0, 1], "mask_column": "scene",
                             "scene_feature": "scene"})
    cases.append(dict(name="mssm_ml", cls=MSSM, cfg=c, vocab=[2, 12, 23, 96, 64, 64, 48, 2], nd=0, scene_last=True))
    # BatchNorm inside DNN (model/utils.py:132-134; the shipped config_census / msl config_amazon set dnn_use_bn)
    c = base_config("mtl", "sharedbottom", ["label2", "label3"], 8, "adam", 0.01,
                    task_names=["ctr", "ctcvr"], task_types=["binary", "binary"], dnn_use_bn=True)
    cases.append(dict(name="sharedbottom_bn", cls=SharedBottom, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=2))
    c = base_config("mtl", "mmoe", ["l1", "l2"], 8, "adagrad", 0.01, task_names=["ctr", "ctcvr"],
                    task_types=["binary", "binary"], dnn_use_bn=True, expert_dnn_hidden_units=[32, 16],
                    gate_dnn_hidden_units=[16], tower_dnn_hidden_units=[16])
    cases.append(dict(name="mmoe_bn", cls=MMOE, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=0))
    # MSSM with BatchNorm in the deeper expert level (the shipped configs_mtl/config_census.json turns dnn_use_bn on)
    c = base_config("mtl", "mssm", ["l1", "l2"], 8, "adam", 0.005, task_names=["ctr", "ctcvr"],
                    task_types=["binary", "binary"], num_experts=3, expert_dnn_hidden_units=[32, 16],
                    tower_dnn_hidden_units=[16], dnn_use_bn=True)
    cases.append(dict(name="mssm_bn", cls=MSSM, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=0))
    # Cross-Stitch with BatchNorm in every block
    c = base_config("mtl", "cross_stitch", ["l1", "l2"], 8, "adagrad", 0.01, task_names=["ctr", "ctcvr"],
                    task_types=["binary", "binary"], shared_hidden_unit=32, dnn_hidden_units=[32, 16],
                    tower_dnn_hidden_units=[16], dnn_use_bn=True)
    cases.append(dict(name="cross_stitch_bn", cls=CrossStitch, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=0))
    # non-zero regulariser (model/basemodel.py:524-540; the code default for an absent l2_reg_embedding key is 1e-5):
    # PLE, so that the dead last-level shared-gate tensors receive their regulariser-only gradient
    c = base_config("mtl", "ple", ["l1", "l2"], 8, "adam", 0.005,
                    task_names=["ctr", "ctcvr"], task_types=["binary", "binary"],
                    expert_dnn_hidden_units=[32], gate_dnn_hidden_units=[16], tower_dnn_hidden_units=[16],
                    l2_reg_dnn=0.05, l2_reg_embedding=0.01)
    cases.append(dict(name="ple_l2", cls=PLE, cfg=c, vocab=[9, 3, 96, 64, 64, 80, 48], nd=0))
    # ESCM (model/escm.py): three outputs for two tasks, the loss branch of basemodel.py:284-292
    c = base_config("mtl", "escm", ["label2", "label3"], 8, "adam", 0.01,
                    task_names=["ctr", "ctcvr"], task_types=["binary", "binary"])
    cases.append(dict(name="escm_ml", cls=ESCM, cfg=c, vocab=[96, 64, 2, 7, 21, 64, 48], nd=2))
    # APG (model/apg.py): per-sample generated [k,k] weights driven by the scene embedding, msl mode
    c = base_config("msl", "apg", ["label", "label"], 8, "adam", 0.005, task_types=["binary", "binary"],
                    dnn_hidden_units=[32, 16])
    c["data_config"].update({"num_domains": 2, "mask_values": [0, 1], "mask_column": "scene",
                             "scene_feature": "scene"})
    cases.append(dict(name="apg_ae", cls=APG, cfg=c, vocab=[96, 64, 48, 32, 24, 10, 2], nd=0, scene_last=True))
    # STAR with its DomainBatchNorm (model/utils.py:553-636): msl mode (heads == domains, the only arrangement in which
    # the reference's mask indexing works), forward / backward WITH a domain mask in training mode, then eval mode on
    # the moved population statistics
    c = base_config("msl", "star", ["label", "label"], 8, "adam", 0.005, task_types=["binary", "binary"],
                    dnn_use_bn=True)
    c["data_config"].update({"num_domains": 2, "mask_values": [0, 1], "mask_column": "scene",
                             "scene_feature": "scene"})
    cases.append(dict(name="star_dbn", cls=STAR, cfg=c, vocab=[2, 12, 23, 96, 64, 64, 48, 2], nd=0, scene_last=True,
                      masked_train=True))
    return cases


def feature_columns(case):
    vocab, emb = case["vocab"], case["cfg"]["model_config"]["emb"]
    names = [f"s{i}" for i in range(len(vocab))]
    if case.get("scene_last"):
        names[-1] = "scene"
    cols = [SparseFeat(n, vocabulary_size=v, embedding_dim=emb) for n, v in zip(names, vocab)]
    dn = [f"d{j}" for j in range(case["nd"])]
    cols += [DenseFeat(n, 1) for n in dn]
    case["cfg"]["data_config"]["dense_columns"] = dn
    return cols, names, dn


def draw_batch(gen, vocab, nd, T, task_name, num_domains):
    cols = []
    for i, v in enumerate(vocab):
        if i % 3 == 0 and v > 4:  # skewed field: many duplicates inside the batch
            u = torch.rand(B, generator=gen)
            idx = torch.floor((v ** u - 1.0)).clamp(0, v - 1).long()
        else:
            idx = torch.randint(0, v, (B,), generator=gen)
        cols.append(idx.float())
    # force the edge rows 0 and V-1 to appear
    cols[0][0] = 0.0
    cols[0][1] = float(vocab[0] - 1)
    X = torch.stack(cols, 1)
    if nd:
        X = torch.cat([X, torch.rand(B, nd, generator=gen)], 1)
    if task_name == "msl":
        lab = (torch.rand(B, 1, generator=gen) < 0.4).float()
        y = lab.repeat(1, num_domains)
    elif task_name == "mtmsl":
        a = (torch.rand(B, 1, generator=gen) < 0.4).float()
        b2 = (torch.rand(B, 1, generator=gen) < 0.3).float()
        y = torch.cat([a.repeat(1, num_domains), b2.repeat(1, num_domains)], 1)
    else:
        y = (torch.rand(B, T, generator=gen) < 0.4).float()
    return X.float(), y.float()


def frozen_star_tensors(model):
    """STAR keeps all-but-the-last domain's specific weights in plain Python lists (utils.py:181-191)."""
    out = {}
    for pfx, mods in (("linears", model.linears), ("final_layers", model.final_layers)):
        for li, m in enumerate(mods):
            for d, (w, b) in enumerate(zip(m.specific_weights, m.specific_biases)):
                out[f"{pfx}.{li}.specific_weights.{d}"] = w.detach().numpy().copy()
                out[f"{pfx}.{li}.specific_biases.{d}"] = b.detach().numpy().copy()
    return out


def ref_loss(model, y_pred, y):
    """The loss expression of the reference's training loop for an unmasked batch (basemodel.py:283-296), including
    its ESCM branch (:284-292), assembled from the reference's own loss functions / counterfact_ipw."""
    lf = model.loss_func
    if model.model_config["model_name"] == "escm":
        loss_0 = lf[0](y_pred[:, 0], y[:, 0], reduction="sum")
        loss_1 = lf[1](y_pred[:, 1], y[:, 1], reduction="sum")
        loss_2 = lf[1](y_pred[:, 2], y[:, 1], reduction="sum")
        loss_1 = model.counterfact_ipw(loss_1, torch.sum(y[:, 0]), y[:, 0].float(), y_pred[:, 0])
        return loss_0 + loss_1 * model.counterfactual_w + loss_2 * model.global_w
    return sum(lf[i](y_pred[:, i], y[:, i], reduction="sum") for i in range(model.num_tasks))


def ref_train_step(model, X, y):
    """The reference's pure step: basemodel.py:268-313 minus logging/metrics."""
    y_pred = model(X, None).squeeze()
    model.optim.zero_grad()
    loss = ref_loss(model, y_pred, y)
    total = loss + model.get_regularization_loss() + model.aux_loss
    total.backward()
    model.optim.step()
    return float(loss.item())


def run_case(case):
    name, cls, cfg = case["name"], case["cls"], case["cfg"]
    cols, names, dn = feature_columns(case)
    torch.manual_seed(0)
    model = cls(cols, device="cpu", config=cfg)
    T = model.num_tasks
    D = cfg["data_config"].get("num_domains", 1)
    task_name = cfg["model_config"]["task_name"]
    gen = torch.Generator().manual_seed(1)
    batches = [draw_batch(gen, case["vocab"], case["nd"], T, task_name, D) for _ in range(3)]
    X0, y0 = batches[0]
    out = {"cfg": np.array(json.dumps(cfg)), "vocab": np.array(case["vocab"], dtype=np.int64),
           "sparse_names": np.array(names), "dense_names": np.array(dn)}
    for i, (X, y) in enumerate(batches):
        out[f"X{i}"] = X.numpy()
        out[f"y{i}"] = y.numpy()
    mask0 = None
    if task_name in ("msl", "mtmsl"):
        scene = X0[:, len(case["vocab"]) - 1]
        mask0 = torch.stack([(scene == v).float() for v in cfg["data_config"]["mask_values"]], 1)
        out["mask0"] = mask0.numpy()

    # as-constructed weights (init_std=1e-4): outputs are ~0.5, recorded for completeness
    model.train()
    with torch.no_grad():
        yp = model(X0, None)
    out["init_y_pred"] = yp.numpy()

    # non-trivial weights: re-draw every weight matrix / table N(0, 0.1); biases keep their init
    g2 = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if ".bn." in k:                                # BatchNorm affine parameters: away from (1, 0)
                p.copy_((1.0 if k.endswith("weight") else 0.0) + torch.randn(p.shape, generator=g2) * 0.2)
            elif k.endswith(".u") and cls is SNR_trans:     # routing parameters must stay inside (0, 1)
                p.copy_(torch.rand(p.shape, generator=g2) * 0.9 + 0.05)
            elif k.endswith(".alpha") and cls in (SNR_trans, MSSM):
                p.copy_(torch.rand(p.shape, generator=g2) + 0.5)
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g2) * 0.1)
            elif k.startswith("out."):
                p.copy_(torch.randn(p.shape, generator=g2) * 0.1)
        if case.get("logit_std"):
            # centre and spread the logits of batch 0: logit_t' = (logit_t - mean_t) * logit_std / std_t, through the last
            # tower layer's weight and the head's bias
            model.eval()
            z = torch.logit(model(X0, None).double())
            model.train()
            named = dict(model.named_parameters())
            for t in range(T):
                f = case["logit_std"] / float(z[:, t].std())
                named[f"tower_dnn_final_layer.{t}.weight"].mul_(f)
                b = named[f"out.{t}.bias"]
                b.copy_((b - float(z[:, t].mean())) * f)
        if cls is STAR:
            for mods in (model.linears, model.final_layers):
                for m in mods:
                    for w in m.specific_weights:
                        w.copy_(1.0 + torch.randn(w.shape, generator=g2) * 0.5)
                    m.shared_weight.copy_(torch.randn(m.shared_weight.shape, generator=g2) * 0.2)
    state0 = copy.deepcopy(model.state_dict())
    for k, v in state0.items():
        out[f"state/{k}"] = v.numpy().copy()
    if cls is STAR:
        for k, v in frozen_star_tensors(model).items():
            out[f"frozen/{k}"] = v
    if cls is MSSM:  # unregistered u vectors and trans_matrix lists (mssm.py:26-36)
        for gname, mod in model.mssm.items():
            if gname.startswith("gate"):
                out[f"frozen/mssm.{gname}.trans_matrix"] = torch.stack(
                    [torch.stack([m.detach() for m in row]) for row in mod.trans_matrix]).numpy().copy()
                out[f"frozen/mssm.{gname}.u"] = torch.stack(
                    [torch.stack([m.detach() for m in row]) for row in mod.u]).numpy().copy()
    if cls is SNR_trans:  # the unregistered trans_matrix lists (snr_trans.py:30-34), stacked [outputs, inputs, d, d]
        for gname, mod in model.trans.items():
            if gname.startswith("gate"):
                out[f"frozen/trans.{gname}.trans_matrix"] = torch.stack(
                    [torch.stack([m.detach() for m in row]) for row in mod.trans_matrix]).numpy().copy()

    # forward in eval mode with the layer-output hooks on
    model.eval()
    model.update_save(True)
    with torch.no_grad():
        yp = model(X0, None)
        if hasattr(model, "layer_output_dict"):
            for k, v in model.layer_output_dict.items():
                out[f"layer/{k}"] = v.numpy().copy()
        out["y_pred"] = yp.numpy().copy()
        if mask0 is not None:
            out["y_pred_masked"] = model(X0, mask0).numpy().copy()
        sl, dl = model.input_from_feature_columns(X0, model.dnn_feature_columns, model.embedding_dict)
        from model.utils import combined_dnn_input
        out["dnn_input"] = combined_dnn_input(sl, dl).numpy().copy()
    model.update_save(False)

    # loss + gradients of one pure step (no optimizer)
    model.train()
    model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], ["auc", "acc"])
    model.zero_grad()
    y_pred = model(X0, None).squeeze()
    loss = ref_loss(model, y_pred, y0)
    (loss + model.get_regularization_loss() + model.aux_loss).backward()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    for k, p in model.named_parameters():
        if p.grad is not None:
            out[f"grad/{k}"] = p.grad.numpy().copy()
        else:
            out[f"nograd/{k}"] = np.array(1)

    if case.get("masked_train"):  # forward(X, mask) in TRAINING mode + its gradients, then eval on the moved statistics
        model.load_state_dict(state0)
        model.train()
        model.zero_grad()
        ypm = model(X0, mask0)
        lossm = sum(model.loss_func[i](ypm[:, i], y0[:, i], reduction="sum") for i in range(T))
        lossm.backward()
        out["mtrain/y_pred"] = ypm.detach().numpy().copy()
        out["mtrain/loss"] = np.array(lossm.item(), dtype=np.float64)
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[f"mtrain/grad/{k}"] = p.grad.numpy().copy()
        out["mtrain/pop_means"] = torch.stack([t.detach() for t in model.domain_bn.pop_means]).numpy().copy()
        out["mtrain/pop_vars"] = torch.stack([t.detach() for t in model.domain_bn.pop_vars]).numpy().copy()
        model.eval()
        with torch.no_grad():
            out["mtrain/y_pred_eval_after"] = model(X0, mask0).numpy().copy()
        model.train()

    # optimizer trajectories: 1 and 3 steps of Adam and Adagrad over batches 0,1,2
    opts = ("adam", "adagrad") + (("rmsprop", "sgd") if name in EXTRA_OPTIMIZER_CASES else ())
    for opt in opts:
        model.load_state_dict(state0)
        model.compile(opt, cfg["optim_config"]["loss"], ["auc", "acc"])
        losses = []
        for i, (X, y) in enumerate(batches):
            losses.append(ref_train_step(model, X, y))
            if (opt in ("adam", "rmsprop", "sgd") and i in (0, 2)) or (opt == "adagrad" and i == 2):
                for k, v in model.state_dict().items():
                    out[f"{opt}{i + 1}/{k}"] = v.numpy().copy()
        out[f"{opt}_losses"] = np.array(losses, dtype=np.float64)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB, loss={out['loss']:.6f}, "
          f"y_pred[0]={out['y_pred'][0]}")


if __name__ == "__main__":
    torch.set_num_threads(1)
    only = set(sys.argv[1:])  # optional: names of the cases to (re)generate
    for case in make_cases():
        if not only or case["name"] in only:
            run_case(case)
