"""Synthetic workloads of SURVEY.md section 8(d): shapes of the BASELINE.json configs with deterministic inputs.

AE-30 (headline): 30 sparse fields, vocab [1e7, 1e6 x2, 1e5 x4, 1e4 x8, 1e3 x8, 1e2 x6, 2] (last = `scene`,
2 domains, task 'msl'), E=8, MMoE 4 experts [256,128], gates [64], towers [64], Adam lr 0.005
(model_config of configs_msl/config_AE.json:32-65 with model_name -> mmoe)."""
import copy

import torch

AE30_VOCAB = [10_000_000] + [1_000_000] * 2 + [100_000] * 4 + [10_000] * 8 + [1_000] * 8 + [100] * 6 + [2]
KUAIREC32_VOCAB = [63, 4, 2, 2, 2, 800, 8, 500, 7, 200, 7, 1500, 7, 2, 7, 50, 1500, 15, 34, 3, 120, 450, 7, 5, 3, 2, 2,
                   2, 2, 2, 7000, 10000]
IJCAI7_VOCAB = [9, 3, 1_100_000, 1_700, 8_500, 430_000, 5_000]
AMAZON8_VOCAB = [2, 12, 23, 2_000_000, 500_000, 600, 30_000, 2]

_BASE = {
    "data_config": {"data_name": "synthetic", "label_columns": ["label", "label"], "dense_columns": []},
    "model_config": {"task_name": "mtl", "model_name": "mmoe", "task": "binary", "emb": 8, "num_experts": 4,
                     "shared_expert_num": 2, "specific_expert_num": 3, "num_levels": 2,
                     "expert_dnn_hidden_units": [256, 128], "dnn_hidden_units": [256, 128, 64],
                     "bottom_dnn_hidden_units": [256, 128], "gate_dnn_hidden_units": [64],
                     "tower_dnn_hidden_units": [64], "l2_reg_linear": 0, "l2_reg_embedding": 0, "l2_reg_dnn": 0,
                     "dnn_use_bn": False, "dnn_dropout": 0.0, "dnn_activation": "relu", "use_cka_loss": False},
    "optim_config": {"lr": 0.005, "optimizer": "adam", "loss": ["binary_crossentropy", "binary_crossentropy"],
                     "metrics": ["auc", "acc"], "early_stop": 3},
    "training_config": {"train_batch_size": 4096, "val_batch_size": 4096, "test_batch_size": 4096, "epochs": 1},
    "save_config": {"save": False, "save_layer_output": False},
}


def workload(name, vocab_scale=1.0):
    """Returns (config dict, sparse names, vocab list, dense names) for a named synthetic workload."""
    cfg = copy.deepcopy(_BASE)
    mc, dc = cfg["model_config"], cfg["data_config"]
    if name in ("mmoe_ae30", "mmoe_ae30d"):
        vocab = list(AE30_VOCAB)
        mc.update(task_name="msl", model_name="mmoe")
        dc.update(num_domains=2, mask_values=[0, 1], mask_column="scene", scene_feature="scene")
        names = [f"c{i}" for i in range(len(vocab) - 1)] + ["scene"]
        dense = [f"n{j}" for j in range(63)] if name.endswith("d") else []
    elif name == "sharedbottom_ml":
        vocab = [6040, 3706, 2, 7, 21, 3439, 301]
        mc.update(model_name="sharedbottom", task_names=["ctr", "ctcvr"], task_types=["binary", "binary"])
        cfg["optim_config"]["lr"] = 0.01
        names, dense = [f"s{i}" for i in range(7)], []
    elif name == "mmoe_kuairec":
        vocab = list(KUAIREC32_VOCAB)
        mc.update(model_name="mmoe", emb=16, expert_dnn_hidden_units=[512, 256], gate_dnn_hidden_units=[128],
                  tower_dnn_hidden_units=[128], task_names=["ctr", "ctcvr"], task_types=["binary", "binary"])
        cfg["optim_config"]["lr"] = 0.001
        names, dense = [f"s{i}" for i in range(32)], []
    elif name == "ple_ijcai":
        vocab = list(IJCAI7_VOCAB)
        mc.update(model_name="ple", expert_dnn_hidden_units=[128], gate_dnn_hidden_units=[64],
                  tower_dnn_hidden_units=[64], task_names=["ctr", "ctcvr"], task_types=["binary", "binary"])
        names, dense = [f"s{i}" for i in range(7)], []
    elif name in ("star_amazon", "pepnet_amazon", "apg_amazon"):
        vocab = list(AMAZON8_VOCAB)
        mc.update(task_name="mtmsl", model_name=name.split("_")[0], dnn_hidden_units=[128, 128],
                  task_types=["binary"] * 4)
        dc.update(label_columns=["label", "label", "label2", "label2"], num_domains=2, mask_values=[0, 1],
                  mask_column="scene", scene_feature="scene")
        cfg["optim_config"].update(optimizer="adagrad", lr=0.01, loss=["binary_crossentropy"] * 4)
        names, dense = [f"s{i}" for i in range(7)] + ["scene"], []
    elif name in ("mlp_ae30", "esmm_ae30", "escm_ae30", "cross_stitch_ae30", "hmoe_ae30", "aitm_ae30", "snr_trans_ae30",
                  "mssm_ae30"):
        # the wider zoo on the AliExpress-shaped tables, as two-task mtl models
        vocab = list(AE30_VOCAB)
        mc.update(model_name=name[:-5], task_names=["ctr", "ctcvr"], task_types=["binary", "binary"],
                  dnn_hidden_units=[256, 128], shared_hidden_unit=256, task_weight_hidden_units=[64])
        names, dense = [f"c{i}" for i in range(len(vocab))], []
    else:
        raise KeyError(name)
    if vocab_scale != 1.0:
        vocab = [max(2, int(v * vocab_scale)) for v in vocab]
    dc["dense_columns"] = dense
    return cfg, names, vocab, dense


def build_model(name, device, vocab_scale=1.0, seed=0, **model_kw):
    from .model import AITM, APG, ESCM, ESMM, HMOE, MLP, MMOE, MSSM, PLE, STAR, SNR_trans, CrossStitch, DenseFeat, PepNet, SharedBottom, SparseFeat
    cfg, names, vocab, dense = workload(name, vocab_scale)
    cfg["model_config"].update(model_kw)
    emb = cfg["model_config"]["emb"]
    cols = [SparseFeat(n, v, embedding_dim=emb) for n, v in zip(names, vocab)] + [DenseFeat(n, 1) for n in dense]
    cls = {"sharedbottom": SharedBottom, "mmoe": MMOE, "ple": PLE, "star": STAR, "pepnet": PepNet, "mlp": MLP,
           "esmm": ESMM, "escm": ESCM, "apg": APG, "cross_stitch": CrossStitch, "hmoe": HMOE, "aitm": AITM, "snr_trans": SNR_trans,
           "mssm": MSSM}[
        cfg["model_config"]["model_name"]]
    torch.manual_seed(seed)
    # build on the host (same generator stream as the reference for a given seed), then move once
    model = cls(cols, device="cpu", config=cfg)
    model = model.to(device)
    return model, cfg, vocab, dense


def num_tasks(cfg):
    mc, dc = cfg["model_config"], cfg["data_config"]
    if mc["task_name"] == "msl":
        return dc["num_domains"]
    if mc["task_name"] == "mtmsl":
        return len(dc["label_columns"])
    return len(mc.get("task_names", ["ctr", "ctcvr"]))


def synth_batch(vocab, n_dense, B, T, seed, dist="zipf", alpha=1.05):
    """X [B, F+Nd] float32 (indices stored as floats, like the reference feeds its models) and y [B,T].
    zipf: bounded Zipf(alpha) over ranks via the inverse CDF r = ((V^(1-a) - 1) u + 1)^(1/(1-a));
    uniform: worst case for the gather."""
    g = torch.Generator().manual_seed(seed)
    cols = []
    for v in vocab:
        u = torch.rand(B, generator=g, dtype=torch.float64)
        if dist == "zipf":
            r = ((float(v) ** (1.0 - alpha) - 1.0) * u + 1.0) ** (1.0 / (1.0 - alpha))
            idx = (r.floor() - 1).clamp_(0, v - 1)
        elif dist == "uniform":
            idx = (u * v).floor().clamp_(0, v - 1)
        else:
            raise ValueError(dist)
        cols.append(idx.to(torch.float32))
    X = torch.stack(cols, 1)
    if n_dense:
        X = torch.cat([X, torch.rand(B, n_dense, generator=g)], 1)
    lab = (torch.rand(B, 1, generator=g) < 0.5).float()
    y = lab.repeat(1, T)
    return X.contiguous(), y.contiguous()


def algorithmic_per_sample(cfg, vocab, n_dense):
    """Per-sample algorithmic bytes / FLOPs of the step (SURVEY.md 8(d) formulas)."""
    mc = cfg["model_config"]
    E, F = mc["emb"], len(vocab)
    out = {"gather_bytes": F * (4 + 8 * E) + 8 * n_dense, "scatter_bytes": F * (4 + 12 * E)}
    K0 = F * E + n_dense
    T = num_tasks(cfg)
    if mc["model_name"] == "mmoe":
        Ne = mc["num_experts"]

        def mlp(k, units):
            f = 0
            for u in units:
                f += 2 * k * u
                k = u
            return f, k
        fe, H = mlp(K0, mc["expert_dnn_hidden_units"])
        fg, G = mlp(K0, mc["gate_dnn_hidden_units"])
        ft, Ht = mlp(H, mc["tower_dnn_hidden_units"])
        fwd = Ne * fe + T * (fg + 2 * G * Ne) + T * (ft + 2 * Ht) + T * 2 * Ne * H
        out["mlp_flops_fwd"] = fwd
        out["mlp_flops_step"] = 3 * fwd
    return out
