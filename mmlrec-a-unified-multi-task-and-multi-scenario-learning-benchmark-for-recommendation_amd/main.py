"""Driver with the reference's CLI and result-CSV surface (reference main.py:1-184), on the MI355X path.

    python -m mmlrec_amd.main --config configs_msl/config_AE.json --run 1 --model_name mmoe [--seed 0] [--device cuda]

Differences from the reference are its documented defects: the metric functions are imported
(D1), label order is the first-occurrence order of `label_columns` instead of a hash-ordered set (D5), `--is_parallel`
starts a real one-process-per-GPU run (D6) and boolean flags parse `1/true/True` (D15).  Like the reference it loops
over seeds [0, 2, 4, 8] unless --seeds is given.
"""
import argparse
import os
import pickle
import random

import numpy as np
import pandas as pd
import torch

from .model import AITM, APG, ESCM, ESMM, HMOE, MLP, MMOE, MSSM, SNR_trans, CrossStitch, PLE, STAR, PepNet, SharedBottom
from .utils.data_utils import ctrdataset, unserialize

MODELS = {"mmoe": MMOE, "pcg": MMOE, "sharedbottom": SharedBottom, "ple": PLE, "star": STAR, "pepnet": PepNet,
          "mlp": MLP, "esmm": ESMM, "escm": ESCM, "apg": APG, "cross_stitch": CrossStitch, "hmoe": HMOE, "aitm": AITM, "snr_trans": SNR_trans, "mssm": MSSM}


def set_seed(seed, re=True):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_model(model_name, df_columns=None, config=None, device="cuda"):
    name = model_name.lower()
    if name not in MODELS:
        raise NotImplementedError(f"model {model_name!r} is outside the MI355X hot path "
                                  f"(available: {sorted(MODELS)})")
    # construct on the host (same random stream as the reference for a seed), then move once
    return MODELS[name](df_columns, device="cpu", config=config).to(device)


def _bool(v):
    return str(v).lower() in ("1", "true", "yes", "y")


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--run", type=_bool, default=False)
    p.add_argument("--model_name", type=str, default="")
    p.add_argument("--config", type=str, default="")
    p.add_argument("--is_parallel", type=_bool, default=False)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seeds", type=str, default="0,2,4,8", help="comma list (the reference hard-codes 0,2,4,8)")
    p.add_argument("--profile", type=_bool, default=False,
                   help="emit roctx ranges (epoch / train_step / kernel) for `rocprofv3 --marker-trace --kernel-trace`")
    return p


def ordered_labels(label_columns):
    seen, out = set(), []
    for c in label_columns:
        if c not in seen:
            seen.add(c)
            out.append(c)
    return out


def evaluate_predictions(model, config, test, target, test_mask, pred_ans):
    """Per-head LogLoss/AUC (+ total_auc for msl/mtmsl) rounded like the reference (main.py:128-172)."""
    from sklearn.metrics import log_loss, roc_auc_score
    dc, mc = config["data_config"], config["model_config"]
    res = {}
    labels = np.asarray(test[target].values)
    total_auc = None
    for i, _ in enumerate(model.task_types):
        if model.task_name in ("msl", "mtmsl"):
            j = i if model.task_name == "msl" else i % dc.get("num_domains", 0)
            m = test_mask[:, j].astype(bool)
            ml, mp = labels[:, i][m].reshape(-1, 1), pred_ans[:, i][m].reshape(-1, 1)
            ll, auc = round(log_loss(ml, mp), 4), round(roc_auc_score(ml, mp), 4)
            if model.task_name == "msl":
                total_auc = roc_auc_score(labels[:, 0], np.sum(pred_ans, axis=-1))
            else:
                l = dc.get("num_domains", 0)
                yt = labels[:, [0, l]]
                yp = np.stack([pred_ans[:, :l].sum(-1), pred_ans[:, l:].sum(-1)], -1)
                total_auc = roc_auc_score(yt, yp)
        else:
            ll = round(log_loss(labels[:, i], pred_ans[:, i]), 4)
            auc = round(roc_auc_score(labels[:, i], pred_ans[:, i]), 4)
        res[f"log_loss_{i}"], res[f"auc_{i}"] = ll, auc
    if total_auc is not None:
        res["total_auc"] = round(total_auc, 4)
    return res


def run(args):
    """One process of a run.  --is_parallel: a rank that loses a collective exits with code 70 at once
    (parallel.exit_on_collective_error) so the launcher can reap the others instead of waiting for the RCCL timeout."""
    if getattr(args, "is_parallel", False):
        from .parallel import exit_on_collective_error
        return exit_on_collective_error(_run, args)
    return _run(args)


def _run(args):
    dist = None
    if getattr(args, "profile", False):
        from . import profiling
        if not profiling.enable():
            print("--profile: no roctx library found, ranges are not emitted")
    if args.is_parallel:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl")
        args.device = f"cuda:{local}"
    device = torch.device(args.device)
    out = []
    for seed in [int(s) for s in args.seeds.split(",")]:
        print("seed:", seed)
        set_seed(seed)
        config = unserialize(args.config)
        dc, mc, oc, tc, sc = (config[k] for k in ("data_config", "model_config", "optim_config", "training_config",
                                                  "save_config"))
        if args.run:
            mc["model_name"] = args.model_name
        model_name = mc.get("model_name", "sharedbottom")
        target = ordered_labels(dc.get("label_columns", ["label"]))
        train, test, test_mask, train_in, test_in, _, df_columns = ctrdataset(config)
        model = get_model(model_name, df_columns, config, device)
        model.compile(optimizer=oc.get("optimizer", "adagrad"),
                      loss=oc.get("loss", ["binary_crossentropy", "binary_crossentropy"]),
                      metrics=oc.get("metrics", ["auc", "acc"]))
        if dist is not None:
            from . import parallel
            parallel.shard_model(model, dist, tc.get("train_batch_size", 4096),
                                 mode=mc.get("parallel_mode", "row_sharded"))  # additive key (parallel.MODES)
        best = model.fit(train_in, train[target].values, batch_size=tc.get("train_batch_size", 4096),
                         epochs=tc.get("epochs", 10), validation_data=(test_in, test[target].values))
        if sc.get("save_layer_output", False):
            best.update_save()
            pred_ans, layer_out = best.predict(test_in, tc.get("test_batch_size", 4096))
            for key, value in layer_out.items():
                fn = dc.get("layer_output_path", "") + f'{model_name}_l2{mc.get("l2_reg_dnn", "0")}_{key}.pkl'
                with open(fn, "wb") as fh:
                    pickle.dump(value, fh)
        else:
            pred_ans = best.predict(test_in, tc.get("test_batch_size", 4096))
        row = {"type": f'{dc.get("data_name", "")}_{mc.get("task_name", "")}_{model_name}_{seed}'}
        row.update(evaluate_predictions(model, config, test, target, test_mask, pred_ans))
        print(row)
        path = dc.get("test_result_path", "")
        if path and (dist is None or dist.get_rank() == 0):
            pd.DataFrame([row]).to_csv(path, mode="a" if os.path.exists(path) else "w", index=False,
                                       header=not os.path.exists(path))
        out.append(row)
        del model, best
    return out


if __name__ == "__main__":
    run(build_parser().parse_args())
