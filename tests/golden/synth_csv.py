"""Deterministic MovieLens-shaped CSV pair shared by the harness-golden generator and the harness test."""
import numpy as np
import pandas as pd

COLUMNS = ["user_tag", "movie_tag", "gender_tag", "age_tag", "occupation_tag", "zip_tag", "genres_tag", "label2",
           "label3"]
VOCAB = [400, 300, 2, 7, 21, 120, 40]


def make_frames(n_train=5120, n_test=1536, seed=11):
    rng = np.random.default_rng(seed)
    n = n_train + n_test
    feats = [np.minimum((v ** rng.random(n)).astype(np.int64), v - 1) for v in VOCAB]
    # raw ids are NOT contiguous (LabelEncoder has work to do)
    raw = [f * 3 + 5 for f in feats]
    w = [rng.standard_normal(v) for v in VOCAB]
    score = sum(wi[f] for wi, f in zip(w, feats)) / np.sqrt(len(VOCAB))
    l2 = (score + 0.5 * rng.standard_normal(n) > 0.3).astype(np.int64)
    l3 = (score + 0.8 * rng.standard_normal(n) > 0.9).astype(np.int64)
    df = pd.DataFrame({c: v for c, v in zip(COLUMNS, raw + [l2, l3])})
    return df[:n_train].reset_index(drop=True), df[n_train:].reset_index(drop=True)


def write_csvs(dirname, **kw):
    import os
    tr, te = make_frames(**kw)
    a, b = os.path.join(dirname, "synth_train.csv"), os.path.join(dirname, "synth_test.csv")
    tr.to_csv(a, index=False)
    te.to_csv(b, index=False)
    return a, b


def config(train_path, test_path, result_path, model_name="sharedbottom"):
    return {
        "data_config": {"data_name": "synthml", "train_dataset_path": train_path, "test_dataset_path": test_path,
                        "test_result_path": result_path, "layer_output_path": "", "all_columns": COLUMNS,
                        "feature_columns": COLUMNS[:7], "dense_columns": [], "ignore_columns": [],
                        "label_columns": ["label2", "label3"], "sample": "random"},
        "model_config": {"task_name": "mtl", "model_name": model_name, "task": "binary", "emb": 8, "num_experts": 4,
                         "shared_expert_num": 2, "specific_expert_num": 3, "num_levels": 2,
                         "expert_dnn_hidden_units": [64, 32], "dnn_hidden_units": [64, 32],
                         "bottom_dnn_hidden_units": [64, 32], "gate_dnn_hidden_units": [16],
                         "tower_dnn_hidden_units": [16], "l2_reg_linear": 0, "l2_reg_embedding": 0, "l2_reg_dnn": 0,
                         "dnn_use_bn": False, "dnn_dropout": 0.0, "dnn_activation": "relu", "use_cka_loss": False},
        "optim_config": {"lr": 0.01, "optimizer": "adam", "loss": ["binary_crossentropy", "binary_crossentropy"],
                         "metrics": ["auc", "acc"], "early_stop": 3},
        "training_config": {"train_batch_size": 256, "val_batch_size": 256, "test_batch_size": 256, "epochs": 2},
        "save_config": {"save": False, "save_path": "", "save_layer_output": False},
    }
