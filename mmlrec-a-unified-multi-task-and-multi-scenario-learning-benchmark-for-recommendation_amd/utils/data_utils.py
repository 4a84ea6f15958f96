"""Config loading and CSV -> model-input preparation (counterpart of the reference's utils/data_utils.py:14-111).

Runs once on the host with pandas / scikit-learn exactly like the reference; it is not part of the accelerated path,
but it DEFINES the X layout the kernels consume: sparse (label-encoded, contiguous ids) columns first, then the
min-max-scaled dense columns (reference :73-77), with `scene_feature` appended to the sparse list (:49-50).
"""
import json
import os
import pickle

import numpy as np
import pandas as pd

from ..model.utils import DenseFeat, SparseFeat, get_feature_names

__all__ = ["ctrdataset", "get_test_mask", "unserialize", "pd", "np", "os", "json"]


def unserialize(path):
    """json / .npy / pickle by file suffix (reference :102-111; its 'ny' typo for numpy files is kept)."""
    suffix = os.path.basename(path).split(".")[-1]
    if suffix in ("ny", "npy"):
        return np.load(path)
    if suffix == "json":
        with open(path, "r") as fh:
            return json.load(fh)
    with open(path, "rb") as fh:
        return pickle.load(fh)


def get_test_mask(domain_values, mask_values, num_domains):
    """float32 one-hot [N, D]: domain_values == mask_values[d] (reference :96-100)."""
    dv = np.tile(np.asarray(domain_values).reshape((-1, 1)), (1, num_domains))
    mv = np.tile(np.asarray(mask_values).reshape((1, -1)), (len(dv), 1))
    return (dv == mv).astype(np.float32)


def ctrdataset(config):
    """Returns (train_df, test_df, test_mask, train_model_input, test_model_input, linear_cols, dnn_cols)."""
    from sklearn.preprocessing import LabelEncoder, MinMaxScaler
    dc, mc = config["data_config"], config["model_config"]
    train_path, test_path = dc.get("train_dataset_path", ""), dc.get("test_dataset_path", "")
    all_columns = dc.get("all_columns", [])
    feature_columns = list(dc.get("feature_columns", []))
    dense_columns = dc.get("dense_columns", [])
    ignore_columns = dc.get("ignore_columns", [])
    label_columns = dc.get("label_columns", ["label"])
    train_df = pd.read_csv(train_path, usecols=all_columns)
    test_df = pd.read_csv(test_path, usecols=all_columns)
    if "kuairec" in train_path:  # dataset special cases of the reference (:27-39)
        for col in all_columns:
            if "onehot" in col:
                train_df[col] = train_df[col].astype(str)
                test_df[col] = test_df[col].astype(str)
        train_df = train_df[train_df["user_active_degree"] != "0"]
    if "iaac" in train_path:
        train_df["predict_category_property"] = train_df["predict_category_property"].astype(str)
        test_df["predict_category_property"] = test_df["predict_category_property"].astype(str)
        test_df = test_df[:-2]
    train_len = len(train_df)
    df = pd.concat([train_df, test_df])

    task_name = mc.get("task_name", "mtl")
    mask_column = dc.get("mask_column", "")
    scene_feature = dc.get("scene_feature", "")
    emb = mc.get("emb", 4)
    if scene_feature != "" and scene_feature not in feature_columns:
        feature_columns.append(scene_feature)
    sparse_features = feature_columns
    for col in all_columns:
        if col in label_columns + ignore_columns:
            continue
        if "amazon_new" in train_path:
            df[col] = df[col].astype(str)
        if col in dense_columns:
            df[col] = MinMaxScaler().fit_transform(df[[col]]).reshape(-1)
        else:
            df[col] = LabelEncoder().fit_transform(df[col])

    # first-occurrence order of the (possibly duplicated) label names: reindex() then yields every duplicate for
    # msl / mtmsl exactly like the reference's column trick (:65-70)
    new_columns = sparse_features + dense_columns + label_columns
    if task_name in ("msl", "mtmsl") and mask_column != "" and mask_column not in feature_columns:
        new_columns += [mask_column]
    df = df.reindex(columns=new_columns)

    cols = [SparseFeat(feat, vocabulary_size=int(df[feat].max()) + 1, embedding_dim=emb) for feat in sparse_features] \
        + [DenseFeat(feat, 1) for feat in dense_columns]
    feature_names = get_feature_names(cols + cols)
    train, test = df[:train_len], df[train_len:]
    train_input = {name: train[name] for name in feature_names}
    test_input = {name: test[name] for name in feature_names}
    test_mask = None
    if task_name in ("msl", "mtmsl") and mask_column != "":
        if mask_column not in feature_columns:
            train_input[mask_column] = train[mask_column]
            test_input[mask_column] = test[mask_column]
        test_mask = get_test_mask(test[mask_column], dc.get("mask_values", []), dc.get("num_domains", 1))
    return train, test, test_mask, train_input, test_input, cols, cols
