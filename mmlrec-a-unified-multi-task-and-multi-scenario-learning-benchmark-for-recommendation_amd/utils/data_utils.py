"""Config loading and CSV -> model-input preparation (counterpart of the reference's utils/data_utils.py:14-111).

Runs once on the host with pandas / scikit-learn, like the reference; it is not part of the accelerated path, but it
DEFINES the X layout the kernels consume: sparse (label-encoded, contiguous ids) columns first, then the
min-max-scaled dense columns (reference :73-77), with `scene_feature` appended to the sparse list (:49-50).
Organised as a small pipeline (read -> per-dataset fixes -> encode -> schema -> split) with the same observable result
as the reference's single function; tests/test_harness.py pins layout, vocabulary sizes and the msl mask.
"""
import json
import os
import pickle

import numpy as np
import pandas as pd

from ..model.utils import DenseFeat, SparseFeat, get_feature_names

__all__ = ["ctrdataset", "get_test_mask", "unserialize", "pd", "np", "os", "json"]

_LOADERS = {"json": lambda p: json.load(open(p, "r")), "npy": np.load, "ny": np.load}  # 'ny': reference typo kept


def unserialize(path):
    """json / .npy / pickle by file suffix (reference :102-111)."""
    loader = _LOADERS.get(os.path.basename(path).rsplit(".", 1)[-1])
    if loader is not None:
        return loader(path)
    with open(path, "rb") as fh:
        return pickle.load(fh)


def get_test_mask(domain_values, mask_values, num_domains):
    """float32 one-hot [N, D]: column d is (domain_values == mask_values[d]) (reference :96-100)."""
    col = np.asarray(domain_values).reshape(-1, 1)
    row = np.asarray(mask_values).reshape(1, -1)
    if row.shape[1] != num_domains:  # the reference's tiled comparison fails to broadcast in this case too
        raise ValueError(f"mask_values has {row.shape[1]} entries for num_domains = {num_domains}")
    return (col == row).astype(np.float32)


def _read_frames(dc):
    usecols = dc.get("all_columns", [])
    return (pd.read_csv(dc.get("train_dataset_path", ""), usecols=usecols),
            pd.read_csv(dc.get("test_dataset_path", ""), usecols=usecols))


def _dataset_fixes(path, columns, tr, te):
    """The per-dataset special cases the reference hard-codes on the file name (:27-39)."""
    if "kuairec" in path:
        for c in (c for c in columns if "onehot" in c):
            tr[c], te[c] = tr[c].astype(str), te[c].astype(str)
        tr = tr[tr["user_active_degree"] != "0"]
    if "iaac" in path:
        c = "predict_category_property"
        tr[c], te[c] = tr[c].astype(str), te[c].astype(str)
        te = te[:-2]
    return tr, te


def _encode(df, columns, skip, dense, stringify):
    """Contiguous ids per sparse column, [0, 1] scaling per dense column -- fitted on train + test together (:53-62)."""
    from sklearn.preprocessing import LabelEncoder, MinMaxScaler
    for c in columns:
        if c in skip:
            continue
        if stringify:
            df[c] = df[c].astype(str)
        df[c] = MinMaxScaler().fit_transform(df[[c]]).reshape(-1) if c in dense else LabelEncoder().fit_transform(df[c])
    return df


def ctrdataset(config):
    """Returns (train_df, test_df, test_mask, train_model_input, test_model_input, linear_cols, dnn_cols)."""
    dc, mc = config["data_config"], config["model_config"]
    path = dc.get("train_dataset_path", "")
    columns = dc.get("all_columns", [])
    dense = dc.get("dense_columns", [])
    labels = dc.get("label_columns", ["label"])
    sparse = list(dc.get("feature_columns", []))
    scene, mask_col = dc.get("scene_feature", ""), dc.get("mask_column", "")
    scenario_mode = mc.get("task_name", "mtl") in ("msl", "mtmsl") and mask_col != ""
    if scene and scene not in sparse:
        sparse.append(scene)

    tr, te = _dataset_fixes(path, columns, *_read_frames(dc))
    n_train = len(tr)
    df = _encode(pd.concat([tr, te]), columns, set(labels) | set(dc.get("ignore_columns", [])), dense,
                 stringify="amazon_new" in path)
    # column order = X layout.  Label names may repeat (msl / mtmsl list one label once per domain): reindex() then
    # yields every duplicate, which is the reference's trick for building y with one column per head (:65-70)
    extra = [mask_col] if scenario_mode and mask_col not in sparse else []
    df = df.reindex(columns=sparse + dense + labels + extra)

    emb = mc.get("emb", 4)
    schema = [SparseFeat(c, vocabulary_size=int(df[c].max()) + 1, embedding_dim=emb) for c in sparse]
    schema += [DenseFeat(c, 1) for c in dense]
    names = get_feature_names(schema + schema)
    train, test = df[:n_train], df[n_train:]
    inputs = [{n: part[n] for n in names} for part in (train, test)]
    test_mask = None
    if scenario_mode:
        for part, d in zip((train, test), inputs):
            if extra:
                d[mask_col] = part[mask_col]
        test_mask = get_test_mask(test[mask_col], dc.get("mask_values", []), dc.get("num_domains", 1))
    return train, test, test_mask, inputs[0], inputs[1], schema, schema
