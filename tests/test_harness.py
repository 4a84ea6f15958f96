"""Harness rows: config / CSV preparation on the host, and fit()/predict() epoch logs on the MI355X against
logs captured from the unmodified reference (tests/golden/make_harness_golden.py)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR

sys.path.insert(0, GOLDEN_DIR)
import synth_csv  # noqa: E402


def test_ctrdataset_layout_and_vocab(tmp_path):
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd.model.utils import SparseFeat
    from mmlrec_amd.utils.data_utils import ctrdataset, unserialize
    a, b = synth_csv.write_csvs(str(tmp_path))
    cfg = synth_csv.config(a, b, str(tmp_path / "res.csv"))
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    cfg2 = unserialize(str(p))
    assert cfg2 == cfg
    train, test, mask, tin, tein, lin, dnn = ctrdataset(cfg2)
    gold = json.load(open(os.path.join(GOLDEN_DIR, "harness_logs.json")))
    assert [f.vocabulary_size for f in dnn] == gold["sharedbottom_ordered"]["vocab"]  # same LabelEncoder ids
    assert all(isinstance(f, SparseFeat) and f.embedding_dim == 8 for f in dnn)
    assert list(tin.keys()) == synth_csv.COLUMNS[:7]
    assert len(train) == 5120 and len(test) == 1536 and mask is None
    for f in dnn:  # contiguous ids starting at 0
        assert int(min(tin[f.name].min(), tein[f.name].min())) == 0


def test_msl_mask_and_duplicate_labels(tmp_path):
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd.utils.data_utils import ctrdataset, get_test_mask
    a, b = synth_csv.write_csvs(str(tmp_path))
    cfg = synth_csv.config(a, b, "")
    cfg["model_config"]["task_name"] = "msl"
    cfg["data_config"].update(label_columns=["label2", "label2"], num_domains=2, mask_values=[0, 1],
                              mask_column="gender_tag", scene_feature="gender_tag")
    train, test, mask, tin, tein, _, dnn = ctrdataset(cfg)
    assert mask.shape == (1536, 2) and mask.dtype == np.float32
    assert np.array_equal(mask.sum(1), np.ones(1536, np.float32))
    assert train[["label2"]].shape[1] == 2  # duplicated label column (reference data_utils.py:65-70)
    m = get_test_mask([0, 1, 1, 5], [0, 1], 2)
    assert m.tolist() == [[1, 0], [0, 1], [0, 1], [0, 0]]


def test_model_construction_on_cpu_and_loud_failure():
    """Models can be built and inspected without a GPU, but never compute there."""
    import torch
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import workloads as W
    from mmlrec_amd._lib import MMLError
    model, cfg, vocab, dense = W.build_model("mmoe_ae30", "cpu", vocab_scale=1e-4)
    keys = list(model.state_dict().keys())
    assert keys[0] == "embedding_dict.c0.weight" and "gate_dnn_final_layer.1.weight" in keys
    with pytest.raises(MMLError):
        model(torch.zeros(4, 30))
    with pytest.raises(ValueError):
        bad = dict(cfg)
        bad["model_config"] = dict(cfg["model_config"], task_types=["binary"])
        type(model)(model.dnn_feature_columns, config=bad)


@pytest.mark.gpu
@pytest.mark.parametrize("table_update", ["auto", "lazy_exact"])
@pytest.mark.parametrize("model_name", ["sharedbottom", "mmoe"])
@pytest.mark.parametrize("shuffle", [False, True])
def test_fit_epoch_logs_match_reference(tmp_path, model_name, shuffle, table_update, capsys):
    import torch
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import main as M
    from mmlrec_amd.utils.data_utils import ctrdataset
    gold = json.load(open(os.path.join(GOLDEN_DIR, "harness_logs.json")))
    key = f"{model_name}_{'shuffle' if shuffle else 'ordered'}"
    want = gold[key]["epoch_logs"]
    a, b = synth_csv.write_csvs(str(tmp_path))
    cfg = synth_csv.config(a, b, str(tmp_path / "res.csv"), model_name)
    cfg["model_config"]["table_update"] = table_update
    M.set_seed(0)
    train, test, mask, tin, tein, _, dfc = ctrdataset(cfg)
    model = M.get_model(model_name, dfc, cfg, "cuda")
    model.compile("adam", cfg["optim_config"]["loss"], ["auc", "acc"])
    target = ["label2", "label3"]
    best = model.fit(tin, train[target].values, batch_size=256, epochs=2,
                     validation_data=(tein, test[target].values), shuffle=shuffle)
    got = model.history
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for k in ("loss", "auc", "acc", "val_auc", "val_acc"):
            assert abs(g[k] - w[k]) < 2e-3, (key, k, g[k], w[k])  # logs are printed with 4 decimals
    pred = best.predict(tein, 256)
    ref = np.load(os.path.join(GOLDEN_DIR, "harness_pred.npz"))[key]
    assert pred.dtype == np.float64 and pred.shape == ref.shape
    assert np.abs(pred - ref).max() < 5e-3
    # result-CSV row of the driver (reference main.py:128-178)
    row = M.evaluate_predictions(model, cfg, test, target, mask, pred)
    assert set(row) == {"log_loss_0", "auc_0", "log_loss_1", "auc_1"}


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["mmoe_ae30", "pepnet_amazon", "sharedbottom_ml"])
def test_device_batch_metrics_equal_host_loop(workload):
    """fit()'s device-side per-batch auc / acc (mml_auc_segments + torch reductions) == the host loop over sklearn
    metrics it replaces (reference model/basemodel.py:316-337), for the msl, mtmsl and mtl reductions."""
    import torch
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import workloads as W
    model, cfg, vocab, dense = W.build_model(workload, torch.device("cuda:0"), vocab_scale=1e-3)
    model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], ["auc", "acc"])
    T = W.num_tasks(cfg)
    n, bs = 3000, 512
    g = torch.Generator().manual_seed(7)
    pred = torch.rand(n, T, generator=g)
    pred[:, 0] = (pred[:, 0] * 16).round() / 16  # ties
    if model.task_name in ("msl", "mtmsl"):      # one live head per sample, like the masked forward
        pred = pred / T
    y = (torch.rand(n, T, generator=g) < 0.4).float()
    perm = torch.randperm(n, generator=g)
    dev = torch.device("cuda:0")
    got = model._device_batch_metrics(pred.to(dev), y.to(dev), perm.to(dev), bs)
    pe, ye = pred.numpy().astype("float64"), y.numpy()[perm.numpy()]
    steps = (n - 1) // bs + 1
    for name, fn in model.metrics.items():
        vals = [model._metric(fn, ye[s * bs:(s + 1) * bs], pe[s * bs:(s + 1) * bs]) for s in range(steps)]
        want = np.sum(vals) / steps
        assert abs(got[name] - want) < 1e-9, (workload, name, got[name], want)
