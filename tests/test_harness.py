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


@pytest.mark.gpu
def test_main_run_writes_result_csv_and_layer_pickles(tmp_path):
    """End to end through the driver (reference main.py:86-178): JSON config -> ctrdataset -> fit with validation ->
    predict with save_layer_output -> layer-output pickles (main.py:115-122) and the result CSV (main.py:128-178:
    one row per seed, columns type, log_loss_i, auc_i), appended on the second seed."""
    import pickle
    import pandas as pd
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import main as M
    a, b = synth_csv.write_csvs(str(tmp_path), n_train=2048, n_test=512)
    res = tmp_path / "res.csv"
    cfg = synth_csv.config(a, b, str(res), "mmoe")
    cfg["data_config"]["layer_output_path"] = str(tmp_path) + os.sep
    cfg["save_config"]["save_layer_output"] = True
    cfg["training_config"]["epochs"] = 1
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    args = M.build_parser().parse_args(["--config", str(p), "--run", "1", "--model_name", "mmoe", "--seeds", "0,2"])
    rows = M.run(args)
    assert [r["type"] for r in rows] == ["synthml_mtl_mmoe_0", "synthml_mtl_mmoe_2"]
    df = pd.read_csv(res)
    assert list(df.columns) == ["type", "log_loss_0", "auc_0", "log_loss_1", "auc_1"]  # reference row schema
    assert len(df) == 2 and list(df["type"]) == [r["type"] for r in rows]
    for r, (_, d) in zip(rows, df.iterrows()):
        for k in ("log_loss_0", "auc_0", "log_loss_1", "auc_1"):
            assert abs(r[k] - d[k]) < 1e-12 and round(r[k], 4) == r[k]        # rounded to 4 decimals like main.py
        assert 0.5 < r["auc_0"] <= 1.0 and 0.0 < r["log_loss_0"] < 1.0          # the model learned something
    # layer-output pickles: <path><model>_l2<l2_reg_dnn>_<key>.pkl, float64 arrays over the whole test set
    want = {"dnn_input": (512, 56), "expert_outputs": (512, 4, 32), "gate_outputs": (512, 2, 4),
            "mmoe_outputs": (512, 2, 32), "tower_outputs": (512, 2, 16)}
    for key, shape in want.items():
        fn = tmp_path / f"mmoe_l20_{key}.pkl"
        assert fn.exists(), key
        arr = pickle.load(open(fn, "rb"))
        assert arr.shape == shape and arr.dtype == np.float64, (key, arr.shape)
    gates = pickle.load(open(tmp_path / "mmoe_l20_gate_outputs.pkl", "rb"))
    assert np.allclose(gates.sum(-1), 1.0, atol=1e-5)  # softmax rows


@pytest.mark.gpu
def test_state_dict_save_load_round_trip(tmp_path):
    """On-disk format: torch.save(model.state_dict()) carries the reference's key names and shapes
    (Appendix C), loads strict=True into a fresh model -- also after lazy_exact training, whose state_dict() first
    replays the deferred table updates -- and reproduces the predictions bit for bit; a reference-keyed checkpoint
    (the golden state) loads the same way."""
    import torch
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import workloads as W
    from conftest import load_golden
    from test_models_gpu import build, load_state
    dev = torch.device("cuda:0")
    model, cfg, vocab, dense = W.build_model("mmoe_ae30", dev, vocab_scale=1e-4, seed=0, table_update="lazy_exact")
    model.compile("adam", cfg["optim_config"]["loss"], ["auc"])
    model.train()
    T = W.num_tasks(cfg)
    for i in range(3):
        X, y = W.synth_batch(vocab, len(dense), 256, T, seed=30 + i)
        step = model.train_step_runner(256)
        step.plan.X.copy_(X.to(dev))
        step.plan.y.copy_(y.to(dev))
        step.run()
    path = tmp_path / "ckpt.pt"
    torch.save(model.state_dict(), path)
    sd = torch.load(path, map_location="cpu")
    assert list(sd.keys()) == list(model.state_dict().keys())
    assert all(k.startswith(("embedding_dict.", "out.", "expert_dnn.", "gate_dnn.", "gate_dnn_final_layer.",
                             "tower_dnn.", "tower_dnn_final_layer.")) for k in sd)
    fresh, _, _, _ = W.build_model("mmoe_ae30", dev, vocab_scale=1e-4, seed=123)
    fresh.load_state_dict(sd, strict=True)
    Xe, _ = W.synth_batch(vocab, len(dense), 512, T, seed=40)
    model.eval()
    fresh.eval()
    with torch.no_grad():
        a, b = model(Xe.to(dev)), fresh(Xe.to(dev))
    assert torch.equal(a, b)
    # a checkpoint written by the reference (the golden state carries its key names / layouts)
    g = load_golden("ple_ijcai")
    m2, _ = build(g)
    load_state(m2, g)
    p2 = tmp_path / "ref_ckpt.pt"
    torch.save({k[6:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("state/")}, p2)
    m3, _ = build(g)
    m3.load_state_dict(torch.load(p2, map_location="cpu"), strict=True)
    m3.eval()
    with torch.no_grad():
        yp = m3(torch.from_numpy(g["X0"]).cuda()).cpu().numpy()
    assert np.abs(yp - g["y_pred"]).max() < 1e-4 * np.abs(g["y_pred"]).max()


def test_profile_ranges_are_balanced_and_off_by_default():
    """--profile (main.py / bench.py): roctx ranges around epochs, steps and C-ABI calls; host-side only.  Without the
    flag nothing is emitted; with it every push has its pop (checked through a recording stub of the roctx calls)."""
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import profiling
    assert not profiling.enabled
    with profiling.range("not recorded"):
        pass
    events = []

    class Stub:
        @staticmethod
        def roctxRangePushA(b):
            events.append(("push", b.decode()))
            return 0

        @staticmethod
        def roctxRangePop():
            events.append(("pop", None))
            return 0
    old = profiling._lib
    try:
        profiling._lib, profiling.enabled = Stub, True
        with profiling.range("epoch 0"):
            profiling.push("train_step")
            profiling.pop()
    finally:
        profiling._lib, profiling.enabled = old, False
    assert events == [("push", "epoch 0"), ("push", "train_step"), ("pop", None), ("pop", None)]
    from mmlrec_amd.main import build_parser
    assert build_parser().parse_args(["--profile", "1"]).profile is True
