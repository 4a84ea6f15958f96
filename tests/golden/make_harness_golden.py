#!/usr/bin/env python3
"""Harness golden (build container only): runs the UNMODIFIED reference's ctrdataset + compile + fit + predict on the
deterministic synthetic CSVs and stores the per-epoch logs and the final predictions (SURVEY 8(c) 'Python harness rows').
Two runs: shuffle=False and shuffle=True (pins the DataLoader permutation draws)."""
import contextlib
import io
import json
import os
import re
import sys
import tempfile

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth_csv  # noqa: E402
import main as ref_main  # noqa: E402  (reference main.py: set_seed, get_model)
from utils.data_utils import ctrdataset  # noqa: E402  (reference)

LOG = re.compile(r"(\w+):\s+(-?[\d.]+(?:e-?\d+)?|nan)")


def run(model_name, shuffle):
    with tempfile.TemporaryDirectory() as d:
        a, b = synth_csv.write_csvs(d)
        cfg = synth_csv.config(a, b, os.path.join(d, "res.csv"), model_name)
        ref_main.set_seed(0)
        ref_main.device = torch.device("cpu")
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            train, test, test_mask, tin, tein, _, dfc = ctrdataset(cfg)
            model = ref_main.get_model(model_name, dfc, cfg)
            model.compile("adam", cfg["optim_config"]["loss"], ["auc", "acc"])
            target = ["label2", "label3"]
            best = model.fit(tin, train[target].values, batch_size=256, epochs=2,
                             validation_data=(tein, test[target].values), shuffle=shuffle)
            pred = best.predict(tein, 256)
        logs = []
        for line in buf.getvalue().splitlines():
            if " - loss:" in line:
                logs.append({k: float(v) for k, v in LOG.findall(line)})
        vocab = [int(f.vocabulary_size) for f in dfc]
        return logs, pred, vocab


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    arrays = {}
    for model_name in ("sharedbottom", "mmoe"):
        for shuffle in (False, True):
            logs, pred, vocab = run(model_name, shuffle)
            key = f"{model_name}_{'shuffle' if shuffle else 'ordered'}"
            out[key] = {"epoch_logs": logs, "vocab": vocab}
            arrays[key] = pred.astype(np.float32)
            print(key, logs)
    json.dump(out, open(os.path.join(HERE, "harness_logs.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "harness_pred.npz"), **arrays)
