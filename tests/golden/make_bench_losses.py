#!/usr/bin/env python3
"""Per-step summed-BCE losses of bench.py's exact step sequence, computed by the ORACLE (oracle/mmlrec_oracle.py, which is
pinned to the reference by tests/golden/*.npz) on the host:

    workload mmoe_ae30, reference initialisation (torch.manual_seed(0), model built on the host), B = 65 536 per step,
    bounded-Zipf batches with seeds 1 + (i mod 4)  (bench.py make_batches at rank 0), dense Adam lr 0.005.

Writes tests/golden/bench_losses_mmoe_ae30.json.  tests/test_fullsize_gpu.py replays the same sequence on the MI355X
through the HIP-graph / two-stream step of bench.py and compares step by step; bench.py itself checks the losses of its
first steps against this file and reports the result in its JSON line ("loss_check").

Run from the repo root (takes ~1 minute on 8 cores):   python tests/golden/make_bench_losses.py [--steps 16]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--workload", default="mmoe_ae30")
    args = ap.parse_args()
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import workloads as W
    from oracle import mmlrec_oracle as orc
    orc.use_fast(True)
    model, cfg, vocab, dense = W.build_model(args.workload, "cpu")
    names = [f.name for f in model._sparse_cols()]
    spec = orc.Spec(cfg, names, vocab, dense)
    params = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    opt = orc.DenseOptimizer(cfg["optim_config"]["optimizer"], cfg["optim_config"]["lr"])
    T = W.num_tasks(cfg)
    batches = [W.synth_batch(vocab, len(dense), args.batch, T, seed=1 + i) for i in range(4)]
    losses = []
    for i in range(args.steps):
        X, y = batches[i % 4]
        losses.append(float(orc.train_step(spec, params, opt, X.numpy(), y.numpy())))
        print(i, losses[-1], losses[-1] / args.batch, flush=True)
    out = {"workload": args.workload, "batch": args.batch, "seeds": "1 + (step mod 4)", "index_dist": "zipf",
           "optimizer": cfg["optim_config"]["optimizer"], "lr": cfg["optim_config"]["lr"], "init": "torch.manual_seed(0)",
           "loss_sum_per_step": losses, "generator": "tests/golden/make_bench_losses.py (oracle/mmlrec_oracle.py)"}
    path = os.path.join(ROOT, "tests", "golden", f"bench_losses_{args.workload}.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
