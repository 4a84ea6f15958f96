#!/usr/bin/env python3
"""Per-step summed-BCE losses of bench.py's exact step sequence, computed by the ORACLE (oracle/mmlrec_oracle.py, which is
pinned to the reference by tests/golden/*.npz) on the host:

    workload mmoe_ae30, reference initialisation (torch.manual_seed(0), model built on the host), B = 65 536 per step,
    bounded-Zipf batches with seeds 1 + (i mod 4)  (bench.py make_batches at rank 0), dense Adam lr 0.005.

Writes tests/golden/bench_losses_mmoe_ae30.json.  tests/test_fullsize_gpu.py replays the same sequence on the MI355X
through the HIP-graph / two-stream step of bench.py and compares step by step; bench.py itself checks the losses of its
first steps against this file and reports the result in its JSON line ("loss_check").

Run from the repo root (takes ~1 minute on 8 cores):   python tests/golden/make_bench_losses.py [--steps 16]

Round 6: the full-size tests of the OTHER configurations (tests/test_fullsize_gpu.py,
test_bench_secondary_configurations_steps_match_oracle) take their free-running loss trajectory from fixtures too, instead
of running a second oracle beside the GPU (the host oracle at B = 65 536 was 85 % of the GPU suite's wall time on a 16-CPU
host).  Their initialisation is conftest.randomize_he(model, 11) and their batches seeds 1, 2, 3 without rotation:

    for w in mmoe_kuairec ple_ijcai star_amazon pepnet_amazon mmoe_ae30d; do
        python tests/golden/make_bench_losses.py --workload $w --batch 32768 --steps 3 --init he11; done

writes tests/golden/bench_losses_<workload>_he11_b32768.json.
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
    ap.add_argument("--init", default="ref0", choices=["ref0", "he11"],
                    help="ref0: the reference's initialisation under torch.manual_seed(0) (bench.py); he11: "
                         "conftest.randomize_he(model, 11) (the full-size step tests)")
    args = ap.parse_args()
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import workloads as W
    from oracle import mmlrec_oracle as orc
    orc.use_fast(True)
    model, cfg, vocab, dense = W.build_model(args.workload, "cpu")
    frozen = None
    if args.init == "he11":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import randomize_he
        frozen = randomize_he(model, 11) or None
    names = [f.name for f in model._sparse_cols()]
    spec = orc.Spec(cfg, names, vocab, dense)
    params = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    opt = orc.DenseOptimizer(cfg["optim_config"]["optimizer"], cfg["optim_config"]["lr"])
    T = W.num_tasks(cfg)
    nb = 4 if args.init == "ref0" else args.steps
    batches = [W.synth_batch(vocab, len(dense), args.batch, T, seed=1 + i) for i in range(nb)]
    losses = []
    for i in range(args.steps):
        X, y = batches[i % nb]
        losses.append(float(orc.train_step(spec, params, opt, X.numpy(), y.numpy(), frozen)))
        print(i, losses[-1], losses[-1] / args.batch, flush=True)
    out = {"workload": args.workload, "batch": args.batch, "seeds": "1 + (step mod %d)" % nb, "index_dist": "zipf",
           "optimizer": cfg["optim_config"]["optimizer"], "lr": cfg["optim_config"]["lr"],
           "init": "torch.manual_seed(0)" if args.init == "ref0" else "conftest.randomize_he(model, 11)",
           "loss_sum_per_step": losses, "generator": "tests/golden/make_bench_losses.py (oracle/mmlrec_oracle.py)"}
    name = f"bench_losses_{args.workload}.json" if args.init == "ref0" else \
        f"bench_losses_{args.workload}_{args.init}_b{args.batch}.json"
    path = os.path.join(ROOT, "tests", "golden", name)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
