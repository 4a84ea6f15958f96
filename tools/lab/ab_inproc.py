"""Two configurations of the training step in ONE process, replayed in alternating blocks: box drift (+-3 % over a
minute on this pool) cancels in the paired differences.  Only knobs read at plan / runner construction can differ
(Python-level MMLREC_* variables; the library's own statics are read once per process).
usage: python tools/lab/ab_inproc.py "<env A>" "<env B>" [--workload W] [--batch B] [--blocks 12] [--steps 25]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("env_a")
ap.add_argument("env_b")
ap.add_argument("--workload", default="mmoe_ae30")
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--blocks", type=int, default=12)
ap.add_argument("--steps", type=int, default=25)
ap.add_argument("--table-update", default="dense_exact")
ap.add_argument("--shared", action="store_true", help="ONE model and plan (same buffers: no placement bias), two TrainStep "
                "call lists over it -- for knobs that only change the lists (fork placement, optimizer launches)")
args = ap.parse_args()

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import workloads as W  # noqa: E402

dev = torch.device("cuda", 0)


def build(env):
    saved = {}
    for kv in env.split():
        k, v = kv.split("=", 1)
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    model, cfg, vocab, dense = W.build_model(args.workload, dev, table_update=args.table_update, use_hip_graph=True)
    model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], cfg["optim_config"]["metrics"])
    model.train()
    T = W.num_tasks(cfg)
    batches = []
    for i in range(4):
        X, y = W.synth_batch(vocab, len(dense), args.batch, T, seed=1 + i, dist="zipf")
        batches.append((X.to(dev), y.to(dev)))
    runner = model.train_step_runner(args.batch, use_graph=True, overlap=False, split_dense=False)
    for i in range(4):  # eager step, capture, two replays
        runner.load(*batches[i % 4])
        runner.run()
    torch.cuda.synchronize()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return model, runner, batches


def block(runner, batches, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        runner.load(*batches[i % 4])
        runner.run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def build_shared(envs):
    from mmlrec_amd.trainer import TrainStep
    model, cfg, vocab, dense = W.build_model(args.workload, dev, table_update=args.table_update, use_hip_graph=True)
    model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], cfg["optim_config"]["metrics"])
    model.train()
    T = W.num_tasks(cfg)
    batches = []
    for i in range(4):
        X, y = W.synth_batch(vocab, len(dense), args.batch, T, seed=1 + i, dist="zipf")
        batches.append((X.to(dev), y.to(dev)))
    out = []
    for env in envs:
        saved = {}
        for kv in env.split():
            k, v = kv.split("=", 1)
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        runner = TrainStep(model, args.batch, True, None, False, False)
        for i in range(4):
            runner.load(*batches[i % 4])
            runner.run()
        torch.cuda.synchronize()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        out.append((model, runner, batches))
    return out


if args.shared:
    A, B = build_shared([args.env_a, args.env_b])
else:
    A = build(args.env_a)
    B = build(args.env_b)
import gc  # noqa: E402
gc.collect()
gc.disable()
da, db = [], []
for r in range(args.blocks):
    da.append(block(A[1], A[2], args.steps))
    db.append(block(B[1], B[2], args.steps))
diffs = sorted(b - a for a, b in zip(da, db))
med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
print("A %-60s median %.4f ms  (min %.4f)" % (args.env_a, med(da), min(da)))
print("B %-60s median %.4f ms  (min %.4f)" % (args.env_b, med(db), min(db)))
print("paired B - A: median %+.1f us, quartiles %+.1f / %+.1f us over %d blocks of %d steps (%s, B = %d)"
      % (med(diffs) * 1e3, diffs[len(diffs) // 4] * 1e3, diffs[(3 * len(diffs)) // 4] * 1e3, args.blocks, args.steps,
         args.workload, args.batch))
