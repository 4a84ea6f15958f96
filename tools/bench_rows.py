#!/usr/bin/env python3
"""Gather / scatter kernels in isolation on the AE-30 tables: B in {65 536, 1 048 576}, Zipf and uniform
indices, HIP-event timed; prints one JSON line per case with the achieved fraction of the 8 TB/s HBM roofline
(algorithmic bytes: gather F*(4+8E), scatter F*(4+12E) per sample).

    python tools/bench_rows.py [--batches 65536,1048576] [--dists zipf,uniform] [--reps 20] [--graph]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="65536,1048576")
    ap.add_argument("--dists", default="zipf,uniform")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--workload", default="mmoe_ae30")
    ap.add_argument("--graph", action="store_true", help="time the launches replayed from ONE HIP graph (device time of "
                                                         "small launches; eager calls are host-bound below ~20 us)")
    args = ap.parse_args()
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import ops, workloads as W
    dev = torch.device("cuda:0")
    cfg, names, vocab, dense = W.workload(args.workload)
    E = cfg["model_config"]["emb"]
    F = len(vocab)
    g = torch.Generator(device="cpu").manual_seed(0)
    tabs = [torch.randn(v, E, generator=g).to(dev) for v in vocab]
    grads = [torch.zeros(v, E, device=dev) for v in vocab]
    cols = list(range(F))
    for B in [int(b) for b in args.batches.split(",")]:
        for dist in args.dists.split(","):
            X, _ = W.synth_batch(vocab, 0, B, 2, seed=1, dist=dist)
            X = X.to(dev)
            d_out = torch.randn(B, F * E, device=dev)
            out = torch.empty(B, F * E, device=dev)

            def timed(fn):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                if args.graph:
                    g_ = torch.cuda.CUDAGraph()
                    st_ = torch.cuda.Stream()
                    with torch.cuda.stream(st_):
                        with torch.cuda.graph(g_, stream=st_):
                            for _ in range(args.reps):
                                fn()
                    g_.replay()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    g_.replay()
                    b.record()
                    torch.cuda.synchronize()
                    return a.elapsed_time(b) / args.reps
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.reps):
                    fn()
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) / args.reps

            t_g = timed(lambda: ops.gather_fwd(tabs, X, cols, out=out))
            t_s = timed(lambda: ops.scatter_bwd(grads, X, cols, d_out))
            for name, t, per in (("gather", t_g, F * (4 + 8 * E)), ("scatter", t_s, F * (4 + 12 * E))):
                gbs = B * per / (t * 1e-3) / 1e9
                print(json.dumps({"kernel": name, "workload": args.workload, "B": B, "dist": dist, "us": round(t * 1e3, 1),
                                  "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 3)}), flush=True)
            # keep the accumulators small in magnitude
            for gr in grads:
                gr.zero_()


if __name__ == "__main__":
    main()
