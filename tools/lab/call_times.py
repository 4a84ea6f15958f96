#!/usr/bin/env python3
"""Device time of every C-ABI call of a training step, each replayed 20x from its own HIP graph (eager launches from
Python are host-bound below ~20 us), next to the bytes / FLOPs the engine books for it.
usage: call_times.py [workload] [batch] [table_update]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402


def main():
    os.environ.setdefault("MMLREC_INNER_FORK", "0")  # (every call of the step in ONE list: no fork / join entries)
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import workloads as W, engine as E
    wl = sys.argv[1] if len(sys.argv) > 1 else "mmoe_ae30"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    tu = sys.argv[3] if len(sys.argv) > 3 else "lazy_exact"
    dev = torch.device("cuda:0")
    model, cfg, vocab, dense = W.build_model(wl, dev, table_update=tu)
    model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], ["auc"])
    model.train()
    T = W.num_tasks(cfg)
    X, y = W.synth_batch(vocab, len(dense), B, T, seed=1)
    st = model.train_step_runner(B)
    st.plan.X.copy_(X.to(dev))
    st.plan.y.copy_(y.to(dev))
    for _ in range(3):
        st.run()
    torch.cuda.synchronize()
    segs = [st.whole] if st.whole is not None else [st.pre, st.early, st.front, st.sideq, st.tail]
    reps = 20
    tot = 0.0
    for seg in segs:
        for kind, item, _g in seg.parts:
            if kind != "c":
                continue
            for c in item:
                meta = c[2] if len(c) > 2 and isinstance(c[2], dict) else {}
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                        for _ in range(reps):
                            E.Plan._run([c])
                g.replay()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                g.replay()
                b.record()
                torch.cuda.synchronize()
                us = a.elapsed_time(b) / reps * 1e3
                tot += us
                by, fl = meta.get("bytes", 0.0), meta.get("flops", 0.0)
                extra = f"{by / us / 1e6:7.2f} TB/s" if by else (f"{fl / us / 1e6:7.1f} TFLOP/s" if fl else "")
                print(f"{us:8.1f} us  {meta.get('kernel', c[0].__name__):42s} {extra}")
    print(f"sum {tot:.1f} us  ({wl}, B = {B}, {tu})")


if __name__ == "__main__":
    main()
