"""Per-field-class time of the scatter (AE-30, B = 65 536, Zipf): the launch restricted to the fields of one class.
dOut keeps its full [B, 240] layout (the kernel reads only the class's columns via `cols`-independent f*E offsets, so the
restricted launch is given a compacted dOut of the class width)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mmlrec_amd
from mmlrec_amd import ops, workloads as W
dev = torch.device("cuda:0")
cfg, names, vocab, dense = W.workload("mmoe_ae30")
E, B = 8, 65536
dist = sys.argv[1] if len(sys.argv) > 1 else "zipf"
X, _ = W.synth_batch(vocab, 0, B, 2, seed=1, dist=dist)
classes = {"V=1e7": [0], "V=1e6 x2": [1, 2], "V=1e5 x4": [3, 4, 5, 6], "V=1e4 x8": list(range(7, 15)),
           "V=1e3 x8": list(range(15, 23)), "V=100 x6 + V=2": list(range(23, 30)), "all 30": list(range(30))}
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g_ = torch.cuda.CUDAGraph(); st_ = torch.cuda.Stream()
    with torch.cuda.stream(st_):
        with torch.cuda.graph(g_, stream=st_):
            for _ in range(reps): fn()
    g_.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g_.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name, fs in classes.items():
    Xc = X[:, fs].contiguous().to(dev)
    grads = [torch.zeros(vocab[f], E, device=dev) for f in fs]
    d_out = torch.randn(B, len(fs) * E, device=dev)
    t = timed(lambda: ops.scatter_bwd(grads, Xc, list(range(len(fs))), d_out))
    distinct = sum(int(torch.unique(Xc[:, i]).numel()) for i in range(len(fs)))
    per = len(fs) * (4 + 12 * E)
    print(json.dumps({"class": name, "fields": len(fs), "us": round(t * 1e3, 1), "us_per_field": round(t * 1e3 / len(fs), 2),
                      "distinct_rows": distinct, "lookups": B * len(fs), "algorithmic_GBps": round(B * per / (t * 1e-3) / 1e9, 1)}), flush=True)
