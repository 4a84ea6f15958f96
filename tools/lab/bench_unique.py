import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import mmlrec_amd
from mmlrec_amd import ops, workloads as W, _lib as L
dev = torch.device("cuda:0")
vocab = W.AE30_VOCAB; F = len(vocab); E = 8
seen = [torch.zeros((v + 31) // 32, dtype=torch.int32, device=dev) for v in vocab]
marks = torch.zeros(ops.marks_bytes(vocab), dtype=torch.uint8, device=dev)
rowbase = np.concatenate([[0], np.cumsum(vocab)]).tolist()
for B in (64, 4096, 65536):
    X, _ = W.synth_batch(vocab, 0, B, 2, seed=1)
    X = X.to(dev)
    touched = torch.zeros(B * F, dtype=torch.int32, device=dev); count = torch.zeros(1, dtype=torch.int32, device=dev)
    def run():
        ops.index_unique(vocab, list(range(F)), E, X, seen, rowbase, touched, count, marks=marks)
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    print("B", B, "index_unique us", a.elapsed_time(b) / 20 * 1e3, "count", int(count.item()))
