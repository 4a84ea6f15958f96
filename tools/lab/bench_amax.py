#!/usr/bin/env python3
"""Stand-alone time of the magnitude pass over the gathered input (x0 [65536, 240]) for several grid caps
(MMLREC_AMAX_BLOCKS is read once per process: run once per value)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mmlrec_amd  # noqa: F401,E402
from mmlrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(65536, 240, device=dev)
big = torch.randn(64 * 1024 * 1024, device=dev)  # evicts the caches between launches
slots = ops.amax_slots(1, dev)
for warm in (True, False):
    ts = []
    for _ in range(20):
        if not warm:
            big.mul_(1.0001)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.amax_batch([(x, slots[0])])
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print("MMLREC_AMAX_BLOCKS=%s  %s caches: median %.1f us, min %.1f" % (os.environ.get("MMLREC_AMAX_BLOCKS"), "warm" if warm else "cold", ts[len(ts) // 2], ts[0]))
assert ops.amax_value(slots[0]) == float(x.abs().max())
