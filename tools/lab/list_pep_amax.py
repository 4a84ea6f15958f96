"""PepNet step at B = 65 536: what the magnitude launches and the copies in front of the GEMMs touch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mmlrec_amd  # noqa: F401
from mmlrec_amd import workloads as W, engine as E, _lib as L
dev = torch.device("cuda:0")
model, cfg, vocab, dense = W.build_model(sys.argv[1] if len(sys.argv) > 1 else "pepnet_amazon", dev, table_update="auto")
model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], cfg["optim_config"]["metrics"])
model.train()
st = model.train_step_runner(65536, use_graph=True, overlap=False, split_dense=False)
lib = L.load()
calls = [c for part in st.whole.parts if part[0] == "c" for c in part[1]]
for i, c in enumerate(calls[:24]):
    m = c[-1] if isinstance(c[-1], dict) else {}
    name = m.get("kernel", getattr(c[0], "__name__", str(c[0])))
    extra = ""
    if "need" in m:
        extra = " need: " + ", ".join("%s->slot" % (tuple(t.shape),) for t, _ in m["need"])
    if c[0] is lib.mml_copy2d_batch:
        arr, n = c[1]
        extra = " copies: " + ", ".join("%dx%d" % (arr[k].rows, arr[k].cols) for k in range(n))
    if c[0] is lib.mml_copy2d:
        extra = " copy: %dx%d" % (c[1][4], c[1][5])
    print(i, name, m.get("bytes"), extra)
