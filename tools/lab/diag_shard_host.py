"""Diagnostic: host issue time vs device time of the forced-shard step (world 1), with and without prefetch."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import mmlrec_amd
from mmlrec_amd import workloads as W, parallel
dev = torch.device("cuda:0")
model, cfg, vocab, dense = W.build_model("mmoe_ae30", dev, table_update="dense_exact", use_hip_graph=True)
model.compile("adam", cfg["optim_config"]["loss"], ["auc"]); model.train()
parallel.shard_model(model, dist, 65536, mode="row_sharded")
B = 65536
batches = [tuple(t.to(dev) for t in W.synth_batch(vocab, 0, B, 2, seed=1 + i)) for i in range(4)]
runner = model.train_step_runner(B, use_graph=True)
for ahead in (False, True):
    runner.drop_prefetch()
    def one(i):
        if not runner._has_next:
            runner.plan.X.copy_(batches[i % 4][0]); runner.plan.y.copy_(batches[i % 4][1])
        runner.run()
        if ahead: runner.prefetch(*batches[(i + 1) % 4])
    for i in range(5): one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for i in range(20):
        h0 = time.perf_counter(); one(5 + i); host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"prefetch={ahead}: step {dt / 20 * 1e3:.3f} ms, host issue time {host / 20 * 1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(20): one(30 + i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
dist.destroy_process_group()
