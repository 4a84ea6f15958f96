#!/usr/bin/env python3
"""Headline benchmark: training-step samples/sec of MMoE on AliExpress-shaped synthetic batches (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process IS a rank; started plainly
(`python bench.py --gpus 8`) it launches its own N ranks as a child `python -m torch.distributed.run ...` BEFORE any GPU
call, relays the child's JSON line and exits with its code (the reference's counterpart, main.py:81-83, is a dead stub).

A step = one pass of the hot path over one batch that is already resident in HBM:
fused gather -> expert/gate/tower MLPs (fp32 MFMA) -> heads + summed BCE -> backward (dgrad/wgrad GEMMs, gate/head
backward, sparse row-scatter) -> optimizer (reference-exact dense Adam over every table row + MLP parameters).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed in a
second, instrumented pass over the same batches) and `cpu_baseline` (the oracle timed on the host, N=1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP runtime defaults of the package (mmlrec_amd/__init__.py: _runtime_defaults), before anything can initialise HIP;
# child ranks inherit them
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="samples per GPU per step")
    ap.add_argument("--workload", default="mmoe_ae30")
    ap.add_argument("--dist", default="zipf", choices=["zipf", "uniform"])
    ap.add_argument("--table-update", default="dense_exact", choices=["dense_exact", "sparse_rows", "lazy_exact", "auto"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile", action="store_true",
                    help="emit roctx ranges (train_step / HIP-graph segment / kernel) for rocprofv3 --marker-trace")
    ap.add_argument("--parallel-mode", default="row_sharded", choices=["row_sharded", "replicated", "table_wise"],
                    help="how the tables are spread over the ranks when --gpus > 1 (mmlrec_amd/parallel.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="1 (default since round 5: the whole step is ONE HIP graph on one stream; kernels do not co-run, "
                         "so a rocprofv3 --kernel-trace reports stand-alone kernel durations) or 2 (forked tail: table "
                         "scatter + table optimizer beside the weight-gradient GEMMs; tools/lab/ab_streams.sh holds the A/B)")
    ap.add_argument("--serial", action="store_true", help="same as --streams 1 (kept for old command lines)")
    ap.add_argument("--no-lazy", action="store_true", help="skip the secondary lazy_exact measurement")
    ap.add_argument("--lazy-epoch-steps", type=int, default=500,
                    help="lazy_exact: also time an epoch of this many steps + its flush (0 = skip)")
    ap.add_argument("--split-dense", action="store_true",
                    help="dense table update as untouched rows beside the forward + touched rows after the scatter, "
                         "instead of ONE launch after the scatter that skips the gradient read of unmarked rows")
    ap.add_argument("--no-split-dense", action="store_true", help="(default since the marked single launch; kept for "
                                                                  "old command lines)")
    ap.add_argument("--alt-batch", type=int, default=4096, help="also report this per-GPU batch (0 = skip)")
    ap.add_argument("--cpu-batch", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-big-steps", type=int, default=3, help="cpu_baseline: steps at the headline batch (0 = skip)")
    ap.add_argument("--scatter-mode", default="atomic", choices=["atomic", "deterministic"],
                    help="deterministic: table gradients as order-independent integer fixed-point sums (bitwise repeatable)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (BASELINE.json's other configurations, a few steps each)")
    ap.add_argument("--no-loss-check", action="store_true")
    return ap.parse_args()


def visible_gpu_count():
    """GPUs this process would see, WITHOUT a HIP call (torch.cuda.device_count() can fall through to hipGetDeviceCount
    on builds without amdsmi, which initialises the runtime in the parent of the ranks: ADVICE r3): the KFD topology
    nodes with SIMDs, cut down by HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES.  None when sysfs says nothing (the ranks then
    fail with their own message)."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(ln.split(None, 1) for ln in f.read().splitlines() if " " in ln)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process (no
    GPU call happens in this process: the device count comes from sysfs), relay its output -- the JSON line of rank 0 stays the last stdout line --
    and return its exit code."""
    import socket
    import subprocess
    share = os.environ.get("MMLREC_BENCH_SHARE_GPU") == "1"
    ndev = visible_gpu_count()  # (from sysfs: nothing in THIS process may initialise the HIP runtime)
    if ndev is not None and ndev < args.gpus and not share:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for attempt in range(3):
        # (the port is probed free here and bound by the launcher a moment later: if something else on the host took it in
        #  between -- EADDRINUSE, seen once in ~30 runs of the test suite -- the launcher dies at once and the launch is
        #  repeated on another port)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        t_launch = time.time()
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)  # (stderr streams: the preflight line, progress)
        # (a launcher that cannot bind its port dies within seconds, before any rank exists)
        if r.returncode == 0 or time.time() - t_launch > 30.0 or '"metric"' in r.stdout:
            break
    lines = r.stdout.splitlines()
    js = [ln for ln in lines if ln.startswith("{") and '"metric"' in ln]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln)
    if js:
        print(js[-1], flush=True)
    if r.returncode == 0 and not js:
        print("bench.py: the ranks exited without printing the result line", file=sys.stderr)
        return 3
    return r.returncode


def loss_tolerance(step):
    """Relative tolerance of the per-step loss against the oracle-made fixture (tests/golden/bench_losses_*.json), or
    None where the comparison says nothing.  Steps 0-11 stay near 2 ln 2 per sample: 1e-4, the contract's tolerance
    (measured <= 3e-6).  Afterwards the model memorises the four rotated batches (the loss falls by ~3 % per step) and a
    free-running trajectory amplifies fp32-level differences -- Adam turns the sign of a noise-level gradient into a
    full lr step, a ReLU flips for one sample: 5e-3 up to step 27 (measured up to 9e-4 at step 24; 1.3e-2 at step 34,
    where the loss has halved), not compared beyond.  The step-wise parity tests carry the proof; this catches a run
    that went wrong."""
    return 1e-4 if step < 12 else (5e-3 if step < 28 else None)


def loss_fixture(args):
    if args.no_loss_check or args.dist != "zipf" or args.table_update not in ("dense_exact", "auto"):
        return None
    path = os.path.join(ROOT, "tests", "golden", f"bench_losses_{args.workload}.json")
    if not os.path.exists(path):
        return None
    fx = json.load(open(path))
    return fx if fx.get("batch") == args.batch else None


def dist_setup(n):
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # MMLREC_BENCH_SHARE_GPU=1: every rank on GPU 0 over gloo (parallel.Comm stages through the host) -- a smoke
        # test of the N > 1 control flow on a one-GPU box, NOT a measurement (RCCL refuses two ranks on one device)
        if os.environ.get("MMLREC_BENCH_SHARE_GPU") == "1":
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            return rank, local, world, dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        return rank, local, world, dist
    torch.cuda.set_device(local)
    return rank, local, world, None


def barrier(dist):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def timed_steps(runner, batches, steps, warmup, dist, flush=None, warm_losses=None):
    nb = len(batches)

    # row-sharded tables: batch i + 1 is handed over while step i is in flight (trainer.TrainStep.prefetch), so its
    # routing -- incl. the count exchange and the host read of the split sizes -- is off the step's critical path
    ahead = getattr(getattr(runner, "par", None), "mode", None) == "row_sharded"

    def one(i):
        if not runner._has_next:
            runner.load(*batches[i % nb])  # (one launch for both tensors)
        runner.run()
        if ahead:
            runner.prefetch(*batches[(i + 1) % nb])

    for i in range(warmup):
        one(i)
        if warm_losses is not None:  # untimed steps: a host read per step costs nothing that is measured
            warm_losses.append(float(runner.plan.loss.item()))
    # Keep the host out of the timed region: a cyclic-GC pass that happens to run here destroys HIP graphs / events of
    # an earlier phase (tens of ms of hipFree + synchronisation; seen as a one-off 60 ms gap in the kernel trace)
    import gc
    gc.collect()
    gc.disable()
    barrier(dist)
    comm = getattr(getattr(runner, "par", None), "comm", None)
    snap0 = comm.stats_snapshot() if comm is not None else None
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    if comm is not None:  # what this rank issued inside the timed region, per step (host-side counters of parallel.Comm)
        runner.collectives_per_step = comm.stats_delta(snap0, comm.stats_snapshot(), per=steps)
    t_flush = 0.0
    if flush is not None:  # lazy_exact: the deferred zero-gradient updates of every untouched row are paid HERE
        torch.cuda.synchronize()
        tf = time.perf_counter()
        flush()
        torch.cuda.synchronize()
        t_flush = time.perf_counter() - tf
    barrier(dist)
    dt = time.perf_counter() - t0
    gc.enable()
    runner.drop_prefetch()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if flush is not None:
        return dt, t_flush
    return dt


def kernel_breakdown(runner, batches, steps):
    """Second, instrumented pass: every C-ABI call bracketed by HIP events on the launch stream."""
    from mmlrec_amd import engine as E
    acc = {}
    # ONE call list per step and no garbage collection while it is issued: the bracket of the first call after a
    # synchronisation starts on an idle stream, so any host hiccup (a generation-2 collection over the thousands of event
    # objects of a large plan) lands in it -- with one list per phase that was the table scatter, first of `bwd_tail`
    # (snr_trans_ae30 once reported 4 ms for an 87 us kernel)
    calls = (list(runner.plan.fwd) + list(runner.plan.head_train) + list(runner.plan.bwd) + list(runner.plan.bwd_tail) +
             list(getattr(runner.plan, "head_side", [])) + list(runner.plan.bwd_side) + list(runner.opt_calls))
    was = gc.isenabled()
    gc.disable()
    try:
        for i in range(steps):
            X, y = batches[i % len(batches)]
            runner.plan.X.copy_(X)
            runner.plan.y.copy_(y)
            E.Plan.run_timed(calls, acc)
    finally:
        if was:
            gc.enable()
    return acc


def roofline_of(acc):
    """Dominant kernel by total time; compute-bound GEMMs are priced against the fp32 MFMA peak
    (157.3 TFLOP/s, MI355X_MICROARCH.md), streaming kernels against the 8 TB/s HBM3E spec."""
    # (collectives of the table-sharded path are timed in `acc` too; the roofline is a statement about a HIP kernel)
    name = max((k for k in acc if not k.startswith(("all_to_all", "all_reduce", "all_gather", "row_sharded_"))),
               key=lambda k: acc[k]["ms"])
    e = acc[name]
    avg_ms = e["ms"] / e["launches"]
    note = None
    if name.startswith("gemm_ws_kernel") and e.get("hbm_bytes", 0.0) > 0:
        # the weight-stationary launches are streams (K, N <= 256: 4 (K + N) bytes per sample and problem against 6 K N
        # plane-product FLOP): priced against HBM by their compulsory bytes (every operand read once, every output written
        # once), per template variant (round 6: VERDICT r5 weak 4 -- the MFMA fraction of an aggregate label said nothing)
        achieved = e["hbm_bytes"] / e["launches"] / (avg_ms * 1e-3) / 1e9
        peak, unit, bound = 8000.0, "GB/s", "hbm"
        note = "compulsory HBM bytes of the launch (operands once, outputs once) / launch time"
        e = dict(e, flops=0.0, bytes=e["hbm_bytes"])
    elif e["flops"] > 0:
        achieved = e["flops"] / e["launches"] / (avg_ms * 1e-3) / 1e12
        unit, bound = "TFLOP/s", "mfma"
        planes = 0
        import re
        m = re.match(r"gemm_pipe_kernel<\w+, \w+, \d+, \d+, (\d+)", name)
        if m and int(m.group(1)) in (1, 2, 3):
            planes = int(m.group(1))
        elif name.startswith("g16_"):  # bf16-storage kernels (csrc/gemm16.hip): one bf16 MFMA per product block
            planes = 1
        elif name.startswith(("gemm_ws_kernel", "gemm_panel_kernel", "gemm_nt_kernel", "gemm_os_kernel")):
            # the weight-stationary / activation-stationary kernels issue the same three f16 MFMAs per product block as
            # gemm_pipe_kernel<..., 2, ...> (csrc/gemm_ws.hip, csrc/gemm_panel.hip), one per block in the bf16 forms
            planes = 2
        if planes:  # fp32 emulated on the 16-bit MFMA pipe: 3 (two fp16 planes) or 6 (three bf16 planes) MFMAs per block
            per = {1: 1, 2: 3, 3: 6}[planes]
            peak = 2500.0 / per
            ins = "v_mfma_f32_32x32x16_f16" if planes == 2 else "v_mfma_f32_32x32x16_bf16"
            note = (f"algorithmic fp32 FLOP/s; the kernel issues {per} {ins} per 32x32x16 product "
                    f"block, so peak = dense 16-bit MFMA peak (2.5 PFLOP/s) / {per}; frac = MFMA pipe utilisation. "
                    f"The fp32 MFMA peak (v_mfma_f32_32x32x2_f32) is 157.3 TFLOP/s.")
        else:
            peak = 157.3
    else:
        achieved = e["bytes"] / e["launches"] / (avg_ms * 1e-3) / 1e9
        peak, unit, bound = 8000.0, "GB/s", "hbm"
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes of this same
    # command; tools/pmc_traffic.sh -> profiles/traffic.json).  bench.py cannot run the profiler on itself.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath)).get(name)
            traffic = round(float(t["hbm_bytes_per_launch"])) if t else None
        except Exception:
            traffic = None
    out = {"kernel": name, "bound": bound, "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": unit,
           "frac": round(achieved / peak, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 5),
           "launches_per_step": None,
           "algorithmic_per_launch": round((e["flops"] if e["flops"] > 0 else e["bytes"]) / e["launches"])}
    if note:
        out["note"] = note
    return out


def cpu_baseline(args):
    """The oracle (kind 'port': numpy restatement, BLAS threads for the GEMMs) on a bounded sample of the SAME
    workload: `cpu_steps` full steps at batch `cpu_batch` incl. the reference's dense Adam over every table row."""
    from oracle import mmlrec_oracle as orc
    import numpy as np
    from mmlrec_amd import workloads as W
    cfg, names, vocab, dense = W.workload(args.workload)
    spec = orc.Spec(cfg, names, vocab, dense)
    fast = orc.use_fast(True)  # C/OpenMP gather / scatter / dense-Adam loops of the oracle, when built
    rng = np.random.default_rng(0)
    params = orc.random_params(spec, rng)
    opt = orc.DenseOptimizer(cfg["optim_config"]["optimizer"], cfg["optim_config"]["lr"])
    T = W.num_tasks(cfg)
    batches = [W.synth_batch(vocab, len(dense), args.cpu_batch, T, seed=100 + i, dist=args.dist) for i in range(2)]
    batches = [(x.numpy(), y.numpy()) for x, y in batches]
    orc.train_step(spec, params, opt, *batches[0])  # warm-up (allocates optimizer state)
    t0 = time.perf_counter()
    for i in range(args.cpu_steps):
        orc.train_step(spec, params, opt, *batches[i % 2])
    dt = time.perf_counter() - t0
    cores = os.cpu_count() or 1
    # The same at the HEADLINE batch (the port was only timed at the reference's 4 096), with the dense
    # table Adam timed by itself: its p, g, m, v reads and p, m, v writes over every table row are 28 bytes per
    # parameter -- the achieved GB/s says how far from the HOST's own memory roofline the port runs.
    big = None
    if args.batch != args.cpu_batch and args.cpu_big_steps > 0:
        Xb, yb = W.synth_batch(vocab, len(dense), args.batch, T, seed=102, dist=args.dist)
        Xb, yb = Xb.numpy(), yb.numpy()
        n_tab = sum(int(np.prod(v.shape)) for k, v in params.items() if k.startswith("embedding_dict."))
        t_fb = t_opt = 0.0
        for i in range(args.cpu_big_steps):
            ta = time.perf_counter()
            loss, grads, _ = orc.loss_and_grads(spec, params, Xb, yb)
            tb = time.perf_counter()
            opt.step(params, grads)
            tc = time.perf_counter()
            t_fb += tb - ta
            t_opt += tc - tb
        nb = args.cpu_big_steps
        big = {"batch": args.batch, "steps": nb, "value": round(args.batch * nb / (t_fb + t_opt), 1), "unit": "samples/s",
               "ms_per_step": round((t_fb + t_opt) / nb * 1e3, 1), "forward_backward_ms": round(t_fb / nb * 1e3, 1),
               "dense_optimizer_ms": round(t_opt / nb * 1e3, 1),
               "dense_optimizer_GB_per_s": round(28.0 * n_tab * nb / t_opt / 1e9, 1),
               "note": "dense_optimizer_GB_per_s = 28 B x table parameters / the optimizer's time (MLP tensors included "
                       "in the time, negligible): the host's DRAM streams a few hundred GB/s"}
    torch_leg = None
    try:
        torch_leg = torch_cpu_baseline(args, cfg, names, vocab, dense, params, T)
    except Exception as e:  # (mmoe / sharedbottom only; never lose the line to the host side)
        torch_leg = {"failed": repr(e)}
    numpy_port = {"value": round(args.cpu_batch * args.cpu_steps / dt, 1), "unit": "samples/s", "cores": cores,
                  "kind": "port", "batch": args.cpu_batch, "at_headline_batch": big,
            "note": "the oracle (numpy / BLAS + C/OpenMP restatement of the reference step, pinned to the reference by "
                    "tests/golden), NOT the reference's PyTorch path: that one measured 8.1 k samples/s on 8 cores in "
                    "the build container (BASELINE.md) and cannot travel to the GPU box",
            "sample": f"{args.cpu_steps} full train steps (fwd+BCE+bwd+dense {cfg['optim_config']['optimizer']}) of "
                      f"{args.workload} at batch {args.cpu_batch}, {args.dist} indices; oracle/mmlrec_oracle.py with "
                      f"multi-threaded BLAS GEMMs and " + ("C/OpenMP" if fast else "numpy (single-thread)") +
                      " gather/scatter/dense-Adam loops"}
    if torch_leg and "value" in torch_leg:
        # the headline CPU figure is the PyTorch one: that is what the reference runs on a CPU host
        # (model/basemodel.py:268-313); the numpy oracle stays beside it
        out = dict(torch_leg)
        out["numpy_port"] = numpy_port
        return out
    numpy_port["torch_cpu"] = torch_leg
    return numpy_port


def effective_cpus():
    """CPUs this process may really use: the affinity mask cut down by the cgroup CPU quota (cpu.max), so that a
    container limited to a few cores does not start one thread per core of the host."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def torch_cpu_baseline(args, cfg, names, vocab, dense, params_np, T, budget_s=45.0):
    """oracle/torch_cpu.py -- the reference's own ATen calls in functional form, golden-checked against reference-made
    fixtures (tests/test_torch_cpu_baseline.py) -- on the host cores: full steps incl. autograd's dense [V, E] table
    gradients and torch.optim over every table row, at the reference's batch (cpu_batch) and at the headline batch.
    Bounded: the thread count is the best of a few candidates (one step each -- with one thread per hardware thread of a
    256-thread host a step of this model took 42 s, round 5), and every leg stops at its share of `budget_s`."""
    from oracle import torch_cpu as tc
    from mmlrec_amd import workloads as W
    n0 = torch.get_num_threads()
    t_start = time.perf_counter()
    ncpu = effective_cpus()
    try:
        spec = tc.Spec(cfg, names, vocab, dense)
        p = tc.params_from_numpy(params_np)
        opt = tc.make_optimizer(cfg["optim_config"]["optimizer"], p, cfg["optim_config"]["lr"])
        b0 = [W.synth_batch(vocab, len(dense), args.cpu_batch, T, seed=100 + i, dist=args.dist) for i in range(2)]
        # thread count: every candidate takes one step at the reference's batch (the first also warms the optimizer state up)
        cands, tried = [], {}
        for c in (8, 16, 32, 64, 128, ncpu):  # ascending: the search stops where more threads stop paying
            if c <= ncpu and c not in cands:
                cands.append(c)
        if not cands:
            cands = [ncpu]
        for c in cands:
            torch.set_num_threads(c)
            t0 = time.perf_counter()
            tc.train_step(spec, p, opt, *b0[0])
            if not tried:   # (the very first step allocates the moments: time a second one)
                t0 = time.perf_counter()
                tc.train_step(spec, p, opt, *b0[1])
            tried[c] = time.perf_counter() - t0
            if tried[c] > 1.3 * min(tried.values()) or time.perf_counter() - t_start > 0.4 * budget_s:
                break
        threads = min(tried, key=tried.get)
        torch.set_num_threads(threads)
        runs = []
        for B, steps, share in ((args.cpu_batch, max(3, min(args.cpu_steps, 8)), 0.3), (args.batch, args.cpu_big_steps, 0.3)):
            if steps <= 0 or (runs and B == runs[0]["batch"]):
                continue
            bs = b0 if B == args.cpu_batch else [W.synth_batch(vocab, len(dense), B, T, seed=102, dist=args.dist)]
            if B != args.cpu_batch:
                tc.train_step(spec, p, opt, *bs[0])  # (first touch of this batch size)
            done, t0 = 0, time.perf_counter()
            while done < steps and (done == 0 or time.perf_counter() - t0 < share * budget_s):
                tc.train_step(spec, p, opt, *bs[done % len(bs)])
                done += 1
            dt = time.perf_counter() - t0
            runs.append({"batch": B, "steps": done, "value": round(B * done / dt, 1), "unit": "samples/s",
                         "ms_per_step": round(dt / done * 1e3, 1)})
    finally:
        torch.set_num_threads(n0)
    return {"value": runs[0]["value"], "unit": "samples/s", "cores": threads, "kind": "port",
            "implementation": "torch-CPU restatement of the reference step (oracle/torch_cpu.py: F.embedding / F.linear / "
                              "softmax / matmul / sigmoid / binary_cross_entropy(sum), autograd's dense table gradients, "
                              f"torch.optim.{cfg['optim_config']['optimizer']} over every table row), torch {torch.__version__}, "
                              f"torch.set_num_threads({threads})",
            "host_cpus": ncpu, "one_step_s_by_threads": {str(k): round(v, 3) for k, v in tried.items()},
            "batch": runs[0]["batch"], "at_headline_batch": runs[1] if len(runs) > 1 else None,
            "sample": f"{runs[0]['steps']} full train steps of {args.workload} at batch {runs[0]['batch']}"
                      + (f" and {runs[1]['steps']} at batch {runs[1]['batch']}" if len(runs) > 1 else "")
                      + f", {args.dist} indices, after warm-up steps; thread count = the fastest of "
                        f"{sorted(tried)} (one step each)",
            "note": "pinned to the unmodified reference by tests/test_torch_cpu_baseline.py (forward bit for bit, "
                    "gradients and Adam / Adagrad steps to 1e-6); the reference itself measured 8.1 k samples/s on the 8 "
                    "cores of the build container (BASELINE.md) and cannot travel to the GPU box"}


def secondary_configs(args, dev):
    """BASELINE.json's other configurations as driver-visible numbers: configs[1] MMoE on
    KuaiRec-shaped batches (fp32-equivalent GEMMs, and the bf16-operand mode the config names), configs[2] PLE on
    Ijcai-shaped batches, configs[4] STAR and PepNet on Amazon-shaped batches -- a short pure-step measurement each at
    B = 65 536 and at the reference's B = 4 096, with the dominant kernel's roofline fraction.  Same protocol as the
    headline (resident batches, HIP-graph replay, barrier + synchronize around the timed steps), fewer steps."""
    import gc
    from mmlrec_amd import _lib
    from mmlrec_amd import workloads as W
    lib = _lib.load()
    mode0 = lib.mml_gemm_get_mode()
    out = []
    plan = [("configs[1] MMoE / KuaiRec-32, E=16", "mmoe_kuairec", None),
            ("configs[1] MMoE / KuaiRec-32, E=16, bf16 (GEMM mode 1, opt-in, outside the 1e-4 contract: bf16 operands, activations "
             "and gradients between GEMMs stored as bf16, fp32 accumulation / master weights / optimizer)", "mmoe_kuairec", 1),
            ("configs[2] PLE / Ijcai-7, 2 levels x (3 specific + 2 shared) experts", "ple_ijcai", None),
            ("configs[4] STAR / Amazon-8", "star_amazon", None),
            ("configs[4] PepNet / Amazon-8", "pepnet_amazon", None)]
    for label, wl, mode in plan:
        try:
            lib.mml_gemm_set_mode(mode if mode is not None else mode0)
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):  # (STAR / MLP print their layer widths like the reference does)
                model, cfg, vocab, dense = W.build_model(wl, dev, table_update="auto", use_hip_graph=not args.no_graph)
            model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], cfg["optim_config"]["metrics"])
            model.train()
            T = W.num_tasks(cfg)
            entry = {"config": label, "workload": describe_workload(wl, cfg, vocab, dense),
                     "table_update": model.optimizer().table_update,
                     "dtype": "bf16 storage + operands, f32 accumulate" if mode == 1 else "f32", "runs": []}
            for B, steps in ((65536, 20), (4096, 40)):
                batches = []
                for i in range(2):
                    X, y = W.synth_batch(vocab, len(dense), B, T, seed=1 + i, dist=args.dist)
                    batches.append((X.to(dev), y.to(dev)))
                runner = model.train_step_runner(B, use_graph=not args.no_graph)
                dt = timed_steps(runner, batches, steps, 3, None)
                acc = kernel_breakdown(runner, batches, 3)
                roof = roofline_of(acc)
                roof["launches_per_step"] = acc[roof["kernel"]]["launches"] / 3
                roof["traffic"] = None  # (PMC passes are taken on the headline workload only)
                entry["runs"].append({"batch_per_gpu": B, "value": round(B * steps / dt, 1), "unit": "samples/s",
                                      "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": 3,
                                      "roofline": {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit",
                                                                        "frac", "avg_launch_ms", "launches_per_step")}})
                del runner
            out.append(entry)
            del model
            gc.collect()
        except Exception as e:  # a secondary configuration must never cost the headline line
            out.append({"config": label, "failed": repr(e)})
        finally:
            lib.mml_gemm_set_mode(mode0)
    return out


def expected_collectives(args, world, main_r):
    """DESIGN section 5's model of one step's exchange for this run (per rank): distinct rows of the local batch that
    live on other ranks x 4 B of keys, x 4 E B of rows each way; the MLP gradient arena through a ring."""
    if args.parallel_mode != "row_sharded":
        return None
    d = main_r.get("distinct_rows")
    if d is None:
        return None
    remote = d * (world - 1) / max(world, 1)
    E = main_r["emb"]
    arena = main_r["arena_bytes"]
    return {"distinct_rows_of_a_batch": int(d), "all_to_all_calls": 4, "all_reduce_calls": 1,
            "keys_MB": round(remote * 4 / 1e6, 4), "rows_MB_each_way": round(remote * 4 * E / 1e6, 4),
            "all_reduce_ring_MB": round(2 * arena * (world - 1) / max(world, 1) / 1e6, 4)}


def check_collectives(cps, exp, world):
    """Do the collectives counted inside the timed region (parallel.Comm's host-side counters, per step) agree with
    DESIGN section 5's model?  Calls exactly; bytes within 15 % (the model prices the batch-0 row set, the four resident
    batches differ by a few per cent; the count exchange's few bytes are not in the model).  None: no model for this mode."""
    if exp is None or cps is None:
        return None
    a2a = {k: v for k, v in cps.items() if k.startswith("all_to_all")}
    ar = {k: v for k, v in cps.items() if k.startswith("all_reduce")}
    calls_a2a = sum(v["calls"] for v in a2a.values())
    calls_ar = sum(v["calls"] for v in ar.values())
    sent = sum(v["bytes_sent"] for v in a2a.values()) / 1e6
    want = exp["keys_MB"] + 2 * exp["rows_MB_each_way"]
    out = {"all_to_all_calls": [round(calls_a2a, 3), exp["all_to_all_calls"]],
           "all_reduce_calls": [round(calls_ar, 3), exp["all_reduce_calls"]],
           "all_to_all_MB_sent": [round(sent, 4), round(want, 4)]}
    ok = abs(calls_a2a - exp["all_to_all_calls"]) < 1e-6 and abs(calls_ar - exp["all_reduce_calls"]) < 1e-6
    if world > 1:
        ok = ok and abs(sent - want) <= 0.15 * want + 0.01
    out["ok"] = bool(ok)
    return out


def describe_workload(name, cfg, vocab, dense):
    mc = cfg["model_config"]
    keys = {"mmoe": ("expert", "gate", "tower"), "ple": ("expert", "gate", "tower"),
            "sharedbottom": ("bottom", "tower")}.get(mc["model_name"], ())
    parts = [f"{k}s {mc[k + '_dnn_hidden_units']}" for k in keys]
    if mc["model_name"] in ("mmoe",):
        parts.insert(0, f"{mc['num_experts']} experts")
    if mc["model_name"] == "ple":
        parts.insert(0, f"{mc.get('num_levels', 2)} levels x ({mc.get('specific_expert_num', 3)} specific + "
                        f"{mc.get('shared_expert_num', 2)} shared) experts")
    if mc["model_name"] in ("esmm", "hmoe", "aitm", "snr_trans", "mssm"):
        parts.append(f"towers/experts {mc['expert_dnn_hidden_units']}")
    elif not keys:
        parts.append(f"layers {mc['dnn_hidden_units']}")
    return (f"{name}: {len(vocab)} sparse fields{f' + {len(dense)} dense' if dense else ''}, "
            f"{sum(vocab) / 1e6:.2f}M rows ({max(vocab):.0e}-row top table), E={mc['emb']}, {mc['model_name']} "
            f"{', '.join(parts)}, {cfg['optim_config']['optimizer']} lr {cfg['optim_config']['lr']}")


def main():
    args = parse()
    forced = os.environ.get("MMLREC_BENCH_FORCE_SHARD") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not forced:
        sys.exit(spawn_ranks(args))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus and not forced:  # a silent 1-GPU measurement labelled otherwise helps nobody
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}", file=sys.stderr)
        sys.exit(2)
    rank, local, world, dist = dist_setup(args.gpus)
    dev = torch.device("cuda", local)
    import mmlrec_amd  # noqa: F401
    from mmlrec_amd import workloads as W
    if args.profile:
        from mmlrec_amd import profiling
        profiling.enable()

    model, cfg, vocab, dense = W.build_model(args.workload, dev, table_update=args.table_update,
                                             use_hip_graph=not args.no_graph, scatter_mode=args.scatter_mode)
    model.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], cfg["optim_config"]["metrics"])
    model.train()
    T = W.num_tasks(cfg)

    allreduce = None
    # MMLREC_BENCH_FORCE_SHARD=1 with --gpus 2 and WORLD_SIZE=1 walks the table-sharded code path (exchange, no HIP
    # graph, per-call timing of the collectives) on a single GPU: a smoke test of the N > 1 bench, not a measurement
    if world > 1 or (dist is not None and forced):
        from mmlrec_amd import parallel
        parallel.shard_model(model, dist, args.batch, mode=args.parallel_mode)

    def make_batches(B):
        out = []
        for i in range(4):
            X, y = W.synth_batch(vocab, len(dense), B, T, seed=1 + i + 1000 * rank, dist=args.dist)
            out.append((X.to(dev), y.to(dev)))
        return out

    results = {}
    for B in [args.batch] + ([args.alt_batch] if args.alt_batch and args.alt_batch != args.batch else []):
        batches = make_batches(B)
        runner = model.train_step_runner(B, use_graph=not args.no_graph, allreduce=allreduce, overlap=(args.streams == 2 and not args.serial),
                                         split_dense="force" if args.split_dense else False)
        if B == args.batch and getattr(model, "_parallel", None) is not None:
            # PREFLIGHT (before anything is timed; every rank computes it, rank 0 prints it): what DESIGN section 5's model
            # says one step of THIS run exchanges per rank -- the JSON line repeats it next to what was counted inside the
            # timed region and says whether the two agree (collectives_per_step.check)
            Xi = batches[0][0][:, :len(vocab)].long()
            pre = dict(distinct_rows=int(sum(torch.unique(Xi[:, f]).numel() for f in range(len(vocab)))),
                       emb=int(cfg["model_config"]["emb"]), arena_bytes=int(runner.store.arena.numel() * 4))
            exp = expected_collectives(args, world, pre)
            if rank == 0:
                print("bench.py preflight: world %d, mode %s, batch %d per rank -> expected per step and rank: %s"
                      % (world, args.parallel_mode, B, json.dumps(exp)), file=sys.stderr, flush=True)
        steps = args.steps if B == args.batch else max(args.steps, 50)
        warm_losses = [] if (B == args.batch and world == 1 and getattr(model, "_parallel", None) is None) else None
        dt = timed_steps(runner, batches, steps, args.warmup, dist, warm_losses=warm_losses)
        results[B] = dict(dt=dt, steps=steps, value=world * B * steps / dt, ms=dt / steps * 1e3,
                          loss=float(runner.plan.loss.item()) / B, warm_losses=warm_losses)
        if B == args.batch:
            runner0 = runner
            if getattr(model, "_parallel", None) is not None:
                Xi = batches[0][0][:, :len(vocab)].long()
                results[B]["distinct_rows"] = int(sum(torch.unique(Xi[:, f]).numel() for f in range(len(vocab))))
                results[B]["emb"] = int(cfg["model_config"]["emb"])
                results[B]["arena_bytes"] = int(runner.store.arena.numel() * 4)
        acc = kernel_breakdown(runner, batches, min(args.steps, 10))
        results[B]["acc"] = acc
        results[B]["bsteps"] = min(args.steps, 10)
        # did the run stay a training run?  (the summed-BCE kernel clamps log p at -100 with fmaxf, which also swallows a
        # NaN: a diverged model keeps reporting a finite loss.  snr_trans_ae30 on two alternating batches overfits to
        # loss 0.004 by step 31 and has all-NaN input gradients by step 41 -- and a scatter that takes 4 ms for them)
        gin = runner.plan.layer_outputs.get("dnn_input") if hasattr(runner.plan, "layer_outputs") else None
        results[B]["finite"] = bool(gin is None or gin.grad is None or torch.isfinite(gin.grad).all().item())

    # forward-only path (predict / evaluate reuse the gather, GEMM, gate and head kernels):
    # model.forward in eval mode under no_grad, as predict() calls it -- includes the input copy, the output clone
    # and the embedding-status check (one host sync per call)
    infer = None
    if world == 1:
        B = args.batch
        batches = make_batches(B)
        model.eval()
        import gc
        with torch.no_grad():
            for i in range(3):
                model(batches[i % 4][0])
            torch.cuda.synchronize()
            gc.collect()   # (as in timed_steps: a cyclic-GC pass that destroys the HIP graphs / events of the earlier
            gc.disable()   # phases inside the timed loop costs tens of ms once)
            n_inf = max(args.steps, 20)
            t0 = time.perf_counter()
            for i in range(n_inf):
                model(batches[i % 4][0])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            gc.enable()
        model.train()
        infer = {"value": round(B * n_inf / dt, 1), "unit": "samples/s", "batch": B, "ms_per_batch": round(dt / n_inf * 1e3, 4),
                 "path": "model.forward(X) in eval mode under no_grad (what predict() runs per batch)"}

    # secondary measurement: the lazy_exact table optimizer (dense-Adam trajectory at touched-row cost); its timed
    # region ENDS with the flush that brings every one of the 12.49 M rows to the reference state
    lazy = {}
    if world == 1 and args.table_update != "lazy_exact" and not args.no_lazy:
        del runner
        model2, _, _, _ = W.build_model(args.workload, dev, table_update="lazy_exact", use_hip_graph=not args.no_graph)
        model2.compile(cfg["optim_config"]["optimizer"], cfg["optim_config"]["loss"], cfg["optim_config"]["metrics"])
        model2.train()
        for B in [args.batch] + ([args.alt_batch] if args.alt_batch and args.alt_batch != args.batch else []):
            batches = make_batches(B)
            r2 = model2.train_step_runner(B, use_graph=not args.no_graph)
            steps = args.steps if B == args.batch else max(args.steps, 50)
            dt, tf = timed_steps(r2, batches, steps, args.warmup, dist, flush=model2.flush_tables)
            lazy[B] = {"batch_per_gpu": B, "unit": "samples/s", "steps": steps,
                       "value_incl_final_flush": round(B * steps / dt, 1),
                       "value_steps_only": round(B * steps / (dt - tf), 1),
                       "ms_per_step_steps_only": round((dt - tf) / steps * 1e3, 4), "final_flush_ms": round(tf * 1e3, 3)}
            if B == args.batch and args.lazy_epoch_steps > 0:
                # steady state: an "epoch" of lazy_epoch_steps steps, then the flush a fit() epoch ends with:
                # rows no batch touched replay that many zero-gradient steps there
                n_ep = args.lazy_epoch_steps
                dt2, tf2 = timed_steps(r2, batches, n_ep, 0, dist, flush=model2.flush_tables)
                lazy[B]["epoch"] = {"steps": n_ep, "ms_per_step_steps_only": round((dt2 - tf2) / n_ep * 1e3, 4),
                                    "flush_ms": round(tf2 * 1e3, 3),
                                    "value_incl_flush": round(B * n_ep / dt2, 1),
                                    "note": "flush amortised over an epoch of this many steps (4 resident batches rotated: "
                                            "the untouched rows skip every step of the epoch)"}
        del model2

    if dist is not None:
        barrier(dist)
        dist.destroy_process_group()
    if rank != 0:
        return
    from mmlrec_amd import _lib
    gmode = _lib.load().mml_gemm_get_mode()
    gemm_dtype = {0: "f32", 4: "f32 (GEMMs: fp32-equivalent emulation on the 16-bit MFMA pipe with f32 accumulate -- two scaled fp16 planes "
                                "per operand where the operand magnitudes travel with the tensors (batches >= 32 768), else three bf16 planes; "
                                "max-norm error vs float64 3.3e-7 / 4.7e-7, fp32 MFMA 4.3e-7)",
                  2: "f32 (GEMMs: as mode 4)",
                  3: "f32 (GEMMs: fp32-equivalent 3-plane bf16 MFMA emulation, f32 accumulate)",
                  1: "bf16 GEMM operands (rounded in registers), f32 accumulate, f32 everywhere else"}.get(
        gmode, "f32 operands, GEMM products from 2 bf16 planes (~1e-5 rel), f32 accumulate")
    main_r = results[args.batch]
    roof = roofline_of(main_r["acc"])
    roof["launches_per_step"] = main_r["acc"][roof["kernel"]]["launches"] / main_r["bsteps"]
    per = W.algorithmic_per_sample(cfg, vocab, len(dense))
    line = {
        "metric": "train-step samples/sec, MMoE AliExpress-shape batch" if args.workload.startswith("mmoe_ae30")
                  else f"train-step samples/sec, {args.workload}",
        "value": round(main_r["value"], 1), "unit": "samples/s", "n_gpus": world, "steps": main_r["steps"],
        "warmup": args.warmup, "ms_per_step": round(main_r["ms"], 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": gemm_dtype, "data": "synthetic",
        "config": {"workload": describe_workload(args.workload, cfg, vocab, dense),
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world, "index_dist": args.dist,
                   "table_update": model.optimizer().table_update,
                   "dense_update_schedule": ("split: untouched rows beside the forward, touched rows after the scatter"
                                             if getattr(runner0, "split_dense", False) else
                                             ("one launch after the scatter, gradients read for marked rows only"
                                              if getattr(runner0, "grad_marks", False) else "one launch after the scatter")),
                   "hip_graph": not args.no_graph, "scatter_mode": args.scatter_mode,
                   "hip_runtime": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")},
                   "streams": 1 if not getattr(runner0, "overlap", True) else (3 if getattr(runner0, "split_dense", False) else 2),
                   "early_fork": int(getattr(runner0, "early_fork", 0) or 0),
                   "tables": "single GPU" if getattr(model, "_parallel", None) is None else
                             {"row_sharded": "row-wise sharded over ranks (owner = (row + field) mod N), one all-to-all "
                                             "per direction: keys / rows / row gradients",
                              "replicated": "replicated, all-gather of (index, row-gradient) pairs",
                              "table_wise": "table-wise sharded over ranks, all-to-all index/row/grad exchange"}[
                                 args.parallel_mode],
                   "algorithmic_per_sample": per},
        "roofline": roof,
        "kernels_ms_per_step": {k: round(v["ms"] / main_r["bsteps"], 4) for k, v in
                                sorted(main_r["acc"].items(), key=lambda kv: -kv[1]["ms"])},
        "mean_loss_per_sample": round(main_r["loss"], 5),
        "gradients_finite": main_r.get("finite", True),
    }
    # the whole step against the HBM roofline: every launch's compulsory bytes -- each distinct
    # operand read once, each output written once; GEMM launches included (engine: `hbm_bytes`) -- over the MEASURED step
    kern = {k: v for k, v in main_r["acc"].items()
            if not k.startswith(("all_to_all", "all_reduce", "all_gather", "row_sharded_"))}
    step_bytes = sum(v.get("hbm_bytes", v["bytes"]) for v in kern.values()) / main_r["bsteps"]
    step_flops = sum(v["flops"] for v in kern.values()) / main_r["bsteps"]
    line["whole_step"] = {
        "algorithmic_hbm_bytes": round(step_bytes), "algorithmic_flops": round(step_flops),
        "achieved_GB_per_s": round(step_bytes / (main_r["ms"] * 1e-3) / 1e9, 1), "peak_GB_per_s": 8000.0,
        "frac_of_hbm_peak": round(step_bytes / (main_r["ms"] * 1e-3) / 8e12, 4),
        "hbm_floor_ms": round(step_bytes / 8e12 * 1e3, 4),
        "note": "sum over the step's launches of their compulsory HBM bytes / ms_per_step / 8 TB/s; the phases of a step "
                "depend on each other, so the floor is the sum of the launches' floors"}
    if not line["gradients_finite"]:
        sys.stderr.write("bench: the input gradients of the last step are not finite -- the model diverged on this "
                         "synthetic stream; the timing is that of a diverged run\n")
    if world > 1 or forced:
        line["rccl_ranks"] = world if os.environ.get("MMLREC_BENCH_SHARE_GPU") != "1" else 0
    # parity of the measured run itself: the losses of the untimed warm-up steps and of the LAST timed step against the
    # oracle's losses for this exact step sequence (tests/golden/make_bench_losses.py; the GPU test
    # test_bench_sequence_losses_match_fixture checks every step).  A mismatch fails the run.
    fx = loss_fixture(args)
    if fx is not None and main_r.get("warm_losses") is not None:
        want = fx["loss_sum_per_step"]
        got = list(enumerate(main_r["warm_losses"]))
        last = args.warmup + main_r["steps"] - 1
        got.append((last, main_r["loss"] * args.batch))
        checked = [(i, v, want[i], abs(v - want[i]) / want[i], loss_tolerance(i)) for i, v in got
                   if i < len(want) and loss_tolerance(i) is not None]
        # the warm-up steps (1e-4) decide: a mismatch there is a wrong result and fails the run.  The last timed step
        # sits where a free-running trajectory has started to amplify rounding noise: reported, never fatal
        bad = [c for c in checked if not c[3] < c[4] and c[0] < 12]
        late = [c for c in checked if not c[3] < c[4] and c[0] >= 12]
        line["loss_check"] = {"against": "tests/golden/bench_losses_%s.json (oracle)" % args.workload,
                              "steps_checked": [c[0] for c in checked],
                              "max_rel_err": max([c[3] for c in checked], default=None), "ok": not bad,
                              "late_step_within_tolerance": not late}
        if bad or late:
            print("bench.py: loss check %s: " % ("FAILED" if bad else "late step outside its tolerance (not fatal)") +
                  json.dumps(bad + late), file=sys.stderr)
            line["loss_check"]["failed"] = [[c[0], c[1], c[2]] for c in bad + late]
    comm = {k: v for k, v in main_r["acc"].items()
            if k.startswith(("all_to_all", "all_reduce", "all_gather", "row_sharded_"))}
    if comm:  # serial, event-bracketed time of the exchange steps of rank 0 (second, instrumented pass)
        line["collectives_ms_per_step"] = {k: round(v["ms"] / main_r["bsteps"], 4) for k, v in comm.items()}
    cps = getattr(runner0, "collectives_per_step", None)
    if cps is not None:
        # rank 0's collectives INSIDE the timed region, per step: number of calls and the bytes that left / reached this
        # rank (all_to_all: what goes to / comes from the other ranks; all_reduce: the ring's 2 (N - 1) / N of the
        # buffer).  To be read against the per-N table of DESIGN section 5: row_sharded = 4 all-to-alls (per-owner
        # counts, keys, rows, row gradients) + 1 all-reduce (the MLP gradient arena) per step.
        line["collectives_per_step"] = {k: {"calls": round(v["calls"], 3), "MB_sent": round(v["bytes_sent"] / 1e6, 4),
                                            "MB_received": round(v["bytes_received"] / 1e6, 4)} for k, v in cps.items()}
        exp = expected_collectives(args, world, main_r)
        line["collectives_per_step"]["expected"] = exp
        line["collectives_per_step"]["check"] = check_collectives(cps, exp, world)
    if args.alt_batch and args.alt_batch in results:
        r = results[args.alt_batch]
        line["alt"] = {"batch_per_gpu": args.alt_batch, "value": round(r["value"], 1), "unit": "samples/s",
                       "ms_per_step": round(r["ms"], 4), "steps": r["steps"]}
        try:  # the reference's own batch size is launch- / latency-bound: its dominant kernel and roofline fraction too
            ra = roofline_of(r["acc"])
            ra["launches_per_step"] = r["acc"][ra["kernel"]]["launches"] / r["bsteps"]
            ra["traffic"] = None  # (the PMC summary under profiles/ is taken at the headline batch)
            line["alt"]["roofline"] = ra
            line["alt"]["launches_per_step"] = sum(v["launches"] for v in r["acc"].values()) / r["bsteps"]
        except Exception:
            pass
    if infer:
        line["inference"] = infer
    if lazy and args.batch in lazy and "epoch" in lazy[args.batch]:
        ep = lazy[args.batch]["epoch"]
        line["table_update_modes"] = {
            "dense_exact": {"value": line["value"], "ms_per_step": line["ms_per_step"],
                            "note": "the primary `value`: the reference's literal schedule (every row of every table "
                                    "every step)"},
            "lazy_exact": {"value": ep["value_incl_flush"], "ms_per_step": ep["ms_per_step_steps_only"],
                           "flush_ms": ep["flush_ms"], "epoch_steps": ep["steps"],
                           "note": "what table_update='auto' (the default of fit() / main.run()) picks for Adam / RMSprop "
                                   "since round 4: the same trajectory (tests: 2e-6), flush of the epoch included"}}
    if lazy:
        line["lazy_exact"] = {"note": "same dense-Adam trajectory (tests: <=2e-6 rel on parameters), table update "
                                      "restricted to the batch's rows + replay of skipped zero-gradient steps; the "
                                      "flush (replay for ALL 12.49M rows, needed only before evaluation/checkpoint: "
                                      "once per epoch in fit()) is timed separately and also folded into "
                                      "value_incl_final_flush over this short run",
                              "runs": list(lazy.values())}
    if world == 1 and not args.no_configs and args.workload == "mmoe_ae30" and not forced:
        line["configs"] = secondary_configs(args, dev)
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:  # never lose the GPU line to a host-side problem
            line["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                                    "sample": f"failed: {e!r}"}
    # RCCL writes its version banner through C stdio (block-buffered when stdout is a pipe or a file): push it out
    # first, so that the JSON line is the LAST line of stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)
    if line.get("loss_check", {}).get("ok") is False:
        sys.exit(4)


if __name__ == "__main__":
    try:
        main()
    except Exception as e:  # a rank that lost a collective must end the whole job: the launcher reaps the others
        if type(e).__name__ == "CollectiveError":
            import traceback
            traceback.print_exc()
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(70)
        raise
