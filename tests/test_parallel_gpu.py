"""Multi-process parity of the parallel modes on REAL kernels: two ranks share the one MI355X of the test box and talk
through gloo (parallel.Comm stages device buffers through the host there; on a multi-GPU node the same code runs over
RCCL).  Contract: an N-rank step on a batch split N ways == the 1-rank step on the whole batch -- and
the 1-rank step is pinned to the reference by the golden trajectories, so the ranks are compared with those directly."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check_state(sd, g, tag, lr, steps):
    bad = []
    for k in sd:
        ref = g[f"{tag}/{k}"].astype(np.float64)
        dv = np.abs(sd[k].cpu().numpy().astype(np.float64) - ref)
        if (dv > RTOL * max(np.abs(ref).max(), 1e-30)).mean() >= 2e-3 or dv.max() > 2.5 * lr * steps:
            bad.append((k, float(dv.max())))
    return bad


def _worker(rank, world, port, jobs, ret):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import mmlrec_amd  # noqa: F401
        from mmlrec_amd import parallel
        from conftest import load_golden
        from test_models_gpu import build, load_state
        out = []
        for job in jobs:
            case_name, mode, kind, tu, graph = job[:5]
            prefetch = len(job) > 5 and job[5]
            g = load_golden(case_name)
            model, cfg = build(g, table_update=tu)
            load_state(model, g)
            model.compile(kind, cfg["optim_config"]["loss"], ["auc"])
            model.train()
            dedup = not mode.endswith("_nodedup")  # row_sharded: distinct rows (default) or every lookup travels
            mode = mode.replace("_nodedup", "")
            par = parallel.shard_model(model, dist, 64 // world, mode=mode, dedup=dedup)
            losses = []
            # prefetch: the index-only half of the next batch's exchange (counts all-to-all included) runs on the routing
            # stream while the step is in flight (TrainStep.prefetch; bench.py's loop) -- on TWO ranks here
            pre = prefetch and mode == "row_sharded" and graph
            batches = [(torch.from_numpy(g[f"X{i}"])[rank::world].contiguous().cuda(),
                        torch.from_numpy(g[f"y{i}"])[rank::world].contiguous().cuda()) for i in range(3)]
            for i in range(3):
                X, y = batches[i]
                step = model.train_step_runner(X.shape[0], use_graph=graph)
                if pre:
                    if not step._has_next:
                        step.load(X, y)
                    else:
                        assert i > 0
                else:
                    step.plan.X.copy_(X)
                    step.plan.y.copy_(y)
                step.run()
                if pre and i + 1 < 3:
                    step.prefetch(*batches[i + 1])
                lt = step.plan.loss.detach().clone().double()
                par.comm.all_reduce(lt)
                losses.append(float(lt.item()))
            if pre:
                step.drop_prefetch()
            sd = model.state_dict()  # flushes lazy rows and synchronises the tables (collective)
            bad = _check_state(sd, g, f"{kind}3", cfg["optim_config"]["lr"], 3)
            ok_loss = bool(np.allclose(losses, g[f"{kind}_losses"], rtol=RTOL))
            # a second state_dict must not start another collective (tables are clean now)
            assert not par.dirty
            out.append((case_name, mode, kind, tu, graph, ok_loss, losses, bad))
        ret[rank] = ("ok", out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        ret[rank] = ("error", traceback.format_exc())
        raise


def _spawn(jobs, world=2):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), jobs, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        status, out = ret[r]
        assert status == "ok", out
        for case_name, mode, kind, tu, graph, ok_loss, losses, bad in out:
            assert ok_loss, (r, case_name, mode, kind, tu, losses)
            assert not bad, (r, case_name, mode, kind, tu, bad)


def _dropout_worker(rank, world, port, ret):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import mmlrec_amd  # noqa: F401
        from mmlrec_amd import parallel
        from conftest import load_golden, table_update_report
        from oracle import mmlrec_oracle as orc
        from test_dropout_gpu import P, _spec
        from test_models_gpu import build, load_state
        out = []
        for mode in ("row_sharded", "replicated"):
            g = load_golden("mmoe_ae30d")
            model, cfg = build(g, dnn_dropout=P, table_update="dense_exact")
            load_state(model, g)
            model.compile("adam", cfg["optim_config"]["loss"], ["auc"])
            model.train()
            B = 64 // world
            par = parallel.shard_model(model, dist, B, mode=mode)
            before = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
            X = torch.from_numpy(g["X0"])[rank * B:(rank + 1) * B].contiguous().cuda()  # consecutive blocks
            y = torch.from_numpy(g["y0"])[rank * B:(rank + 1) * B].contiguous().cuda()
            step = model.train_step_runner(B, use_graph=True)
            step.plan.X.copy_(X)
            step.plan.y.copy_(y)
            step.run()
            lt = step.plan.loss.detach().clone().double()
            par.comm.all_reduce(lt)
            after = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
            # the one-rank step on the WHOLE batch under the same mask stream (oracle)
            params = {k: v.copy() for k, v in before.items()}
            orc.set_dropout(P, seed=model.dropout_seed, step=1)
            try:
                ref_loss = orc.train_step(_spec(g, P), params, orc.DenseOptimizer("adam", cfg["optim_config"]["lr"]),
                                          g["X0"], g["y0"])
            finally:
                orc.set_dropout(0)
            bad = []
            if abs(float(lt.item()) - ref_loss) / ref_loss >= RTOL:
                bad.append(("loss", float(lt.item()), ref_loss))
            for k, r in params.items():
                b, a = before[k], after[k]
                b2 = b.reshape(b.shape[0], -1) if b.ndim > 1 else b.reshape(1, -1)
                rows = np.nonzero(np.abs(r.reshape(b2.shape) - b2).max(1) + np.abs(a.reshape(b2.shape) - b2).max(1))[0]
                if len(rows) and table_update_report(b2, a.reshape(b2.shape), r.reshape(b2.shape), rows)[0] >= 2e-3:
                    bad.append((k,))
            out.append((mode, bad))
        ret[rank] = ("ok", out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        ret[rank] = ("error", traceback.format_exc())
        raise


@pytest.mark.timeout(900)
def test_world2_dropout_step_is_the_one_rank_step():
    """Dropout on two data-parallel ranks: the ranks' masks are the two halves of the one-rank mask (mml_dropout row0),
    so the summed loss and the parameters after a fused step equal the oracle's step on the whole batch."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dropout_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in range(2):
        status, out = ret[r]
        assert status == "ok", out
        for mode, bad in out:
            assert not bad, (r, mode, bad)


@pytest.mark.timeout(900)
def test_world2_steps_match_reference_trajectories():
    jobs = []
    # (star_amazon: BASELINE configs[4] names STAR "over sharded tables" -- VERDICT r3)
    for case_name in ("mmoe_ae30d", "pepnet_amazon", "star_amazon"):
        for mode in ("row_sharded", "replicated", "table_wise"):
            jobs.append((case_name, mode, "adam", "dense_exact", True))
            jobs.append((case_name, mode, "adagrad", "sparse_rows", False))
        jobs.append((case_name, "row_sharded", "adam", "lazy_exact", True))
        jobs.append((case_name, "row_sharded", "adam", "dense_exact", True, True))   # + prefetched routing
        jobs.append((case_name, "row_sharded_nodedup", "adam", "dense_exact", True))
        jobs.append((case_name, "row_sharded_nodedup", "adagrad", "sparse_rows", False))
        jobs.append((case_name, "replicated", "adam", "lazy_exact", False))
    _spawn(jobs)


def _fit_worker(rank, world, port, mode, ret):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        torch.cuda.set_device(0)
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import mmlrec_amd  # noqa: F401
        from mmlrec_amd import parallel, workloads as W
        torch.manual_seed(0)
        model, cfg, vocab, dense = W.build_model("mmoe_ae30", torch.device("cuda:0"), vocab_scale=1e-4, seed=0)
        model.compile("adagrad", cfg["optim_config"]["loss"], ["auc", "acc"])
        T = W.num_tasks(cfg)
        X, y = W.synth_batch(vocab, len(dense), 1001, T, seed=3)   # 1001: not a multiple of world * batch
        Xv, yv = W.synth_batch(vocab, len(dense), 300, T, seed=4)
        if world > 1:
            parallel.shard_model(model, dist, 128, mode=mode)
        torch.manual_seed(5)
        cols = lambda M: [M[:, j].numpy() for j in range(M.shape[1])]  # fit() takes a list of feature columns
        best = model.fit(cols(X), y.numpy(), batch_size=128 // world, epochs=2, shuffle=False,
                         validation_data=(cols(Xv), yv.numpy()))
        pred = best.predict(cols(Xv), 128)
        sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
        ret[rank] = ("ok", (model.history, pred, sd, getattr(best, "_parallel", None) is None))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    except Exception:
        ret[rank] = ("error", traceback.format_exc())
        raise


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["row_sharded", "replicated"])
def test_world2_fit_with_validation(mode):
    """fit() on two ranks (deepcopy of the best model, per-rank data slices, synchronised tables): runs to
    the end, both ranks agree, and the result tracks the 1-rank run on the same data (same global batches up to the
    order inside a batch; Adagrad, so trajectories stay close)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret2, ret1 = mgr.dict(), mgr.dict()
    mp.spawn(_fit_worker, args=(2, _free_port(), mode, ret2), nprocs=2, join=True)
    mp.spawn(_fit_worker, args=(1, _free_port(), mode, ret1), nprocs=1, join=True)
    for r in ret2.values():
        assert r[0] == "ok", r[1]
    assert ret1[0][0] == "ok", ret1[0][1]
    (h0, p0, sd0, plain0), (h1, p1, sd1, plain1) = ret2[0][1], ret2[1][1]
    assert plain0 and plain1  # the returned best model is a plain single-GPU snapshot
    assert np.array_equal(p0, p1)
    for k in sd0:
        if mode == "row_sharded" or not k.startswith("embedding_dict."):
            assert np.array_equal(sd0[k], sd1[k]), k
    for a, b in zip(h0, h1):
        assert a.keys() == b.keys()
        assert all(np.isclose(a[k], b[k], rtol=1e-12, equal_nan=True) for k in a), (a, b)
    h_ref, p_ref, sd_ref, _ = ret1[0][1]
    # world 2 with batch 64 per rank sees the same 128-sample global batches as world 1 with batch 128, except that the
    # final ragged batch is padded by one repeated sample (DistributedSampler semantics)
    assert abs(h0[-1]["loss"] - h_ref[-1]["loss"]) < 2e-3 * abs(h_ref[-1]["loss"])
    assert np.abs(p0 - p_ref).max() < 5e-3


@pytest.mark.timeout(900)
def test_bench_two_ranks_control_flow():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU, rank 0 prints ONE JSON
    line), here with both ranks sharing the test box's GPU over gloo (MMLREC_BENCH_SHARE_GPU=1: a control-flow smoke
    test of the row-sharded benchmark path -- barriers, max-over-ranks timing, the instrumented pass with its
    collectives -- not a measurement)."""
    import json
    import subprocess
    env = dict(os.environ, MMLREC_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for attempt in range(3):
        # (the rendezvous port is probed free and bound a second later by the launcher: once in ~30 suite runs something else
        #  on the box takes it in between -- EADDRINUSE, a fault of this harness, not of the run: another port)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
               "--warmup", "2", "--batch", "4096", "--alt-batch", "0"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8192
    assert "row-wise sharded" in d["config"]["tables"]
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    assert any(k.startswith("row_sharded_") for k in d.get("collectives_ms_per_step", {}))
    # what rank 0 issued inside the timed region, per step, against DESIGN section 5's model of the exchange (VERDICT r4
    # item 10: the first multi-GPU run is checked against this line by line): four all-to-alls -- per-owner counts of the
    # NEXT batch (prefetched routing), keys, rows, row gradients -- and one all-reduce of the MLP gradient arena
    cps = d["collectives_per_step"]
    exp = cps["expected"]
    assert cps["all_to_all"]["calls"] == exp["all_to_all_calls"] == 4
    assert cps["all_reduce"]["calls"] == exp["all_reduce_calls"] == 1
    assert abs(cps["all_reduce"]["MB_sent"] - exp["all_reduce_ring_MB"]) < 1e-3
    want = exp["keys_MB"] + 2 * exp["rows_MB_each_way"]   # (+ 2 x 4 bytes of counts; the four batches differ by a few %)
    assert abs(cps["all_to_all"]["MB_sent"] - want) < 0.1 * want, (cps, exp)
    assert abs(cps["all_to_all"]["MB_received"] - want) < 0.1 * want, (cps, exp)
    # round 6: the run checks ITSELF -- a preflight line on stderr before anything is timed (what DESIGN 5's model expects
    # for this world size) and the verdict of counted against expected in the JSON line
    pre = [l for l in r.stderr.splitlines() if l.startswith("bench.py preflight: world 2")]
    assert len(pre) == 1 and '"all_to_all_calls": 4' in pre[0] and "row_sharded" in pre[0], r.stderr[-1500:]
    chk = cps["check"]
    assert chk["ok"] is True and chk["all_to_all_calls"] == [4.0, 4] and chk["all_reduce_calls"] == [1.0, 1], chk
    assert abs(chk["all_to_all_MB_sent"][0] - chk["all_to_all_MB_sent"][1]) <= 0.15 * chk["all_to_all_MB_sent"][1] + 0.01


@pytest.mark.timeout(900)
def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (WORLD_SIZE unset): bench.py starts its two ranks itself as a
    child torch.distributed.run, relays ONE JSON line and reports n_gpus = 2 on row-wise sharded tables (VERDICT r2:
    the plain command used to measure one GPU silently).  Both ranks share the test box's GPU over gloo
    (MMLREC_BENCH_SHARE_GPU=1): control flow, not a measurement."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MMLREC_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch",
           "4096", "--alt-batch", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().endswith(lines[0]), r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192
    assert "row-wise sharded" in d["config"]["tables"]
    assert "rccl_ranks" in d and "collectives_ms_per_step" in d


def test_bench_refuses_a_world_that_is_not_gpus():
    """A launcher that started a different number of ranks than --gpus names is an error, not a silent 1-GPU run."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    env.pop("MMLREC_BENCH_FORCE_SHARD", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
