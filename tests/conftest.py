import os
import sys


def _host_threads(cap=16):
    """Threads the host-side checkers (numpy / OpenBLAS, the OpenMP loops of oracle/fast.c, torch-CPU) may use: the CPUs
    this process may run on -- its affinity mask AND its cgroup CPU quota (a container limited to 16 CPUs of a 256-thread
    host by quota still sees 256 in os.cpu_count(): 256 spinning OpenMP threads on 16 CPUs' worth of time is what turns a
    3 s oracle step into 30 s) -- and never more than `cap` (measured on a 256-thread MI355X host: the GPU suite takes
    145 s with 16 threads, 230 s with 256)."""
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
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


# (before numpy / torch load their thread pools; an explicit setting in the environment wins)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(_host_threads()))

# the package's HIP runtime defaults (mmlrec_amd/__init__.py), before a test module's `import torch` can initialise HIP
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The engine gives GEMM launches their operand magnitudes (two-plane fp16 arithmetic) only for batches >= 32 768, where
# it pays; the fixtures hold 64 samples, so the suite switches it on for every batch -- the tests that name an `arith`
# parameter run both forms.
os.environ.setdefault("MMLREC_AMAX", "1")

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["sharedbottom_ml", "mmoe_kuairec", "ple_ijcai", "mmoe_ae30", "mmoe_ae30d", "star_amazon",
                "pepnet_amazon", "mlp_ml", "mlp_ae", "esmm_ml",
                "cross_stitch_ae", "hmoe_ml", "aitm_ml", "snr_trans_ae",
                "mssm_ml", "sharedbottom_bn", "mmoe_bn",
                "mssm_bn", "cross_stitch_bn", "ple_l2", "escm_ml", "apg_ae", "star_dbn",
                # round 4: logits of +-4 (p = 0.0003 .. 0.994) and a partly SATURATED head (p exactly 0 / 1, clamped BCE)
                "mmoe_ae30_s4", "mmoe_ae30_sat"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return request.param, load_golden(request.param)


def bn_noise_keys(keys):
    """Keys whose comparison is noise-driven in a model with BatchNorm: the bias of a Linear that feeds a BatchNorm has
    a mathematically ZERO gradient (the batch mean is subtracted again), so the reference's own gradient is fp32 noise
    (~1e-7) -- and Adam / Adagrad turn the SIGN of that noise into full lr-sized steps.  The model output does not
    depend on these biases; the BatchNorm's running_mean follows them (it is checked strictly after the first step,
    whose forward still saw the initial bias)."""
    keys = set(keys)
    bias = {k for k in keys if ".linears." in k and k.endswith(".bias")
            and k.replace(".linears.", ".bn.").replace(".bias", ".weight") in keys}
    rmean = {k.replace(".linears.", ".bn.").replace(".bias", ".running_mean") for k in bias}
    return bias, rmean & keys


# ---------------------------------------------------------------------------------------------------------------
# Table updates are judged over the rows the batch TOUCHED (at B = 8 192 the batch touches 0.045 % of the
# 1e7-row table, so an outlier allowance counted over all rows accepts anything).
# ---------------------------------------------------------------------------------------------------------------
def table_update_report(before, got, ref, rows):
    """Element-wise comparison of the UPDATE (after - before) of the touched rows.  Tolerance per element:
    5 % of the reference update, floored at 1e-3 of the largest update of the table (first-step Adam moves every element
    by ~lr, so this is the "5 % of lr" criterion there; it stays meaningful when |g| << eps and updates are << lr) and at
    two ulps of the parameter (an update below the parameter's spacing is quantised).  Returns (share of touched elements
    outside the tolerance, largest error / largest update)."""
    b = before[rows].astype(np.float64)
    d_ref = ref[rows].astype(np.float64) - b
    d_got = got[rows].astype(np.float64) - b
    scale = max(float(np.abs(d_ref).max()), 1e-30)
    tol = np.maximum(np.maximum(0.05 * np.abs(d_ref), 1e-3 * scale), 2.0 * np.spacing(np.abs(before[rows])))
    err = np.abs(d_got - d_ref)
    bad = err > tol
    if bad.sum() <= 2:  # (a 99-row table has 792 elements: two noise-level sign flips must not fail it)
        return float(bad.sum()) * 1e-9, float(err.max() / scale)
    # (round 6: the two free flips are free for every tensor -- a 62-row table of KuaiRec-32 (992 elements) failed the
    # 2e-3 share with THREE flipped elements in 3 of 36 soak runs, fork or no fork: scatter-atomics-order noise under Adam)
    return float(bad.sum() - 2) / float(bad.size), float(err.max() / scale)


def check_tables(vocab, names, X, before, got, ref, allow=2e-3, moved=True):
    """Every table: untouched rows bit-unchanged (valid while no row outside `X` has optimizer state), touched rows
    updated like the oracle's (outliers counted over touched elements only)."""
    worst = 0.0
    for f, v in enumerate(vocab):
        k = f"embedding_dict.{names[f]}.weight"
        rows = np.unique(X[:, f].astype(np.int64))
        mask = np.ones(v, bool)
        mask[rows] = False
        assert np.array_equal(got[k][mask], before[k][mask]), f"{k}: an untouched row moved"
        share, rel = table_update_report(before[k], got[k], ref[k], rows)
        assert share < allow, (k, share, rel, len(rows))
        if moved:
            assert np.abs(got[k][rows] - before[k][rows]).max() > 0, k
        worst = max(worst, share)
    return worst


def check_update(name, before, got, ref, allow=2e-3):
    """The same update criterion for a tensor every element of which is touched (MLP weights and biases)."""
    rows = np.arange(before.shape[0]) if before.ndim else np.arange(1)
    b = before if before.ndim else before.reshape(1)
    share, rel = table_update_report(b, got.reshape(b.shape), ref.reshape(b.shape), rows)
    assert share < allow, (name, share, rel)
    return share


# ---------------------------------------------------------------------------------------------------------------
# The full-size step tests' initialisation (also used by tests/golden/make_bench_losses.py for the loss fixtures)
# ---------------------------------------------------------------------------------------------------------------
def randomize_he(model, seed):
    """He-scaled weights / 0.05-scaled tables drawn on the host and copied into the model (and its unregistered STAR
    tensors), so that logits are far from the 0.5 the reference's 1e-4 init gives."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("embedding_dict."):
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(p.device))
            elif p.dim() == 2:
                fan_in = p.shape[0] if (".shared_weight" in n or ".specific_weight" in n) else p.shape[1]
                scale = (2.0 / fan_in) ** 0.5
                if ".specific_weight" in n:  # multiplies the shared weight elementwise: keep the product He-scaled
                    p.copy_((1.0 + 0.25 * torch.randn(p.shape, generator=g)).to(p.device))
                else:
                    p.copy_((torch.randn(p.shape, generator=g) * scale).to(p.device))
            elif not n.startswith("out."):
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(p.device))
        frozen = {}
        for pfx in ("linears", "final_layers"):
            for li, mod in enumerate(getattr(model, pfx, [])):
                if not hasattr(mod, "specific_weights"):
                    continue
                for d in range(len(mod.specific_weights) - 1):  # the last one IS the registered parameter
                    w, b = mod.specific_weights[d], mod.specific_biases[d]
                    w.data.copy_((1.0 + 0.25 * torch.randn(w.shape, generator=g)).to(w.device))
                    b.data.copy_((torch.randn(b.shape, generator=g) * 0.05).to(b.device))
                    frozen[f"{pfx}.{li}.specific_weights.{d}"] = w.detach().cpu().numpy().copy()
                    frozen[f"{pfx}.{li}.specific_biases.{d}"] = b.detach().cpu().numpy().copy()
    return frozen
