"""CPU oracle for the MMLRec hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-numpy (float32) restatement of the reference's training step for the five hot-path models:
multi-field embedding gather -> concat -> expert/gate/tower MLPs -> sigmoid heads -> summed BCE ->
hand-derived backward (dense [V,E] table gradients) -> dense optimizer.  Every function cites the
reference file:line (relative to the alipay/MMLRec tree) it restates.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
shipped package never does (it fails loudly when the HIP library is missing instead).

Parity status: PINNED -- tests/test_oracle_golden.py checks this file against fixtures in
tests/golden/*.npz that were produced by importing and running the unmodified reference
(tests/golden/make_golden.py).  The reference's own tree holds no tests or golden vectors
(SURVEY.md section 4), and its arithmetic lives in PyTorch ATen (un-vendored; README pins "PyTorch 1.11.0",
the fixtures were produced with torch 2.10.0 CPU), so those fixtures are the only pin there is.
"""
from __future__ import annotations

import json
from collections import OrderedDict

import numpy as np

F32 = np.float32

# optional C/OpenMP versions of the memory-bound loops (oracle/fast.c); numpy is always the fallback
_FAST = None


def use_fast(enable=True):
    """Load oracle/_build/liboracle_fast.so (built by __graft_entry__.build()); returns True when active."""
    global _FAST
    if not enable:
        _FAST = None
        return False
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle_fast.so")
    if not os.path.exists(path):
        return False
    lib = ctypes.CDLL(path)
    lib.gather_fields.restype = lib.scatter_fields.restype = lib.adam_dense.restype = lib.adagrad_dense.restype = None
    _FAST = lib
    return True


def _fp(a):
    import ctypes
    return a.ctypes.data_as(ctypes.c_void_p)


def _ptrs(arrays):
    import ctypes
    return (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


# ----------------------------------------------------------------------------------------------
# model description
# ----------------------------------------------------------------------------------------------
class Spec:
    """Static description of one model instance: feature schema + config (model/utils.py:328-431)."""

    def __init__(self, cfg, sparse_names, vocab, dense_names=()):
        self.cfg = cfg
        self.mc = cfg["model_config"]
        self.dc = cfg["data_config"]
        self.sparse_names = list(sparse_names)
        self.vocab = [int(v) for v in vocab]
        self.dense_names = list(dense_names)
        self.emb = int(self.mc.get("emb", 8))
        self.F = len(self.sparse_names)
        self.Nd = len(self.dense_names)
        self.K0 = self.F * self.emb + self.Nd  # compute_input_dim, basemodel.py:489-508
        self.model_name = self.mc.get("model_name", "sharedbottom").lower()
        self.task_name = self.mc.get("task_name", "mtl")
        self.num_domains = int(self.dc.get("num_domains", 1))
        # basemodel.py:93-102
        if self.task_name == "msl":
            self.T = self.num_domains
        elif self.task_name == "mtmsl":
            self.T = len(self.dc["label_columns"])
        else:
            self.T = len(self.mc.get("task_names", ["ctr", "ctcvr"]))
        self.dropout = float(self.mc.get("dnn_dropout", 0))  # applied when set_dropout() gives the mask stream a key
        if self.mc.get("dnn_activation", "relu") != "relu":
            raise NotImplementedError("oracle covers relu towers only")

    @staticmethod
    def from_golden(g):
        cfg = json.loads(str(g["cfg"]))
        return Spec(cfg, [str(s) for s in g["sparse_names"]], g["vocab"], [str(s) for s in g["dense_names"]])


# ----------------------------------------------------------------------------------------------
# K1: multi-field gather + concat
# ----------------------------------------------------------------------------------------------
def gather_dnn_input(spec, params, X):
    """input_from_feature_columns (model/basemodel.py:461-487) + combined_dnn_input (model/utils.py:434-446).

    idx = X[:, f].long() truncates toward zero (basemodel.py:476); out[b, f*E:(f+1)*E] = table_f[idx];
    dense columns follow the sparse block unchanged.  Bit-exact copy semantics.
    """
    X = np.asarray(X, dtype=F32)
    B = X.shape[0]
    E = spec.emb
    out = np.empty((B, spec.K0), dtype=F32)
    idx_all = np.ascontiguousarray(np.trunc(X[:, :spec.F]).astype(np.int64))
    tabs = [params[f"embedding_dict.{name}.weight"] for name in spec.sparse_names]
    for f, name in enumerate(spec.sparse_names):
        idx = idx_all[:, f]
        if idx.min() < 0 or idx.max() >= tabs[f].shape[0]:
            raise IndexError(f"index out of range in field {name}")  # nn.Embedding raises IndexError on CPU
    if _FAST is not None and all(t.flags.c_contiguous and t.dtype == F32 for t in tabs):
        import ctypes
        _FAST.gather_fields(_ptrs(tabs), _fp(idx_all), ctypes.c_int64(B), spec.F, E, _fp(out),
                            ctypes.c_int64(out.strides[0] // 4))
    else:
        for f in range(spec.F):
            out[:, f * E:(f + 1) * E] = tabs[f][idx_all[:, f]]
    if spec.Nd:
        out[:, spec.F * E:] = X[:, spec.F:spec.F + spec.Nd]
    return out, idx_all


def scatter_table_grads(spec, d_dnn_input, idx_all, params):
    """embedding_dense_backward triggered by sparse=False (model/basemodel.py:122, model/utils.py:476):
    gradW_f = zeros[V_f,E]; gradW_f[idx] += g[:, f*E:(f+1)*E] with duplicates accumulating in batch order."""
    E = spec.emb
    grads = {}
    keys = [f"embedding_dict.{name}.weight" for name in spec.sparse_names]
    if _FAST is not None:
        import ctypes
        gs = [np.zeros(params[k].shape, dtype=F32) for k in keys]
        d = np.ascontiguousarray(d_dnn_input, dtype=F32)
        _FAST.scatter_fields(_ptrs(gs), _fp(np.ascontiguousarray(idx_all)), ctypes.c_int64(d.shape[0]), spec.F, E,
                             _fp(d), ctypes.c_int64(d.shape[1]))
        return dict(zip(keys, gs))
    for f, key in enumerate(keys):
        g = np.zeros_like(params[key], dtype=F32)
        np.add.at(g, idx_all[:, f], d_dnn_input[:, f * E:(f + 1) * E])
        grads[key] = g
    return grads


# ----------------------------------------------------------------------------------------------
# dense building blocks (forward returns what backward needs)
# ----------------------------------------------------------------------------------------------
def linear_fwd(x, W, b=None):
    """nn.Linear: y = x @ W^T + b with W [out,in] (model/utils.py:130, :151)."""
    y = x @ W.T
    if b is not None:
        y = y + b
    return y.astype(F32, copy=False)


def linear_bwd(x, W, dy, has_bias=True):
    dx = (dy @ W).astype(F32, copy=False)
    dW = (dy.T @ x).astype(F32, copy=False)
    db = dy.sum(0).astype(F32) if has_bias else None
    return dx, dW, db


def relu(x):
    return np.maximum(x, F32(0))


def sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def softmax_rows(z):
    m = z.max(1, keepdims=True)
    e = np.exp(z - m, dtype=F32)
    return (e / e.sum(1, keepdims=True)).astype(F32)


BN_EPS, BN_MOMENTUM = 1e-5, 0.1
_MODE = {"training": False}  # module mode of the restated model: BatchNorm uses batch statistics when True


def set_training(flag):
    """model.train() / model.eval() of the restatement (only BatchNorm layers care, model/utils.py:153-154)."""
    _MODE["training"] = bool(flag)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the published
    algorithm, pinned by its known-answer vectors in tests/test_dropout_cpu.py) on arrays of counters: four uint32
    words out per (c0, c1, c2, c3) under the key (k0, k1)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xffffffff for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xffffffff, int(k1) & 0xffffffff
    for _ in range(rounds):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0), p1 & np.uint64(0xffffffff),
                          (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1), p0 & np.uint64(0xffffffff))
        k0, k1 = (k0 + 0x9E3779B9) & 0xffffffff, (k1 + 0xBB67AE85) & 0xffffffff
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def dropout_scale(rows, cols, p, seed, step, site, row0=0):
    """The factor nn.Dropout multiplies a [rows, cols] activation with (model/utils.py:159; torch: input * noise with
    noise = bernoulli(1 - p) / (1 - p)), under this build's mask stream (include/mmlrec.h, mml_dropout): element
    (r, c) is kept iff word c % 4 of philox4x32(counter = (row0 + r, c // 4, step, site), key = seed) >= floor(p * 2^32)
    (row0: position of the first row in the global batch of a data-parallel step)."""
    c4 = (cols + 3) // 4
    r = np.repeat(np.arange(row0, row0 + rows, dtype=np.uint64), c4)
    q = np.tile(np.arange(c4, dtype=np.uint64), rows)
    w = philox4x32(r, q, np.full(r.shape, step, dtype=np.uint64), np.full(r.shape, site, dtype=np.uint64),
                   seed & 0xffffffff, (seed >> 32) & 0xffffffff)
    words = np.stack(w, axis=1).reshape(rows, 4 * c4)[:, :cols]
    thr = min(int(float(np.float32(p)) * 4294967296.0), 0xffffffff)
    scale = F32(1.0) / (F32(1.0) - F32(p))
    return np.where(words < np.uint32(thr), F32(0), scale).astype(F32)


def set_dropout(p=0.0, seed=0, step=0):
    """Key of the dropout mask stream for the next forward (p = 0 switches dropout off): the model's dnn_dropout, the
    seed the engine holds (BaseModel.dropout_seed) and the optimizer's step counter as the forward reads it (1 in the
    first fused step)."""
    _MODE["dropout"] = dict(p=float(p), seed=int(seed), step=int(step)) if p else None


def _layer_site(name):
    import zlib
    return zlib.crc32(name.encode()) & 0xffffffff


def dnn_fwd(params, prefix, x):
    """DNN.forward (model/utils.py:146-161): [Linear -> (BatchNorm1d) -> ReLU -> Dropout]* (dropout only in training
    mode and when set_dropout() armed it; the reference's MLP model builds its blocks without it, model/mlp.py).
    BatchNorm1d as torch does it: batch mean / biased variance in training mode, with the running statistics moved by
    momentum 0.1 (unbiased variance) and num_batches_tracked incremented IN `params`; running statistics in eval mode.
    Returns output and the per-layer records for backward."""
    acts = []
    l = 0
    h = x
    while f"{prefix}.linears.{l}.weight" in params:
        W = params[f"{prefix}.linears.{l}.weight"]
        b = params[f"{prefix}.linears.{l}.bias"]
        z = linear_fwd(h, W, b)
        bn = None
        gk = f"{prefix}.bn.{l}.weight"
        if gk in params:
            g, be = params[gk], params[f"{prefix}.bn.{l}.bias"]
            if _MODE["training"]:
                z64 = z.astype(np.float64)
                mu, var = z64.mean(0), z64.var(0)
                B = z.shape[0]
                rm, rv = f"{prefix}.bn.{l}.running_mean", f"{prefix}.bn.{l}.running_var"
                params[rm] = ((1 - BN_MOMENTUM) * params[rm] + BN_MOMENTUM * mu).astype(F32)
                params[rv] = ((1 - BN_MOMENTUM) * params[rv] + BN_MOMENTUM * var * B / max(B - 1, 1)).astype(F32)
                nb = f"{prefix}.bn.{l}.num_batches_tracked"
                params[nb] = np.asarray(params[nb]) + 1
            else:
                mu = params[f"{prefix}.bn.{l}.running_mean"].astype(np.float64)
                var = params[f"{prefix}.bn.{l}.running_var"].astype(np.float64)
            rstd = 1.0 / np.sqrt(var + BN_EPS)
            xhat = ((z - mu) * rstd).astype(F32)
            bn = (xhat, rstd.astype(F32))
            z = (xhat * g + be).astype(F32)
        y = relu(z)
        acts.append((h, y, bn))
        dr = _MODE.get("dropout")
        if dr and _MODE["training"] and not prefix.startswith("mlp_layers."):
            ms = dropout_scale(y.shape[0], y.shape[1], dr["p"], dr["seed"], dr["step"], _layer_site(f"{prefix}.{l}"))
            acts[-1] = (h, y, bn, ms)
            y = (y * ms).astype(F32)
        h = y
        l += 1
    if l == 0:
        raise KeyError(prefix)
    return h, acts


def dnn_bwd(params, prefix, acts, dy, grads):
    """Backward of dnn_fwd; accumulates into grads[key]; returns d(input)."""
    for l in range(len(acts) - 1, -1, -1):
        ms = acts[l][3] if len(acts[l]) == 4 else None
        x, y, bn = acts[l][:3] if len(acts[l]) >= 3 else (*acts[l], None)
        if ms is not None:  # dropout backward: the same factor on the gradient
            dy = (dy * ms).astype(F32)
        dz = (dy * (y > 0)).astype(F32)
        if bn is not None:  # BatchNorm backward with batch statistics
            xhat, rstd = bn
            g = params[f"{prefix}.bn.{l}.weight"]
            dbeta, dgamma = dz.sum(0), (dz * xhat).sum(0)
            _acc(grads, f"{prefix}.bn.{l}.weight", dgamma.astype(F32))
            _acc(grads, f"{prefix}.bn.{l}.bias", dbeta.astype(F32))
            dz = (g * rstd * (dz - (dbeta + xhat * dgamma) / dz.shape[0])).astype(F32)
        W = params[f"{prefix}.linears.{l}.weight"]
        dx, dW, db = linear_bwd(x, W, dz)
        _acc(grads, f"{prefix}.linears.{l}.weight", dW)
        _acc(grads, f"{prefix}.linears.{l}.bias", db)
        dy = dx
    return dy


def _acc(grads, key, val):
    if key in grads:
        grads[key] = (grads[key] + val).astype(F32)
    else:
        grads[key] = np.asarray(val, dtype=F32)


def gate_mix_fwd(logits, experts):
    """softmax over experts then bmm([B,1,Ne],[B,Ne,H]) (model/mmoe.py:86-88, model/ple.py:139-140)."""
    p = softmax_rows(logits)
    mix = np.einsum("be,beh->bh", p, experts).astype(F32)
    return p, mix


def gate_mix_bwd(p, experts, dmix):
    dexperts = (p[:, :, None] * dmix[:, None, :]).astype(F32)
    dp = np.einsum("bh,beh->be", dmix, experts).astype(F32)
    dlogits = (p * (dp - (p * dp).sum(1, keepdims=True))).astype(F32)
    return dlogits, dexperts


# ----------------------------------------------------------------------------------------------
# heads + loss
# ----------------------------------------------------------------------------------------------
def head_fwd(params, t, h, w_key):
    """Linear(H->1, bias=False) then PredictionLayer: sigmoid(x + bias) (model/utils.py:242-248)."""
    logit = linear_fwd(h, params[w_key]) + params[f"out.{t}.bias"]
    return sigmoid(logit[:, 0])


def bce_sum(p, y):
    """F.binary_cross_entropy(reduction='sum') (model/basemodel.py:294-296): log terms clamped at -100."""
    lp = np.maximum(np.log(p, dtype=F32), F32(-100))
    l1p = np.maximum(np.log1p(-p, dtype=F32), F32(-100))
    return float((-(y * lp + (F32(1) - y) * l1p)).astype(np.float64).sum())


def bce_sigmoid_bwd(p, y):
    """d loss / d logit: BCE backward (p-y)/max(p(1-p),1e-12) times sigmoid' p(1-p)."""
    pq = p * (F32(1) - p)
    return ((p - y) / np.maximum(pq, F32(1e-12)) * pq).astype(F32)


def apply_mask(spec, y_pred, mask):
    """output * domain_mask[:, i] (msl) or [:, i % D] (mtmsl) (model/mmoe.py:101-106)."""
    if mask is None or spec.task_name not in ("msl", "mtmsl"):
        return y_pred
    out = y_pred.copy()
    for i in range(spec.T):
        j = i if spec.task_name == "msl" else i % spec.num_domains
        out[:, i] = out[:, i] * mask[:, j]
    return out


# ----------------------------------------------------------------------------------------------
# model forwards / backwards.  Each forward returns (probabilities [B,T], cache); each backward takes
# dlogit [B,T] and returns (grads of dense params, d dnn_input).
# ----------------------------------------------------------------------------------------------
def _towers_fwd(spec, params, streams):
    """tower DNN + final layer + PredictionLayer per task (model/mmoe.py:91-108, sharedbottom.py:58-76)."""
    T = spec.T
    has_tower = "tower_dnn.0.linears.0.weight" in params
    ps, cache = [], []
    for t in range(T):
        if has_tower:
            h, acts = dnn_fwd(params, f"tower_dnn.{t}", streams[t])
        else:
            h, acts = streams[t], []
        ps.append(head_fwd(params, t, h, f"tower_dnn_final_layer.{t}.weight"))
        cache.append((h, acts))
    return np.stack(ps, 1).astype(F32), cache


def _towers_bwd(spec, params, cache, dlogit, grads):
    dstreams = []
    for t in range(spec.T):
        h, acts = cache[t]
        dz = dlogit[:, t:t + 1]
        W = params[f"tower_dnn_final_layer.{t}.weight"]
        _acc(grads, f"out.{t}.bias", dz.sum(0))
        dh, dW, _ = linear_bwd(h, W, dz, has_bias=False)
        _acc(grads, f"tower_dnn_final_layer.{t}.weight", dW)
        if acts:
            dh = dnn_bwd(params, f"tower_dnn.{t}", acts, dh, grads)
        dstreams.append(dh)
    return dstreams


def sharedbottom_fwd(spec, params, x):
    """SharedBottom.forward (model/sharedbottom.py:52-86)."""
    bottom, acts = dnn_fwd(params, "bottom_dnn", x)
    p, tc = _towers_fwd(spec, params, [bottom] * spec.T)
    layers = {"shared_bottom_outputs": bottom}
    if tc[0][1]:
        layers["tower_outputs"] = np.stack([c[0] for c in tc], 1)
    return p, dict(acts=acts, towers=tc, layers=layers)


def sharedbottom_bwd(spec, params, cache, dlogit):
    grads = {}
    ds = _towers_bwd(spec, params, cache["towers"], dlogit, grads)
    dbottom = sum(ds).astype(F32)
    dx = dnn_bwd(params, "bottom_dnn", cache["acts"], dbottom, grads)
    return grads, dx


def mlp_fwd(spec, params, x):
    """MLP.forward (model/mlp.py:36-66): a chain of one-layer DNN blocks, ONE final Linear(H -> 1, no bias) shared by
    all tasks, then a PredictionLayer per task applied to that SAME logit tensor.  PredictionLayer adds its bias IN
    PLACE (`output = X; output += self.bias`, model/utils.py:242-245), so the biases pile up on the shared tensor:
    head t sees z + b_0 + ... + b_t.  Restated as the reference behaves."""
    n = len(spec.mc.get("dnn_hidden_units", [256, 128]))
    h, acts, layers = x, [], {}
    for i in range(n):
        h, a = dnn_fwd(params, f"mlp_layers.{i}", h)
        acts.append(a)
        layers[f"mlp_output_{i}"] = h
    z = linear_fwd(h, params["final_layer.weight"])[:, 0]
    ps = []
    for t in range(spec.T):
        z = (z + params[f"out.{t}.bias"][0]).astype(F32)
        ps.append(sigmoid(z))
    return np.stack(ps, 1).astype(F32), dict(acts=acts, last=h, layers=layers)


def mlp_bwd(spec, params, cache, dlogit):
    grads = {}
    for j in range(spec.T):  # bias j reaches the logits of heads j, j+1, ...
        _acc(grads, f"out.{j}.bias", dlogit[:, j:].sum(axis=(0, 1), keepdims=False).reshape(1))
    dz = dlogit.sum(1, keepdims=True).astype(F32)  # every head reads the same logit
    dh, dW, _ = linear_bwd(cache["last"], params["final_layer.weight"], dz, has_bias=False)
    _acc(grads, "final_layer.weight", dW)
    for i in reversed(range(len(cache["acts"]))):
        dh = dnn_bwd(params, f"mlp_layers.{i}", cache["acts"][i], dh, grads)
    return grads, dh


def snr_z(u, alpha, beta=0.9, gamma=-0.1, eps=1.1):
    """Hard-concrete routing coefficients and their derivatives (model/snr_trans.py:40-43)."""
    u64, a64 = u.astype(np.float64), float(alpha[0])
    s = 1.0 / (1.0 + np.exp(-(np.log(u64) - np.log(1 - u64) + np.log(a64) / beta)))
    sb = s * (eps - gamma) + gamma
    live = (sb > 0) & (sb <= 1)
    z = np.where(sb <= 0, 0.0, np.where(sb > 1, 1.0, sb))
    ds = np.where(live, (eps - gamma) * s * (1 - s), 0.0)
    return z.astype(F32), (ds * (1 / u64 + 1 / (1 - u64))).astype(F32), (ds / (a64 * beta)).astype(F32)


def snr_trans_fwd(spec, params, x, frozen):
    """SNR_trans.forward (model/snr_trans.py:120-163): per level Ne one-layer experts, then the routing gate
    out_o = sum_j z_oj (x_j @ M_oj) with the frozen trans_matrix blocks M (never registered: :30-34)."""
    Ne = int(spec.mc.get("num_experts", 4))
    units = spec.mc.get("expert_dnn_hidden_units", [256, 128])
    # MSSM (model/mssm.py:128-179) is the same network with other key names and per-COLUMN coefficients whose seeds u
    # are unregistered too ([No, Ne, d], from the fixture's frozen set); a [d] z scales the columns of x_j @ M_oj
    dct, stem = ("mssm", "mssm.expert") if spec.model_name == "mssm" else ("trans", "trans.trans")
    ins, levels = [x] * Ne, []
    for i in range(len(units)):
        hs, acts = [], []
        for j in range(Ne):
            h, a = dnn_fwd(params, f"{stem}{i + 1}.{j}", ins[j])
            hs.append(h)
            acts.append(a)
        M = frozen[f"{dct}.gate{i + 1}.trans_matrix"]  # [No, Ne, d, d]
        ukey = f"{dct}.gate{i + 1}.u"
        z, dzu, dza = snr_z(params[ukey] if ukey in params else frozen[ukey], params[f"{dct}.gate{i + 1}.alpha"])
        prods = [[(hs[j] @ M[o, j]).astype(F32) for j in range(Ne)] for o in range(M.shape[0])]
        outs = [sum(prods[o][j] * z[o, j] for j in range(Ne)).astype(F32) for o in range(M.shape[0])]
        levels.append(dict(hs=hs, acts=acts, M=M, z=z, dzu=dzu, dza=dza, prods=prods))
        ins = outs
    p, tc = _towers_fwd(spec, params, ins)
    return p, dict(levels=levels, towers=tc, x=x, layers={})


def snr_trans_bwd(spec, params, cache, dlogit):
    Ne = int(spec.mc.get("num_experts", 4))
    grads = {}
    douts = _towers_bwd(spec, params, cache["towers"], dlogit, grads)
    dx = np.zeros_like(cache["x"])
    nlev = len(cache["levels"])
    for i in reversed(range(nlev)):
        c = cache["levels"][i]
        M, z = c["M"], c["z"]
        No = M.shape[0]
        per_col = z.ndim == 3
        dz = np.array([[(douts[o].astype(np.float64) * c["prods"][o][j]).sum(0) if per_col else
                        float((douts[o].astype(np.float64) * c["prods"][o][j]).sum()) for j in range(Ne)]
                       for o in range(No)])
        dct, stem = ("mssm", "mssm.expert") if spec.model_name == "mssm" else ("trans", "trans.trans")
        if not per_col:  # SNR-trans: u is a registered parameter; MSSM's u never learns
            _acc(grads, f"{dct}.gate{i + 1}.u", (dz * c["dzu"]).astype(F32))
        _acc(grads, f"{dct}.gate{i + 1}.alpha", np.array([(dz * c["dza"]).sum()], dtype=F32))
        dhs = [sum((douts[o] * z[o, j]) @ M[o, j].T for o in range(No)).astype(F32) for j in range(Ne)]
        dins = [dnn_bwd(params, f"{stem}{i + 1}.{j}", c["acts"][j], dhs[j], grads) for j in range(Ne)]
        if i == 0:
            dx = sum(dins).astype(F32)
        else:
            douts = dins
    return grads, dx


def aitm_fwd(spec, params, x):
    """AITM.forward (model/aitm.py:75-111): bottoms, then for task i >= 1 the attention over the two tokens
    p = g_{i-1}(feat[i-1]) and q = feat[i] with V = h1(.), K = h2(.), Q = h3(.) (:84-93), towers."""
    T = spec.T
    H = spec.mc.get("expert_dnn_hidden_units", [256, 128])[-1]
    sq = F32(np.sqrt(H))
    feat, bacts, att = [], [], []
    for i in range(T):
        h, a = dnn_fwd(params, f"bottom.{i}", x)
        feat.append(h)
        bacts.append(a)
    raw = list(feat)
    for i in range(1, T):
        p = linear_fwd(feat[i - 1], params[f"g.{i - 1}.weight"], params[f"g.{i - 1}.bias"])
        q = feat[i]
        tok = np.stack([p, q], 1)  # [B,2,H]
        V = (tok @ params["h1.weight"].T + params["h1.bias"]).astype(F32)
        K = (tok @ params["h2.weight"].T + params["h2.bias"]).astype(F32)
        Q = (tok @ params["h3.weight"].T + params["h3.bias"]).astype(F32)
        s = ((K * Q).sum(2, keepdims=True, dtype=F32) / sq).astype(F32)
        a = softmax_rows(s[:, :, 0])[:, :, None]
        att.append(dict(prev=feat[i - 1], tok=tok, V=V, K=K, Q=Q, a=a))
        feat[i] = (a * V).sum(1).astype(F32)
    probs, tc = _towers_fwd(spec, params, feat)
    return probs, dict(bacts=bacts, att=att, towers=tc, sq=sq, layers={})


def aitm_bwd(spec, params, cache, dlogit):
    T = spec.T
    grads = {}
    dfeat = _towers_bwd(spec, params, cache["towers"], dlogit, grads)
    for i in reversed(range(1, T)):
        c = cache["att"][i - 1]
        a, V, K, Q, tok = c["a"], c["V"], c["K"], c["Q"], c["tok"]
        do = dfeat[i][:, None, :]
        dV = (a * do).astype(F32)
        da = (do * V).sum(2, keepdims=True)
        ds = (a * (da - (a * da).sum(1, keepdims=True)) / cache["sq"]).astype(F32)
        dK, dQ = (ds * Q).astype(F32), (ds * K).astype(F32)
        dtok = np.zeros_like(tok)
        B2 = tok.reshape(-1, tok.shape[2])
        for nm, d in (("h1", dV), ("h2", dK), ("h3", dQ)):
            d2 = d.reshape(-1, d.shape[2])
            _acc(grads, f"{nm}.weight", (d2.T @ B2).astype(F32))
            _acc(grads, f"{nm}.bias", d2.sum(0).astype(F32))
            dtok += (d @ params[f"{nm}.weight"]).astype(F32)
        dp, dq = dtok[:, 0], dtok[:, 1]
        dprev, dW, db = linear_bwd(c["prev"], params[f"g.{i - 1}.weight"], dp, has_bias=True)
        _acc(grads, f"g.{i - 1}.weight", dW)
        _acc(grads, f"g.{i - 1}.bias", db)
        dfeat[i] = dq
        dfeat[i - 1] = (dfeat[i - 1] + dprev).astype(F32)
    dx = None
    for i in range(T):
        d = dnn_bwd(params, f"bottom.{i}", cache["bacts"][i], dfeat[i], grads)
        dx = d if dx is None else dx + d
    return grads, dx.astype(F32)


def cross_stitch_fwd(spec, params, x):
    """CrossStitch.forward (model/cross_stitch.py:82-121): shared layer, then per level T one-layer DNNs and a stitch
    (cat of the T activations times cross_stitch_weight [in, out], model/cross_stitch.py:17-19), then towers."""
    T = spec.T
    units = spec.mc.get("dnn_hidden_units", [256, 128])
    sh, a_sh = dnn_fwd(params, "shared_layer", x)
    ins, levels = [sh] * T, []
    for i, d in enumerate(units):
        hs, acts = [], []
        for j in range(T):
            h, a = dnn_fwd(params, f"cross_stitch.task_layer_{i}.{j}", ins[j])
            hs.append(h)
            acts.append(a)
        cat = np.concatenate(hs, 1)
        mix = (cat @ params[f"cross_stitch.gate_{i}.cross_stitch_weight"]).astype(F32)
        levels.append((acts, cat))
        ins = [mix[:, j * d:(j + 1) * d] for j in range(T)]
    p, tc = _towers_fwd(spec, params, ins)
    layers = {"cross_stitch_outputs": np.stack(ins, 1)}
    if tc[0][1]:
        layers["tower_outputs"] = np.stack([c[0] for c in tc], 1)
    return p, dict(a_sh=a_sh, levels=levels, towers=tc, layers=layers)


def cross_stitch_bwd(spec, params, cache, dlogit):
    T = spec.T
    units = spec.mc.get("dnn_hidden_units", [256, 128])
    grads = {}
    dins = _towers_bwd(spec, params, cache["towers"], dlogit, grads)
    for i in reversed(range(len(units))):
        d = units[i]
        acts, cat = cache["levels"][i]
        dmix = np.concatenate(dins, 1).astype(F32)
        W = params[f"cross_stitch.gate_{i}.cross_stitch_weight"]
        _acc(grads, f"cross_stitch.gate_{i}.cross_stitch_weight", (cat.T @ dmix).astype(F32))
        dcat = (dmix @ W.T).astype(F32)
        dins = [dnn_bwd(params, f"cross_stitch.task_layer_{i}.{j}", acts[j], dcat[:, j * d:(j + 1) * d], grads)
                for j in range(T)]
    dsh = sum(dins).astype(F32)  # every task's first layer reads the same shared activation
    dx = dnn_bwd(params, "shared_layer", cache["a_sh"], dsh, grads)
    return grads, dx


def esmm_fwd(spec, params, x):
    """ESMM.forward (model/esmm.py:46-71): ctr / cvr towers, both through the single PredictionLayer `out`
    (model/basemodel.py:132), outputs [ctr, ctr * cvr]."""
    hc, ac = dnn_fwd(params, "ctr_dnn", x)
    hv, av = dnn_fwd(params, "cvr_dnn", x)
    b = params["out.bias"]
    c = sigmoid((linear_fwd(hc, params["ctr_dnn_final_layer.weight"]) + b)[:, 0])
    v = sigmoid((linear_fwd(hv, params["cvr_dnn_final_layer.weight"]) + b)[:, 0])
    p = np.stack([c, c * v], 1).astype(F32)
    return p, dict(ac=ac, av=av, hc=hc, hv=hv, c=c, v=v, layers={"target0_output": hc, "target1_output": hv})


def esmm_bwd(spec, params, cache, dprob):
    """dprob = dLoss / d[ctr, ctcvr] (NOT d/dlogit: the second output is a product, model/esmm.py:58)."""
    c, v = cache["c"], cache["v"]
    dc = (dprob[:, 0] + dprob[:, 1] * v).astype(F32)
    dv = (dprob[:, 1] * c).astype(F32)
    dzc = (dc * c * (F32(1) - c)).astype(F32)[:, None]
    dzv = (dv * v * (F32(1) - v)).astype(F32)[:, None]
    grads = {}
    _acc(grads, "out.bias", (dzc.sum(0) + dzv.sum(0)).astype(F32))
    dhc, dW, _ = linear_bwd(cache["hc"], params["ctr_dnn_final_layer.weight"], dzc, has_bias=False)
    _acc(grads, "ctr_dnn_final_layer.weight", dW)
    dhv, dW, _ = linear_bwd(cache["hv"], params["cvr_dnn_final_layer.weight"], dzv, has_bias=False)
    _acc(grads, "cvr_dnn_final_layer.weight", dW)
    dx = dnn_bwd(params, "ctr_dnn", cache["ac"], dhc, grads) + dnn_bwd(params, "cvr_dnn", cache["av"], dhv, grads)
    return grads, dx.astype(F32)


def escm_fwd(spec, params, x):
    """ESCM.forward (model/escm.py:74-96): ESMM's towers, outputs [ctr, cvr, ctr * cvr]."""
    p2, cache = esmm_fwd(spec, params, x)
    c, v = cache["c"], cache["v"]
    cache["layers"] = {}
    return np.stack([c, v, c * v], 1).astype(F32), cache


def escm_loss_and_dprob(p, y, cf_w=0.1, global_w=1.0):
    """The ESCM branch of BaseModel.fit (model/basemodel.py:284-292) with counterfact_ipw (model/escm.py:98-112):
    loss_0 = BCE_sum(ctr, y0); loss_1 = BCE_sum(cvr, y1) (a scalar); loss_2 = BCE_sum(ctcvr, y1);
    ips = clip(1 / max(ctr * sum(y0), 1e-6), -15, 15) * B;  loss = loss_0 + cf_w * mean(loss_1 * ips * y0) + global_w * loss_2.
    (`ips.stop_gradient = True` is a no-op in torch: the gradient flows through ips.)  Returns (loss, dLoss/d[ctr, cvr, ctcvr])
    with the ctr / cvr entries holding ONLY the direct terms (the product's chain rule is esmm-style, done by the caller)."""
    c, v, p2 = p[:, 0].astype(F32), p[:, 1].astype(F32), p[:, 2].astype(F32)
    y0, y1 = y[:, 0].astype(F32), y[:, 1].astype(F32)
    B = F32(len(c))
    N = F32(y0.sum())
    L0, L1, L2 = bce_sum(c, y0), bce_sum(v, y1), bce_sum(p2, y1)
    ps = np.maximum(c * N, F32(1e-6))
    r = (F32(1) / ps).astype(F32)
    clip = np.clip(r, F32(-15), F32(15))
    S = float((y0 * clip).sum())
    loss = L0 + cf_w * (L1 * S) + global_w * L2
    inside = (c * N > F32(1e-6)) & (r >= F32(-15)) & (r <= F32(15))
    dclip = np.where(inside, -N * r * r, F32(0)).astype(F32)
    d = np.zeros((len(c), 3), dtype=F32)
    d[:, 0] = bce_prob_bwd(c, y0) + F32(cf_w * L1) * y0 * dclip
    d[:, 1] = F32(cf_w * S) * bce_prob_bwd(v, y1)
    d[:, 2] = F32(global_w) * bce_prob_bwd(p2, y1)
    return float(loss), d


def escm_bwd(spec, params, cache, dprob3):
    c, v = cache["c"], cache["v"]
    d2 = np.stack([dprob3[:, 0] + dprob3[:, 2] * v, dprob3[:, 1] + dprob3[:, 2] * c], 1).astype(F32)
    # esmm_bwd expects d/d[ctr, ctcvr]: feed d/d ctr directly and fold d/d cvr through a unit "ctcvr" slot
    dzc = (d2[:, 0] * c * (F32(1) - c)).astype(F32)[:, None]
    dzv = (d2[:, 1] * v * (F32(1) - v)).astype(F32)[:, None]
    grads = {}
    _acc(grads, "out.bias", (dzc.sum(0) + dzv.sum(0)).astype(F32))
    dhc, dW, _ = linear_bwd(cache["hc"], params["ctr_dnn_final_layer.weight"], dzc, has_bias=False)
    _acc(grads, "ctr_dnn_final_layer.weight", dW)
    dhv, dW, _ = linear_bwd(cache["hv"], params["cvr_dnn_final_layer.weight"], dzv, has_bias=False)
    _acc(grads, "cvr_dnn_final_layer.weight", dW)
    dx = dnn_bwd(params, "ctr_dnn", cache["ac"], dhc, grads) + dnn_bwd(params, "cvr_dnn", cache["av"], dhv, grads)
    return grads, dx.astype(F32)


def apg_fwd(spec, params, x):
    """APG.forward (model/apg.py:146-193) with APGLayer.forward's use_uv_shared / no-mf_p branch (:100-106, :116-118):
    per layer  o1 = x Wnk + bnk;  W_b = reshape(Linear_kk(s_b)) [k,k], c_b = Linear_bias(s_b);  o2_b = o1_b W_b + c_b;
    out = relu(o2 Wkm + bkm).  s = the detached scene embedding (:152-153).  The per-sample weights ARE materialised
    here (the checker follows the reference's formulation, not the GEMM rewrite of csrc/apg.hip)."""
    E = spec.emb
    pos = spec.sparse_names.index(spec.dc["scene_feature"])
    s = x[:, pos * E:(pos + 1) * E]
    h, acts, layers = x, [], {}
    nl = len(spec.mc.get("dnn_hidden_units", [256, 128]))
    for i in range(nl):
        pf = f"apg_layers.{i}"
        Wnk, bnk = params[f"{pf}.shared_weight_nk"], params[f"{pf}.shared_bias_nk"]
        Wkm, bkm = params[f"{pf}.shared_weight_km"], params[f"{pf}.shared_bias_km"]
        k = Wnk.shape[1]
        Wg = linear_fwd(s, params[f"{pf}.specific_weight_kk.linears.0.weight"],
                        params[f"{pf}.specific_weight_kk.linears.0.bias"]).reshape(-1, k, k)
        cg = linear_fwd(s, params[f"{pf}.specific_bias_kk.linears.0.weight"],
                        params[f"{pf}.specific_bias_kk.linears.0.bias"])
        o1 = (h @ Wnk + bnk).astype(F32)
        o2 = (np.einsum("bi,bij->bj", o1, Wg) + cg).astype(F32)
        o3 = relu((o2 @ Wkm + bkm).astype(F32))
        acts.append((h, o1, Wg, o2, o3))
        layers[f"apg_output_{i}"] = o3
        h = o3
    ps = [sigmoid((linear_fwd(h, params[f"final_layer.{t}.weight"]) + params[f"out.{t}.bias"])[:, 0])
          for t in range(spec.T)]
    return np.stack(ps, 1).astype(F32), dict(acts=acts, s=s, h=h, layers=layers)


def apg_bwd(spec, params, cache, dlogit):
    grads = {}
    h = cache["h"]
    dh = np.zeros_like(h)
    for t in range(spec.T):
        dz = dlogit[:, t:t + 1]
        _acc(grads, f"out.{t}.bias", dz.sum(0))
        d, dW, _ = linear_bwd(h, params[f"final_layer.{t}.weight"], dz, has_bias=False)
        _acc(grads, f"final_layer.{t}.weight", dW)
        dh = dh + d
    s = cache["s"]
    for i in reversed(range(len(cache["acts"]))):
        pf = f"apg_layers.{i}"
        hin, o1, Wg, o2, o3 = cache["acts"][i]
        k = o1.shape[1]
        d3 = (dh * (o3 > 0)).astype(F32)
        _acc(grads, f"{pf}.shared_weight_km", (o2.T @ d3).astype(F32))
        _acc(grads, f"{pf}.shared_bias_km", d3.sum(0))
        d2 = (d3 @ params[f"{pf}.shared_weight_km"].T).astype(F32)
        dWg = np.einsum("bi,bj->bij", o1, d2).reshape(len(o1), k * k).astype(F32)   # d/d generated weights
        _, dW, db = linear_bwd(s, params[f"{pf}.specific_weight_kk.linears.0.weight"], dWg)
        _acc(grads, f"{pf}.specific_weight_kk.linears.0.weight", dW)
        _acc(grads, f"{pf}.specific_weight_kk.linears.0.bias", db)
        _, dW, db = linear_bwd(s, params[f"{pf}.specific_bias_kk.linears.0.weight"], d2)
        _acc(grads, f"{pf}.specific_bias_kk.linears.0.weight", dW)
        _acc(grads, f"{pf}.specific_bias_kk.linears.0.bias", db)
        d1 = np.einsum("bj,bij->bi", d2, Wg).astype(F32)
        _acc(grads, f"{pf}.shared_weight_nk", (hin.T @ d1).astype(F32))
        _acc(grads, f"{pf}.shared_bias_nk", d1.sum(0))
        dh = (d1 @ params[f"{pf}.shared_weight_nk"].T).astype(F32)
    return grads, dh.astype(F32)


def bce_prob_bwd(p, y):
    """d BCE / d p as PyTorch evaluates it: (p - y) / max(p (1 - p), 1e-12)."""
    return ((p - y) / np.maximum(p * (F32(1) - p), F32(1e-12))).astype(F32)


def mmoe_fwd(spec, params, x):
    """MMOE.forward (model/mmoe.py:65-119)."""
    Ne = int(spec.mc.get("num_experts", 4))
    eo, eacts = [], []
    for e in range(Ne):
        h, a = dnn_fwd(params, f"expert_dnn.{e}", x)
        eo.append(h)
        eacts.append(a)
    experts = np.stack(eo, 1)  # [B,Ne,H]
    has_gate_dnn = "gate_dnn.0.linears.0.weight" in params
    gates, mixes = [], []
    for t in range(spec.T):
        if has_gate_dnn:
            g, gacts = dnn_fwd(params, f"gate_dnn.{t}", x)
        else:
            g, gacts = x, []
        logits = linear_fwd(g, params[f"gate_dnn_final_layer.{t}.weight"])
        p, mix = gate_mix_fwd(logits, experts)
        gates.append((g, gacts, p))
        mixes.append(mix)
    probs, tc = _towers_fwd(spec, params, mixes)
    layers = {"expert_outputs": experts, "mmoe_outputs": np.stack(mixes, 1),
              "gate_outputs": np.stack([g[2] for g in gates], 1)}
    if tc[0][1]:
        layers["tower_outputs"] = np.stack([c[0] for c in tc], 1)
    return probs, dict(x=x, experts=experts, eacts=eacts, gates=gates, towers=tc, layers=layers)


def mmoe_bwd(spec, params, cache, dlogit):
    grads = {}
    Ne = cache["experts"].shape[1]
    dmix = _towers_bwd(spec, params, cache["towers"], dlogit, grads)
    dexperts = np.zeros_like(cache["experts"])
    dx = np.zeros_like(cache["x"])
    for t in range(spec.T):
        g, gacts, p = cache["gates"][t]
        dlogits, de = gate_mix_bwd(p, cache["experts"], dmix[t])
        dexperts += de
        W = params[f"gate_dnn_final_layer.{t}.weight"]
        dg, dW, _ = linear_bwd(g, W, dlogits, has_bias=False)
        _acc(grads, f"gate_dnn_final_layer.{t}.weight", dW)
        if gacts:
            dx += dnn_bwd(params, f"gate_dnn.{t}", gacts, dg, grads)
        else:
            dx += dg
    for e in range(Ne):
        dx += dnn_bwd(params, f"expert_dnn.{e}", cache["eacts"][e], dexperts[:, e], grads)
    return grads, dx.astype(F32)


def hmoe_fwd(spec, params, x):
    """HMOE.forward (model/hmoe.py:83-153): the MMoE body up to the tower DNNs, then per task a softmax task-weight gate
    over the T tower outputs (the others detached, :124-129) in front of the final Linear + PredictionLayer."""
    T = spec.T
    p0, c = mmoe_fwd_body(spec, params, x)
    towers, tws = c["tower_outs"], []
    has_tw = "task_weight.0.linears.0.weight" in params
    stack = np.stack(towers, 1)  # [B,T,Ht]
    ps, outs = [], []
    for i in range(T):
        if has_tw:
            g, gacts = dnn_fwd(params, f"task_weight.{i}", x)
        else:
            g, gacts = x, []
        logits = linear_fwd(g, params[f"task_weight_final_layer.{i}.weight"])
        w, mix = gate_mix_fwd(logits, stack)
        tws.append((g, gacts, w))
        outs.append(mix)
        ps.append(head_fwd(params, i, mix, f"tower_dnn_final_layer.{i}.weight"))
    c.update(tws=tws, task_outs=outs, tstack=stack)
    return np.stack(ps, 1).astype(F32), c


def mmoe_fwd_body(spec, params, x):
    """experts, gates, mixtures and tower DNNs of MMoE / HMoE without the heads."""
    Ne = int(spec.mc.get("num_experts", 4))
    eo, eacts = [], []
    for e in range(Ne):
        h, a = dnn_fwd(params, f"expert_dnn.{e}", x)
        eo.append(h)
        eacts.append(a)
    experts = np.stack(eo, 1)
    has_gate_dnn = "gate_dnn.0.linears.0.weight" in params
    gates, mixes = [], []
    for t in range(spec.T):
        g, gacts = dnn_fwd(params, f"gate_dnn.{t}", x) if has_gate_dnn else (x, [])
        p, mix = gate_mix_fwd(linear_fwd(g, params[f"gate_dnn_final_layer.{t}.weight"]), experts)
        gates.append((g, gacts, p))
        mixes.append(mix)
    has_tower = "tower_dnn.0.linears.0.weight" in params
    touts, tacts = [], []
    for t in range(spec.T):
        h, a = dnn_fwd(params, f"tower_dnn.{t}", mixes[t]) if has_tower else (mixes[t], [])
        touts.append(h)
        tacts.append(a)
    layers = {"expert_outputs": experts, "mmoe_outputs": np.stack(mixes, 1),
              "gate_outputs": np.stack([g[2] for g in gates], 1)}
    if has_tower:
        layers["tower_outputs"] = np.stack(touts, 1)
    return None, dict(x=x, experts=experts, eacts=eacts, gates=gates, tower_outs=touts, tower_acts=tacts,
                      layers=layers)


def hmoe_bwd(spec, params, cache, dlogit):
    T = spec.T
    grads = {}
    x = cache["x"]
    dx = np.zeros_like(x)
    dtower = [None] * T
    for i in range(T):
        g, gacts, w = cache["tws"][i]
        dz = dlogit[:, i:i + 1]
        _acc(grads, f"out.{i}.bias", dz.sum(0))
        dmix, dW, _ = linear_bwd(cache["task_outs"][i], params[f"tower_dnn_final_layer.{i}.weight"], dz, has_bias=False)
        _acc(grads, f"tower_dnn_final_layer.{i}.weight", dW)
        dlogits, dstack = gate_mix_bwd(w, cache["tstack"], dmix)
        dtower[i] = dstack[:, i]  # towers j != i are detached in task i's mixture
        Wt = params[f"task_weight_final_layer.{i}.weight"]
        dg, dWt, _ = linear_bwd(g, Wt, dlogits, has_bias=False)
        _acc(grads, f"task_weight_final_layer.{i}.weight", dWt)
        dx += dnn_bwd(params, f"task_weight.{i}", gacts, dg, grads) if gacts else dg
    Ne = cache["experts"].shape[1]
    dexperts = np.zeros_like(cache["experts"])
    for t in range(T):
        dm = dnn_bwd(params, f"tower_dnn.{t}", cache["tower_acts"][t], dtower[t], grads) if cache["tower_acts"][t] \
            else dtower[t]
        g, gacts, p = cache["gates"][t]
        dlogits, de = gate_mix_bwd(p, cache["experts"], dm)
        dexperts += de
        dg, dW, _ = linear_bwd(g, params[f"gate_dnn_final_layer.{t}.weight"], dlogits, has_bias=False)
        _acc(grads, f"gate_dnn_final_layer.{t}.weight", dW)
        dx += dnn_bwd(params, f"gate_dnn.{t}", gacts, dg, grads) if gacts else dg
    for e in range(Ne):
        dx += dnn_bwd(params, f"expert_dnn.{e}", cache["eacts"][e], dexperts[:, e], grads)
    return grads, dx.astype(F32)


def ple_fwd(spec, params, x):
    """PLE.forward / cgc_net (model/ple.py:107-198).  Quirks kept: `specific_expert_num` shared
    experts are constructed but only `shared_expert_num` are used (ple.py:47 vs :120-121); the last level's
    shared gate output is never consumed."""
    T = spec.T
    S = int(spec.mc.get("specific_expert_num", 3))
    Sh = int(spec.mc.get("shared_expert_num", 1))
    L = int(spec.mc.get("num_levels", 1))
    has_gate_dnn = "specific_gate_dnn.0.0.0.linears.0.weight" in params
    inputs = [x] * (T + 1)
    levels = []
    outs_per_level = []
    for lv in range(L):
        spec_out, spec_acts = [], []
        for i in range(T):
            for j in range(S):
                h, a = dnn_fwd(params, f"specific_experts.{lv}.{i}.{j}", inputs[i])
                spec_out.append(h)
                spec_acts.append(a)
        sh_out, sh_acts = [], []
        for k in range(Sh):
            h, a = dnn_fwd(params, f"shared_experts.{lv}.0.{k}", inputs[-1])
            sh_out.append(h)
            sh_acts.append(a)
        gates = []
        outs = []
        for i in range(T + 1):
            if i < T:
                ex = np.stack(spec_out[i * S:(i + 1) * S] + sh_out, 1)
                gpre = f"specific_gate_dnn.{lv}.{i}.0"
                fkey = f"specific_gate_dnn_final_layer.{lv}.{i}.weight"
            else:
                ex = np.stack(spec_out + sh_out, 1)
                gpre = f"shared_gate_dnn.{lv}"
                fkey = f"shared_gate_dnn_final_layer.{lv}.weight"
            if has_gate_dnn:
                g, gacts = dnn_fwd(params, gpre, inputs[i])
            else:
                g, gacts = inputs[i], []
            p, mix = gate_mix_fwd(linear_fwd(g, params[fkey]), ex)
            gates.append((g, gacts, p, ex, gpre, fkey))
            outs.append(mix)
        levels.append(dict(inputs=inputs, spec_acts=spec_acts, sh_acts=sh_acts, gates=gates))
        outs_per_level.append(np.stack(outs, 1))
        inputs = outs
    probs, tc = _towers_fwd(spec, params, inputs[:T])
    layers = {f"ple_output_{i}": o for i, o in enumerate(outs_per_level)}
    if tc[0][1]:
        layers["tower_outputs"] = np.stack([c[0] for c in tc], 1)
    return probs, dict(levels=levels, towers=tc, layers=layers, T=T, S=S, Sh=Sh)


def ple_bwd(spec, params, cache, dlogit):
    grads = {}
    T, S, Sh = cache["T"], cache["S"], cache["Sh"]
    dstreams = _towers_bwd(spec, params, cache["towers"], dlogit, grads) + [None]
    for lv in range(len(cache["levels"]) - 1, -1, -1):
        L = cache["levels"][lv]
        inputs = L["inputs"]
        dinputs = [np.zeros_like(inputs[0]) for _ in range(T + 1)]
        dspec = [None] * (T * S)
        dsh = [None] * Sh

        def add(lst, i, v):
            lst[i] = v if lst[i] is None else (lst[i] + v).astype(F32)

        for i in range(T + 1):
            if dstreams[i] is None:  # last level's shared gate: output unused (ple.py:146-152)
                continue
            g, gacts, p, ex, gpre, fkey = L["gates"][i]
            dlogits, de = gate_mix_bwd(p, ex, dstreams[i])
            if i < T:
                for j in range(S):
                    add(dspec, i * S + j, de[:, j])
                for k in range(Sh):
                    add(dsh, k, de[:, S + k])
            else:
                for j in range(T * S):
                    add(dspec, j, de[:, j])
                for k in range(Sh):
                    add(dsh, k, de[:, T * S + k])
            dg, dW, _ = linear_bwd(g, params[fkey], dlogits, has_bias=False)
            _acc(grads, fkey, dW)
            if gacts:
                dinputs[i] += dnn_bwd(params, gpre, gacts, dg, grads)
            else:
                dinputs[i] += dg
        for i in range(T):
            for j in range(S):
                if dspec[i * S + j] is not None:
                    dinputs[i] += dnn_bwd(params, f"specific_experts.{lv}.{i}.{j}", L["spec_acts"][i * S + j],
                                          dspec[i * S + j], grads)
        for k in range(Sh):
            if dsh[k] is not None:
                dinputs[T] += dnn_bwd(params, f"shared_experts.{lv}.0.{k}", L["sh_acts"][k], dsh[k], grads)
        dstreams = dinputs
    dx = sum(dstreams).astype(F32)  # level-0 inputs are all dnn_input (ple.py:161)
    return grads, dx


def _star_w(params, frozen, pfx, li, d, T):
    """SharedSpecificLinear (model/utils.py:163-223): only the LAST domain's specific weight is a registered
    parameter; domains d < T-1 use frozen tensors."""
    if d == T - 1:
        return params[f"{pfx}.{li}.specific_weight"], params[f"{pfx}.{li}.specific_bias"]
    return frozen[f"{pfx}.{li}.specific_weights.{d}"], frozen[f"{pfx}.{li}.specific_biases.{d}"]


def star_fwd(spec, params, x, frozen):
    """STAR.forward (model/star.py:39-80): h = relu(h @ (Wspec_i * Wsh) + bspec_i + bsh) per layer,
    then the per-head final star layer [H->1] and PredictionLayer.  Weights are stored [in,out]."""
    T = spec.T
    nl = len(spec.mc.get("dnn_hidden_units", [256, 128]))
    ps, heads = [], []
    star_layers = [[None] * T for _ in range(nl)]
    for i in range(T):
        h = x
        acts = []
        for j in range(nl):
            ws, bs = _star_w(params, frozen, "linears", j, i, T)
            wsh, bsh = params[f"linears.{j}.shared_weight"], params[f"linears.{j}.shared_bias"]
            y = relu((h @ (ws * wsh) + bs + bsh).astype(F32))
            acts.append((h, y, ws))
            star_layers[j][i] = y
            h = y
        ws, bs = _star_w(params, frozen, "final_layers", i, i, T)
        wsh, bsh = params[f"final_layers.{i}.shared_weight"], params[f"final_layers.{i}.shared_bias"]
        logit = (h @ (ws * wsh) + bs + bsh).astype(F32) + params[f"out.{i}.bias"]
        ps.append(sigmoid(logit[:, 0]))
        heads.append((h, acts, ws))
    layers = {f"star_output_{j}": np.stack(star_layers[j], 1) for j in range(nl)}
    return np.stack(ps, 1).astype(F32), dict(heads=heads, layers=layers, x=x)


def star_bwd(spec, params, cache, dlogit):
    grads = {}
    T = spec.T
    dx = np.zeros_like(cache["x"])
    for i in range(T):
        h, acts, ws_f = cache["heads"][i]
        dz = dlogit[:, i:i + 1]
        _acc(grads, f"out.{i}.bias", dz.sum(0))
        wsh = params[f"final_layers.{i}.shared_weight"]
        dWeff = (h.T @ dz).astype(F32)
        _acc(grads, f"final_layers.{i}.shared_weight", dWeff * ws_f)
        _acc(grads, f"final_layers.{i}.shared_bias", dz.sum(0))
        if i == T - 1:
            _acc(grads, f"final_layers.{i}.specific_weight", dWeff * wsh)
            _acc(grads, f"final_layers.{i}.specific_bias", dz.sum(0))
        dh = (dz @ (ws_f * wsh).T).astype(F32)
        for j in range(len(acts) - 1, -1, -1):
            xin, y, ws = acts[j]
            wsh = params[f"linears.{j}.shared_weight"]
            dzz = dh * (y > 0)
            dWeff = (xin.T @ dzz).astype(F32)
            _acc(grads, f"linears.{j}.shared_weight", dWeff * ws)
            _acc(grads, f"linears.{j}.shared_bias", dzz.sum(0))
            if i == T - 1:
                _acc(grads, f"linears.{j}.specific_weight", dWeff * wsh)
                _acc(grads, f"linears.{j}.specific_bias", dzz.sum(0))
            dh = (dzz @ (ws * wsh).T).astype(F32)
        dx += dh
    return grads, dx.astype(F32)


def _gatenn_fwd(params, pfx, gin):
    """GateNN (model/pepnet.py:8-32): 2 * sigmoid(Linear(relu(Linear(x))))."""
    h1 = relu(linear_fwd(gin, params[f"{pfx}.gate.0.weight"], params[f"{pfx}.gate.0.bias"]))
    s = sigmoid(linear_fwd(h1, params[f"{pfx}.gate.2.weight"], params[f"{pfx}.gate.2.bias"]))
    return (F32(2) * s).astype(F32), (gin, h1, s)


def _gatenn_bwd(params, pfx, c, dgw, grads):
    gin, h1, s = c
    dz2 = (F32(2) * dgw * s * (F32(1) - s)).astype(F32)
    dh1, dW2, db2 = linear_bwd(h1, params[f"{pfx}.gate.2.weight"], dz2)
    _acc(grads, f"{pfx}.gate.2.weight", dW2)
    _acc(grads, f"{pfx}.gate.2.bias", db2)
    dz1 = dh1 * (h1 > 0)
    _, dW1, db1 = linear_bwd(gin, params[f"{pfx}.gate.0.weight"], dz1)  # gate input is detached
    _acc(grads, f"{pfx}.gate.0.weight", dW1)
    _acc(grads, f"{pfx}.gate.0.bias", db1)


def pepnet_fwd(spec, params, x):
    """PepNet.forward (model/pepnet.py:121-157) with user_sf/item_sf empty: task_sf_emb = scene_emb."""
    E = spec.emb
    scene_pos = spec.sparse_names.index(spec.dc["scene_feature"])  # pepnet.py:97,:126 (offset == position)
    scene = x[:, scene_pos * E:(scene_pos + 1) * E]
    fg, fgc = _gatenn_fwd(params, "feature_gate", np.concatenate([x, scene], 1))
    x2 = (fg * x).astype(F32)
    gin = np.concatenate([x2, scene], 1)
    nl = len(spec.mc.get("dnn_hidden_units", [256, 128]))
    ps, tasks = [], []
    for t in range(spec.T):
        h = x2
        lay = []
        for l in range(nl + 1):
            gw, gc = _gatenn_fwd(params, f"ppn.{t}.gate_layers.{l}", gin)
            hin = (h * gw).astype(F32)
            if l < nl:
                y = relu(linear_fwd(hin, params[f"ppn.{t}.mlp_layers.{l}.0.weight"],
                                    params[f"ppn.{t}.mlp_layers.{l}.0.bias"]))
            else:
                y = linear_fwd(hin, params[f"ppn.{t}.mlp_layers.{l}.weight"], params[f"ppn.{t}.mlp_layers.{l}.bias"])
            lay.append((h, gw, gc, hin, y))
            h = y
        ps.append(sigmoid((h + params[f"out.{t}.bias"])[:, 0]))
        tasks.append(lay)
    return np.stack(ps, 1).astype(F32), dict(x=x, fg=fg, fgc=fgc, x2=x2, tasks=tasks, nl=nl, layers={})


def pepnet_bwd(spec, params, cache, dlogit):
    grads = {}
    nl = cache["nl"]
    dx2 = np.zeros_like(cache["x2"])
    for t in range(spec.T):
        dy = dlogit[:, t:t + 1]
        _acc(grads, f"out.{t}.bias", dy.sum(0))
        for l in range(nl, -1, -1):
            h, gw, gc, hin, y = cache["tasks"][t][l]
            if l < nl:
                wk, bk = f"ppn.{t}.mlp_layers.{l}.0.weight", f"ppn.{t}.mlp_layers.{l}.0.bias"
                dz = dy * (y > 0)
            else:
                wk, bk = f"ppn.{t}.mlp_layers.{l}.weight", f"ppn.{t}.mlp_layers.{l}.bias"
                dz = dy
            dhin, dW, db = linear_bwd(hin, params[wk], dz)
            _acc(grads, wk, dW)
            _acc(grads, bk, db)
            _gatenn_bwd(params, f"ppn.{t}.gate_layers.{l}", gc, (dhin * h).astype(F32), grads)
            dy = (dhin * gw).astype(F32)
        dx2 += dy
    _gatenn_bwd(params, "feature_gate", cache["fgc"], (dx2 * cache["x"]).astype(F32), grads)
    dx = (dx2 * cache["fg"]).astype(F32)
    return grads, dx


_FWD = {"apg": apg_fwd, "escm": escm_fwd, "aitm": aitm_fwd, "hmoe": hmoe_fwd, "cross_stitch": cross_stitch_fwd, "esmm": esmm_fwd, "mlp": mlp_fwd, "sharedbottom": sharedbottom_fwd, "mmoe": mmoe_fwd, "pcg": mmoe_fwd, "ple": ple_fwd, "pepnet": pepnet_fwd}
_BWD = {"apg": apg_bwd, "snr_trans": snr_trans_bwd, "mssm": snr_trans_bwd, "aitm": aitm_bwd, "hmoe": hmoe_bwd, "cross_stitch": cross_stitch_bwd, "mlp": mlp_bwd, "sharedbottom": sharedbottom_bwd, "mmoe": mmoe_bwd, "pcg": mmoe_bwd, "ple": ple_bwd, "pepnet": pepnet_bwd}


def forward(spec, params, X, mask=None, frozen=None):
    """model(X, domain_mask) -> probabilities [B,T]; also returns the cache for backward and the
    save_layer_output tensors (e.g. model/mmoe.py:110-118)."""
    x, idx = gather_dnn_input(spec, params, X)
    if spec.model_name == "star":
        p, cache = star_fwd(spec, params, x, frozen)
    elif spec.model_name in ("snr_trans", "mssm"):
        p, cache = snr_trans_fwd(spec, params, x, frozen)
    else:
        p, cache = _FWD[spec.model_name](spec, params, x)
    cache["dnn_input"] = x
    cache["idx"] = idx
    cache["p"] = p
    if spec.model_name == "mlp" and spec.task_name != "msl":  # model/mlp.py:53-54 masks in the msl mode only
        mask = None
    if spec.model_name == "aitm" and spec.task_name != "msl":  # model/aitm.py:104-105
        mask = None
    return apply_mask(spec, p, mask), cache


def loss_and_grads(spec, params, X, y, frozen=None):
    """One pure reference step without the optimizer (model/basemodel.py:268-312, mask=None per SURVEY D3):
    returns (sum-BCE loss, {state_dict key: gradient})."""
    y = np.asarray(y, dtype=F32)
    was = _MODE["training"]
    set_training(True)  # basemodel.py:261 model.train()
    try:
        p, cache = forward(spec, params, X, None, frozen)
    finally:
        set_training(was)
    if spec.model_name == "escm":
        loss, d3 = escm_loss_and_dprob(p, y)
        grads, dx = escm_bwd(spec, params, cache, d3)
        grads.update(scatter_table_grads(spec, dx, cache["idx"], params))
        cache["d_dnn_input"] = dx
        return loss, grads, cache
    loss = sum(bce_sum(p[:, t], y[:, t]) for t in range(spec.T))
    dlogit = bce_sigmoid_bwd(p, y)
    if spec.model_name == "star":
        grads, dx = star_bwd(spec, params, cache, dlogit)
    elif spec.model_name == "esmm":
        grads, dx = esmm_bwd(spec, params, cache, bce_prob_bwd(p, y))
    else:
        grads, dx = _BWD[spec.model_name](spec, params, cache, dlogit)
    grads.update(scatter_table_grads(spec, dx, cache["idx"], params))
    cache["d_dnn_input"] = dx
    return loss, grads, cache


# ----------------------------------------------------------------------------------------------
# optimizers: torch.optim defaults, dense over every parameter (model/basemodel.py:569-584)
# ----------------------------------------------------------------------------------------------
class DenseOptimizer:
    def __init__(self, kind, lr):
        self.kind, self.lr, self.t = kind, F32(lr), 0
        self.state = {}

    def step(self, params, grads):
        """In-place update of params[key] for every key with a gradient (params without grad are skipped
        like torch.optim does for p.grad is None)."""
        self.t += 1
        lr = self.lr
        for k, g in grads.items():
            p = params[k]
            st = self.state.setdefault(k, {})
            if self.kind == "sgd":
                p -= lr * g
            elif self.kind == "adam":  # betas (0.9, 0.999), eps 1e-8, no amsgrad, no weight decay
                if "m" not in st:
                    st["m"], st["v"] = np.zeros_like(p), np.zeros_like(p)
                m, v = st["m"], st["v"]
                b1, b2 = 0.9, 0.999
                if _FAST is not None and p.size >= 1 << 16 and p.flags.c_contiguous and g.flags.c_contiguous:
                    import ctypes
                    _FAST.adam_dense(_fp(p), _fp(g), _fp(m), _fp(v), ctypes.c_int64(p.size), ctypes.c_float(lr),
                                     ctypes.c_float(b1), ctypes.c_float(b2), ctypes.c_float(1e-8), self.t)
                    continue
                m *= F32(b1)
                m += F32(1 - b1) * g
                v *= F32(b2)
                v += F32(1 - b2) * g * g
                bc1 = 1.0 - b1 ** self.t
                bc2 = 1.0 - b2 ** self.t
                step_size = F32(float(lr) / bc1)
                denom = np.sqrt(v) / F32(np.sqrt(bc2)) + F32(1e-8)
                p -= step_size * (m / denom)
            elif self.kind == "adagrad":  # lr_decay 0, eps 1e-10, initial accumulator 0
                if "sum" not in st:
                    st["sum"] = np.zeros_like(p)
                s = st["sum"]
                if _FAST is not None and p.size >= 1 << 16 and p.flags.c_contiguous and g.flags.c_contiguous:
                    import ctypes
                    _FAST.adagrad_dense(_fp(p), _fp(g), _fp(s), ctypes.c_int64(p.size), ctypes.c_float(lr),
                                        ctypes.c_float(1e-10))
                    continue
                s += g * g
                p -= lr * g / (np.sqrt(s) + F32(1e-10))
            elif self.kind == "rmsprop":  # alpha 0.99, eps 1e-8, no momentum
                if "sq" not in st:
                    st["sq"] = np.zeros_like(p)
                sq = st["sq"]
                sq *= F32(0.99)
                sq += F32(0.01) * g * g
                p -= lr * g / (np.sqrt(sq) + F32(1e-8))
            else:
                raise NotImplementedError(self.kind)  # basemodel.py:581


def reg_map(spec, params):
    """name -> (l1, l2) as the reference registers them (basemodel.py:129-130 for the tables; for sharedbottom / mmoe /
    ple every sub-network's `'weight' in name and 'bn' not in name` parameters get l2_reg_dnn: sharedbottom.py:36-47,
    mmoe.py:36-62, ple.py:57-103 -- including PLE's dead last-level shared-gate tensors).  The key
    defaults are the reference's: 1e-5 for the tables when the key is absent, 0 for the DNNs."""
    if spec.model_name not in ("sharedbottom", "mmoe", "pcg", "ple"):
        raise NotImplementedError("reg_map restates sharedbottom / mmoe / ple only")
    l2e, l2d = spec.mc.get("l2_reg_embedding", 1e-5), spec.mc.get("l2_reg_dnn", 0)
    out = {}
    for k in params:
        if k.startswith("embedding_dict."):
            if l2e > 0:
                out[k] = (0.0, float(l2e))
        elif "weight" in k and "bn" not in k and l2d > 0:
            out[k] = (0.0, float(l2d))
    return out


def add_reg_grads(params, grads, reg):
    """d/dp [l1 |p| + l2 p^2] added to (or creating) each regularised parameter's gradient (basemodel.py:524-540)."""
    for k, (l1, l2) in reg.items():
        p = params[k]
        g = np.zeros_like(p)
        if l2 > 0:
            g = g + F32(2.0 * l2) * p
        if l1 > 0:
            g = g + F32(l1) * np.sign(p)
        grads[k] = (grads[k] + g).astype(F32) if grads.get(k) is not None else g.astype(F32)
    return grads


def train_step(spec, params, opt, X, y, frozen=None, reg=None):
    """basemodel.py:268-313: forward, summed BCE, backward (of loss + regulariser), dense optimizer step.
    Returns the BCE loss (what the reference logs, :307)."""
    loss, grads, _ = loss_and_grads(spec, params, X, y, frozen)
    if reg:
        add_reg_grads(params, grads, reg)
    opt.step(params, grads)
    return loss


def params_from_golden(g, prefix="state/"):
    return OrderedDict((k[len(prefix):], np.array(g[k], dtype=F32)) for k in g.files if k.startswith(prefix))


# ----------------------------------------------------------------------------------------------
# parameter shapes (restating the constructors: model/mmoe.py:9-63, sharedbottom.py:10-50, ple.py:11-105,
# star.py:9-37, pepnet.py:80-119) -- used to build random models at sizes no fixture covers
# ----------------------------------------------------------------------------------------------
def param_shapes(spec):
    mc = spec.mc
    T, K0, E = spec.T, spec.K0, spec.emb
    shapes = OrderedDict()
    for name, v in zip(spec.sparse_names, spec.vocab):
        shapes[f"embedding_dict.{name}.weight"] = (v, E)

    def dnn(prefix, k, units):
        for l, u in enumerate(units):
            shapes[f"{prefix}.linears.{l}.weight"] = (u, k)
            shapes[f"{prefix}.linears.{l}.bias"] = (u,)
            k = u
        return k

    def towers(in_dim):
        tu = mc.get("tower_dnn_hidden_units", [64])
        for t in range(T):
            h = dnn(f"tower_dnn.{t}", in_dim, tu) if tu else in_dim
            shapes[f"tower_dnn_final_layer.{t}.weight"] = (1, h)
        for t in range(T):
            shapes[f"out.{t}.bias"] = (1,)

    name = spec.model_name
    if name == "cross_stitch":
        units = mc.get("dnn_hidden_units", [256, 128])
        k = dnn("shared_layer", K0, [mc.get("shared_hidden_unit", 256)])
        for i, d in enumerate(units):
            for j in range(T):
                dnn(f"cross_stitch.task_layer_{i}.{j}", k, [d])
            shapes[f"cross_stitch.gate_{i}.cross_stitch_weight"] = (T * d, T * d)
            k = d
        towers(k)
    elif name in ("esmm", "escm"):
        shapes["out.bias"] = (1,)
        for twr in ("ctr", "cvr"):
            h = dnn(f"{twr}_dnn", K0, mc.get("expert_dnn_hidden_units", [256, 128]))
        shapes["ctr_dnn_final_layer.weight"] = (1, h)
        shapes["cvr_dnn_final_layer.weight"] = (1, h)
    elif name == "mlp":
        k = K0
        for i, u in enumerate(mc.get("dnn_hidden_units", [256, 128])):
            k = dnn(f"mlp_layers.{i}", k, [u])
        shapes["final_layer.weight"] = (1, k)
        for t in range(T):
            shapes[f"out.{t}.bias"] = (1,)
    elif name == "sharedbottom":
        h = dnn("bottom_dnn", K0, mc.get("bottom_dnn_hidden_units", [256, 128]))
        towers(h)
    elif name in ("mmoe", "pcg"):
        Ne = mc.get("num_experts", 4)
        for e in range(Ne):
            H = dnn(f"expert_dnn.{e}", K0, mc.get("expert_dnn_hidden_units", [256, 128]))
        gu = mc.get("gate_dnn_hidden_units", [64])
        G = K0
        for t in range(T):
            if gu:
                G = dnn(f"gate_dnn.{t}", K0, gu)
        for t in range(T):
            shapes[f"gate_dnn_final_layer.{t}.weight"] = (Ne, G)
        towers(H)
    elif name == "ple":
        S, Sh, Lv = mc.get("specific_expert_num", 3), mc.get("shared_expert_num", 1), mc.get("num_levels", 1)
        eu, gu = mc.get("expert_dnn_hidden_units", [256, 128]), mc.get("gate_dnn_hidden_units", [64])
        H = eu[-1]
        for lv in range(Lv):
            kin = K0 if lv == 0 else H
            for i in range(T):
                for j in range(S):
                    dnn(f"specific_experts.{lv}.{i}.{j}", kin, eu)
        for lv in range(Lv):
            kin = K0 if lv == 0 else H
            for j in range(S):  # built with specific_expert_num, ple.py:47
                dnn(f"shared_experts.{lv}.0.{j}", kin, eu)
        for lv in range(Lv):
            kin = K0 if lv == 0 else H
            for i in range(T):
                dnn(f"specific_gate_dnn.{lv}.{i}.0", kin, gu)
        for lv in range(Lv):
            for i in range(T):
                shapes[f"specific_gate_dnn_final_layer.{lv}.{i}.weight"] = (S + Sh, gu[-1])
        for lv in range(Lv):
            dnn(f"shared_gate_dnn.{lv}", K0 if lv == 0 else H, gu)
        for lv in range(Lv):
            shapes[f"shared_gate_dnn_final_layer.{lv}.weight"] = (T * S + Sh, gu[-1])
        towers(H)
    else:
        raise NotImplementedError(f"param_shapes for {name} (use a golden fixture)")
    return shapes


def random_params(spec, rng, w_std=None, table_std=0.05):
    """He-scaled random weights (so activations neither vanish nor explode) for parity runs at arbitrary sizes."""
    params = OrderedDict()
    for k, shp in param_shapes(spec).items():
        if k.startswith("embedding_dict."):
            params[k] = (rng.standard_normal(shp, dtype=np.float32) * F32(table_std))
        elif len(shp) == 2:
            std = w_std if w_std is not None else float(np.sqrt(2.0 / shp[1]))
            params[k] = (rng.standard_normal(shp, dtype=np.float32) * F32(std))
        elif k.startswith("out."):
            params[k] = np.zeros(shp, dtype=F32)
        else:
            params[k] = (rng.standard_normal(shp, dtype=np.float32) * F32(0.05))
    return params
