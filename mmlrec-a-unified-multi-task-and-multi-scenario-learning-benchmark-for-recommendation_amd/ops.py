"""Thin torch-tensor front end of the C ABI (include/mmlrec.h): argument marshalling only, no arithmetic.

torch is used for device memory and the current HIP stream; every number is produced by libmmlrec_hip.so.
All functions require CUDA(HIP) tensors and raise MMLError otherwise -- there is no CPU fallback.
"""
import ctypes as C

import os

import torch

from . import _lib as L

_workspaces = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


_masked_streams = {}


def cu_range_stream(device, cu_lo, cu_hi):
    """torch view of a HIP stream confined to compute units [cu_lo, cu_hi) (mml_stream_create_cu_range); one per
    (device, range), kept for the life of the process."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, int(cu_lo), int(cu_hi))
    st = _masked_streams.get(key)
    if st is None:
        h = C.c_void_p()
        L.check(L.load().mml_stream_create_cu_range(idx, int(cu_lo), int(cu_hi), C.byref(h)))
        st = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", idx))
        _masked_streams[key] = st
    return st


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise L.MMLError("mmlrec_amd ops need tensors on an MI355X (got a CPU tensor); there is no CPU fallback")


def _f32_2d(t, name):
    if t.dtype != torch.float32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise L.MMLError(f"{name}: expected a float32 2-D tensor with unit inner stride, got {t.dtype} {tuple(t.shape)} "
                         f"strides {t.stride()}")
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


_SPIN_CAL = {}  # device index -> (spin ticks for ~0.2 ms, measured ms of one such spin)


def concurrent_stream(device, tries=6):
    """A stream whose kernels really run beside those of the current stream.  HIP hands streams one of a few
    hardware queues round-robin (four by default); a process that also holds RCCL's streams and the routing stream of
    the row-sharded tables can find its side stream on the SAME queue as the main stream, and the forked tail then
    runs one launch after the other (seen in the kernel trace of the forced row-sharded step: weight-gradient GEMMs
    and table update on one queue, 2.10 ms where the unsharded step takes 1.88).  So: probe -- a spin kernel on each
    stream, concurrent if the pair takes the time of one -- and keep the first candidate that passes.
    MMLREC_SIDE_PROBE=0 takes the first stream unprobed."""
    main = torch.cuda.current_stream(device)
    first = torch.cuda.Stream(device=device)
    if os.environ.get("MMLREC_SIDE_PROBE", "1") == "0" or not hasattr(torch.cuda, "_sleep"):
        return first
    try:
        def timed(fn):
            torch.cuda.synchronize(device)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main)
            fn()
            b.record(main)
            torch.cuda.synchronize(device)
            return a.elapsed_time(b)

        # torch.cuda._sleep counts s_memtime ticks (a 100 MHz constant clock on gfx9, not shader cycles): the spin length
        # for ~0.2 ms is MEASURED once per device and cached (400 000 ticks were ~4 ms per probe)
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        cal = _SPIN_CAL.get(key)
        if cal is None:
            torch.cuda._sleep(1000)  # (loads the spin kernel)
            probe = 20_000
            ms = max(timed(lambda: torch.cuda._sleep(probe)), 1e-3)
            spin = int(min(max(probe * 0.2 / ms, 2_000), 2_000_000))
            one = timed(lambda: torch.cuda._sleep(spin))
            cal = _SPIN_CAL[key] = (spin, one)
        spin, one = cal

        def pair_ms(s):
            j = torch.cuda.Event()

            def both():
                a0 = torch.cuda.Event()
                a0.record(main)
                s.wait_event(a0)
                with torch.cuda.stream(s):
                    torch.cuda._sleep(spin)
                    j.record(s)
                torch.cuda._sleep(spin)
                main.wait_event(j)
            return timed(both)

        cand = first
        for _ in range(tries):
            if min(pair_ms(cand), pair_ms(cand)) < 1.5 * one:
                return cand
            cand = torch.cuda.Stream(device=device)
    except Exception:  # (a probe must never cost the step)
        pass
    return first


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (kept alive here; kernels only see the raw pointer)."""
    key = (device.type, device.index)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _ptr_array(tensors):
    arr = (L.fp * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def new_status(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def check_status(status, what="embedding lookup"):
    """Turn the device status word into the exception nn.Embedding raises on CPU (IndexError)."""
    v = int(status.item())
    if v:
        status.zero_()
        raise IndexError(f"index out of range in {what} (status bits {v:#x}: 1 = negative, 2 = >= vocabulary)")


# ---------------------------------------------------------------------------------------------- K1 / K2
def gather_fwd(tables, X, cols, dense_col0=0, nd=0, out=None, status=None):
    """dnn_input = concat_f table_f[X[:, cols[f]].long()] ++ X[:, dense_col0:dense_col0+nd]."""
    lib = L.load()
    _need_gpu(X, *tables)
    X = _f32_2d(X, "X")
    F = len(tables)
    E = tables[0].shape[1]
    B = X.shape[0]
    K0 = F * E + nd
    if out is None:
        out = torch.empty((B, K0), dtype=torch.float32, device=X.device)
    vocab = (L.i64 * F)(*[t.shape[0] for t in tables])
    col = (L.i32 * F)(*cols)
    rc = lib.mml_gather_fwd(_ptr_array(tables), vocab, col, F, E, X.data_ptr(), _ld(X), dense_col0, nd, B,
                            out.data_ptr(), _ld(out), L.ptr(status), _stream())
    L.check(rc, "mml_gather_fwd")
    return out


def gather_fwd_wgmax(tables, X, cols, dense_col0=0, nd=0, out=None, status=None):
    """gather_fwd that also returns the per-workgroup maxima of |out| (include/mmlrec.h: mml_gather_fwd_wgmax)."""
    lib = L.load()
    _need_gpu(X, *tables)
    X = _f32_2d(X, "X")
    F, E, B = len(tables), tables[0].shape[1], X.shape[0]
    if out is None:
        out = torch.empty((B, F * E + nd), dtype=torch.float32, device=X.device)
    n = int(lib.mml_gather_wgmax_len(F, E, nd, B))
    wg = torch.full((1, max(n, 1)), float("nan"), dtype=torch.float32, device=X.device)
    vocab = (L.i64 * F)(*[t.shape[0] for t in tables])
    col = (L.i32 * F)(*cols)
    rc = lib.mml_gather_fwd_wgmax(_ptr_array(tables), vocab, col, F, E, X.data_ptr(), _ld(X), dense_col0, nd, B,
                                  out.data_ptr(), _ld(out), wg.data_ptr(), n, L.ptr(status), _stream())
    L.check(rc, "mml_gather_fwd_wgmax")
    return out, wg


def gather_fwd_idx32(tables, idx, dense=None, out=None, status=None):
    lib = L.load()
    _need_gpu(idx, *tables)
    F, E, B = len(tables), tables[0].shape[1], idx.shape[0]
    nd = 0 if dense is None else dense.shape[1]
    if out is None:
        out = torch.empty((B, F * E + nd), dtype=torch.float32, device=idx.device)
    vocab = (L.i64 * F)(*[t.shape[0] for t in tables])
    rc = lib.mml_gather_fwd_idx32(_ptr_array(tables), vocab, F, E, idx.data_ptr(), idx.stride(0), L.ptr(dense),
                                  0 if dense is None else dense.stride(0), nd, B, out.data_ptr(), _ld(out),
                                  L.ptr(status), _stream())
    L.check(rc, "mml_gather_fwd_idx32")
    return out


def marks_bytes(vocab):
    """Size of the row-mark scratch map of mml_scatter_bwd / mml_index_unique (include/mmlrec.h: row_marks)."""
    return 32 * sum((int(v) + 31) // 32 for v in vocab)


def scatter_bwd(grad_tables, X, cols, d_out, seen=None, rowbase=None, touched=None, touched_count=None, status=None,
                marks=None):
    """grad_tables[f][X[:, cols[f]].long()] += d_out[:, f*E:(f+1)*E] (float atomics)."""
    lib = L.load()
    _need_gpu(X, d_out, *grad_tables)
    F, E, B = len(grad_tables), grad_tables[0].shape[1], X.shape[0]
    vocab = (L.i64 * F)(*[t.shape[0] for t in grad_tables])
    col = (L.i32 * F)(*cols)
    seen_arr = _ptr_array(seen) if seen is not None else None
    rb = (L.i64 * (F + 1))(*rowbase) if rowbase is not None else None
    rc = lib.mml_scatter_bwd(_ptr_array(grad_tables), vocab, col, F, E, X.data_ptr(), _ld(X), B, d_out.data_ptr(),
                             _ld(d_out), seen_arr, rb, L.ptr(touched), L.ptr(touched_count),
                             0 if touched is None else touched.numel(), L.ptr(marks), L.ptr(status), _stream())
    L.check(rc, "mml_scatter_bwd")


def scatter_bwd_det(grad_tables, X, cols, d_out, acc64, marks, amax_slot=None, clear_marks=True, status=None):
    """Deterministic scatter (mml_scatter_bwd_det): acc64 = per-table int64 [V, E] accumulators (all zero between calls),
    marks = the byte map of marks_bytes(vocab).  Bitwise repeatable, independent of the order of the samples."""
    lib = L.load()
    _need_gpu(X, d_out, *grad_tables)
    F, E, B = len(grad_tables), grad_tables[0].shape[1], X.shape[0]
    vocab = (L.i64 * F)(*[t.shape[0] for t in grad_tables])
    col = (L.i32 * F)(*cols)
    if amax_slot is None:
        amax_slot = amax_slots(1, X.device)[0]
    rc = lib.mml_scatter_bwd_det(_ptr_array(grad_tables), vocab, col, F, E, X.data_ptr(), _ld(X), B, d_out.data_ptr(),
                                 _ld(d_out), _ptr_array(acc64), amax_slot.data_ptr(), marks.data_ptr(),
                                 int(bool(clear_marks)), L.ptr(status), _stream())
    L.check(rc, "mml_scatter_bwd_det")


def scatter_bwd_idx32(grad_tables, idx, d_out, seen=None, rowbase=None, touched=None, touched_count=None, status=None,
                      marks=None):
    """grad_tables[f][idx[:, f]] += d_out[:, f*E:(f+1)*E] with native int32 indices [B, F]."""
    lib = L.load()
    _need_gpu(idx, d_out, *grad_tables)
    F, E, B = len(grad_tables), grad_tables[0].shape[1], idx.shape[0]
    vocab = (L.i64 * F)(*[t.shape[0] for t in grad_tables])
    seen_arr = _ptr_array(seen) if seen is not None else None
    rb = (L.i64 * (F + 1))(*rowbase) if rowbase is not None else None
    rc = lib.mml_scatter_bwd_idx32(_ptr_array(grad_tables), vocab, F, E, idx.data_ptr(), idx.stride(0), B,
                                   d_out.data_ptr(), _ld(d_out), seen_arr, rb, L.ptr(touched), L.ptr(touched_count),
                                   0 if touched is None else touched.numel(), L.ptr(marks), L.ptr(status), _stream())
    L.check(rc, "mml_scatter_bwd_idx32")


def route(X, cols, vocab, keybase, world, status=None):
    """Row-sharded routing of a batch (csrc/shard.hip): returns (counts [world] int32, send_keys [B*F] int32 grouped by
    owner, pos [B, F] int32)."""
    lib = L.load()
    _need_gpu(X)
    F, B = len(vocab), X.shape[0]
    col = (L.i32 * F)(*cols)
    voc = (L.i64 * F)(*vocab)
    kb = (L.i64 * F)(*keybase[:F])
    counters = torch.zeros(2 * world, dtype=torch.int32, device=X.device)
    keys = torch.empty(B * F, dtype=torch.int32, device=X.device)
    pos = torch.empty(B, F, dtype=torch.int32, device=X.device)
    s = _stream()
    if X.dtype == torch.int32:
        xa = (None, 0, X.data_ptr(), X.stride(0))
    else:
        xa = (_f32_2d(X, "X").data_ptr(), _ld(X), None, 0)
    L.check(lib.mml_route_count(*xa, col, voc, F, B, world, counters.data_ptr(), L.ptr(status), s), "mml_route_count")
    counts = counters[:world].clone()
    L.check(lib.mml_route_place(*xa, col, voc, kb, F, B, world, counters.data_ptr(), keys.data_ptr(), pos.data_ptr(),
                                L.ptr(status), s), "mml_route_place")
    return counts, keys, pos


def rows_permute(src, pos, E, dst):
    """dst[pos[b, f]] = src[b, f*E:(f+1)*E]."""
    B, F = pos.shape
    L.check(L.load().mml_rows_permute(src.data_ptr(), _ld(src), pos.data_ptr(), F, E, B, dst.data_ptr(), _stream()),
            "mml_rows_permute")


# ---------------------------------------------------------------------------------------------- K3
def make_fwd_descs(problems):
    """problems: dicts with A [M,K], W ([N,K] or [K,N] if w_kn), bias or None, C [M,N], act, w_kn."""
    arr = (L.GemmFwdDesc * len(problems))()
    for d, p in zip(arr, problems):
        A, W, Cc = p["A"], p["W"], p["C"]
        w_kn = int(p.get("w_kn", 0))
        d.A, d.W, d.C = A.data_ptr(), W.data_ptr(), Cc.data_ptr()
        d.bias = L.ptr(p.get("bias"))
        d.lda, d.ldw, d.ldc = _ld(A), _ld(W), _ld(Cc)
        d.M, d.K = A.shape
        d.N = W.shape[1] if w_kn else W.shape[0]
        d.act = int(p.get("act", L.ACT_NONE))
        d.w_kn = w_kn
        m = p.get("mask")
        d.relu_mask = L.ptr(m)
        d.ldmask = _ld(m) if m is not None else 0
        d.amax_a, d.amax_w, d.amax_out = L.ptr(p.get("amax_a")), L.ptr(p.get("amax_w")), L.ptr(p.get("amax_out"))
        d.w_planes, d.w_kexp = L.ptr(p.get("w_planes")), L.ptr(p.get("w_kexp"))  # pre-cut weight (planes_cut)
        mul, prod = p.get("mul"), p.get("prod")  # K7: prod = C * mul from the same epilogue
        if mul is not None:
            d.mul, d.prod, d.ldmul, d.ldprod = mul.data_ptr(), prod.data_ptr(), _ld(mul), _ld(prod)
            d.amax_prod = L.ptr(p.get("amax_prod"))
    return arr


# ---- operand magnitudes (include/mmlrec.h): slots of MML_AMAX_WORDS words --------------------------------------
AMAX_WORDS = 8
PLANES_ROWS, PLANES_COLS = 0, 1


def make_planes_descs(items):
    """items: (W [rows, cols] view, planes int32 tensor of W's shape and pitch, layout, [magnitude slots], kexp int32
    [1] tensor) -> the descriptor block of mml_gemm_planes_cut."""
    arr = (L.PlanesDesc * len(items))()
    for d, (W, planes, layout, slots, kexp) in zip(arr, items):
        # K6: W = (A, B) -> the planes of the element-wise product A * B; slots = [(slot of A_i, slot of B_i)] for the
        # matrices of the group that shares the exponent
        W2 = None
        if isinstance(W, tuple):
            W, W2 = W
            if W2.shape != W.shape:
                raise L.MMLError("factors of a derived weight must have one shape")
            if 2 * len(slots) > L.MAX_SRC:
                raise L.MMLError("too many product weights share one exponent")
        # planes: the weight's shape (pitch may differ), or -- for the zero-padded operand of a reduction that is not a
        # multiple of 16 -- the same rows with the columns rounded up (a zero-initialised buffer)
        if planes.shape[0] != W.shape[0] or planes.shape[1] < W.shape[1]:
            raise L.MMLError("planes buffer must have the weight's rows and at least its columns")
        d.W, d.planes = W.data_ptr(), planes.data_ptr()
        d.rows, d.cols = W.shape
        d.ld = _ld(W)
        d.ldp = _ld(planes)
        d.layout = int(layout)
        d.n_amax = len(slots)
        if W2 is not None:
            d.W2, d.ld2 = W2.data_ptr(), _ld(W2)
            for a, (sa, sb) in enumerate(slots):
                d.amax[a] = sa.data_ptr()
                d.amax[len(slots) + a] = sb.data_ptr()
        else:
            for a, sl in enumerate(slots):
                d.amax[a] = sl.data_ptr()
        d.kexp = kexp.data_ptr()
    return arr


def planes_cut(items):
    """Cut weight matrices into their two fp16 planes once (include/mmlrec.h: mml_gemm_planes_cut)."""
    arr = make_planes_descs(items)
    L.check(L.load().mml_gemm_planes_cut(arr, len(items), _stream()), "mml_gemm_planes_cut")


def amax_slots(n, device):
    """n zeroed magnitude slots as an int32 [n, AMAX_WORDS] tensor (row i = slot i)."""
    return torch.zeros(n, AMAX_WORDS, dtype=torch.int32, device=device)


def make_amax_descs(pairs):
    """pairs: (tensor [rows, cols] with unit column stride, slot) -- one mml_amax_batch call raises every slot."""
    arr = (L.AmaxDesc * len(pairs))()
    for d, (x, slot) in zip(arr, pairs):
        if x.dim() == 1:
            x = x.view(1, -1)
        d.x, d.rows, d.cols, d.ld, d.slot = x.data_ptr(), x.shape[0], x.shape[1], _ld(x), slot.data_ptr()
    return arr


def amax_batch(pairs):
    arr = make_amax_descs(pairs)
    L.check(L.load().mml_amax_batch(arr, len(pairs), _stream()), "mml_amax_batch")


def amax_value(slot):
    """The slot's value as a Python float (host read: tests / diagnostics)."""
    return float(slot.view(-1).max().view(1).view(torch.float32).item())


def _measured(tensors, cache):
    """slot per tensor (by storage pointer + shape), measured now with the stand-alone kernel"""
    todo = []
    for t in tensors:
        key = (t.data_ptr(), tuple(t.shape), t.stride(0) if t.dim() > 1 else 0)
        if key not in cache:
            cache[key] = amax_slots(1, t.device)[0]
            todo.append((t, cache[key]))
    if todo:
        amax_batch(todo)
    return cache


def gemm_fwd(problems, amax=False):
    """amax=True: measure both operands of every problem first (stand-alone mml_amax_batch launches), which lets the
    launch run the two-plane fp16 arithmetic -- what engine.py arranges with magnitudes produced along the way."""
    lib = L.load()
    if amax:
        cache = _measured([p["A"] for p in problems] + [p["W"] for p in problems], {})
        problems = [dict(p, amax_a=cache[(p["A"].data_ptr(), tuple(p["A"].shape), p["A"].stride(0))],
                         amax_w=cache[(p["W"].data_ptr(), tuple(p["W"].shape), p["W"].stride(0))]) for p in problems]
    arr = make_fwd_descs(problems)
    L.check(lib.mml_gemm_grouped_fwd(arr, len(problems), _stream()), "mml_gemm_grouped_fwd")


def make_dgrad_descs(problems):
    """problems: dicts with dA [M,K], Y (or None), act, accumulate, srcs = [(dC [M,N], W, w_kn), ...]."""
    arr = (L.GemmDgradDesc * len(problems))()
    for d, p in zip(arr, problems):
        gate = p.get("gate")  # K7 backward: dict(h, g, dh, dg, act_h, act_g, acc_h, acc_g[, amax_dh, amax_dg]); no dA
        if gate is not None:
            d.gate_h, d.gate_g, d.d_h, d.d_g = (gate[k].data_ptr() for k in ("h", "g", "dh", "dg"))
            d.ld_h, d.ld_g, d.ld_dh, d.ld_dg = (_ld(gate[k]) for k in ("h", "g", "dh", "dg"))
            d.act_h, d.act_g, d.acc_h, d.acc_g = (int(gate[k]) for k in ("act_h", "act_g", "acc_h", "acc_g"))
            d.amax_dh, d.amax_dg = L.ptr(gate.get("amax_dh")), L.ptr(gate.get("amax_dg"))
        dA = p["dA"] if gate is None else gate["dh"]
        d.dA = dA.data_ptr() if gate is None else None
        Y = p.get("Y")
        d.Y = L.ptr(Y)
        d.ldda = _ld(dA)
        d.ldy = _ld(Y) if Y is not None else 0
        d.M, d.K = dA.shape
        d.act = int(p.get("act", L.ACT_NONE)) if Y is not None else L.ACT_NONE
        d.accumulate = int(p.get("accumulate", 0))
        srcs = p["srcs"]
        d.n_src = len(srcs)
        for s, src in enumerate(srcs):
            dC, W, w_kn = src[:3]
            d.dC[s], d.W[s] = dC.data_ptr(), W.data_ptr()
            d.lddc[s], d.ldw[s] = _ld(dC), _ld(W)
            d.N[s] = dC.shape[1]
            d.w_kn[s] = int(w_kn)
            if len(src) > 3:  # (dC, W, w_kn, magnitude slot of dC, magnitude slot of W[, planes of W, their exponent])
                d.amax_dc[s], d.amax_w[s] = L.ptr(src[3]), L.ptr(src[4])
            if len(src) > 5:
                d.w_planes[s], d.w_kexp[s] = L.ptr(src[5]), L.ptr(src[6])
        m = p.get("mask")
        d.relu_mask = L.ptr(m)
        d.ldmask = _ld(m) if m is not None else 0
        d.amax_out = L.ptr(p.get("amax_out"))
    return arr


def gemm_dgrad(problems, amax=False):
    lib = L.load()
    if amax:
        ts = [t for p in problems for src in p["srcs"] for t in src[:2]]
        cache = _measured(ts, {})
        k = lambda t: (t.data_ptr(), tuple(t.shape), t.stride(0))  # noqa: E731
        problems = [dict(p, srcs=[(dC, W, kn, cache[k(dC)], cache[k(W)]) for dC, W, kn in
                                  [src[:3] for src in p["srcs"]]]) for p in problems]
    arr = make_dgrad_descs(problems)
    L.check(lib.mml_gemm_grouped_dgrad(arr, len(problems), _stream()), "mml_gemm_grouped_dgrad")


def make_wgrad_descs(problems):
    """problems: dicts with dC [M,N], A [M,K], dW ([N,K] or [K,N] if w_kn), dbias or None, accumulate, w_kn."""
    arr = (L.GemmWgradDesc * len(problems))()
    for d, p in zip(arr, problems):
        dC, A, dW = p["dC"], p["A"], p["dW"]
        d.dC, d.A, d.dW = dC.data_ptr(), A.data_ptr(), dW.data_ptr()
        d.dbias = L.ptr(p.get("dbias"))
        d.lddc, d.lda, d.lddw = _ld(dC), _ld(A), _ld(dW)
        d.M, d.N = dC.shape
        d.K = A.shape[1]
        d.accumulate = int(p.get("accumulate", 0))
        d.w_kn = int(p.get("w_kn", 0))
        d.amax_dc, d.amax_a = L.ptr(p.get("amax_dc")), L.ptr(p.get("amax_a"))
    return arr


def gemm_wgrad(problems, amax=False):
    lib = L.load()
    if amax:
        cache = _measured([p["dC"] for p in problems] + [p["A"] for p in problems], {})
        k = lambda t: (t.data_ptr(), tuple(t.shape), t.stride(0))  # noqa: E731
        problems = [dict(p, amax_dc=cache[k(p["dC"])], amax_a=cache[k(p["A"])]) for p in problems]
    arr = make_wgrad_descs(problems)
    dev = problems[0]["dC"].device
    nbytes = lib.mml_gemm_grouped_wgrad_workspace_bytes(arr, len(problems))
    ws = workspace(nbytes, dev)
    L.check(lib.mml_gemm_grouped_wgrad(arr, len(problems), ws.data_ptr(), ws.numel(), _stream()),
            "mml_gemm_grouped_wgrad")


# ---------------------------------------------------------------------------------------------- K4
def make_gate_group(experts, gates, B, H, d_experts=None, e_relu=True):
    """experts: list of [B,H] tensors; gates: dicts with G, Wg, P, mix, expert (index list) and, for backward,
    dmix, dG, dWg, g_relu, active."""
    g = L.GateGroup()
    g.n_experts, g.n_gates, g.H, g.B, g.e_relu = len(experts), len(gates), H, B, int(e_relu)
    for x, e in enumerate(experts):
        g.E[x], g.lde[x] = e.data_ptr(), _ld(e)
        if d_experts is not None:
            g.dE[x], g.ldde[x] = d_experts[x].data_ptr(), _ld(d_experts[x])
    for i, q in enumerate(gates):
        d = g.gate[i]
        G, Wg, P = q["G"], q["Wg"], q["P"]
        d.G, d.Wg, d.P = G.data_ptr(), Wg.data_ptr(), P.data_ptr()
        d.ldg, d.ldp = _ld(G), _ld(P)
        d.ne, d.Gd = Wg.shape
        if not Wg.is_contiguous():
            raise L.MMLError("gate weight must be contiguous")
        if q.get("mix") is not None:
            d.mix, d.ldmix = q["mix"].data_ptr(), _ld(q["mix"])
        d.active = int(q.get("active", 1))
        d.g_relu = int(q.get("g_relu", 1))
        if q.get("dmix") is not None:
            d.dmix, d.lddmix = q["dmix"].data_ptr(), _ld(q["dmix"])
        if q.get("dG") is not None:
            d.dG, d.lddg = q["dG"].data_ptr(), _ld(q["dG"])
        if q.get("dWg") is not None:
            d.dWg = q["dWg"].data_ptr()
        for s, x in enumerate(q["expert"]):
            d.expert[s] = x
    # bf16-storage path: tensors only GEMMs read may be bf16 buffers (include/mmlrec.h: mml_gate_group.out_bf16)
    def _all16(ts, what):
        ts = [t for t in ts if t is not None]
        n16 = sum(t.dtype == torch.bfloat16 for t in ts)
        if n16 not in (0, len(ts)):
            raise L.MMLError(f"gate group: {what} must be all bf16 or all fp32")
        return bool(ts) and n16 == len(ts)
    bits = 0
    if _all16([q.get("mix") for q in gates], "mix"):
        bits |= L.GATE_MIX_BF16
    if d_experts is not None and _all16(list(d_experts), "dE"):
        bits |= L.GATE_DE_BF16
    if _all16([q.get("dG") for q in gates], "dG"):
        bits |= L.GATE_DG_BF16
    if _all16(list(experts), "E"):
        bits |= L.GATE_E_BF16
    g.out_bf16 = bits
    return g


def gate_mix_fwd(group):
    lib = L.load()
    L.check(lib.mml_gate_mix_fwd(C.byref(group), _stream()), "mml_gate_mix_fwd")


def gate_mix_bwd(group, device, phases=False):
    """phases: the two-launch form (row kernel, then the reduction of its partial sums)."""
    lib = L.load()
    ws = workspace(lib.mml_gate_mix_bwd_workspace_bytes(C.byref(group)), device)
    if phases:
        for ph in (1, 2):
            L.check(lib.mml_gate_mix_bwd_phase(C.byref(group), ws.data_ptr(), ws.numel(), ph, _stream()),
                    "mml_gate_mix_bwd_phase")
        return
    L.check(lib.mml_gate_mix_bwd(C.byref(group), ws.data_ptr(), ws.numel(), _stream()), "mml_gate_mix_bwd")


# ---------------------------------------------------------------------------------------------- K5
def make_head_group(heads, prob, y=None, mask=None, loss=None, dprob=None):
    """heads: dicts with Hin [B,H], w [H] (any shape with H elements), bias [1], optional w2, bias2 (list of 1-element
    tensors packed by the caller into one tensor), dH, dw, dbias, h_relu, mask_col; gated heads also gate [B,H], dgate,
    gate_act."""
    g = L.HeadGroup()
    g.n_heads = len(heads)
    g.B = prob.shape[0]
    g.prob, g.ldprob = prob.data_ptr(), _ld(prob)
    if y is not None:
        g.y, g.ldy = y.data_ptr(), _ld(y)
    if mask is not None:
        g.mask, g.ldmask = mask.data_ptr(), _ld(mask)
    g.loss = L.ptr(loss)
    if dprob is not None:
        g.dprob, g.lddprob = dprob.data_ptr(), _ld(dprob)
    for t, q in enumerate(heads):
        d = g.head[t]
        Hin = q["Hin"]
        d.Hin, d.ldh, d.H = Hin.data_ptr(), _ld(Hin), Hin.shape[1]
        d.w, d.bias = q["w"].data_ptr(), q["bias"].data_ptr()
        d.w2 = L.ptr(q.get("w2"))
        b2 = q.get("bias2")
        d.bias2 = L.ptr(b2)
        d.n_bias2 = 0 if b2 is None else b2.numel()
        if q.get("dH") is not None:
            d.dH, d.lddh = q["dH"].data_ptr(), _ld(q["dH"])
        d.dw, d.dbias = L.ptr(q.get("dw")), L.ptr(q.get("dbias"))
        d.h_relu = int(q.get("h_relu", 1))
        d.mask_col = int(q.get("mask_col", -1))
        if q.get("gate") is not None:  # gated head: the input is Hin (.) gate (include/mmlrec.h: mml_head_desc.gate)
            d.gate, d.ldgate = q["gate"].data_ptr(), _ld(q["gate"])
            d.gate_act = int(q.get("gate_act", L.ACT_NONE))
            if q.get("dgate") is not None:
                d.dgate, d.lddgate = q["dgate"].data_ptr(), _ld(q["dgate"])
    dhs = [q["dH"] for q in heads if q.get("dH") is not None]
    n16 = sum(t.dtype == torch.bfloat16 for t in dhs)
    if n16 not in (0, len(dhs)):
        raise L.MMLError("head group: dH must be all bf16 or all fp32")
    g.dh_bf16 = int(bool(dhs) and n16 == len(dhs))  # (include/mmlrec.h: mml_head_group.dh_bf16)
    return g


def make_tower_head_group(tasks, prob, y, mask=None, loss=None):
    """K5' (include/mmlrec.h): tasks = dicts with A [M, K], amax_a, planes_fwd / kexp_fwd (MML_PLANES_ROWS image of the tower
    weight [N, K]), planes_bwd / kexp_bwd (MML_PLANES_COLS), bias1 [N] or None, w [N], hbias [1], hbias2 or None, dH [M, N],
    dA [M, K], dw [N], dhbias [1], optional amax_dH / amax_dA, mask_col, head (column of prob / y)."""
    g = L.TowerHeadGroup()
    g.n, g.M = len(tasks), prob.shape[0]
    g.prob, g.ldprob = prob.data_ptr(), _ld(prob)
    g.y, g.ldy = y.data_ptr(), _ld(y)
    if mask is not None:
        g.mask, g.ldmask = mask.data_ptr(), _ld(mask)
    g.loss = L.ptr(loss)
    for t, q in enumerate(tasks):
        d = g.t[t]
        A = q["A"]
        d.A, d.lda, d.K = A.data_ptr(), _ld(A), A.shape[1]
        d.N = q["dH"].shape[1]
        d.amax_a = q["amax_a"].data_ptr()
        d.w_planes_fwd, d.ldpf, d.kexp_fwd = q["planes_fwd"].data_ptr(), _ld(q["planes_fwd"]), q["kexp_fwd"].data_ptr()
        d.w_planes_bwd, d.ldpb, d.kexp_bwd = q["planes_bwd"].data_ptr(), _ld(q["planes_bwd"]), q["kexp_bwd"].data_ptr()
        d.bias1 = L.ptr(q.get("bias1"))
        d.w, d.hbias = q["w"].data_ptr(), q["hbias"].data_ptr()
        b2 = q.get("hbias2")
        d.hbias2, d.n_hbias2 = L.ptr(b2), (0 if b2 is None else b2.numel())
        d.dH, d.lddh = q["dH"].data_ptr(), _ld(q["dH"])
        d.dA, d.ldda = q["dA"].data_ptr(), _ld(q["dA"])
        d.dw, d.dhbias = q["dw"].data_ptr(), q["dhbias"].data_ptr()
        d.amax_dH, d.amax_dA = L.ptr(q.get("amax_dH")), L.ptr(q.get("amax_dA"))
        d.mask_col, d.head = int(q.get("mask_col", -1)), int(q.get("head", t))
    return g


def tower_head_fwd_bwd(group, device, phases=False):
    lib = L.load()
    n = int(lib.mml_tower_head_workspace_bytes(C.byref(group)))
    if n < 0:
        L.check(-1, "mml_tower_head_workspace_bytes")
    ws = torch.empty(max(n, 256), dtype=torch.uint8, device=device)
    for ph in ((1, 2) if phases else (0,)):
        L.check(lib.mml_tower_head_fwd_bwd(C.byref(group), ws.data_ptr(), ws.numel(), ph, _stream()), "mml_tower_head_fwd_bwd")
    return ws


def head_fwd(group):
    lib = L.load()
    L.check(lib.mml_head_fwd(C.byref(group), _stream()), "mml_head_fwd")


def head_bce_fwd_bwd(group, device, phases=False):
    lib = L.load()
    ws = workspace(lib.mml_head_workspace_bytes(C.byref(group)), device)
    if phases:
        for ph in (1, 2):
            L.check(lib.mml_head_bce_fwd_bwd_phase(C.byref(group), ws.data_ptr(), ws.numel(), ph, _stream()),
                    "mml_head_bce_fwd_bwd_phase")
        return
    L.check(lib.mml_head_bce_fwd_bwd(C.byref(group), ws.data_ptr(), ws.numel(), _stream()), "mml_head_bce_fwd_bwd")


def rows_phase1_then_batched_reduce(head_groups, gate_groups, device):
    """The row kernels of the given head / gate groups (phase 1, each into a workspace of its own), then ALL their
    reductions in one launch (include/mmlrec.h: mml_rows_reduce_batch)."""
    lib = L.load()
    items = (L.RowsReduceItem * (len(head_groups) + len(gate_groups)))()
    keep, k = [], 0
    for kind, groups in ((L.ROWS_REDUCE_HEAD, head_groups), (L.ROWS_REDUCE_GATE, gate_groups)):
        for g in groups:
            if kind == L.ROWS_REDUCE_HEAD:
                ws = torch.empty(max(int(lib.mml_head_workspace_bytes(C.byref(g))), 256), dtype=torch.uint8, device=device)
                L.check(lib.mml_head_bce_fwd_bwd_phase(C.byref(g), ws.data_ptr(), ws.numel(), 1, _stream()),
                        "mml_head_bce_fwd_bwd_phase")
            else:
                ws = torch.empty(max(int(lib.mml_gate_mix_bwd_workspace_bytes(C.byref(g))), 256), dtype=torch.uint8,
                                 device=device)
                L.check(lib.mml_gate_mix_bwd_phase(C.byref(g), ws.data_ptr(), ws.numel(), 1, _stream()),
                        "mml_gate_mix_bwd_phase")
            keep.append(ws)
            items[k].kind, items[k].group = kind, C.addressof(g)
            items[k].workspace, items[k].workspace_bytes = ws.data_ptr(), ws.numel()
            k += 1
    L.check(lib.mml_rows_reduce_batch(items, k, _stream()), "mml_rows_reduce_batch")
    return keep


# ---------------------------------------------------------------------------------------------- K6/K7
def ew_mul(a, b, out):
    L.check(L.load().mml_ew_mul(a.data_ptr(), b.data_ptr(), out.data_ptr(), out.numel(), _stream()), "mml_ew_mul")


def ew_mul_bwd(dout, a, b, da=None, db=None, acc_a=False, acc_b=False):
    L.check(L.load().mml_ew_mul_bwd(dout.data_ptr(), L.ptr(a), L.ptr(b), L.ptr(da), L.ptr(db), int(acc_a), int(acc_b),
                                    dout.numel(), _stream()), "mml_ew_mul_bwd")


def ew_add_n(inputs, out):
    L.check(L.load().mml_ew_add_n(_ptr_array(inputs), len(inputs), out.data_ptr(), out.numel(), _stream()),
            "mml_ew_add_n")


def dropout(x, out, p, seed, site=0, step=0, step_dev=None, accumulate=False, row0=0):
    """out (+)= x * keep / (1 - p) on [rows, cols] views (row pitches may differ; out may be x); the backward is the same
    call on the gradient (include/mmlrec.h: mml_dropout; reference model/utils.py:159)."""
    L.check(L.load().mml_dropout(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), x.shape[0], x.shape[1], int(row0), float(p),
                                 int(seed) & ((1 << 64) - 1), int(site) & 0xffffffff, L.ptr(step_dev), int(step),
                                 int(accumulate), _stream()), "mml_dropout")


def copy2d(src, dst, accumulate=False):
    """dst[:, :] (+)= src[:, :] for 2-D row-strided views (column slices of wider buffers are fine)."""
    rows, cols = src.shape
    L.check(L.load().mml_copy2d(src.data_ptr(), _ld(src), dst.data_ptr(), _ld(dst), rows, cols, int(accumulate),
                                _stream()), "mml_copy2d")


def act_bwd(y, dy, dst, act):
    L.check(L.load().mml_act_bwd(y.data_ptr(), dy.data_ptr(), dst.data_ptr(), y.numel(), act, _stream()), "mml_act_bwd")


# ---------------------------------------------------------------------------------------------- K8
def make_hyper(kind, lr, step=1, step_dev=None, zero_grad=False, max_blocks=0):
    h = L.OptHyper()
    h.kind = L.OPT_KINDS[kind] if isinstance(kind, str) else int(kind)
    h.step = int(step)
    h.step_dev = L.ptr(step_dev)
    h.lr = float(lr)
    h.beta1, h.beta2 = 0.9, 0.999
    h.eps = {L.OPT_ADAM: 1e-8, L.OPT_ADAGRAD: 1e-10, L.OPT_RMSPROP: 1e-8}.get(h.kind, 0.0)
    h.alpha = 0.99
    h.zero_grad = int(zero_grad)
    h.max_blocks = int(max_blocks)
    return h


def make_opt_tensors(entries):
    """entries: (param, grad, state1 or None, state2 or None[, (l1, l2)[, skip bitmap[, gradient marks]]]) with equal
    element counts, contiguous.  A skip bitmap (int32 words, one bit per table row) turns the entry into the untouched-rows half of the
    split dense table update (include/mmlrec.h: mml_opt_tensor.skip_rows)."""
    arr = (L.OptTensor * len(entries))()
    for d, ent in zip(arr, entries):
        p, g, s1, s2 = ent[:4]
        d.param, d.grad, d.state1, d.state2, d.n = p.data_ptr(), g.data_ptr(), L.ptr(s1), L.ptr(s2), p.numel()
        d.l1, d.l2 = ent[4] if len(ent) > 4 and ent[4] else (0.0, 0.0)
        if len(ent) > 5 and ent[5] is not None:
            d.skip_rows, d.row_elems, d.zero_grads = ent[5].data_ptr(), p.shape[1], 1
        if len(ent) > 6 and ent[6] is not None:  # byte marks of the rows whose gradient is non-zero (uint8 view)
            d.grad_marks, d.row_elems = ent[6].data_ptr(), p.shape[1]
    return arr


def opt_step_dense(entries, hyper):
    arr = make_opt_tensors(entries)
    L.check(L.load().mml_opt_step_dense(arr, len(entries), C.byref(hyper), _stream()), "mml_opt_step_dense")


def opt_step_rows(tables, grad_tables, state1, state2, seen, rowbase, touched, touched_count, hyper, last=None):
    F, E = len(tables), tables[0].shape[1]
    rb = (L.i64 * (F + 1))(*rowbase)
    L.check(L.load().mml_opt_step_rows(_ptr_array(tables), _ptr_array(grad_tables),
                                       _ptr_array(state1) if state1 is not None else None,
                                       _ptr_array(state2) if state2 is not None else None,
                                       _ptr_array(seen), rb, F, E, touched.data_ptr(), touched_count.data_ptr(),
                                       touched.numel(), _ptr_array(last) if last is not None else None,
                                       C.byref(hyper), _stream()), "mml_opt_step_rows")


def index_unique(vocab, cols, E, X, seen, rowbase, touched, touched_count, status=None, marks=None):
    F = len(vocab)
    L.check(L.load().mml_index_unique((L.i64 * F)(*vocab), (L.i32 * F)(*cols), F, E, X.data_ptr(), _ld(X), X.shape[0],
                                      _ptr_array(seen), (L.i64 * (F + 1))(*rowbase), touched.data_ptr(),
                                      touched_count.data_ptr(), touched.numel(), L.ptr(marks), L.ptr(status), _stream()),
            "mml_index_unique")


def opt_catchup_rows(tables, state1, state2, last, rowbase, touched, touched_count, hyper):
    F, E = len(tables), tables[0].shape[1]
    L.check(L.load().mml_opt_catchup_rows(_ptr_array(tables), _ptr_array(state1),
                                          _ptr_array(state2) if state2 is not None else None, _ptr_array(last),
                                          (L.i64 * (F + 1))(*rowbase), F, E, touched.data_ptr(),
                                          touched_count.data_ptr(), touched.numel(), C.byref(hyper), _stream()),
            "mml_opt_catchup_rows")


def opt_catchup_dense(table, state1, state2, last, hyper):
    L.check(L.load().mml_opt_catchup_dense(table.data_ptr(), L.ptr(state1), L.ptr(state2), last.data_ptr(),
                                           table.shape[0], table.shape[1], C.byref(hyper), _stream()),
            "mml_opt_catchup_dense")


def counter_update(counter, delta=1, reset=False):
    L.check(L.load().mml_counter_update(counter.data_ptr(), int(delta), int(reset), _stream()), "mml_counter_update")


def auc_segments(pred, y, seg):
    """AUC of every (segment of `seg` rows, column) pair of device matrices pred / y [n, C] -> float64 [ceil(n/seg), C]
    (NaN where a segment holds one class).  Device counterpart of the per-step sklearn roc_auc_score of the reference
    loop (model/basemodel.py:316-331)."""
    lib = L.load()
    n, c = pred.shape
    out = torch.empty(((n + seg - 1) // seg, c), dtype=torch.float64, device=pred.device)
    L.check(lib.mml_auc_segments(pred.data_ptr(), pred.stride(0), y.data_ptr(), y.stride(0), n, c, int(seg),
                                 out.data_ptr(), _stream()), "mml_auc_segments")
    return out


# ---------------------------------------------------------------------------------------------- K3' (bf16 storage)
def _is16(t):
    return t.dtype == torch.bfloat16


def _need16(t, name):
    if t.dtype != torch.bfloat16 or t.dim() != 2 or t.stride(1) != 1:
        raise L.MMLError(f"{name}: expected a bf16 2-D tensor with unit inner stride, got {t.dtype} {tuple(t.shape)}")
    return t


def make_cast16_descs(items):
    """items: (src fp32 [r, c], dst bf16 [r, c] or, transposed, [c, r], transpose flag)."""
    arr = (L.Cast16Desc * len(items))()
    for d, (src, dst, tr) in zip(arr, items):
        _f32_2d(src, "cast16 source")
        _need16(dst, "cast16 destination")
        want = (src.shape[1], src.shape[0]) if tr else tuple(src.shape)
        if tuple(dst.shape) != want:
            raise L.MMLError(f"cast16: destination shape {tuple(dst.shape)}, expected {want}")
        d.src, d.dst, d.rows, d.cols = src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1]
        d.lds, d.ldd, d.transpose = _ld(src), _ld(dst), int(bool(tr))
    return arr


def cast16(src, transpose=False, out=None):
    _need_gpu(src)
    if out is None:
        shape = (src.shape[1], src.shape[0]) if transpose else tuple(src.shape)
        out = torch.empty(shape, dtype=torch.bfloat16, device=src.device)
    arr = make_cast16_descs([(src, out, transpose)])
    L.check(L.load().mml_cast16_batch(arr, 1, _stream()), "mml_cast16_batch")
    return out


def gather16_fwd(tables, X, cols, dense_col0=0, nd=0, out=None, status=None):
    """gather_fwd writing dnn_input as bf16 (mml_gather16_fwd)."""
    lib = L.load()
    _need_gpu(X, *tables)
    X = _f32_2d(X, "X")
    F, E, B = len(tables), tables[0].shape[1], X.shape[0]
    if out is None:
        out = torch.empty((B, F * E + nd), dtype=torch.bfloat16, device=X.device)
    vocab = (L.i64 * F)(*[t.shape[0] for t in tables])
    col = (L.i32 * F)(*cols)
    L.check(lib.mml_gather16_fwd(_ptr_array(tables), vocab, col, F, E, X.data_ptr(), _ld(X), dense_col0, nd, B,
                                 out.data_ptr(), _ld(out), L.ptr(status), _stream()), "mml_gather16_fwd")
    return out


def make_g16_tn_descs(problems):
    """problems: dicts with srcs [(A bf16 [M, K_s], B bf16 [N, K_s]), ...], C ([M, N] bf16 or fp32), optional bias [N]
    fp32, act, accumulate, mask_out / mask_in (int32 [M, ceil(N / 32)])."""
    arr = (L.G16TnDesc * len(problems))()
    for d, q in zip(arr, problems):
        Cm = q["C"]
        d.M, d.N = Cm.shape
        d.n_src = len(q["srcs"])
        if not 1 <= d.n_src <= L.MAX_SRC:
            raise L.MMLError("g16_tn: 1 .. MAX_SRC sources")
        for s, (A, Bm) in enumerate(q["srcs"]):
            _need16(A, "g16_tn A")
            _need16(Bm, "g16_tn B")
            if A.shape[0] != d.M or Bm.shape[0] != d.N or A.shape[1] != Bm.shape[1]:
                raise L.MMLError(f"g16_tn: source {s} shapes {tuple(A.shape)} x {tuple(Bm.shape)} for C {tuple(Cm.shape)}")
            d.A[s], d.B[s], d.lda[s], d.ldb[s], d.K[s] = A.data_ptr(), Bm.data_ptr(), _ld(A), _ld(Bm), A.shape[1]
        d.act = int(q.get("act", L.ACT_NONE))
        d.bias = L.ptr(q.get("bias"))
        d.C, d.ldc, d.c_bf16 = Cm.data_ptr(), _ld(Cm), int(_is16(Cm))
        if not d.c_bf16 and Cm.dtype != torch.float32:
            raise L.MMLError("g16_tn: C must be bf16 or fp32")
        d.accumulate = int(q.get("accumulate", 0))
        mo, mi = q.get("mask_out"), q.get("mask_in")
        d.mask_out, d.mask_in = L.ptr(mo), L.ptr(mi)
        m = mo if mo is not None else mi
        d.ldmask = _ld(m) if m is not None else 0
    return arr


def g16_tn(problems):
    arr = make_g16_tn_descs(problems)
    L.check(L.load().mml_g16_tn(arr, len(problems), _stream()), "mml_g16_tn")


def make_g16_wgrad_descs(problems):
    """problems: dicts with dC bf16 [M, N], A bf16 [M, K], dW fp32 [N, K], optional dbias fp32 [N], accumulate."""
    arr = (L.G16WgradDesc * len(problems))()
    for d, q in zip(arr, problems):
        dC, A, dW = _need16(q["dC"], "g16_wgrad dC"), _need16(q["A"], "g16_wgrad A"), _f32_2d(q["dW"], "g16_wgrad dW")
        d.M, d.N, d.K = dC.shape[0], dC.shape[1], A.shape[1]
        if A.shape[0] != d.M or tuple(dW.shape) != (d.N, d.K):
            raise L.MMLError("g16_wgrad: shape mismatch")
        d.dC, d.A, d.dW, d.dbias = dC.data_ptr(), A.data_ptr(), dW.data_ptr(), L.ptr(q.get("dbias"))
        d.lddc, d.lda, d.lddw = _ld(dC), _ld(A), _ld(dW)
        d.accumulate = int(q.get("accumulate", 0))
    return arr


def g16_wgrad(problems, phases=False):
    lib = L.load()
    arr = make_g16_wgrad_descs(problems)
    n = int(lib.mml_g16_wgrad_workspace_bytes(arr, len(problems)))
    if n < 0:
        L.check(-1, "mml_g16_wgrad_workspace_bytes")
    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=problems[0]["dW"].device)
    for ph in ((1, 2) if phases else (0,)):
        L.check(lib.mml_g16_wgrad(arr, len(problems), ws.data_ptr(), ws.numel(), ph, _stream()), "mml_g16_wgrad")
