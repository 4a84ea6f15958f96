"""Static step plans: a model's forward is recorded ONCE per batch size as a short list of C-ABI calls on
pre-allocated device buffers, its backward is derived from that record, and a training step is
`fwd calls + fused head/BCE call + bwd calls + optimizer calls` -- about 25 kernel launches for MMoE instead of
the ~1 000 ATen ops the reference issues per step.  The call list is replayable inside a HIP graph
(torch.cuda.CUDAGraph is used purely as the capture/replay plumbing).

Gradient convention: `Val.grad` holds dL/d(pre-activation) once every consumer has contributed.  A consumer that is
the ONLY consumer of a value folds the activation derivative into its own kernel epilogue (dgrad / gate / head
kernels do this); otherwise consumers add raw contributions and one `mml_act_bwd` pass finalises the sum.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import profiling
from . import ops

ACT = {"none": L.ACT_NONE, None: L.ACT_NONE, "linear": L.ACT_NONE, "relu": L.ACT_RELU, "sigmoid": L.ACT_SIGMOID,
       "sigmoid2": L.ACT_SIGMOID2}


class PVal:
    """A parameter tensor (or a tensor derived from parameters) with an optional gradient buffer."""

    def __init__(self, data, grad=None, name="", needs_grad=True, is_table=False):
        self.data, self.grad, self.name = data, grad, name
        self.needs_grad = needs_grad and grad is not None
        self.is_table = is_table
        self.written = 0  # build-time counter: first writer overwrites, later writers accumulate
        # True: a trained parameter (ParamStore) that changes only in the optimizer -- its magnitude is measured ONCE at
        # the start of a step.  False: a tensor some op of the step produces (STAR's effective weights, APG's generated
        # ones): measured where it is used.
        self.stable = False


class Val:
    """A [B, n] activation living in a (possibly column-sliced) device buffer."""

    def __init__(self, buf, act=L.ACT_NONE, needs_grad=True, name=""):
        self.buf, self.act, self.needs_grad, self.name = buf, act, needs_grad, name
        self.grad = None
        self.consumers = []  # ops that would write a gradient into this value
        self.written = 0
        self.deriv_applied = False
        self.mask = None  # int32 [B, ceil(n/32)] relu sign bits (training plans, written by the producing GEMM)
        # > 0: the value owns its rows up to column kpad (a multiple of 16) and the columns [n, kpad) are zero and
        # never written -- a GEMM may then read it as a [B, kpad] operand (LinearGroupOp: reduction lengths that
        # are not a multiple of the 16-wide k-step, e.g. 30 x 8 embedding columns + 63 dense columns = 303)
        self.kpad = 0
        # > 0: only the first grad_cols columns of the gradient have a reader (dnn_input: the table scatter reads the
        # embedding columns, nobody the dense features' -- model/basemodel.py:461-487 concatenates them behind the
        # embeddings); LinearGroupOp then forms the input gradient for those columns only
        self.grad_cols = 0
        # operand magnitudes for the two-plane fp16 GEMMs (include/mmlrec.h): slot of the value, slot of its gradient,
        # and how many of the gradient's writers raised that slot (valid iff it equals `written`)
        self.amax = None
        self.gamax = None
        self.gamax_writers = 0
        # bf16-storage path (include/mmlrec.h K3'): the value lives in a bf16 buffer -- only GEMMs read it; producer16:
        # a bf16-storage layer group produced it and will read its GRADIENT as a GEMM operand (Plan.grad_of)
        self.producer16 = False

    @property
    def is16(self):
        return self.buf.dtype == torch.bfloat16

    @property
    def n(self):
        return self.buf.shape[1]


INLINE = object()  # call-list marker: (INLINE, python_callable, args) run in place like a PY entry but do NOT cut a HIP graph
# (stream fork / join of trainer.InnerFork: event record / wait, capturable)
PY = object()  # call-list marker: (PY, python_callable, args[, meta]) entries (collectives) next to (c_fn, args[, meta])


def _claim(x):
    acc = 1 if x.written else 0
    x.written += 1
    return acc


class Plan:
    def __init__(self, device, B, training):
        self.device, self.B, self.training = device, int(B), bool(training)
        self.ops = []
        self.fwd, self.head_infer, self.head_train, self.head_bwd, self.bwd = [], [], [], [], []
        # backward is kept in three pieces so a trainer can overlap them on two HIP streams: `bwd` = the critical
        # chain (dgrads, gate/elementwise backward), `bwd_tail` = the table scatter (needs d(dnn_input), feeds the
        # HBM-bound table optimizer), `bwd_side` = every weight-gradient GEMM (MFMA-bound, needs only values the
        # chain has already produced).  Sequential order bwd -> bwd_tail -> bwd_side is always valid.
        self.bwd_tail, self.bwd_side = [], []
        self.head_side = []  # the deferred reduction of the fused head call (dw / dbias / loss): beside `bwd_side`
        self.keep = []  # ctypes descriptor blocks + buffers referenced by raw pointer
        # dropout (DropoutOp): on iff the MODULE is in training mode; the step word of its mask stream is read from
        # `step_dev` (the optimizer's device step counter when the model has one, else a counter of the plan's own)
        self.dropout_on = False
        self.row0 = 0  # first row of this plan's batch in the global batch (data-parallel ranks: rank * B)
        self.dropout_seed = 0
        self.step_dev = None
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.vals = {}
        self.prob = None
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)
        self.y = None
        self.mask = None
        self.dprob = None
        self.X = None
        self.layer_outputs = {}
        # operand magnitudes (ops.amax_slots): one pool per plan, zeroed at the start of every forward; the magnitudes of
        # the stable weights are measured by ONE launch right after (amax_pre, put in front of `fwd` by finish())
        # Large batches: the GEMMs are throughput-bound and the two-plane fp16 form pays (AE-30 at 65 536: GEMM family
        # 1.14 -> 1.0 ms, step -5 %).  Same-box A/B at smaller batches: level at 32 768 and 16 384 (+-2 %: the ~50 us of
        # magnitude launches per step against the arithmetic saved), a loss below (8 192: 0.839 -> 0.854 ms; lazy_exact at
        # 4 096: 0.27 -> 0.35 ms -- a step there is a chain of ~25 short launches).  With the cheaper cut and the pre-cut
        # weights of the end of round 3: 32 768: 1.272 -> 1.244 ms (lazy_exact 0.927 -> 0.894), 16 384: 0.994 -> 1.036,
        # 8 192: 0.760 -> 0.802, 4 096: 0.696 -> 0.720.  So: on from 32 768 samples per step, the three-plane bf16 form
        # below.  MMLREC_AMAX=0 / 1 forces.
        env = os.environ.get("MMLREC_AMAX", "")
        self.use_amax = (env != "0") and (env == "1" or self.B >= 32768)
        # bf16-STORAGE path (round 5, include/mmlrec.h K3'; BASELINE.json configs[1]): under the opt-in reduced-precision
        # GEMM mode 1 (operands rounded to bf16) the values and gradients that only GEMMs read are STORED as bf16 and the
        # layer groups that read them run csrc/gemm16.hip -- same products, half the activation traffic, no conversion in
        # the kernels.  Models mark such values (Plan.val(store16=True)); MMLREC_BF16_STORAGE=0 keeps fp32 buffers.
        self.bf16 = (device.type == "cuda" and os.environ.get("MMLREC_BF16_STORAGE", "1") != "0" and
                     L.load().mml_gemm_get_mode() == 1)
        if self.bf16:
            self.use_amax = False  # (mode 1 reads no operand magnitudes)
        self.cast16_items = []   # (fp32 weight, bf16 copy, transposed): refreshed by ONE launch at the start of a step
        self.cast16_cache = {}
        self.amax_pool = ops.amax_slots(1024, device) if (device.type == "cuda" and self.use_amax) else None
        self.amax_next = 0
        self.amax_weights = {}   # (data_ptr, shape) -> slot
        self.amax_wlist = []     # (tensor, slot) of the stable weights
        # pre-cut weights (mml_gemm_planes_cut): the two fp16 planes of every stable weight the forward / input-gradient
        # GEMMs read, cut once at the start of a step instead of by every wave that stages a fragment of it
        self.planes_items = []   # (W, planes, layout, [slots], kexp)
        self.planes_cache = {}
        self.n_pre = 0           # entries of `fwd` that precede the first op's calls

    # ---- buffers -----------------------------------------------------------------------------
    def empty(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    def zeros(self, *shape, dtype=torch.float32):
        t = torch.zeros(*shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    def _rows(self, n, pad_k=False):
        """[B, n] view of a fresh buffer with 16-byte aligned rows.  pad_k: when n is not a multiple of 16 the row is
        padded with ZERO columns up to the next multiple (returns the padded width, else 0)."""
        if pad_k and n % 16:
            kp = (n + 15) // 16 * 16
            return self.zeros(self.B, kp)[:, :n], kp
        ld = (n + 3) // 4 * 4
        return (self.empty(self.B, ld)[:, :n] if ld != n else self.empty(self.B, n)), 0

    def val(self, n, act=L.ACT_NONE, needs_grad=True, name="", buf=None, pad_k=False, store16=False):
        """pad_k: for values that feed a GEMM as the reduction operand (see Val.kpad).  store16: the value is read by
        bf16-storage layer groups ONLY (the caller vouches for it; finish() checks) -- a bf16 buffer when the plan runs
        the bf16-storage path."""
        kp = 0
        if buf is None:
            if store16 and self.bf16 and n % 8 == 0:
                buf = self.empty(self.B, n, dtype=torch.bfloat16)
            else:
                buf, kp = self._rows(n, pad_k)
        v = Val(buf, act, needs_grad, name)
        v.kpad = kp
        return v

    def _grad16(self, v):
        """dL/dv as a bf16 buffer: a bf16-storage layer group produced v (it reads the gradient as a GEMM operand), and v's
        one consumer writes the gradient exactly once, in bf16."""
        return (self.bf16 and v.producer16 and v.n % 8 == 0 and len(v.consumers) == 1 and
                v.consumers[0].writes_grad16(self, v))

    def weight16(self, pv, transposed):
        """bf16 copy of a weight ([N, K], or transposed [K, N]: the input gradient reads W^T rows), refreshed by the
        cast launch that opens the step."""
        key = (pv.data.data_ptr(), tuple(pv.data.shape), bool(transposed))
        if key not in self.cast16_cache:
            W = pv.data
            if W.dim() != 2 or W.stride(1) != 1:
                raise L.MMLError("bf16-storage path: 2-D weights with unit inner stride")
            shape = (W.shape[1], W.shape[0]) if transposed else tuple(W.shape)
            dst = torch.zeros(shape, dtype=torch.bfloat16, device=self.device)
            self.cast16_cache[key] = dst
            self.cast16_items.append((W, dst, bool(transposed)))
        return self.cast16_cache[key]

    def grad_of(self, v):
        """Allocate v.grad on first use (same row pitch as the value)."""
        if v.grad is None:
            if self._grad16(v):
                v.grad = self.empty(self.B, v.n, dtype=torch.bfloat16)
            elif v.is16:  # (a bf16 value whose gradient is fp32: the gather's output -- the table scatter reads it)
                v.grad = self.empty(self.B, v.n)
            elif v.kpad:
                v.grad = self.zeros(self.B, v.kpad)[:, :v.n]
            else:
                ld = (v.n + 3) // 4 * 4
                v.grad = self.empty(self.B, ld)[:, :v.n] if ld != v.n else self.empty(self.B, v.n)
        return v.grad

    # ---- operand magnitudes --------------------------------------------------------------------
    def new_amax(self):
        if self.amax_pool is None:
            return None
        if self.amax_next >= self.amax_pool.shape[0]:
            raise L.MMLError("operand-magnitude pool exhausted")
        self.amax_next += 1
        return self.amax_pool[self.amax_next - 1]

    def weight_amax(self, pv, tensor, need):
        """Slot of a GEMM weight operand.  `tensor` may be a zero-padded copy of pv.data (same magnitude).  Stable
        parameters join the start-of-step launch; anything else is appended to `need` (measured before the launch that
        is being recorded)."""
        if self.amax_pool is None:
            return None
        key = (pv.data.data_ptr(), tuple(pv.data.shape))
        if key in self.amax_weights:
            return self.amax_weights[key]
        slot = self.new_amax()
        if getattr(pv, "stable", False):
            self.amax_weights[key] = slot
            self.amax_wlist.append((pv.data, slot))
        else:
            need.append((pv.data, slot))  # (not cached: re-measured by every launch that reads it -- it may change)
        return slot

    def weight_planes(self, q, layout, group=None, padded=False):
        """(planes, kexp) of problem q's weight for the forward (layout ROWS: its own exponent) or as one source of an
        input-gradient problem (layout COLS: `group` = the problems whose weights feed the same output, ONE exponent), or
        (None, None): only stable weights in nn.Linear layout whose magnitude is taken at the start of the step.
        padded: the launch reads the zero-padded operand (q["Wp"], reduction extent rounded up to 16): the planes are cut
        from the weight itself into a zero-initialised buffer of the padded shape."""
        if self.amax_pool is None or os.environ.get("MMLREC_GEMM_PLANES", "1") == "0":
            return None, None
        qs = [q] if group is None else group
        slots = []
        # K6: a derived weight W = A (.) B whose factors are stable parameters (STAR: W_specific (.) W_shared, reference
        # model/utils.py:214-218) is cut straight from its factors; all members of a group must be of one kind
        prod = [getattr(g["W"], "factors", None) is not None for g in qs]
        if any(prod) and not all(prod):
            return None, None
        for g in qs:
            W = g["W"]
            kn = int(g.get("w_kn", 0))
            # a [K, N] matrix (w_kn = 1) swaps the roles: the forward's reduction runs down its rows
            lay = layout if not kn else (ops.PLANES_COLS if layout == ops.PLANES_ROWS else ops.PLANES_ROWS)
            red = W.data.shape[1] if lay == ops.PLANES_ROWS else W.data.shape[0]
            if (("Wp" in g) != bool(padded) or W.data.dim() != 2 or W.data.stride(1) != 1 or
                    (red % 16 and not (padded and lay == ops.PLANES_ROWS)) or
                    W.data.data_ptr() % 16 or (W.data.stride(0) % 4 and not padded)):
                return None, None
            fac = getattr(W, "factors", None)
            if fac is not None:
                if padded or not all(getattr(f, "stable", False) and f.data.shape == W.data.shape and
                                     f.data.stride(1) == 1 for f in fac):
                    return None, None
                sl = tuple(self.weight_amax(f, f.data, None) for f in fac)  # (stable: the start-of-step launch)
                slots.append(sl)
                continue
            key = (W.data.data_ptr(), tuple(W.data.shape))
            if kn or not getattr(W, "stable", False) or key not in self.amax_weights:
                return None, None
            slots.append(self.amax_weights[key])
        if any(prod) and 2 * len(slots) > L.MAX_SRC:
            return None, None
        Wq = q["W"]
        W = Wq.data
        kn = int(q.get("w_kn", 0))
        lay = layout if not kn else (ops.PLANES_COLS if layout == ops.PLANES_ROWS else ops.PLANES_ROWS)
        shape = tuple(q["Wp"].shape) if padded else (W.shape[0], W.stride(0))
        flat = tuple(s_.data_ptr() for sl in slots for s_ in (sl if isinstance(sl, tuple) else (sl,)))
        ck = (W.data_ptr(), tuple(W.shape), lay, bool(padded), flat)
        if ck not in self.planes_cache:
            gk = ("kexp", lay, bool(padded), flat)  # one exponent word per group
            if gk not in self.planes_cache:
                self.planes_cache[gk] = torch.zeros(1, dtype=torch.int32, device=self.device)
            planes = torch.zeros(shape, dtype=torch.int32, device=self.device)
            if not padded:
                planes = planes[:, :W.shape[1]]
            self.planes_cache[ck] = (planes, self.planes_cache[gk])
            src = W if not any(prod) else tuple(f.data for f in Wq.factors)
            self.planes_items.append((src, planes, lay, slots, self.planes_cache[gk]))
        return self.planes_cache[ck]

    def value_amax(self, v, view, need):
        """Slot of a forward value used as a GEMM operand: the producer's, or measured now (once)."""
        if self.amax_pool is None:
            return None
        if v.amax is None:
            v.amax = self.new_amax()
            # (a producer may have left partial maxima of the whole value: a bound of any view of it)
            src = getattr(v, "amax_src", None)
            need.append((src if src is not None else view, v.amax))
        return v.amax

    def grad_amax(self, v, need):
        """Slot of v.grad as a GEMM operand, called when every writer of the gradient has been recorded: the slot the
        writers raised if ALL of them did, else measured now."""
        if self.amax_pool is None:
            return None
        if v.gamax is not None and v.gamax_writers == v.written and v.written > 0:
            return v.gamax
        v.gamax = self.new_amax()
        v.gamax_writers = v.written
        need.append((v.grad, v.gamax))
        return v.gamax

    def amax_call(self, need, **meta):
        arr = ops.make_amax_descs(need)
        self.keep.append(arr)
        m = dict(kernel="amax_kernel", bytes=4.0 * sum(t.numel() for t, _ in need), need=list(need))
        m.update(meta)
        return (L.load().mml_amax_batch, (arr, len(need)), m)

    # ---- execution ---------------------------------------------------------------------------
    @staticmethod
    def _run(calls):
        s = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
        if profiling.enabled:  # --profile: one roctx range per call, named after its kernel
            for c in calls:
                if c[0] is INLINE:  # (stream fork / join: the calls it issues open their own ranges)
                    c[1](*c[2])
                    continue
                meta = c[3] if (c[0] is PY and len(c) > 3) else (c[2] if (c[0] is not PY and len(c) > 2) else {})
                with profiling.range(meta.get("kernel") or getattr(c[1] if c[0] is PY else c[0], "__name__", "call")):
                    if c[0] is PY:
                        c[1](*c[2])
                    else:
                        rc = c[0](*c[1], s)
                        if rc:
                            L.check(rc, c[0].__name__)
            return
        for c in calls:
            if c[0] is PY or c[0] is INLINE:
                c[1](*c[2])
                continue
            rc = c[0](*c[1], s)
            if rc:
                L.check(rc, c[0].__name__)

    @staticmethod
    def run_timed(calls, acc):
        """Diagnostic replay: brackets every C-ABI call with HIP events on the launch stream and adds
        (milliseconds, launches) per call label into `acc` (used by bench.py for the roofline line)."""
        s = torch.cuda.current_stream()
        evs = []
        for c in calls:
            if c[0] is INLINE:  # (stream fork / join: run in place, untimed -- what it issues is not this stream's time)
                c[1](*c[2])
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            if c[0] is PY:
                c[1](*c[2])
            else:
                rc = c[0](*c[1], s.cuda_stream)
                if rc:
                    L.check(rc, c[0].__name__)
            b.record(s)
            label = None
            meta0 = c[2] if (c[0] is not PY and len(c) > 2 and isinstance(c[2], dict)) else {}
            if str(meta0.get("kernel", "")).startswith("gemm<"):
                label = L.load().mml_gemm_last_kernel().decode() or None
            evs.append((c, a, b, label))
        torch.cuda.synchronize()
        for c, a, b, label in evs:
            if c[0] is PY:
                meta = c[3] if len(c) > 3 else {"kernel": getattr(c[1], "__name__", "python")}
            else:
                meta = c[2] if len(c) > 2 else {"kernel": c[0].__name__}
            e = acc.setdefault(label or meta["kernel"], {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0,
                                                         "hbm_bytes": 0.0})
            e["ms"] += a.elapsed_time(b)
            e["launches"] += 1
            e["flops"] += meta.get("flops", 0.0)
            e["bytes"] += meta.get("bytes", 0.0)
            # compulsory HBM bytes of the call (every distinct operand read once, every output written once): the GEMM
            # launches are priced in FLOPs (`flops`) but still move their operands -- bench.py's whole-step fraction
            e["hbm_bytes"] = e.get("hbm_bytes", 0.0) + meta.get("hbm_bytes", meta.get("bytes", 0.0))

    def run_forward(self):
        self._run(self.fwd)
        self._run(self.head_infer)

    def run_train_fwd_bwd(self):
        """forward + summed-BCE loss + backward (gradients land in the PVal.grad buffers)."""
        self._run(self.fwd)
        self._run(self.head_train)
        self._run(self.bwd)
        self._run(self.bwd_tail)
        self._run(self.head_side)
        self._run(self.bwd_side)

    def run_backward_from_dprob(self):
        self._run(self.head_bwd)
        self._run(self.bwd)
        self._run(self.bwd_tail)
        self._run(self.bwd_side)

    # ---- graph recording ---------------------------------------------------------------------
    def add(self, op):
        self.ops.append(op)
        seen = set()
        for v in op.inputs():
            if isinstance(v, Val) and v.needs_grad and id(v) not in seen:
                seen.add(id(v))
                v.consumers.append(op)
        self.fwd.extend(op.fwd_calls(self))
        return op

    def finish(self, head_op):
        """Record the head op and derive the backward call list."""
        self.head_op = head_op
        for v in head_op.inputs():
            if isinstance(v, Val) and v.needs_grad:
                v.consumers.append(head_op)
        self.head_infer = head_op.infer_calls(self)
        if not self.training:
            self._amax_prologue()
            self._cast16_prologue()
            return
        calls = head_op.train_calls(self, use_dprob=False)
        is_side = lambda c: isinstance(c[-1], dict) and c[-1].get("side")  # noqa: E731
        self.head_train = [c for c in calls if not is_side(c)]
        # (belongs to head_train, not to the backward every path shares: run_backward_from_dprob has its own head call)
        self.head_side = [c for c in calls if is_side(c)]
        self.head_bwd = head_op.train_calls(self, use_dprob=True, claim=False)
        for op in reversed(self.ops):
            for v in op.outputs():
                if isinstance(v, Val) and v.grad is not None and v.act != L.ACT_NONE and not v.deriv_applied:
                    if v.is16 or v.grad.dtype != torch.float32:
                        raise L.MMLError(f"bf16 value {v.name!r}: its activation derivative must fold into its one consumer")
                    self.bwd.append((L.load().mml_act_bwd, (v.buf.data_ptr(), v.grad.data_ptr(), v.grad.data_ptr(),
                                                            self._flat_numel(v), v.act)))
                    v.deriv_applied = True
            mine = []
            for c in op.bwd_calls(self):
                meta = c[-1] if isinstance(c[-1], dict) else {}
                if meta.get("side"):
                    self.bwd_side.append(c)
                    mine.append(c)
                elif meta.get("tail"):
                    self.bwd_tail.append(c)
                else:
                    self.bwd.append(c)
            # a side call reads dL/d(this op's outputs) and forward values only: it may start once the chain has issued
            # everything up to and including this op's own entries (TrainStep's early fork of the side stream)
            for c in mine:
                c[-1]["ready"] = len(self.bwd)
        self._amax_prologue()
        self._cast16_prologue()
        if os.environ.get("MMLREC_MERGE_COPIES", "1") != "0":
            for name in ("fwd", "bwd", "bwd_tail", "bwd_side", "head_train", "head_bwd"):
                where = [] if name == "bwd" else None
                setattr(self, name, self._merge_copies(getattr(self, name), where=where))
                if name == "bwd" and where:
                    # the side calls' `ready` tags count entries of the UNMERGED chain: into the merged list's index
                    # space (ready = k: the first k entries have been issued -> everything up to the merged entry that
                    # holds old entry k - 1)
                    for c in list(self.bwd_side) + list(self.head_side):
                        m = c[-1] if isinstance(c[-1], dict) else None
                        if m is not None and m.get("ready"):
                            m["ready"] = where[min(m["ready"], len(where)) - 1] + 1
        # (Measured on MI355X: issuing every weight-gradient partial-product GEMM before the first reduction -- the
        # phased wgrad entry point allows it -- makes the step SLOWER, 2.35 ms vs 2.19 ms: the GEMMs then run next to
        # the table scatter and the dense table optimizer for longer and all of them are HBM-bound together.  The
        # list stays in program order: partial products and reduction of one layer back to back.)

    def _merge_copies(self, calls, lib=None, where=None):
        """Runs of neighbouring strided copies (mml_copy2d / mml_copy2d_batch: concat / split of feature blocks, gradient
        hand-overs of shared parameters) as ONE launch each -- inside a step's graph every launch takes >= 4.6 us from
        start to end, and PepNet's step had four of them in a row three times.  A copy joins the run only if it touches
        nothing an earlier copy of the run writes, and writes nothing an earlier one reads (one launch has no order).
        where: a list that receives, per input call, the index of the output entry it went into."""
        lib = lib or L.load()  # (tests/test_plan_passes_cpu.py passes stand-ins: only the functions' identity is used)
        f1, fb = lib.mml_copy2d, lib.mml_copy2d_batch

        def descs_of(c):
            if c[0] is f1:
                src, lds, dst, ldd, rows, cols, acc = c[1]
                return [(src, lds, dst, ldd, rows, cols, acc, None)]
            arr, n = c[1]
            # (the 8th entry: an item's optional magnitude slot, mml_copy2d_desc.amax_out -- stand-in arrays of the CPU
            #  tests need not have the field)
            return [(arr[k].src, arr[k].lds, arr[k].dst, arr[k].ldd, arr[k].rows, arr[k].cols, arr[k].accumulate,
                     getattr(arr[k], "amax_out", None)) for k in range(n)]

        def span(ptr, ld, rows, cols):
            return (ptr, ld, rows, cols)

        def hits(a, b):
            """Do the [rows, cols] regions a, b (pointer, pitch, rows, cols; float32) share an element?  Exact for regions of
            one pitch (column blocks of one buffer: the concat / split case), the byte-interval test otherwise."""
            (pa, la, ra, ca), (pb, lb, rb, cb) = a, b
            ea, eb = pa + 4 * ((max(ra, 1) - 1) * la + ca), pb + 4 * ((max(rb, 1) - 1) * lb + cb)
            if not (pa < eb and pb < ea):
                return False
            if la == lb and la > 0 and (pb - pa) % 4 == 0:
                delta = (pb - pa) // 4
                q, r = divmod(delta, la)   # b's origin in a's grid: row q, column r (Python's floor semantics)
                if ca <= la and r + cb <= la:
                    rows_meet = q < ra and q + rb > 0
                    cols_meet = r < ca and r + cb > 0
                    return rows_meet and cols_meet
            return True

        out, run, meta_run = [], [], []
        pos = [0] * len(calls)
        run_src = []

        def flush():
            if not run:
                return
            for i_ in run_src:
                pos[i_] = len(out)
            run_src.clear()
            if len(meta_run) == 1:
                out.append(meta_run[0])
            else:
                arr = (L.Copy2dDesc * len(run))()
                for d, (src, lds, dst, ldd, rows, cols, acc, am) in zip(arr, run):
                    d.src, d.lds, d.dst, d.ldd, d.rows, d.cols, d.accumulate = src, lds, dst, ldd, rows, cols, acc
                    if am:
                        d.amax_out = am
                self.keep.append(arr)
                meta = dict(kernel="copy2d_batch_kernel",
                            bytes=sum(8.0 * r[4] * r[5] for r in run))
                for c in meta_run:  # (the scheduling tags of the merged calls: they were neighbours of ONE list)
                    m = c[-1] if isinstance(c[-1], dict) else {}
                    for k in ("side", "tail", "rank", "ready"):
                        if k in m:
                            meta[k] = max(meta.get(k, m[k]), m[k]) if k == "ready" else m[k]
                out.append((fb, (arr, len(run)), meta))
            run.clear()
            meta_run.clear()

        for ci, c in enumerate(calls):
            if c[0] is f1 or c[0] is fb:
                ds = descs_of(c)
                ok = len(run) + len(ds) <= 32
                for (src, lds, dst, ldd, rows, cols, acc, _am) in ds:
                    rs, ws = span(src, lds, rows, cols), span(dst, ldd, rows, cols)
                    for (s2, l2, d2, ld2, r2, c2, a2, _am2) in run:
                        rs2, ws2 = span(s2, l2, r2, c2), span(d2, ld2, r2, c2)
                        if hits(rs, ws2) or hits(ws, rs2) or hits(ws, ws2):
                            ok = False
                if not ok:
                    flush()
                run.extend(ds)
                meta_run.append(c)
                run_src.append(ci)
            else:
                flush()
                pos[ci] = len(out)
                out.append(c)
        flush()
        if where is not None:
            where[:] = pos
        return out

    def _cast16_prologue(self):
        """bf16-storage path: the bf16 copies of the weights, ONE launch in front of everything else; and the promise
        behind every bf16 value -- only bf16-storage layer groups read it -- is checked."""
        for op in list(self.ops) + [getattr(self, "head_op", None)]:
            for v in (op.inputs() if op is not None else []):
                if isinstance(v, Val) and v.is16 and isinstance(op, GateGroupOp) and any(v is e for e in op.experts):
                    continue  # (the fast gate kernels read bf16 expert outputs: mml_gate_group.out_bf16 bit 3)
                if isinstance(v, Val) and v.is16 and not (isinstance(op, LinearGroupOp) and op.use16):
                    raise L.MMLError(f"bf16 value {v.name!r} is read by {type(op).__name__}: only bf16-storage layer "
                                     "groups may read a store16 value")
        if not self.cast16_items or getattr(self, "_cast16_done", False):
            return
        arr = ops.make_cast16_descs(self.cast16_items)
        self.keep.append(arr)
        n = sum(w.numel() for w, _, _ in self.cast16_items)
        self.fwd.insert(0, (L.load().mml_cast16_batch, (arr, len(self.cast16_items)),
                            dict(kernel="cast16_kernel", bytes=6.0 * n)))
        self.n_pre += 1
        self._cast16_done = True

    def _amax_prologue(self):
        """Zero EVERY magnitude slot of the plan (forward and backward ones: the producers only ever raise them) and
        measure the stable weights, as the first entries of `fwd`.  Called when the whole plan has been recorded."""
        if self.amax_pool is None or not self.amax_next or self.n_pre:
            return
        lib = L.load()
        pre = [(lib.mml_amax_reset, (self.amax_pool.data_ptr(), self.amax_next),
                dict(kernel="amax_reset", bytes=32.0 * self.amax_next))]
        cut = []
        if self.planes_items:  # (after the magnitudes of the weights: the cut reads them)
            arr = ops.make_planes_descs(self.planes_items)
            self.keep.append(arr)
            cut.append((lib.mml_gemm_planes_cut, (arr, len(self.planes_items)),
                        dict(kernel="planes_cut_kernel", bytes=8.0 * sum((it[0][0] if isinstance(it[0], tuple) else it[0]).numel()
                                                                        for it in self.planes_items))))
        # The weights' magnitudes ride in the magnitude launch that stands in front of the first GEMM anyway (the pass over
        # the gathered input): one launch fewer at the head of the step (~6 us of a 1.7 ms step).  Nothing in front of the
        # first GEMM reads a weight's slot or planes.  MMLREC_AMAX_MERGE=0: the separate launch of round 3.
        first_gemm = next((i for i, c in enumerate(self.fwd) if c[0] in (lib.mml_gemm_grouped_fwd, lib.mml_pep_gate_fwd)),
                          len(self.fwd))
        host = next((i for i, c in enumerate(self.fwd[:first_gemm]) if c[0] is lib.mml_amax_batch and
                     isinstance(c[-1], dict) and "need" in c[-1]), None)
        if self.amax_wlist and host is not None and os.environ.get("MMLREC_AMAX_MERGE", "1") != "0":
            c = self.fwd[host]
            merged = self.amax_call(c[-1]["need"] + self.amax_wlist,
                                    **{k: v for k, v in c[-1].items() if k not in ("kernel", "bytes", "need")})
            self.fwd = self.fwd[:host] + [merged] + cut + self.fwd[host + 1:]
        else:
            if self.amax_wlist:
                pre.append(self.amax_call(self.amax_wlist))
            pre += cut
        self.fwd = pre + self.fwd
        self.n_pre = len(pre)

    def merge_wgrad(self):
        """Small batches: every weight-gradient GEMM of the step in ONE grouped launch (+ one reduction) instead of one
        pair per layer.  At M = 4 096 a layer's launch fills a fraction of the chip for ~15 us; together they take the
        time of the longest (lazy_exact step on AE-30: 0.327 -> see DESIGN 10.12).  Only when every problem writes its
        own dW (a weight shared by two layers is written by two launches in order).  At large batches the per-layer
        order stays: each launch fills the chip by itself and the reductions interleave with the next GEMM."""
        lib = L.load()
        fn = lib.mml_gemm_grouped_wgrad_phase
        idx = [i for i, c in enumerate(self.bwd_side) if c[0] is fn]
        if len(idx) <= 2:
            return False
        descs, groups = [], []
        for i in idx:
            c = self.bwd_side[i]
            if c[1][4] == 1:  # the partial-product phase carries the problems (phase 2 repeats them)
                groups.append([c[1][0][k] for k in range(c[1][1])])
                descs += groups[-1]
        chunks = [descs]  # (small batches: ONE call, the library splits it into groups of MML_MAX_GROUP for the tile kernel)
        if self.B >= 16384:
            # Large batches (gemm_nt_kernel: 128 x 128 output tiles, the batch cut into `slabs` pieces so that tiles x
            # slabs fill the chip's 512 workgroup slots once; at most MML_MAX_GROUP problems per launch): merge when the
            # merged launches take fewer batch steps than the per-layer launches together, a launch + reduction pair
            # priced at ~8 steps.  AE-30: 20 / 8 / 2 tiles -> 82 + 32 + 32 steps per layer against 121 merged (measured
            # 1.65 -> 1.565 ms); KuaiRec-32: 72 / 32 / 4 tiles -> 293 + 128 + 32 against 512 merged (108 tiles x 4 slabs
            # leave 80 slots idle: 3.33 -> 3.38 ms merged, so it stays per layer); PepNet's 40-odd small problems go
            # into launches of 16 (2.21 -> 2.13 ms already as ONE call that fell to the tile kernel).
            steps = self.B // 32
            # (what gemm_nt_kernel takes -- csrc/gemm_nt.hip, mml_gemm_nt_try_wgrad -- goes together: ONE problem it does
            # not take, e.g. a final layer with a single output row, would send its whole launch to the tile kernel)
            serves = [bool(lib.mml_gemm_nt_serves(C.byref(d))) for d in descs]  # (the library's own predicate)
            fits = [d for d, ok in zip(descs, serves) if ok]
            other = [d for d, ok in zip(descs, serves) if not ok]
            chunks = []
            # (round 6: gemm_nt_kernel takes 48 problems per launch; MMLREC_NT_GROUP=16 restores the launches of round 5)
            nt_group = min(L.NT_MAX_GROUP, max(1, int(os.environ.get("MMLREC_NT_GROUP", str(L.NT_MAX_GROUP)))))
            for part in (fits, other):
                if part:
                    nch = -(-len(part) // (nt_group if part is fits else L.MAX_GROUP))
                    per = -(-len(part) // nch)
                    chunks += [part[i:i + per] for i in range(0, len(part), per)]

            def cost(tiles):
                sl = max(1, min(512 // max(tiles, 1), steps // 8, 64))
                return -(-steps // sl) * -(-tiles * sl // 512) + 8

            def tiles_of(g):
                return sum(-(-d.N // 128) * -(-d.K // 128) for d in g)

            if sum(cost(tiles_of(g)) for g in chunks) >= sum(cost(tiles_of(g)) for g in groups):
                return False
        targets = [d.dW for d in descs] + [d.dbias for d in descs if d.dbias]
        if len(set(targets)) != len(targets) or any(d.accumulate for d in descs):
            return False
        flops = sum(self.bwd_side[i][2].get("flops", 0.0) for i in idx)
        hbm = sum(self.bwd_side[i][2].get("hbm_bytes", 0.0) for i in idx)
        merged, reduces = [], []
        for ch in chunks:
            arr = (L.GemmWgradDesc * len(ch))()
            for k, d in enumerate(ch):
                C.memmove(C.byref(arr[k]), C.byref(d), C.sizeof(L.GemmWgradDesc))
            nbytes = lib.mml_gemm_grouped_wgrad_workspace_bytes(arr, len(ch))
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self.keep += [arr, ws]
            share = len(ch) / float(len(descs))
            merged.append((fn, (arr, len(ch), ws.data_ptr(), ws.numel(), 1),
                           dict(kernel=_gemm_symbol(False, False, [], 2), flops=flops * share, hbm_bytes=hbm * share,
                                side=True, rank=0)))
            reduces.append((fn, (arr, len(ch), ws.data_ptr(), ws.numel(), 2),
                            dict(kernel="slab_reduce", bytes=float(nbytes), side=True, rank=1)))
        rest = [c for i, c in enumerate(self.bwd_side) if i not in set(idx)]  # (un-padding copies: after the reduction)
        # (each reduction right behind its launch: the next launch's first tiles start beside it)
        self.bwd_side = [c for pair in zip(merged, reduces) for c in pair] + rest
        return True

    def fuse_tower_head(self):
        """K5' (csrc/tower_head.hip): the last tower layer of every task, the heads + summed BCE and the towers' input gradient
        -- three launches of the recorded step -- as ONE launch (+ its share of the batched reduction), when the recorded
        lists hold exactly that pattern: a forward launch of T Linear + ReLU problems whose outputs are the T heads' inputs and
        nothing else's, the deferred head launch, and an input-gradient launch of T single-source problems over the heads'
        dH.  Rewrites fwd / head_train / head_side / bwd of THIS plan (a TrainStep's own: the forward-only and the dL/dprob
        lists of a model's cached plans are never touched).  MMLREC_TOWER_HEAD=0: off."""
        lib = L.load()
        if (os.environ.get("MMLREC_TOWER_HEAD", "1") == "0" or self.amax_pool is None or self.bf16 or
                self.device.type != "cuda"):
            return False
        fh, ff, fd = lib.mml_head_bce_fwd_bwd_phase, lib.mml_gemm_grouped_fwd, lib.mml_gemm_grouped_dgrad
        if len(self.head_train) != 1 or len(self.head_side) != 1:
            return False
        hc, hs = self.head_train[0], self.head_side[0]
        if hc[0] is not fh or hs[0] is not fh or hc[1][3] != 1 or hs[1][3] != 2:
            return False
        grp = hc[1][0]._obj
        T = int(grp.n_heads)
        if grp.dh_bf16 or grp.dprob or not grp.y or not grp.prob or T < 1:
            return False
        heads = [grp.head[t] for t in range(T)]
        if any(h.gate or h.w2 or not h.dH or not h.h_relu or not h.dw or not h.dbias for h in heads):
            return False
        hin = {int(h.Hin): t for t, h in enumerate(heads)}
        dh = {int(h.dH): t for t, h in enumerate(heads)}
        if len(hin) != T or len(dh) != T:
            return False
        # the forward launch that writes the heads' inputs
        fi = None
        for i in range(len(self.fwd) - 1, -1, -1):
            c = self.fwd[i]
            if c[0] is ff and c[1][1] == T and all(int(c[1][0][k].C or 0) in hin for k in range(T)):
                fi = i
                break
        if fi is None:
            return False
        fdesc = self.fwd[fi][1][0]
        # the input-gradient launch over the heads' dH
        di = None
        for i, c in enumerate(self.bwd):
            if c[0] is fd and c[1][1] == T and all(c[1][0][k].n_src == 1 and int(c[1][0][k].dC[0] or 0) in dh for k in range(T)):
                di = i
                break
        if di is None:
            return False
        ddesc = self.bwd[di][1][0]
        by_t_f = {hin[int(fdesc[k].C)]: fdesc[k] for k in range(T)}
        by_t_d = {dh[int(ddesc[k].dC[0])]: ddesc[k] for k in range(T)}
        if len(by_t_f) != T or len(by_t_d) != T:
            return False
        g = L.TowerHeadGroup()
        g.n, g.M = T, int(grp.B)
        g.prob, g.ldprob, g.y, g.ldy, g.mask, g.ldmask, g.loss = grp.prob, grp.ldprob, grp.y, grp.ldy, grp.mask, grp.ldmask, grp.loss
        for t in range(T):
            f, d, h = by_t_f[t], by_t_d[t], heads[t]
            if (f.act != L.ACT_RELU or f.w_kn or f.mul or not f.w_planes or not f.w_kexp or not f.amax_a or f.M != g.M or
                    f.N != h.H or int(f.ldc) != int(h.ldh)):
                return False
            if (d.gate_h or d.Y or d.relu_mask or d.act != L.ACT_NONE or d.accumulate or d.w_kn[0] or not d.w_planes[0] or
                    not d.w_kexp[0] or not d.dA or d.K != f.K or d.N[0] != f.N or int(d.W[0] or 0) != int(f.W or 0) or
                    int(d.lddc[0]) != int(h.lddh)):
                return False
            q = g.t[t]
            q.A, q.lda, q.amax_a, q.K, q.N = f.A, f.lda, f.amax_a, f.K, f.N
            q.w_planes_fwd, q.ldpf, q.kexp_fwd = f.w_planes, f.ldw, f.w_kexp
            q.w_planes_bwd, q.ldpb, q.kexp_bwd = d.w_planes[0], d.ldw[0], d.w_kexp[0]
            q.bias1, q.w, q.hbias, q.hbias2, q.n_hbias2 = f.bias, h.w, h.bias, h.bias2, h.n_bias2
            q.dH, q.lddh, q.dA, q.ldda, q.dw, q.dhbias = h.dH, h.lddh, d.dA, d.ldda, h.dw, h.dbias
            q.amax_dH, q.amax_dA = grp.amax_dH, d.amax_out
            q.mask_col, q.head = h.mask_col, t
        if not lib.mml_tower_head_serves(C.byref(g)):
            return False
        nws = int(lib.mml_tower_head_workspace_bytes(C.byref(g)))
        ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=self.device)
        self.keep += [g, ws]
        K, N = int(g.t[0].K), int(g.t[0].N)
        byts = 4.0 * g.M * T * (2 * K + N + 3)
        fused = (lib.mml_tower_head_fwd_bwd, (C.byref(g), ws.data_ptr(), ws.numel(), 1),
                 dict(kernel="tower_head_kernel", bytes=byts, hbm_bytes=byts))
        red = (lib.mml_tower_head_fwd_bwd, (C.byref(g), ws.data_ptr(), ws.numel(), 2),
               dict(kernel="slab_reduce", bytes=float(nws), side=True, rank=1, ready=0))
        del self.fwd[fi]
        del self.bwd[di]
        for c in list(self.bwd_side) + list(self.head_side):  # (`ready` counts entries of the backward chain)
            m = c[-1] if isinstance(c[-1], dict) else None
            if m is not None and m.get("ready", 0) > di:
                m["ready"] -= 1
        self.head_train = [fused]
        self.head_side = [red]
        self.tower_head = g
        return True

    def merge_row_reduces(self, lib=None):
        """The deferred reductions of the head / gate kernels' partial sums (`head_side`, and the gate groups' entries of
        `bwd_side`: only the optimizer and the host read their results) as ONE launch in front of the weight gradients
        (as few as the launch's segment capacity allows)."""
        lib = lib or L.load()  # (tests/test_plan_passes_cpu.py passes stand-ins: only the functions' identity is used)
        fh, fg = lib.mml_head_bce_fwd_bwd_phase, lib.mml_gate_mix_bwd_phase
        ft = getattr(lib, "mml_tower_head_fwd_bwd", None)  # (K5': Plan.fuse_tower_head)
        is_red = lambda c: c[0] in (fh, fg, ft) and c[0] is not None and c[1][3] == 2  # noqa: E731
        picked = [c for c in list(self.head_side) + list(self.bwd_side) if is_red(c)]
        if len(picked) < 2:
            return False

        def segments(c):
            """Reduction segments the C side makes of this item (csrc/gate_head.hip, phase 2): a head group one per dw and
            dbias of every head plus the loss, a gate group one per active gate's dWg."""
            g = c[1][0]._obj
            if c[0] is fh:
                return 2 * int(g.n_heads) + (1 if g.loss else 0)
            if c[0] is ft:
                return 2 * int(g.n) + (1 if g.loss else 0)
            return sum(1 for k in range(int(g.n_gates)) if g.gate[k].active)

        # one launch takes at most MAX_REDUCE_SEGS segments (csrc/reduce.hpp): a deep PLE (7 tasks x 4 levels: 15 + 8 + 8 +
        # 8 + 7 = 46) goes into as many launches as it needs, in list order
        chunks, cur, nseg = [], [], 0
        for c in picked:
            s = segments(c)
            if s > L.MAX_REDUCE_SEGS:
                return False  # (a single group beyond the launch's capacity: leave every reduction where it was)
            if cur and nseg + s > L.MAX_REDUCE_SEGS:
                chunks.append(cur)
                cur, nseg = [], 0
            cur.append(c)
            nseg += s
        chunks.append(cur)
        calls = []
        for ch in chunks:
            if len(ch) == 1:  # (nothing to merge it with: its own phase-2 call, on the side list)
                calls.append(ch[0])
                continue
            items = (L.RowsReduceItem * len(ch))()
            for it, c in zip(items, ch):
                grp, ws, nbytes, _ = c[1]
                it.kind = (L.ROWS_REDUCE_HEAD if c[0] is fh else
                           (L.ROWS_REDUCE_TOWER_HEAD if c[0] is ft else L.ROWS_REDUCE_GATE))
                it.group = C.addressof(grp._obj)  # (the ops pass C.byref(group); the group itself lives in plan.keep)
                it.workspace, it.workspace_bytes = ws, nbytes
            self.keep.append(items)
            calls.append((lib.mml_rows_reduce_batch, (items, len(ch)),
                          dict(kernel="slab_reduce", bytes=sum(c[2].get("bytes", 0.0) for c in ch), side=True, rank=1,
                               ready=max(c[2].get("ready", 0) for c in ch))))
        self.head_side = [c for c in self.head_side if not is_red(c)]
        self.bwd_side = calls + [c for c in self.bwd_side if not is_red(c)]
        return True

    def merge_wgrad16(self):
        """bf16-storage path (csrc/gemm16.hip: mml_g16_wgrad, at most G16_MAX_GROUP problems per launch, 128 x 128 tiles
        of 64-row steps, at most 32 slabs): neighbouring weight-gradient launches go together where the tile model says
        the merged launch takes fewer steps -- KuaiRec-32: towers (4 tiles -> 128 workgroups for 512 slots) + second
        expert layers (32 tiles) as one launch of 36 tiles."""
        lib = L.load()
        fn = lib.mml_g16_wgrad
        idx = [i for i, c in enumerate(self.bwd_side) if c[0] is fn and c[1][4] == 1]
        if len(idx) < 2 or any(self.bwd_side[i + 1][0] is not fn or self.bwd_side[i + 1][1][4] != 2 for i in idx):
            return False
        steps = max(self.B // 64, 1)

        def cost(g):
            tiles = sum((d.N // 128) * (d.K // 128) for d in g)
            sl = max(1, min(512 // max(tiles, 1), steps // 4, 32))
            return -(-steps // sl) * -(-tiles * sl // 512) + 8

        groups = [[self.bwd_side[i][1][0][k] for k in range(self.bwd_side[i][1][1])] for i in idx]
        metas = [self.bwd_side[i][2] for i in idx]
        out, om = [groups[0]], [dict(metas[0])]
        for g, m in zip(groups[1:], metas[1:]):
            cur = out[-1]
            tg = [d.dW for d in cur + g] + [d.dbias for d in cur + g if d.dbias]
            if (len(cur) + len(g) <= L.G16_MAX_GROUP and cost(cur + g) < cost(cur) + cost(g) and
                    len(set(tg)) == len(tg) and not any(d.accumulate for d in cur + g)):
                out[-1] = cur + g
                for k in ("flops", "hbm_bytes"):
                    om[-1][k] = om[-1].get(k, 0.0) + m.get(k, 0.0)
            else:
                out.append(g)
                om.append(dict(m))
        if len(out) == len(groups):
            return False
        calls = []
        for g, m in zip(out, om):
            arr = (L.G16WgradDesc * len(g))()
            for k, d in enumerate(g):
                C.memmove(C.byref(arr[k]), C.byref(d), C.sizeof(L.G16WgradDesc))
            nbytes = int(lib.mml_g16_wgrad_workspace_bytes(arr, len(g)))
            if nbytes < 0:
                L.check(-1, "mml_g16_wgrad_workspace_bytes")
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
            self.keep += [arr, ws]
            m["kernel"] = "g16_nt_kernel(wgrad %d problems)" % len(g)
            calls.append((fn, (arr, len(g), ws.data_ptr(), ws.numel(), 1), m))
            calls.append((fn, (arr, len(g), ws.data_ptr(), ws.numel(), 2),
                          dict(kernel="g16_reduce_kernel", bytes=float(nbytes), side=True, rank=1)))
        drop = set(idx) | {i + 1 for i in idx}
        first = idx[0]
        self.bwd_side = ([c for i, c in enumerate(self.bwd_side) if i < first and i not in drop] + calls +
                         [c for i, c in enumerate(self.bwd_side) if i > first and i not in drop])
        return True

    def _flat_numel(self, v):
        # act_bwd is a flat kernel: value and gradient must share the padded pitch (they do by construction)
        if v.buf.stride(0) != v.grad.stride(0):
            raise L.MMLError("value / gradient pitch mismatch for " + v.name)
        if v.buf.stride(0) != v.n and v.buf.storage_offset() % v.buf.stride(0) != 0:
            raise L.MMLError("cannot finalise a column-sliced value: " + v.name)
        return self.B * v.buf.stride(0) if v.buf.stride(0) != v.n else self.B * v.n


# ==================================================================================================
# ops
# ==================================================================================================
def _distinct_bytes(tensors):
    """Bytes of the distinct buffers among `tensors` (torch tensors / None; a view counts its own elements)."""
    seen, n = set(), 0
    for t in tensors:
        if t is None:
            continue
        key = (t.data_ptr(), tuple(t.shape))
        if key not in seen:
            seen.add(key)
            n += t.numel() * t.element_size()
    return float(n)


def _gemm_symbol(arc, brc, cols, epi, kreds=(), tensors=(), nrc_extents=()):
    """Provisional label of a grouped GEMM launch; Plan.run_timed replaces it with the symbol the C side actually
    launched (mml_gemm_last_kernel), which depends on the tile / arithmetic choice made in csrc/gemm.hip."""
    return "gemm<%s>" % ("fwd", "dgrad", "wgrad")[epi]


def scatter_symbol(E):
    """Kernel symbol of a scatter / index-unique launch (csrc/gather_scatter.hip: scatter_impl's choice)."""
    return "scatter_fold_kernel" if E in (4, 8, 16) else ("scatter_hash_kernel" if E <= 16 else "scatter_atomic_kernel")


def _opt_dense_symbol(numel, ntensors, form=0):
    """Kernel symbol of a dense optimizer launch (csrc/optim_ew.hip: mml_opt_step_dense's choice).  form: the loop form
    of the streaming kernel -- 0 plain, 1 two chunks per iteration (MMLREC_OPT_VARIANT bit 1), 2 untouched rows of the
    split update under a capped grid, 3 marked gradients under a capped grid."""
    if not (numel >= (1 << 24) and (ntensors <= 4 or form == 3)):
        return "opt_flat_kernel"
    u = {0: 1, 1: 1, 2: 4, 3: int(os.environ.get("MMLREC_OPT_U", "2"))}[form]
    return "opt_dense_kernel<true, %d, %d>" % (form, u)


class Op:
    def writes_grad16(self, plan, v):
        """bf16-storage path: this op, as the ONLY consumer of v, writes dL/dv once and can write it as bf16."""
        return False

    def inputs(self):
        return []

    def outputs(self):
        return []

    def fwd_calls(self, plan):
        return []

    def bwd_calls(self, plan):
        return []


class GatherOp(Op):
    """K1/K2: multi-field gather (+dense copy) and its scatter backward."""

    def __init__(self, tables, X, cols, dense_col0, nd, out, sparse_rows=None):
        self.tables, self.X, self.cols, self.dense_col0, self.nd, self.out = tables, X, cols, dense_col0, nd, out
        self.sparse_rows = sparse_rows  # TableRows bookkeeping (seen bitmaps, touched list) or None
        self.mark_rows = None  # TableRows: the forward marks the rows it reads and lists them (split dense update)
        # uint8 map (ops.marks_bytes layout): the scatter marks every row it adds to, the dense table optimizer skips
        # the gradient read of unmarked rows and clears the marks (mml_opt_tensor.grad_marks)
        self.grad_marks = None
        # True: the backward is mml_scatter_bwd_det -- order-independent integer fixed-point sums, bitwise repeatable
        # (BaseModel.scatter_mode = "deterministic"); needs store.ensure_det(tables)
        self.deterministic = None

    def outputs(self):
        return [self.out]

    def index_view(self, plan):
        """(index matrix, rows) the table update of this step is built from (the global batch under replication)."""
        return self.X, plan.B

    def pre_index_calls(self, plan):
        return []

    def fwd_calls(self, plan):
        lib = L.load()
        F = len(self.tables)
        E = self.tables[0].data.shape[1]
        tabs = ops._ptr_array([t.data for t in self.tables])
        vocab = (L.i64 * F)(*[t.data.shape[0] for t in self.tables])
        col = (L.i32 * F)(*self.cols)
        plan.keep += [tabs, vocab, col]
        meta = dict(kernel="gather_vec4_kernel" if E % 4 == 0 else "gather_scalar_kernel",
                    bytes=float(plan.B) * (F * (4 + 8 * E) + 8 * self.nd))  # index + row read + row write
        mr = self.mark_rows
        if self.out.is16:  # bf16-storage path: dnn_input leaves the gather as the first layers' GEMM operand
            if mr is not None:
                raise L.MMLError("the row-marking gather has no bf16 form (split dense update)")
            meta16 = dict(kernel="gather16_kernel", bytes=float(plan.B) * (F * (4 + 6 * E) + 6 * self.nd))
            return [(lib.mml_gather16_fwd, (tabs, vocab, col, F, E, self.X.data_ptr(), ops._ld(self.X), self.dense_col0,
                                            self.nd, plan.B, self.out.buf.data_ptr(), ops._ld(self.out.buf),
                                            plan.status.data_ptr()), meta16)]
        if mr is not None:
            ps = ops._ptr_array(mr.seen)
            rb = (L.i64 * (F + 1))(*mr.rowbase)
            plan.keep += [ps, rb]
            return [(lib.mml_gather_fwd_mark, (tabs, vocab, col, F, E, self.X.data_ptr(), ops._ld(self.X),
                                               self.dense_col0, self.nd, plan.B, self.out.buf.data_ptr(),
                                               ops._ld(self.out.buf), mr.marks.data_ptr(), plan.status.data_ptr()), meta),
                    (lib.mml_rows_compact, (ps, vocab, rb, F, mr.touched.data_ptr(), mr.count.data_ptr(),
                                            mr.touched.numel(), mr.marks.data_ptr()),
                     dict(kernel="rows_compact_kernel", bytes=float(mr.marks.numel())))]
        out = self.out.buf
        nwg = int(lib.mml_gather_wgmax_len(F, E, self.nd, plan.B)) if plan.amax_pool is not None else 0
        if (nwg > 0 and os.environ.get("MMLREC_GATHER_WGMAX", "1") != "0" and ops._ld(out) % 4 == 0 and
                out.data_ptr() % 16 == 0 and all(t.data.data_ptr() % 16 == 0 for t in self.tables)):
            # the magnitude of the gathered input comes out of the gather itself: one value per workgroup, read by the
            # magnitude launch in front of the first GEMM instead of a pass over the whole output (25 -> 6 us at 65 536)
            wg = plan.zeros(1, nwg)
            self.out.amax_src = wg
            return [(lib.mml_gather_fwd_wgmax, (tabs, vocab, col, F, E, self.X.data_ptr(), ops._ld(self.X),
                                                self.dense_col0, self.nd, plan.B, out.data_ptr(), ops._ld(out),
                                                wg.data_ptr(), nwg, plan.status.data_ptr()), meta)]
        return [(lib.mml_gather_fwd, (tabs, vocab, col, F, E, self.X.data_ptr(), ops._ld(self.X), self.dense_col0,
                                      self.nd, plan.B, out.data_ptr(), ops._ld(out), plan.status.data_ptr()), meta)]

    def bwd_calls(self, plan):
        if self.out.grad is None or not any(t.needs_grad for t in self.tables):
            return []
        lib = L.load()
        F = len(self.tables)
        E = self.tables[0].data.shape[1]
        gt = ops._ptr_array([t.grad for t in self.tables])
        vocab = (L.i64 * F)(*[t.data.shape[0] for t in self.tables])
        col = (L.i32 * F)(*self.cols)
        plan.keep += [gt, vocab, col]
        for t in self.tables:
            _claim(t)
        sr = self.sparse_rows
        if sr is not None:
            seen = ops._ptr_array(sr.seen)
            rb = (L.i64 * (F + 1))(*sr.rowbase)
            plan.keep += [seen, rb]
            extra = (seen, rb, sr.touched.data_ptr(), sr.count.data_ptr(), sr.touched.numel(), sr.marks.data_ptr())
        else:
            extra = (None, None, None, None, 0, L.ptr(self.grad_marks))
        meta = dict(kernel=scatter_symbol(E),
                    bytes=float(plan.B) * F * (4 + 12 * E), tail=True)  # idx + grad read + row RMW
        det = self.deterministic
        if det is not None:
            # deterministic scatter: 64-bit integer row totals, then fp32 (two launches + the magnitude of d(dnn_input));
            # the marks feed the marked dense update (left set) or the touched-row list (compaction, which clears them)
            acc = ops._ptr_array(det["acc64"])
            marks = self.grad_marks if self.grad_marks is not None else (sr.marks if sr is not None else det["marks"])
            keep_marks = self.grad_marks is not None or sr is not None
            slot = det["slot"]
            plan.keep += [acc]
            calls = [(lib.mml_scatter_bwd_det, (gt, vocab, col, F, E, self.X.data_ptr(), ops._ld(self.X), plan.B,
                                                self.out.grad.data_ptr(), ops._ld(self.out.grad), acc, slot.data_ptr(),
                                                marks.data_ptr(), 0 if keep_marks else 1, plan.status.data_ptr()),
                      dict(meta, kernel="scatter_fold_kernel<det>"))]
            if sr is not None:
                calls.append((lib.mml_rows_compact, (seen, vocab, rb, F, sr.touched.data_ptr(), sr.count.data_ptr(),
                                                     sr.touched.numel(), sr.marks.data_ptr()),
                              dict(kernel="rows_compact_kernel", bytes=float(sr.marks.numel()), tail=True)))
            return calls
        return [(lib.mml_scatter_bwd, (gt, vocab, col, F, E, self.X.data_ptr(), ops._ld(self.X), plan.B,
                                       self.out.grad.data_ptr(), ops._ld(self.out.grad)) + extra +
                 (plan.status.data_ptr(),), meta)]


def g16_layer_ok(B, K, N, training):
    """A Linear(K -> N) at batch B can run on the bf16-storage kernels (include/mmlrec.h K3': tile-aligned extents; the
    weight gradient's tiles are 128 x 128)."""
    ok = B % 128 == 0 and K % 64 == 0 and N % 64 == 0 and K > 0 and N > 0
    if training:
        ok = ok and K % 128 == 0 and N % 128 == 0
    return ok


def _fast_row_width_ok(n):
    return n % 4 == 0 and 0 < n <= 256  # (csrc/rows_fast.hip: gate_fast_config / head_fast_config)


def _padded_view(buf, kp):
    """[B, kp] view over a value / gradient allocated by Plan._rows (columns [n, kp) are zero)."""
    if buf.stride(0) < kp or buf.stride(1) != 1:
        raise L.MMLError("value does not own zero-padded rows")
    return buf.as_strided((buf.shape[0], kp), (buf.stride(0), 1), buf.storage_offset())


def _copy2d_batch_call(plan, pairs, amax=None):
    """One launch copying src[:, :cols] -> dst[:, :cols] for every (src, dst) pair (cols = the narrower of the two);
    amax: per pair a magnitude slot the copy raises with max |x| of what it stores, or None."""
    arr = (L.Copy2dDesc * len(pairs))()
    for k, (d, (src, dst)) in enumerate(zip(arr, pairs)):
        d.src, d.lds, d.dst, d.ldd = src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0)
        d.rows, d.cols, d.accumulate = src.shape[0], min(src.shape[1], dst.shape[1]), 0
        if amax is not None and amax[k] is not None:
            d.amax_out = amax[k].data_ptr()
            plan.keep.append(amax[k])
    plan.keep.append(arr)
    return (L.load().mml_copy2d_batch, (arr, len(pairs)),
            dict(kernel="copy2d_batch_kernel", bytes=8.0 * sum(min(a.numel(), b.numel()) for a, b in pairs)))


def _kpad_of(q):
    x = q["x"]
    if x.kpad and x.n % 16 and not q.get("w_kn", 0) and x.buf.stride(0) >= x.kpad:
        return x.kpad
    return 0


class LinearGroupOp(Op):
    """K3: a set of independent Linear(+activation) problems launched together.
    problems: dicts with x (Val), W (PVal), b (PVal or None), out (Val; out.act is the activation), w_kn.
    K7 (PepNet, reference model/pepnet.py:72-78, :139-140): a problem may also carry mul (Val) and prod (Val) -- the launch
    then stores prod = out * mul from the same epilogue (mml_gemm_fwd_desc.mul / prod), and the product's backward is
    folded into the input-gradient launch of the layer that READS prod (gate mode of mml_gemm_grouped_dgrad): neither
    direction makes a pass of its own over memory.  prod must feed exactly one LinearGroupOp."""

    def __init__(self, problems):
        self.p = problems
        self.use16 = False  # bf16-storage path (csrc/gemm16.hip): decided when the forward is recorded
        for q in self.p:
            # prod_bwd = "ext": only the FORWARD of the product rides in this launch's epilogue; its backward belongs to
            # another op (MulBatchOp(fwd_fused=True)) -- products no Linear layer reads (PepNet's gated input, the
            # products in front of the heads)
            if q.get("mul") is not None and q.get("prod_bwd") != "ext":
                q["prod"].gate = (q["mul"], q["out"])  # (h, g): factors of the product, for the consumer's dgrad

    def writes_grad16(self, plan, v):
        return self.use16 and v.is16

    # ---- bf16-storage path (include/mmlrec.h K3') --------------------------------------------------------------
    def _fwd16(self, plan):
        lib = L.load()
        probs = []
        for q in self.p:
            out = q["out"]
            if (not q["x"].is16 or q.get("w_kn") or q.get("mul") is not None or q["x"].kpad or
                    not g16_layer_ok(plan.B, q["x"].n, out.n, plan.training)):
                raise L.MMLError("bf16-storage layer group: every problem needs a bf16 input, a plain nn.Linear weight and "
                                 f"tile-aligned extents (got {q['x'].n} -> {out.n} at batch {plan.B})")
            if out.act not in (L.ACT_NONE, L.ACT_RELU):
                raise NotImplementedError("bf16-storage layer group: relu / linear activations")
            if plan.training and out.act == L.ACT_RELU and out.mask is None:
                out.mask = torch.zeros(plan.B, (out.n + 31) // 32, dtype=torch.int32, device=plan.device)
            out.producer16 = True
            probs.append(dict(srcs=[(q["x"].buf, plan.weight16(q["W"], False))], C=out.buf,
                              bias=q["b"].data if q.get("b") else None, act=out.act,
                              mask_out=out.mask if out.act == L.ACT_RELU else None))
        calls = []
        for i in range(0, len(probs), L.G16_MAX_GROUP):
            ch, qs = probs[i:i + L.G16_MAX_GROUP], self.p[i:i + L.G16_MAX_GROUP]
            descs = ops.make_g16_tn_descs(ch)
            plan.keep.append(descs)
            meta = dict(kernel="g16_tn_kernel(fwd %dx%d->%d)" % (len(qs), qs[0]["x"].n, max(q["out"].n for q in qs)),
                        flops=sum(2.0 * plan.B * q["out"].n * q["x"].n for q in qs),
                        hbm_bytes=_distinct_bytes([q["x"].buf for q in qs] + [q["out"].buf for q in qs] +
                                                  [q["out"].mask for q in qs]) +
                        2.0 * sum(q["W"].data.numel() for q in qs))
            calls.append((lib.mml_g16_tn, (descs, len(ch)), meta))
        return calls

    def _bwd16(self, plan):
        lib = L.load()
        calls = []
        live = [q for q in self.p if q["out"].grad is not None]
        # the incoming gradients as bf16 operands: written that way by their producer (Plan.grad_of), else cast here
        casts = []
        for q in live:
            g = q["out"].grad
            if g.dtype == torch.bfloat16:
                q["dC16"] = g
            else:
                if q["out"].act != L.ACT_NONE and not q["out"].deriv_applied:
                    raise L.MMLError("bf16-storage layer group: the activation derivative must be applied upstream")
                q["dC16"] = plan.empty(plan.B, q["out"].n, dtype=torch.bfloat16)
                casts.append((g, q["dC16"], False))
        if casts:
            arr = ops.make_cast16_descs(casts)
            plan.keep.append(arr)
            calls.append((lib.mml_cast16_batch, (arr, len(casts)),
                          dict(kernel="cast16_kernel", bytes=6.0 * sum(g.numel() for g, _, _ in casts))))
        # weight / bias gradients (side list: only the optimizer reads them)
        wg = []
        for q in live:
            W, b = q["W"], q.get("b")
            if not W.needs_grad:
                continue
            acc = _claim(W)
            if b is not None and b.needs_grad and _claim(b) != acc:
                raise L.MMLError("weight and bias of one layer must be written in the same order")
            wg.append(dict(dC=q["dC16"], A=q["x"].buf, dW=W.grad, dbias=b.grad if (b and b.needs_grad) else None,
                           accumulate=acc))
        for i in range(0, len(wg), L.G16_MAX_GROUP):
            ch = wg[i:i + L.G16_MAX_GROUP]
            descs = ops.make_g16_wgrad_descs(ch)
            nbytes = int(lib.mml_g16_wgrad_workspace_bytes(descs, len(ch)))
            if nbytes < 0:
                L.check(-1, "mml_g16_wgrad_workspace_bytes")
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=plan.device)
            plan.keep += [descs, ws]
            meta = dict(kernel="g16_nt_kernel(wgrad %dx%dx%d)" % (len(ch), ch[0]["dW"].shape[0], ch[0]["dW"].shape[1]),
                        flops=sum(2.0 * plan.B * q["dW"].numel() for q in ch), side=True,
                        rank=0, hbm_bytes=_distinct_bytes([q["dC"] for q in ch] + [q["A"] for q in ch] +
                                                          [q["dW"] for q in ch]))
            calls.append((lib.mml_g16_wgrad, (descs, len(ch), ws.data_ptr(), ws.numel(), 1), meta))
            calls.append((lib.mml_g16_wgrad, (descs, len(ch), ws.data_ptr(), ws.numel(), 2),
                          dict(kernel="g16_reduce_kernel", bytes=float(nbytes), side=True, rank=1)))
        # input gradients: one problem per distinct input value, its layers as the sources
        by_x = {}
        for q in live:
            if q["x"].needs_grad:
                by_x.setdefault(id(q["x"]), (q["x"], []))[1].append(q)
        dg = []
        for x, qs in by_x.values():
            if len(qs) > L.MAX_SRC:
                raise NotImplementedError("bf16-storage layer group: more than MAX_SRC layers on one input")
            plan.grad_of(x)
            fuse = len(x.consumers) == 1 and x.act == L.ACT_RELU and x.mask is not None
            if x.act != L.ACT_NONE and not fuse:
                raise L.MMLError(f"bf16 value {x.name!r}: its activation derivative must fold into its one consumer")
            acc = _claim(x)
            if acc and x.grad.dtype != torch.float32:
                raise L.MMLError("bf16 gradients cannot accumulate")
            dg.append((dict(srcs=[(q["dC16"], plan.weight16(q["W"], True)) for q in qs], C=x.grad, accumulate=acc,
                            mask_in=x.mask if fuse else None), x, qs))
            if fuse:
                x.deriv_applied = True
        for i in range(0, len(dg), L.G16_MAX_GROUP):
            ch = dg[i:i + L.G16_MAX_GROUP]
            descs = ops.make_g16_tn_descs([c[0] for c in ch])
            plan.keep.append(descs)
            meta = dict(kernel="g16_tn_kernel(dgrad %dx%d<-%d)" % (len(ch), ch[0][1].n, sum(q["out"].n for q in ch[0][2])),
                        flops=sum(2.0 * plan.B * x.n * sum(q["out"].n for q in qs) for _, x, qs in ch),
                        hbm_bytes=_distinct_bytes([x.grad for _, x, _ in ch] + [x.mask for _, x, _ in ch] +
                                                  [q["dC16"] for _, _, qs in ch for q in qs]) +
                        2.0 * sum(q["W"].data.numel() for _, _, qs in ch for q in qs))
            calls.append((lib.mml_g16_tn, (descs, len(ch)), meta))
        return calls

    def inputs(self):
        return [q["x"] for q in self.p] + [q["mul"] for q in self.p
                                           if q.get("mul") is not None and q.get("prod_bwd") != "ext"]

    def outputs(self):
        return [q["out"] for q in self.p] + [q["prod"] for q in self.p if q.get("mul") is not None]

    def fwd_calls(self, plan):
        if plan.bf16 and any(q["x"].is16 for q in self.p):
            self.use16 = True
            return self._fwd16(plan)
        # training plans: a ReLU output also leaves its sign bits (1 bit per element) for the dgrad that will apply
        # relu' to its gradient -- 32x less to re-read than the activations themselves
        for q in self.p:
            out = q["out"]
            if plan.training and out.act == L.ACT_RELU and out.mask is None:
                out.mask = torch.zeros(plan.B, (out.n + 31) // 32, dtype=torch.int32, device=plan.device)
        # reduction length not a multiple of 16: run the GEMM on the zero-padded operand pair (the value's own padded
        # rows, a padded copy of the weight refreshed every step) instead of dropping to the register-staged kernel
        pads = []
        for q in self.p:
            kp = _kpad_of(q)
            if kp:
                q["Ap"] = _padded_view(q["x"].buf, kp)
                q["Wp"] = plan.zeros(q["W"].data.shape[0], kp)
                pads.append((q["W"].data, q["Wp"]))
        pre = [_copy2d_batch_call(plan, pads)] if pads else []
        # operand magnitudes: the input's (its producer's, or measured here), the weight's (start of the step), and the
        # output's is produced by this launch for the GEMMs that read it
        need = []
        for q in self.p:
            q["amax_a"] = plan.value_amax(q["x"], q.get("Ap", q["x"].buf), need)
            q["amax_w"] = plan.weight_amax(q["W"], q.get("Wp", q["W"].data), need)
            if q["out"].amax is None:
                q["out"].amax = plan.new_amax()
            if q.get("mul") is not None and q["prod"].amax is None:
                q["prod"].amax = plan.new_amax()
        if need:
            pre.append(plan.amax_call(need))
        # pre-cut weights: all problems of the launch or none (the kernel takes the planes form per launch)
        wp = [plan.weight_planes(q, ops.PLANES_ROWS, padded="Wp" in q) for q in self.p]
        if any(pl is None for pl, _ in wp):
            wp = [(None, None)] * len(self.p)
        descs = ops.make_fwd_descs([dict(A=q.get("Ap", q["x"].buf), W=q.get("Wp", q["W"].data),
                                         bias=q["b"].data if q.get("b") else None,
                                         C=q["out"].buf, act=q["out"].act, w_kn=q.get("w_kn", 0),
                                         mask=q["out"].mask, amax_a=q["amax_a"], amax_w=q["amax_w"],
                                         amax_out=q["out"].amax, w_planes=pl, w_kexp=kx,
                                         mul=q["mul"].buf if q.get("mul") is not None else None,
                                         prod=q["prod"].buf if q.get("mul") is not None else None,
                                         amax_prod=q["prod"].amax if q.get("mul") is not None else None)
                                    for q, (pl, kx) in zip(self.p, wp)])
        plan.keep.append(descs)
        kn = self.p[0].get("w_kn", 0)
        meta = dict(kernel=_gemm_symbol(True, not kn, [q["out"].n for q in self.p], 0,
                                        kreds=[q.get("Ap", q["x"].buf).shape[1] for q in self.p],
                                        tensors=[q.get("Ap", q["x"].buf) for q in self.p] +
                                                [q.get("Wp", q["W"].data) for q in self.p],
                                        nrc_extents=[q["out"].n for q in self.p] if kn else []),
                    flops=sum(2.0 * plan.B * q["out"].n * q["x"].n for q in self.p),
                    hbm_bytes=_distinct_bytes([q["x"].buf for q in self.p] + [q["W"].data for q in self.p] +
                                              [q["out"].buf for q in self.p] + [q["out"].mask for q in self.p] +
                                              [q["prod"].buf for q in self.p if q.get("mul") is not None] +
                                              [q["mul"].buf for q in self.p if q.get("mul") is not None]))
        return pre + [(L.load().mml_gemm_grouped_fwd, (descs, len(self.p)), meta)]

    def bwd_calls(self, plan):
        if self.use16:
            return self._bwd16(plan)
        lib = L.load()
        calls = []
        live = [q for q in self.p if q["out"].grad is not None]
        # magnitudes of the incoming gradients (every writer of out.grad has been recorded by now): raised by the GEMM
        # that wrote them, else measured here, on the main chain, before the input-gradient and weight-gradient launches
        need = []
        for q in live:
            q["amax_dc"] = plan.grad_amax(q["out"], need)
            if q.get("amax_w") is None:
                q["amax_w"] = plan.weight_amax(q["W"], q.get("Wp", q["W"].data), need)
        if need:
            calls.append(plan.amax_call(need))
        # weight / bias gradients
        wg = []
        for q in live:
            W, b = q["W"], q.get("b")
            if not W.needs_grad:
                continue
            acc = _claim(W)
            if b is not None and b.needs_grad:
                if _claim(b) != acc:
                    raise L.MMLError("weight and bias of one layer must be written in the same order")
            if "Ap" in q and not acc:
                q["dWp"] = plan.empty(W.data.shape[0], q["Ap"].shape[1])
                wg.append(dict(dC=q["out"].grad, A=q["Ap"], dW=q["dWp"], dbias=b.grad if (b and b.needs_grad) else None,
                               accumulate=0, w_kn=0, unpad=(q["dWp"], W.grad), amax_dc=q["amax_dc"],
                               amax_a=q.get("amax_a")))
                continue
            wg.append(dict(dC=q["out"].grad, A=q["x"].buf, dW=W.grad, dbias=b.grad if (b and b.needs_grad) else None,
                           accumulate=acc, w_kn=q.get("w_kn", 0), amax_dc=q["amax_dc"], amax_a=q.get("amax_a")))
        if wg:
            descs = ops.make_wgrad_descs(wg)
            nbytes = lib.mml_gemm_grouped_wgrad_workspace_bytes(descs, len(wg))
            # own workspace: all partial-product launches of the step run before the first reduction (see Plan)
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=plan.device)
            plan.keep += [descs, ws]
            meta = dict(kernel=_gemm_symbol(False, False, [q["dW"].shape[1] for q in wg], 2, kreds=[plan.B],
                                            tensors=[q["dC"] for q in wg] + [q["A"] for q in wg],
                                            nrc_extents=[d for q in wg for d in q["dW"].shape]),
                        flops=sum(2.0 * plan.B * q["dW"].numel() for q in wg), side=True, rank=0,
                        hbm_bytes=_distinct_bytes([q["dC"] for q in wg] + [q["A"] for q in wg] + [q["dW"] for q in wg]))
            calls.append((lib.mml_gemm_grouped_wgrad_phase, (descs, len(wg), ws.data_ptr(), ws.numel(), 1), meta))
            calls.append((lib.mml_gemm_grouped_wgrad_phase, (descs, len(wg), ws.data_ptr(), ws.numel(), 2),
                          dict(kernel="slab_reduce", bytes=float(nbytes), side=True, rank=1)))
            unpad = [q["unpad"] for q in wg if "unpad" in q]  # padded weight gradients -> the parameters' [N, K]
            if unpad:
                c = _copy2d_batch_call(plan, unpad)
                c[2]["side"] = True
                calls.append(c)
        # input gradients: one dgrad problem per distinct input value
        by_x = {}
        for q in live:
            if q["x"].needs_grad:
                by_x.setdefault(id(q["x"]), (q["x"], []))[1].append(q)
        waves = []  # chunk k of every input goes into launch k: chunks of ONE input must not run concurrently
        post = []   # sums of split input gradients, after the launches
        gate_wave = {}  # K7: id(factor value) -> last launch that writes its gradient
        for x, qs in by_x.values():
            chunks = [qs[i:i + L.MAX_SRC] for i in range(0, len(qs), L.MAX_SRC)]
            gate = getattr(x, "gate", None)
            if gate is not None:
                # K7: x = h (.) g was written by the epilogue of the GEMM that produced g; its gradient is not stored,
                # this launch writes the gradients of the two factors (gate mode)
                if len(chunks) != 1 or len(x.consumers) != 1 or x.kpad:
                    raise L.MMLError("a fused gate product must feed exactly one Linear group (<= MAX_SRC layers)")
                h, g = gate
                gd = dict(h=h.buf, g=g.buf)
                # h shared by several products and able to take its gradient in parts (MulBatchOp(fwd_fused)): this
                # problem writes a buffer of its own -- no accumulation, no order between the sharing problems
                parts = getattr(h, "grad_parts", None)
                # (at most seven: the summing launch takes eight terms per target -- parts + the plain gradient buffer a
                # further reader would add to)
                use_part = (parts is not None and h.act == L.ACT_NONE and len(parts) < 7 and
                            h.buf.stride(0) == h.n and os.environ.get("MMLREC_GRAD_PARTS", "1") != "0")
                for key, v, nc in (("h", h, 1), ("g", g, 0)):
                    if key == "h" and use_part:
                        part = plan.empty(plan.B, h.n)
                        parts.append(part)
                        gd.update(dh=part, act_h=L.ACT_NONE, acc_h=0, amax_dh=None)
                        continue
                    gv = plan.grad_of(v)
                    acc = _claim(v)
                    fold = (not acc and v.act != L.ACT_NONE and not v.deriv_applied and len(v.consumers) == nc)
                    if fold:
                        v.deriv_applied = True
                    slot = None
                    if plan.amax_pool is not None:
                        if v.gamax is None:
                            v.gamax = plan.new_amax()
                        if v.gamax_writers == v.written - 1:
                            v.gamax_writers += 1
                            slot = v.gamax
                    gd.update({"d" + key: gv, "act_" + key: v.act if fold else L.ACT_NONE, "acc_" + key: acc,
                               "amax_d" + key: slot})
                ch = chunks[0]
                gp = [plan.weight_planes(q, ops.PLANES_COLS, ch) for q in ch]
                if any(pl is None for pl, _ in gp):
                    gp = [(None, None)] * len(ch)
                # products that share a factor (the gated input feeds every task's first product) write the SAME
                # gradient buffer, the first overwriting, the others adding: they must not run in one launch
                wi = max(-1 if use_part else gate_wave.get(id(h), -1), gate_wave.get(id(g), -1)) + 1
                gate_wave[id(g)] = wi
                if not use_part:
                    gate_wave[id(h)] = wi
                while len(waves) <= wi:
                    waves.append([])
                waves[wi].append(dict(dA=gd["dh"], gate=gd, Y=None, act=L.ACT_NONE, mask=None, accumulate=0,
                                     srcs=[(q["out"].grad, q["W"].data, q.get("w_kn", 0), q["amax_dc"], q["amax_w"]) +
                                           ((pl, kx) if pl is not None else ()) for q, (pl, kx) in zip(ch, gp)]))
                continue
            plan.grad_of(x)
            fuse = len(chunks) == 1 and len(x.consumers) == 1 and x.act != L.ACT_NONE
            # columns of the input gradient somebody reads (Val.grad_cols): a narrower launch over the same buffers
            gc = x.grad_cols if (0 < x.grad_cols < x.n and x.grad_cols % 16 == 0 and x.act == L.ACT_NONE and
                                 all(isinstance(c, LinearGroupOp) for c in x.consumers) and
                                 not any(q.get("w_kn", 0) for q in qs) and
                                 os.environ.get("MMLREC_GRAD_COLS", "1") != "0") else 0
            cut = (lambda t: t[:, :gc]) if gc else (lambda t: t)
            # Small batches: an input fed by many layers (dnn_input: every expert and gate) is ONE problem with a long
            # reduction -- 128 tiles of 72 k-steps at B = 4 096 on AE-30, 41 us on a chip with 256 CUs.  Its sources are
            # dealt to up to four problems of the same launch (partial sums into scratch, one add afterwards): 4x the
            # tiles, a quarter of the steps.  Only without an activation derivative in the epilogue (input layers).
            ktot = sum(q["out"].n for q in qs)
            if (plan.B <= 8192 and len(chunks) == 1 and len(qs) >= 2 and x.act == L.ACT_NONE and ktot >= 512 and
                    os.environ.get("MMLREC_SPLIT_DGRAD", "1") != "0"):
                nparts = min(4, len(qs))
                parts = [[] for _ in range(nparts)]
                for q in sorted(qs, key=lambda q_: -q_["out"].n):  # longest first into the lightest part
                    min(parts, key=lambda p_: sum(r["out"].n for r in p_)).append(q)
                acc = _claim(x)
                padded = x.kpad and all("Wp" in q for q in qs)
                g0 = cut(_padded_view(x.grad, x.kpad) if padded else x.grad)
                pitch = x.grad.stride(0)
                targets = [g0]
                for _ in parts[1:]:
                    t_ = plan.zeros(plan.B, pitch)
                    targets.append(t_.as_strided(g0.shape, (pitch, 1)))
                while len(waves) < 1:
                    waves.append([])
                for part, dst in zip(parts, targets):
                    waves[0].append(dict(dA=dst, Y=None, act=L.ACT_NONE, mask=None, accumulate=acc if dst is g0 else 0,
                                         srcs=[(q["out"].grad, cut(q["Wp"] if padded else q["W"].data), q.get("w_kn", 0),
                                                q["amax_dc"], q["amax_w"]) for q in part]))
                arr = ops._ptr_array([x.grad] + [t_ for t_ in targets[1:]])
                plan.keep.append(arr)
                post.append((lib.mml_ew_add_n, (arr, len(targets), x.grad.data_ptr(), plan.B * pitch),
                             dict(kernel="ew_add_n_kernel", bytes=4.0 * plan.B * pitch * (len(targets) + 1))))
                continue
            for ci, ch in enumerate(chunks):
                acc = _claim(x)
                while len(waves) <= ci:
                    waves.append([])
                padded = x.kpad and all("Wp" in q for q in ch)  # every source reads the zero-padded weight copy
                # this launch raises the magnitude slot of x.grad with what it stores (after derivative / accumulation)
                if x.gamax is None and plan.amax_pool is not None:
                    x.gamax = plan.new_amax()
                if x.gamax is not None and x.gamax_writers == x.written - 1:
                    x.gamax_writers += 1
                    out_slot = x.gamax
                else:
                    out_slot = None
                # the weights that feed this input gradient, cut as one group (common exponent)
                gp = [plan.weight_planes(q, ops.PLANES_COLS, ch, padded=bool(padded)) for q in ch]
                if any(pl is None for pl, _ in gp):
                    gp = [(None, None)] * len(ch)
                waves[ci].append(dict(dA=cut(_padded_view(x.grad, x.kpad) if padded else x.grad),
                                      Y=x.buf if fuse else None, act=x.act if fuse else L.ACT_NONE,
                                      mask=x.mask if (fuse and x.act == L.ACT_RELU) else None,
                                      accumulate=acc, amax_out=out_slot,
                                      srcs=[(q["out"].grad, cut(q["Wp"] if padded else q["W"].data), q.get("w_kn", 0),
                                             q["amax_dc"], q["amax_w"]) + ((pl, kx) if pl is not None else ())
                                            for q, (pl, kx) in zip(ch, gp)]))
            if fuse:
                x.deriv_applied = True
        for dg in waves:
            descs = ops.make_dgrad_descs(dg)
            plan.keep.append(descs)
            kn = dg[0]["srcs"][0][2]
            meta = dict(kernel=_gemm_symbol(True, bool(kn), [q["dA"].shape[1] for q in dg], 1,
                                            kreds=[sr[0].shape[1] for q in dg for sr in q["srcs"]],
                                            tensors=[t_ for q in dg for sr in q["srcs"] for t_ in sr[:2]],
                                            nrc_extents=[] if kn else [q["dA"].shape[1] for q in dg]),
                        flops=sum(2.0 * plan.B * q["dA"].shape[1] * sum(sr[0].shape[1] for sr in q["srcs"])
                                  for q in dg),
                        hbm_bytes=_distinct_bytes([q["dA"] for q in dg] + [q.get("mask") for q in dg] +
                                                  [t_ for q in dg for sr in q["srcs"] for t_ in sr[:2]]))
            calls.append((lib.mml_gemm_grouped_dgrad, (descs, len(dg)), meta))
        return calls + post


class GateGroupOp(Op):
    """K4: gates (skinny linear + softmax) mixing a shared list of expert outputs.
    gates: dicts with G (Val), Wg (PVal), mix (Val), expert (indices into `experts`)."""

    def __init__(self, experts, gates, H):
        self.experts, self.gates, self.H = experts, gates, H

    def writes_grad16(self, plan, v):
        # (the fast row kernels write dE / dG as bf16: mml_gate_group.out_bf16; their shape conditions)
        return (_fast_row_width_ok(self.H) and all(_fast_row_width_ok(g["G"].n) for g in self.gates) and
                len(self.experts) * max(len(self.gates), 2) <= 32)

    def inputs(self):
        return list(self.experts) + [g["G"] for g in self.gates]

    def outputs(self):
        return [g["mix"] for g in self.gates]

    def fwd_calls(self, plan):
        for g in self.gates:
            g["P"] = plan.empty(plan.B, len(g["expert"]))
        grp = ops.make_gate_group([e.buf for e in self.experts],
                                  [dict(G=g["G"].buf, Wg=g["Wg"].data, P=g["P"], mix=g["mix"].buf, expert=g["expert"])
                                   for g in self.gates], plan.B, self.H)
        slot = plan.new_amax()  # ONE magnitude slot for all mixtures (an upper bound for each of them)
        if slot is not None:
            grp.amax_mix = slot.data_ptr()
            for g in self.gates:
                g["mix"].amax = slot
        plan.keep.append(grp)
        byts = 4.0 * plan.B * (len(self.experts) * self.H + sum(g["G"].n + len(g["expert"]) + self.H for g in self.gates))
        return [(L.load().mml_gate_mix_fwd, (C.byref(grp),), dict(kernel="gate_fwd_kernel", bytes=byts))]

    def bwd_calls(self, plan):
        lib = L.load()
        if all(g["mix"].grad is None for g in self.gates):
            return []
        for e in self.experts:
            if e.needs_grad and (len(e.consumers) != 1 or e.written):
                raise NotImplementedError("an expert output feeding something besides one gate group")
        e_relu = all(e.act == L.ACT_RELU for e in self.experts)
        if not e_relu and any(e.act != L.ACT_NONE for e in self.experts):
            raise NotImplementedError("gate group over experts with mixed activations")
        gl = []
        for g in self.gates:
            G, active = g["G"], g["mix"].grad is not None
            q = dict(G=G.buf, Wg=g["Wg"].data, P=g["P"], expert=g["expert"], active=int(active))
            if active:
                if G.needs_grad and (len(G.consumers) != 1 or G.written):
                    raise NotImplementedError("a gate input feeding something besides its gate")
                if _claim(g["Wg"]):
                    raise NotImplementedError("a gate weight shared between gates")
                fuse = G.act == L.ACT_RELU
                if G.act not in (L.ACT_RELU, L.ACT_NONE):
                    raise NotImplementedError("gate input activation")
                q.update(dmix=g["mix"].grad, dG=plan.grad_of(G), dWg=g["Wg"].grad, g_relu=int(fuse))
                _claim(G)
                G.deriv_applied = True
            gl.append(q)
        dE = []
        for e in self.experts:
            dE.append(plan.grad_of(e))
            _claim(e)
            e.deriv_applied = True
        grp = ops.make_gate_group([e.buf for e in self.experts], gl, plan.B, self.H, d_experts=dE, e_relu=e_relu)
        # magnitudes of what the kernel stores: one slot for every expert gradient, one for every gate-input gradient
        # (this op is the only writer of each of them: checked above)
        s_de, s_dg = plan.new_amax(), plan.new_amax()
        if s_de is not None:
            grp.amax_dE, grp.amax_dG = s_de.data_ptr(), s_dg.data_ptr()
            for e in self.experts:
                e.gamax, e.gamax_writers = s_de, e.written
            for g in self.gates:
                if g["mix"].grad is not None:
                    g["G"].gamax, g["G"].gamax_writers = s_dg, g["G"].written
        nws = int(lib.mml_gate_mix_bwd_workspace_bytes(C.byref(grp)))
        # (deferred reduction: the partial sums must survive until the side list runs -- a buffer of this op's own, not the
        # shared scratch every other call overwrites)
        ws = (torch.empty(max(nws, 256), dtype=torch.uint8, device=plan.device) if _defer_reduce()
              else ops.workspace(nws, plan.device))
        plan.keep += [grp, ws]
        act = [g for g in self.gates if g["mix"].grad is not None]
        byts = 4.0 * plan.B * (2 * len(self.experts) * self.H + sum(2 * g["G"].n + len(g["expert"]) + self.H for g in act))
        if _defer_reduce():
            # only the optimizer reads dWg: the reduction of the per-workgroup partial sums leaves the backward chain and
            # runs beside the weight-gradient GEMMs (the workspace is this op's own)
            return [(lib.mml_gate_mix_bwd_phase, (C.byref(grp), ws.data_ptr(), ws.numel(), 1),
                     dict(kernel="gate_bwd_kernel", bytes=byts)),
                    (lib.mml_gate_mix_bwd_phase, (C.byref(grp), ws.data_ptr(), ws.numel(), 2),
                     dict(kernel="slab_reduce", bytes=float(ws.numel()), side=True, rank=1))]
        return [(lib.mml_gate_mix_bwd, (C.byref(grp), ws.data_ptr(), ws.numel()),
                 dict(kernel="gate_bwd_kernel", bytes=byts))]


_DEFER = None  # set by trainer.TrainStep around the recording of its plan (one stream: True)


def _defer_reduce():
    """MMLREC_DEFER_REDUCE=1: the reductions of the gate / head kernels' partial sums (only the optimizer reads their
    results) leave the backward chain and run beside the weight-gradient GEMMs (mml_gate_mix_bwd_phase,
    mml_head_bce_fwd_bwd_phase).  Off by default: two launches fewer on the chain (-18 us in a serial trace of the AE-30
    step), but the two-stream step measured 1.702 against 1.666 ms with it (three interleaved pairs, round 4) -- the step
    is bound by what its kernels take from HBM and the CUs together, not by the length of the chain."""
    e = os.environ.get("MMLREC_DEFER_REDUCE")
    if e is not None:
        return e == "1"
    # Round 5, ONE stream: deferred, and Plan.merge_row_reduces turns the deferred reductions of a step into ONE launch
    # (mml_rows_reduce_batch): MMoE one launch fewer per step, a PLE of two levels two.
    return bool(_DEFER)


class HeadOp(Op):
    """K5: prediction heads (+ summed BCE and its backward when training).
    heads: dicts with Hin (Val), w (PVal with H elements), bias (PVal [1]), bias2 (PVal [1] or None); a gated head (round 6,
    PepNet's last PPNet layer, reference model/pepnet.py:72-78) also carries gate (Val: the input of the head is
    Hin (.) gate, formed inside the kernel; this op must be the only consumer of both values)."""

    def __init__(self, heads, mask_cols=None):
        self.heads = heads
        self.mask_cols = mask_cols

    def writes_grad16(self, plan, v):
        # (the fast head kernel writes dH as bf16: mml_head_group.dh_bf16 -- all heads or none)
        return (type(self) is HeadOp and all(_fast_row_width_ok(h["Hin"].n) for h in self.heads) and
                all(h["Hin"].producer16 and h["Hin"].n % 8 == 0 and len(h["Hin"].consumers) == 1 for h in self.heads))

    def inputs(self):
        return [h["Hin"] for h in self.heads] + [h["gate"] for h in self.heads if h.get("gate") is not None]

    def _group(self, plan, train, use_dprob, claim):
        T = len(self.heads)
        if plan.prob is None:
            plan.prob = plan.empty(plan.B, T)
        prob_buf, dprob_buf = self._prob_buffers(plan, use_dprob)
        hl, post = [], []
        lib = L.load()
        for t, h in enumerate(self.heads):
            Hin = h["Hin"]
            q = dict(Hin=Hin.buf, w=h["w"].data, bias=h["bias"].data,
                     bias2=h["bias2"].data if h.get("bias2") is not None else None,
                     mask_col=(self.mask_cols[t] if (self.mask_cols and plan.mask is not None) else -1))
            G = h.get("gate")
            if G is not None:
                if G.n != Hin.n or not _fast_row_width_ok(Hin.n) or Hin.is16 or G.is16:
                    raise L.MMLError("gated head: gate and input of one width the fast row kernel serves, fp32")
                q.update(gate=G.buf, gate_act=G.act)
            if train:
                sole = len(Hin.consumers) == 1
                if G is not None:
                    if not sole or len(G.consumers) != 1 or G.act not in (L.ACT_NONE, L.ACT_SIGMOID, L.ACT_SIGMOID2):
                        raise L.MMLError("gated head: the head must be the only consumer of its input and of its gate")
                    q["dgate"] = plan.grad_of(G)
                    if claim:
                        if _claim(G):
                            raise L.MMLError("gated head: the gate's gradient has another writer")
                        G.deriv_applied = True   # (the kernel multiplies act'(gate) in)
                        if plan.amax_pool is not None:  # ONE slot for all gates' gradients
                            if getattr(self, "_amax_dG", None) is None:
                                self._amax_dG = plan.new_amax()
                            G.gamax, G.gamax_writers = self._amax_dG, 1
                if sole:
                    q["dH"] = plan.grad_of(Hin)
                    q["h_relu"] = int(Hin.act == L.ACT_RELU)
                    if Hin.act not in (L.ACT_RELU, L.ACT_NONE):
                        raise NotImplementedError("head input activation")
                    if claim:
                        _claim(Hin)
                        Hin.deriv_applied = True
                        if Hin.written == 1 and plan.amax_pool is not None:  # the kernel raises ONE slot for all dH
                            if getattr(self, "_amax_dH", None) is None:
                                self._amax_dH = plan.new_amax()
                            Hin.gamax, Hin.gamax_writers = self._amax_dH, 1
                else:
                    tmp = h.setdefault("_dH_tmp", plan.empty(plan.B, Hin.n))
                    q["dH"], q["h_relu"] = tmp, 0
                    plan.grad_of(Hin)
                    acc = _claim(Hin) if claim else h["_acc"]
                    h["_acc"] = acc
                    post.append((lib.mml_copy2d, (tmp.data_ptr(), ops._ld(tmp), Hin.grad.data_ptr(), ops._ld(Hin.grad),
                                                  plan.B, Hin.n, acc)))
                if claim:
                    h["_acc_w"] = _claim(h["w"])
                    h["_acc_b"] = _claim(h["bias"])
                if h["_acc_b"]:  # a bias shared by several heads (reference model/esmm.py:55-56: one PredictionLayer)
                    tmpb = h.setdefault("_db_tmp", plan.empty(1))
                    q["dbias"] = tmpb
                    post.append((lib.mml_copy2d, (tmpb.data_ptr(), 1, h["bias"].grad.data_ptr(), 1, 1, 1, 1)))
                else:
                    q["dbias"] = h["bias"].grad
                if h["_acc_w"]:
                    # a weight shared by several heads (reference model/mlp.py:28): this head's gradient goes to a
                    # scratch row that is added to the parameter's gradient after the launch
                    tmpw = h.setdefault("_dw_tmp", plan.empty(h["w"].data.numel()))
                    q["dw"] = tmpw
                    n = tmpw.numel()
                    post.append((lib.mml_copy2d, (tmpw.data_ptr(), n, h["w"].grad.data_ptr(), n, 1, n, 1)))
                else:
                    q["dw"] = h["w"].grad
                b2 = h.get("bias2")
                if b2 is not None and b2.needs_grad:
                    acc = _claim(b2) if claim else h["_acc_b2"]
                    h["_acc_b2"] = acc
                    post.append((lib.mml_copy2d, (h["bias"].grad.data_ptr(), 1, b2.grad.data_ptr(), 1, 1, 1, acc)))
            hl.append(q)
        grp = ops.make_head_group(hl, prob_buf, y=plan.y if (train and not use_dprob) else None, mask=plan.mask,
                                  loss=plan.loss if (train and not use_dprob) else None,
                                  dprob=dprob_buf if use_dprob else None)
        if train and getattr(self, "_amax_dH", None) is not None:
            grp.amax_dH = self._amax_dH.data_ptr()
        if train and getattr(self, "_amax_dG", None) is not None:
            grp.amax_dG = self._amax_dG.data_ptr()
        plan.keep.append(grp)
        return grp, post

    def _prob_buffers(self, plan, use_dprob):
        """(where the heads write their probabilities, where they read dL/dprob from)."""
        return plan.prob, plan.dprob

    def infer_calls(self, plan):
        grp, _ = self._group(plan, False, False, False)
        return [(L.load().mml_head_fwd, (C.byref(grp),),
                 dict(kernel="head_kernel", bytes=4.0 * plan.B * sum(h["Hin"].n + 1 for h in self.heads)))]

    def train_calls(self, plan, use_dprob, claim=True):
        lib = L.load()
        if use_dprob and plan.dprob is None:
            plan.dprob = plan.empty(plan.B, len(self.heads))
        grp, post = self._group(plan, True, use_dprob, claim)
        defer = not use_dprob and not post and _defer_reduce()
        nws = int(lib.mml_head_workspace_bytes(C.byref(grp)))
        ws = (torch.empty(max(nws, 256), dtype=torch.uint8, device=plan.device) if defer  # (its own: see GateGroupOp)
              else ops.workspace(nws, plan.device))
        plan.keep.append(ws)
        byts = 4.0 * plan.B * sum((4 if h.get("gate") is not None else 2) * h["Hin"].n + 2 for h in self.heads)
        if defer:
            # (dw / dbias / loss are read by the optimizer and the host only: their reduction runs beside the
            # weight-gradient GEMMs; Plan.finish moves the call tagged `side` out of the head's list)
            return [(lib.mml_head_bce_fwd_bwd_phase, (C.byref(grp), ws.data_ptr(), ws.numel(), 1),
                     dict(kernel="head_kernel", bytes=byts)),
                    (lib.mml_head_bce_fwd_bwd_phase, (C.byref(grp), ws.data_ptr(), ws.numel(), 2),
                     dict(kernel="slab_reduce", bytes=float(ws.numel()), side=True, rank=1, ready=0))]
        return [(lib.mml_head_bce_fwd_bwd, (C.byref(grp), ws.data_ptr(), ws.numel()),
                 dict(kernel="head_kernel", bytes=byts))] + post


class EsmmHeadOp(HeadOp):
    """ESMM (reference model/esmm.py:46-62): two sigmoid heads (ctr, cvr) whose product is the second output.
    The heads run on a raw probability buffer; mml_esmm_combine turns it into [ctr, ctr*cvr], evaluates the summed BCE
    and hands dLoss/d(ctr, cvr) back to the head kernel's dprob path."""
    N_OUT = 2
    KERNEL = "esmm_combine_kernel"

    def _group(self, plan, *a):
        if plan.prob is None:
            plan.prob = plan.empty(plan.B, self.N_OUT)
        return super()._group(plan, *a)

    def _prob_buffers(self, plan, use_dprob):
        if getattr(self, "_raw", None) is None or self._raw.shape[0] != plan.B:
            self._raw = plan.empty(plan.B, 2)
            self._draw = plan.empty(plan.B, 2)
        return self._raw, self._draw

    def _fn_extra(self):
        return L.load().mml_esmm_combine, ()

    def _combine(self, plan, y, dout, draw, loss):
        fn, extra = self._fn_extra()
        return (fn,
                (self._raw.data_ptr(), 2, L.ptr(y), 2 if y is not None else 0, L.ptr(dout),
                 ops._ld(dout) if dout is not None else 0, plan.prob.data_ptr(), ops._ld(plan.prob), L.ptr(draw),
                 2 if draw is not None else 0, L.ptr(loss), plan.B) + extra,
                dict(kernel=self.KERNEL, bytes=4.0 * plan.B * 8))

    def infer_calls(self, plan):
        calls = super().infer_calls(plan)
        return calls + [self._combine(plan, None, None, None, None)]

    def train_calls(self, plan, use_dprob, claim=True):
        lib = L.load()
        if use_dprob:  # autograd hands dL/d(outputs); the raw probabilities are those of the forward pass
            if plan.dprob is None:
                plan.dprob = plan.empty(plan.B, self.N_OUT)
            pre = [self._combine(plan, None, plan.dprob, self._prob_buffers(plan, True)[1], None)]
        else:
            fwd, _ = self._group(plan, False, False, False)
            if plan.y.stride(0) != 2:
                raise L.MMLError("ESMM / ESCM expect a contiguous [B, 2] label buffer")
            pre = [(lib.mml_head_fwd, (C.byref(fwd),), dict(kernel="head_kernel")),
                   self._combine(plan, plan.y, None, self._draw, plan.loss)]
        grp, post = self._group(plan, True, True, claim)
        ws = ops.workspace(lib.mml_head_workspace_bytes(C.byref(grp)), plan.device)
        plan.keep.append(ws)
        byts = 4.0 * plan.B * sum(2 * h["Hin"].n + 2 for h in self.heads)
        return pre + [(lib.mml_head_bce_fwd_bwd, (C.byref(grp), ws.data_ptr(), ws.numel()),
                       dict(kernel="head_kernel", bytes=byts))] + post


class EscmHeadOp(EsmmHeadOp):
    """ESCM (reference model/escm.py:74-112): outputs [ctr, cvr, ctr*cvr]; the training loss is the reference's
    special branch (model/basemodel.py:284-292: BCE(ctr) + 0.1 * counterfactual-IPW(cvr) + BCE(ctcvr)), evaluated with
    its gradient by mml_escm_combine."""
    N_OUT = 3
    KERNEL = "escm_combine_kernel"

    def __init__(self, heads, cf_w, global_w, mask_cols=None):
        super().__init__(heads, mask_cols)
        self.cf_w, self.global_w = float(cf_w), float(global_w)

    def _fn_extra(self):
        return L.load().mml_escm_combine, (self.cf_w, self.global_w)


class BNOp(Op):
    """BatchNorm1d between a layer's Linear and its activation (reference model/utils.py:153-157): z -> y = act(bn(z)).
    Batch statistics (and the in-place running-statistics update) when the module is in training mode, running
    statistics otherwise -- `plan.bn_training`, NOT plan.training: a no_grad forward of a model in train() mode still
    normalises with the batch and moves the running statistics, exactly like torch."""
    EPS, MOMENTUM = 1e-5, 0.1

    def __init__(self, z, y, gamma, beta, module):
        self.z, self.y, self.gamma, self.beta, self.m = z, y, gamma, beta, module

    def inputs(self):
        return [self.z]

    def outputs(self):
        return [self.y]

    def _ws(self, plan):
        nbytes = int(L.load().mml_bn_workspace_bytes(plan.B, self.z.n))
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=plan.device)
        plan.keep.append(ws)
        return ws

    def fwd_calls(self, plan):
        n = self.z.n
        self.mean, self.rstd = plan.empty(n), plan.empty(n)
        ws = self._ws(plan)
        m = self.m
        return [(L.load().mml_bn_fwd,
                 (self.z.buf.data_ptr(), ops._ld(self.z.buf), self.gamma.data.data_ptr(), self.beta.data.data_ptr(),
                  m.running_mean.data_ptr(), m.running_var.data_ptr(), m.num_batches_tracked.data_ptr(),
                  self.mean.data_ptr(), self.rstd.data_ptr(), self.y.buf.data_ptr(), ops._ld(self.y.buf), plan.B, n,
                  self.y.act, int(getattr(plan, "bn_training", plan.training)), self.EPS, self.MOMENTUM, ws.data_ptr(),
                  ws.numel()),
                 dict(kernel="bn_fwd", bytes=4.0 * plan.B * n * 3))]

    def bwd_calls(self, plan):
        if self.y.grad is None:
            return []
        if not getattr(plan, "bn_training", plan.training):
            raise L.MMLError("backward through BatchNorm in eval mode is not supported")
        plan.grad_of(self.z)
        _claim(self.z)
        acc = _claim(self.gamma)
        if _claim(self.beta) != acc:
            raise L.MMLError("BatchNorm weight and bias must be written in the same order")
        ws = self._ws(plan)
        n = self.z.n
        return [(L.load().mml_bn_bwd,
                 (self.y.grad.data_ptr(), ops._ld(self.y.grad), self.z.buf.data_ptr(), ops._ld(self.z.buf),
                  self.gamma.data.data_ptr(), self.mean.data_ptr(), self.rstd.data_ptr(), self.z.grad.data_ptr(),
                  ops._ld(self.z.grad), self.gamma.grad.data_ptr(), self.beta.grad.data_ptr(), acc, plan.B, n,
                  ws.data_ptr(), ws.numel()),
                 dict(kernel="bn_bwd", bytes=4.0 * plan.B * n * 5))]


class DropoutOp(Op):
    """nn.Dropout after a DNN layer's activation (reference model/utils.py:121, :159): x -> y = x * keep / (1 - p) while
    the module is in training mode (`plan.dropout_on`, recorded like BatchNorm's mode; in eval mode the model does not
    add the op).  The mask is regenerated, not stored (mml_dropout): the backward is the same call on dL/dy with the
    step counter the forward read -- the optimizer's device counter, bumped at the start of a fused step."""

    def __init__(self, x, y, p, seed, site):
        self.x, self.y, self.p, self.seed, self.site = x, y, float(p), int(seed) & ((1 << 64) - 1), int(site) & 0xffffffff

    def inputs(self):
        return [self.x]

    def outputs(self):
        return [self.y]

    def _call(self, plan, src, dst, acc):
        return (L.load().mml_dropout,
                (src.data_ptr(), ops._ld(src), dst.data_ptr(), ops._ld(dst), plan.B, self.x.n, plan.row0, self.p, self.seed,
                 self.site, plan.step_dev.data_ptr(), 0, int(acc)),
                dict(kernel="dropout_kernel", bytes=4.0 * plan.B * self.x.n * (3 if acc else 2)))

    def fwd_calls(self, plan):
        return [self._call(plan, self.x.buf, self.y.buf, 0)]

    def bwd_calls(self, plan):
        if self.y.grad is None or not self.x.needs_grad:
            return []
        g = plan.grad_of(self.x)
        return [self._call(plan, self.y.grad, g, _claim(self.x))]


def dropout_site(name):
    """The `site` word of a dropout layer's mask stream: CRC-32 of the layer's name (prefix.layer), so that the engine
    and the CPU restatement (oracle/mmlrec_oracle.py) agree without sharing a counter."""
    import zlib
    return zlib.crc32(name.encode()) & 0xffffffff


class DomainBNOp(Op):
    """DomainBatchNorm of STAR (reference model/utils.py:553-636, applied after the first star layer's activation when
    forward() is given a domain mask, model/star.py:50-51): x (a post-activation value) -> y.  gamma / beta are the
    reference's unregistered constants (1, 0).  Training mode = whole-batch statistics (mml_bn_fwd) plus the per-domain
    population-statistics update; eval mode = per-domain normalisation with the population statistics.

    Training mode assumes ONE-HOT mask rows (what get_mask, model/utils.py:639-645, produces): the reference's
    sum_d mask[b, d] * BN_batch(x[b]) then equals BN_batch(x[b]).  A row whose mask is all zero (or holds several
    ones) would need the factor sum_d mask[b, d]; such masks are outside the contract of this op."""
    EPS, DECAY = 1e-5, 0.99

    def __init__(self, x, y, module, mask):
        self.x, self.y, self.m, self.mask = x, y, module, mask

    def inputs(self):
        return [self.x]

    def outputs(self):
        return [self.y]

    def _ws(self, plan):
        nbytes = int(L.load().mml_bn_workspace_bytes(plan.B, self.x.n))
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=plan.device)
        plan.keep.append(ws)
        return ws

    def fwd_calls(self, plan):
        lib, n, D = L.load(), self.x.n, self.mask.shape[1]
        pm, pv = self.m.population(plan.device)
        if pm.shape != (D, n):
            raise L.MMLError(f"DomainBatchNorm holds {tuple(pm.shape)} statistics, the mask has {D} domains")
        x, y = self.x.buf, self.y.buf
        if not getattr(plan, "bn_training", plan.training):
            return [(lib.mml_domain_bn_eval, (x.data_ptr(), ops._ld(x), self.mask.data_ptr(), ops._ld(self.mask),
                                              pm.data_ptr(), pv.data_ptr(), y.data_ptr(), ops._ld(y), plan.B, n, D,
                                              self.EPS), dict(kernel="domain_bn_eval_kernel"))]
        self.ones, self.zeros = torch.ones(n, device=plan.device), torch.zeros(n, device=plan.device)
        self.mean, self.rstd = plan.empty(n), plan.empty(n)
        self.scratch = [plan.zeros(n), plan.zeros(n), plan.zeros(1, dtype=torch.int64)]  # F.batch_norm's discarded running stats
        ws = self._ws(plan)
        plan.keep += [self.ones, self.zeros]
        return [(lib.mml_domain_bn_update, (x.data_ptr(), ops._ld(x), self.mask.data_ptr(), ops._ld(self.mask), plan.B, n,
                                            D, pm.data_ptr(), pv.data_ptr(), self.DECAY),
                 dict(kernel="domain_bn_update_kernel")),
                (lib.mml_bn_fwd, (x.data_ptr(), ops._ld(x), self.ones.data_ptr(), self.zeros.data_ptr(),
                                  self.scratch[0].data_ptr(), self.scratch[1].data_ptr(), self.scratch[2].data_ptr(),
                                  self.mean.data_ptr(), self.rstd.data_ptr(), y.data_ptr(), ops._ld(y), plan.B, n,
                                  L.ACT_NONE, 1, self.EPS, 0.1, ws.data_ptr(), ws.numel()), dict(kernel="bn_fwd"))]

    def bwd_calls(self, plan):
        if self.y.grad is None:
            return []
        if not getattr(plan, "bn_training", plan.training):
            raise L.MMLError("backward through DomainBatchNorm in eval mode is not supported")
        plan.grad_of(self.x)
        if _claim(self.x):
            raise NotImplementedError("DomainBatchNorm input with another consumer")
        n = self.x.n
        dscr = plan.empty(2 * n)
        ws = self._ws(plan)
        return [(L.load().mml_bn_bwd,
                 (self.y.grad.data_ptr(), ops._ld(self.y.grad), self.x.buf.data_ptr(), ops._ld(self.x.buf),
                  self.ones.data_ptr(), self.mean.data_ptr(), self.rstd.data_ptr(), self.x.grad.data_ptr(),
                  ops._ld(self.x.grad), dscr.data_ptr(), dscr[n:].data_ptr(), 0, plan.B, n, ws.data_ptr(), ws.numel()),
                 dict(kernel="bn_bwd"))]


class ApgFeatOp(Op):
    """z = [o1 (x) s | o1 | s | 0] (csrc/apg.hip): the feature row that turns APG's per-sample generated [k,k] weight
    into one GEMM (reference model/apg.py:77-80, :100-104).  s is the detached scene embedding (a column slice of
    dnn_input): no gradient flows into it."""

    def __init__(self, o1, s, z, k, E):
        self.o1, self.s, self.z, self.k, self.E = o1, s, z, k, E

    def inputs(self):
        return [self.o1]

    def outputs(self):
        return [self.z]

    def fwd_calls(self, plan):
        o1, z = self.o1.buf, self.z.buf
        return [(L.load().mml_apg_features_fwd, (o1.data_ptr(), ops._ld(o1), self.s.data_ptr(), ops._ld(self.s),
                                                 z.data_ptr(), ops._ld(z), plan.B, self.k, self.E, self.z.n),
                 dict(kernel="apg_features_fwd_kernel", bytes=4.0 * plan.B * (self.z.n + self.k + self.E)))]

    def bwd_calls(self, plan):
        if self.z.grad is None or not self.o1.needs_grad:
            return []
        if self.o1.act != L.ACT_NONE:
            raise NotImplementedError("ApgFeatOp input must be a plain linear output")
        do1 = plan.grad_of(self.o1)
        acc = _claim(self.o1)
        dz = self.z.grad
        return [(L.load().mml_apg_features_bwd, (dz.data_ptr(), ops._ld(dz), self.s.data_ptr(), ops._ld(self.s),
                                                 do1.data_ptr(), ops._ld(do1), plan.B, self.k, self.E, acc),
                 dict(kernel="apg_features_bwd_kernel", bytes=4.0 * plan.B * (self.z.n + self.k + self.E)))]


class ApgWeightsOp(Op):
    """Derived [Kf, k] weight of an APG layer's middle GEMM, re-laid-out from the generator Linear(E -> k*k) (weight
    Wkk, bias bkk) and the weight of the generator Linear(E -> k) of the per-sample bias (csrc/apg.hip); backward
    unpacks its gradient into theirs."""

    def __init__(self, Wkk, bkk, Wb, Wcat, k, E):
        self.Wkk, self.bkk, self.Wb, self.Wcat, self.k, self.E = Wkk, bkk, Wb, Wcat, k, E

    def fwd_calls(self, plan):
        return [(L.load().mml_apg_weights, (self.Wkk.data.data_ptr(), self.bkk.data.data_ptr(), self.Wb.data.data_ptr(),
                                            self.Wcat.data.data_ptr(), self.k, self.k, self.E, 0, 0, 0, 0),
                 dict(kernel="apg_weights_kernel"))]

    def bwd_calls(self, plan):
        if not self.Wcat.written:
            return []
        a, b, c = _claim(self.Wkk), _claim(self.bkk), _claim(self.Wb)
        return [(L.load().mml_apg_weights, (self.Wkk.grad.data_ptr(), self.bkk.grad.data_ptr(), self.Wb.grad.data_ptr(),
                                            self.Wcat.grad.data_ptr(), self.k, self.k, self.E, 1, a, b, c),
                 dict(kernel="apg_weights_kernel", side=True))]


class Attn2Op(Op):
    """Two-token attention of AITM (model/aitm.py:84-93): tokens = [(V0, K0, Q0), (V1, K1, Q1)] of [B, H] values -> out."""

    def __init__(self, tokens, out, sqrt_h):
        self.tokens, self.out, self.sqrt_h = tokens, out, float(sqrt_h)

    def inputs(self):
        return [v for tok in self.tokens for v in tok]

    def outputs(self):
        return [self.out]

    def _desc(self, plan):
        d = L.Attn2Desc()
        for t, (V, K, Q) in enumerate(self.tokens):
            d.V[t], d.K[t], d.Q[t] = V.buf.data_ptr(), K.buf.data_ptr(), Q.buf.data_ptr()
            d.ldv[t], d.ldk[t], d.ldq[t] = ops._ld(V.buf), ops._ld(K.buf), ops._ld(Q.buf)
        d.out, d.ldo = self.out.buf.data_ptr(), ops._ld(self.out.buf)
        d.A = self.A.data_ptr()
        d.B, d.H, d.sqrt_h = plan.B, self.out.n, self.sqrt_h
        return d

    def fwd_calls(self, plan):
        self.A = plan.empty(plan.B, 2)
        d = self._desc(plan)
        plan.keep.append(d)
        return [(L.load().mml_attn2_fwd, (C.byref(d),), dict(kernel="attn2_fwd_kernel", bytes=4.0 * plan.B * 7 * self.out.n))]

    def bwd_calls(self, plan):
        if self.out.grad is None:
            return []
        d = self._desc(plan)
        d.dout, d.lddo = self.out.grad.data_ptr(), ops._ld(self.out.grad)
        for t, (V, K, Q) in enumerate(self.tokens):
            for v in (V, K, Q):
                if v.act != L.ACT_NONE or len(v.consumers) != 1:
                    raise NotImplementedError("Attn2Op inputs must be plain linear outputs with this op as sole consumer")
                plan.grad_of(v)
                _claim(v)
            d.dV[t], d.dK[t], d.dQ[t] = V.grad.data_ptr(), K.grad.data_ptr(), Q.grad.data_ptr()
            d.lddv[t], d.lddk[t], d.lddq[t] = ops._ld(V.grad), ops._ld(K.grad), ops._ld(Q.grad)
        plan.keep.append(d)
        return [(L.load().mml_attn2_bwd, (C.byref(d),), dict(kernel="attn2_bwd_kernel", bytes=4.0 * plan.B * 14 * self.out.n))]


class JoinOp(Op):
    """parts[j] are column slices of whole.buf (torch.cat of the reference done by writing in place, e.g.
    model/cross_stitch.py:18): no launch; on the way back the parts' gradients ARE the column slices of whole.grad."""

    def __init__(self, parts, whole):
        self.parts, self.whole = parts, whole
        off = 0
        for v in parts:
            if v.buf.data_ptr() != whole.buf[:, off:off + v.n].data_ptr() or v.buf.stride(0) != whole.buf.stride(0):
                raise L.MMLError("JoinOp: part is not the expected column slice of the whole")
            if v.act != whole.act:
                raise L.MMLError("JoinOp: parts and whole must carry the same activation")
            off += v.n

    def inputs(self):
        return list(self.parts)

    def outputs(self):
        return [self.whole]

    def fwd_calls(self, plan):
        return []

    def bwd_calls(self, plan):
        if self.whole.grad is None:
            return []
        off = 0
        for v in self.parts:
            v.grad = self.whole.grad[:, off:off + v.n]
            v.deriv_applied = True  # Plan.finish / the consumer's dgrad has applied act' on the whole already
            v.written = 1
            off += v.n
        return []


class SplitOp(Op):
    """The inverse: parts[j] are column slices of whole (torch slicing, model/cross_stitch.py:22-27); their gradient
    buffers are slices of whole.grad from the start, so consumers write straight into it."""

    def __init__(self, plan, whole, widths):
        self.whole, self.parts = whole, []
        if plan.training and whole.needs_grad:
            plan.grad_of(whole)
        off = 0
        for j, w in enumerate(widths):
            v = Val(whole.buf[:, off:off + w], L.ACT_NONE, whole.needs_grad, f"{whole.name}.{j}")
            if whole.grad is not None:
                v.grad = whole.grad[:, off:off + w]
            self.parts.append(v)
            off += w

    def inputs(self):
        return [self.whole]

    def outputs(self):
        return list(self.parts)

    def fwd_calls(self, plan):
        return []

    def bwd_calls(self, plan):
        if any(v.written for v in self.parts):
            if not all(v.written for v in self.parts):
                raise L.MMLError("SplitOp: every slice needs a consumer that writes its gradient")
            self.whole.written = 1
        else:
            self.whole.grad = None
        return []


class MulOp(Op):
    """out = a * b on whole contiguous [B,n] buffers (PepNet gating, model/pepnet.py:77, :140)."""

    def __init__(self, a, b, out):
        self.a, self.b, self.out = a, b, out
        self.flat = self._flat_numel((a, b, out))

    @staticmethod
    def _flat_numel(vals):
        """The flat kernels run over whole buffers: contiguous [B, n], or rows of one common zero-padded pitch."""
        if all(v.buf.is_contiguous() for v in vals):
            return vals[0].buf.numel()
        st = vals[0].buf.stride(0)
        if all(v.kpad == st and v.buf.stride(0) == st and v.n == vals[0].n for v in vals):
            return vals[0].buf.shape[0] * st
        raise L.MMLError("MulOp needs contiguous buffers (or one common zero-padded pitch)")

    def inputs(self):
        return [self.a, self.b]

    def outputs(self):
        return [self.out]

    def fwd_calls(self, plan):
        return [(L.load().mml_ew_mul, (self.a.buf.data_ptr(), self.b.buf.data_ptr(), self.out.buf.data_ptr(),
                                       self.flat), dict(kernel="ew_mul_kernel", bytes=12.0 * self.flat))]

    def bwd_calls(self, plan):
        if self.out.grad is None:
            return []
        if self.out.grad.stride(0) != self.out.buf.stride(0):
            raise L.MMLError("MulOp: value / gradient pitch mismatch")
        da = db = None
        acc_a = acc_b = 0
        if self.a.needs_grad:
            da = plan.grad_of(self.a)
            acc_a = _claim(self.a)
        if self.b.needs_grad:
            db = plan.grad_of(self.b)
            acc_b = _claim(self.b)
        if da is None and db is None:
            return []
        # an operand that is an activation's output and feeds nothing else: fold act' into the gradient written here
        # (Plan.finish would otherwise add a separate read-modify-write pass over it)
        acts = []
        for v, g, acc in ((self.a, da, acc_a), (self.b, db, acc_b)):
            fold = (g is not None and not acc and v.act != L.ACT_NONE and not v.deriv_applied and
                    len(v.consumers) == 1)
            acts.append(v.act if fold else L.ACT_NONE)
            if fold:
                v.deriv_applied = True
        return [(L.load().mml_ew_mul_bwd_act, (self.out.grad.data_ptr(), self.a.buf.data_ptr(), self.b.buf.data_ptr(),
                                               L.ptr(da), L.ptr(db), acc_a, acc_b, self.flat, acts[0], acts[1]),
                 dict(kernel="ew_mul_bwd_kernel", bytes=4.0 * self.flat * (3 + (da is not None) + (db is not None))))]


class MulBatchOp(Op):
    """out_j = a_j * b_j for several independent [B, n] products in ONE launch each way (PepNet: the gate products of
    every task of a layer, model/pepnet.py:72-78, :139-140; one launch per product before).  The backward writes each
    operand's gradient once, summed over the products it appears in (the gated input feeds every task's first layer),
    with the derivative of the activation that produced the operand folded in when this op is its only consumer.
    items: (a Val, b Val, out Val)."""

    def __init__(self, items, fwd_fused=False):
        self.items = items
        self.flat = [MulOp._flat_numel(it) for it in items]
        # True: the products themselves leave the epilogue of the GEMM that produces b (LinearGroupOp problems with mul /
        # prod and prod_bwd = "ext": K7 forward); this op only contributes the backward
        self.fwd_fused = bool(fwd_fused)
        if self.fwd_fused and len(items) <= 8:
            # A product whose forward left a GEMM's epilogue and that several gate-mode input-gradient problems read as
            # their factor h (PepNet's gated input: every task's first product) may receive its gradient as PARTS, one
            # buffer per reader, summed by this op's backward launch inside its sums of products -- instead of one buffer
            # the readers add to one after the other (four launches in a row for Amazon-8's four tasks)
            for _, _, o in items:
                o.grad_parts = []

    def inputs(self):
        return [v for a, b, _ in self.items for v in (a, b)]

    def outputs(self):
        return [o for _, _, o in self.items]

    @staticmethod
    def _descs(plan, rows):
        arr = (L.SumProdDesc * len(rows))()
        for d, row in zip(arr, rows):
            out, n, terms, acc, act, deriv_of = row[:6]
            if len(terms) > 8:
                raise NotImplementedError("more than 8 products share one operand")
            d.out, d.n, d.n_terms, d.accumulate = out.data_ptr(), n, len(terms), int(acc)
            d.amax_out = L.ptr(row[6]) if len(row) > 6 else None
            d.act, d.deriv_of = int(act), (deriv_of.data_ptr() if act != L.ACT_NONE else None)
            for k, (x, y) in enumerate(terms):
                d.x[k], d.y[k] = x.data_ptr(), y.data_ptr()
        plan.keep.append(arr)
        return arr

    def fwd_calls(self, plan):
        if self.fwd_fused:
            return []
        rows = []
        for (a, b, o), n in zip(self.items, self.flat):
            # the product's magnitude for the GEMMs that read it (only when the flat launch covers no padding columns)
            slot = plan.new_amax() if (o.amax is None and o.buf.stride(0) == o.n) else None
            if slot is not None:
                o.amax = slot
            rows.append((o.buf, n, [(a.buf, b.buf)], 0, L.ACT_NONE, None, slot))
        return [(L.load().mml_sumprod_batch, (self._descs(plan, rows), len(rows)),
                 dict(kernel="sumprod_batch_kernel", bytes=12.0 * sum(self.flat)))]

    def bwd_calls(self, plan):
        targets = {}  # id(operand) -> [operand, flat numel, [(dout, other factor)]]
        for (a, b, o), n in zip(self.items, self.flat):
            douts = list(getattr(o, "grad_parts", None) or [])
            if o.grad is not None:
                douts.append(o.grad)
            if not douts:
                continue
            if any(d_.stride(0) != o.buf.stride(0) for d_ in douts):
                raise L.MMLError("MulBatchOp: value / gradient pitch mismatch")
            for v, other in ((a, b), (b, a)):
                if v.needs_grad:
                    targets.setdefault(id(v), [v, n, []])[2].extend((d_, other.buf) for d_ in douts)
        rows, nbytes = [], 0.0
        for v, n, terms in targets.values():
            g = plan.grad_of(v)
            acc = _claim(v)
            fold = (not acc and v.act != L.ACT_NONE and not v.deriv_applied and len(v.consumers) == 1)
            if fold:
                v.deriv_applied = True
            slot = None  # magnitude of the gradient as stored (tracked like a dgrad GEMM's amax_out)
            if plan.amax_pool is not None and g.stride(0) == v.n:
                if v.gamax is None:
                    v.gamax = plan.new_amax()
                if v.gamax_writers == v.written - 1:
                    v.gamax_writers += 1
                    slot = v.gamax
            rows.append((g, n, terms, acc, v.act if fold else L.ACT_NONE, v.buf, slot))
            nbytes += 4.0 * n * (2 * len(terms) + 1 + (1 if (acc or fold) else 0))
        if not rows:
            return []
        return [(L.load().mml_sumprod_batch, (self._descs(plan, rows), len(rows)),
                 dict(kernel="sumprod_batch_kernel", bytes=nbytes))]


class CopyColsOp(Op):
    """dst[:, :] = src[:, :] for column-sliced views; no gradient flows (used for detached concatenations,
    model/pepnet.py:72, :139)."""

    def __init__(self, src, dst, amax_out=None):
        self.src, self.dst, self.amax_out = src, dst, amax_out

    def fwd_calls(self, plan):
        if self.amax_out is not None:
            # round 6: the copy raises the magnitude slot of the operand it assembles (mml_copy2d_desc.amax_out) -- PepNet's
            # two gate inputs cost a pass of mml_amax_batch each (17-20 us of a 1.67 ms step) right behind their copies
            return [_copy2d_batch_call(plan, [(self.src, self.dst)], amax=[self.amax_out])]
        return [(L.load().mml_copy2d, (self.src.data_ptr(), ops._ld(self.src), self.dst.data_ptr(), ops._ld(self.dst),
                                       plan.B, self.src.shape[1], 0))]


class PMulOp(Op):
    """Derived parameter out = a * b (STAR: specific * shared weight, model/utils.py:215)."""

    def __init__(self, a, b, out):
        self.a, self.b, self.out = a, b, out

    def fwd_calls(self, plan):
        return [(L.load().mml_ew_mul, (self.a.data.data_ptr(), self.b.data.data_ptr(), self.out.data.data_ptr(),
                                       self.out.data.numel()))]

    def bwd_calls(self, plan):
        if not self.out.written:
            return []
        da = self.a.grad if self.a.needs_grad else None
        db = self.b.grad if self.b.needs_grad else None
        if da is None and db is None:
            return []
        acc_a = _claim(self.a) if da is not None else 0
        acc_b = _claim(self.b) if db is not None else 0
        # consumes a weight gradient -> belongs behind the wgrad GEMMs on the side list
        return [(L.load().mml_ew_mul_bwd, (self.out.grad.data_ptr(), self.a.data.data_ptr(), self.b.data.data_ptr(),
                                           L.ptr(da), L.ptr(db), acc_a, acc_b, self.out.data.numel()),
                 dict(kernel="ew_mul_bwd_kernel", side=True))]


class SumProdBatchOp(Op):
    """Derived parameters as ONE launch each way: out_j = sum_k x_jk * y_jk (y optional).  STAR: W_eff = W_spec * W_shared
    and b_eff = b_spec + b_shared for every head and layer (model/utils.py:214-216); the backward sums every factor's
    gradient over the items it appears in (d W_shared = sum_heads dW_eff * W_spec) -- fixed order, no atomics.
    items: (out PVal, [(x PVal, y PVal or None), ...])."""

    def __init__(self, items):
        self.items = items

    @staticmethod
    def _descs(plan, rows):
        arr = (L.SumProdDesc * len(rows))()
        for d, (out, terms, acc) in zip(arr, rows):
            if len(terms) > 8:
                raise NotImplementedError("more than 8 terms in one derived-parameter sum")
            d.out, d.n, d.n_terms, d.accumulate = out.data_ptr(), out.numel(), len(terms), int(acc)
            for k, (x, y) in enumerate(terms):
                if x.numel() != out.numel() or (y is not None and y.numel() != out.numel()):
                    raise L.MMLError("SumProdBatchOp: operand size mismatch")
                d.x[k], d.y[k] = x.data_ptr(), (y.data_ptr() if y is not None else None)
        plan.keep.append(arr)
        return arr

    def fwd_calls(self, plan):
        rows = [(out.data, [(x.data, y.data if y is not None else None) for x, y in terms], 0)
                for out, terms in self.items]
        return [(L.load().mml_sumprod_batch, (self._descs(plan, rows), len(rows)),
                 dict(kernel="sumprod_batch_kernel", bytes=12.0 * sum(o.data.numel() for o, _ in self.items)))]

    def bwd_calls(self, plan):
        grads = {}  # id(param) -> (param, [(dout, other factor or None)])
        for out, terms in self.items:
            if not out.written:
                continue
            for x, y in terms:
                for p, other in ((x, y), (y, x)):
                    if p is not None and p.needs_grad:
                        grads.setdefault(id(p), (p, []))[1].append((out.grad, other.data if other is not None else None))
        rows = [(p.grad, terms, _claim(p)) for p, terms in grads.values()]
        if not rows:
            return []
        return [(L.load().mml_sumprod_batch, (self._descs(plan, rows), len(rows)),
                 dict(kernel="sumprod_batch_kernel", side=True))]


class SnrWeightsOp(Op):
    """Routing weights of an SNR-trans / MSSM gate (model/snr_trans.py:38-50, model/mssm.py:40-58): W[o][j] =
    M[o][j] scaled by the hard-concrete z(u, alpha) -- one coefficient per block (u: PVal [No, Ne], learned) or one per
    output column (u: frozen tensor [No, Ne, units], MSSM).  views[o] are PVals over W[o] seen as the [n_in*units,
    units] ([K,N]) weight of output o's routing GEMM.  Backward turns their weight gradients into du and dalpha."""
    BETA, GAMMA, EPS = 0.9, -0.1, 1.1

    def __init__(self, u, alpha, M, W, dW, views):
        self.u, self.alpha, self.M, self.W, self.dW, self.views = u, alpha, M, W, dW, views
        self.n_blocks = M.shape[0] * M.shape[1]
        self.block = M.shape[2] * M.shape[3]
        self.u_learned = isinstance(u, PVal)
        self.u_data = u.data if self.u_learned else u
        self.zw = 1 if self.u_data.numel() == self.n_blocks else M.shape[3]
        if self.u_data.numel() != self.n_blocks * self.zw:
            raise L.MMLError("SnrWeightsOp: u must hold one coefficient per block or per block column")

    def fwd_calls(self, plan):
        return [(L.load().mml_snr_gate_weights_fwd,
                 (self.u_data.data_ptr(), self.alpha.data.data_ptr(), self.M.data_ptr(), self.W.data_ptr(),
                  self.n_blocks, self.block, self.zw, self.BETA, self.GAMMA, self.EPS),
                 dict(kernel="snr_weights_fwd_kernel", bytes=8.0 * self.W.numel()))]

    def bwd_calls(self, plan):
        if not any(v.written for v in self.views):
            return []
        if not all(v.written for v in self.views):
            raise L.MMLError("SnrWeightsOp: every routing weight needs its gradient")
        du, acc_u = (self.u.grad.data_ptr(), _claim(self.u)) if self.u_learned else (None, 0)
        return [(L.load().mml_snr_gate_weights_bwd,
                 (self.dW.data_ptr(), self.M.data_ptr(), self.u_data.data_ptr(), self.alpha.data.data_ptr(), du,
                  self.alpha.grad.data_ptr(), acc_u, _claim(self.alpha), self.n_blocks, self.block, self.zw, self.BETA,
                  self.GAMMA, self.EPS, plan.empty(self.n_blocks).data_ptr()),
                 dict(kernel="snr_weights_bwd_kernel", bytes=8.0 * self.W.numel(), side=True))]


class PAddOp(Op):
    """Derived parameter out = sum(inputs) (STAR bias sums, model/utils.py:216)."""

    def __init__(self, ins, out):
        self.ins, self.out = ins, out

    def fwd_calls(self, plan):
        arr = ops._ptr_array([p.data for p in self.ins])
        plan.keep.append(arr)
        return [(L.load().mml_ew_add_n, (arr, len(self.ins), self.out.data.data_ptr(), self.out.data.numel()))]

    def bwd_calls(self, plan):
        if not self.out.written:
            return []
        calls = []
        n = self.out.data.numel()
        for p in self.ins:
            if p.needs_grad:
                calls.append((L.load().mml_copy2d, (self.out.grad.data_ptr(), n, p.grad.data_ptr(), n, 1, n, _claim(p)),
                              dict(kernel="copy2d_kernel", side=True)))
        return calls


# ==================================================================================================
# parameter store + optimizer state shared by all plans of one model
# ==================================================================================================
class TableRows:
    """Bookkeeping for the sparse-row table update: per-table `seen` bitmaps + the touched-row list."""

    def __init__(self, vocab, device, cap):
        self.rowbase = [0]
        for v in vocab:
            self.rowbase.append(self.rowbase[-1] + int(v))
        self.seen = [torch.zeros((int(v) + 31) // 32, dtype=torch.int32, device=device) for v in vocab]
        # one byte per row, all-zero between launches: rows are marked with plain stores, a compaction pass turns the
        # marks into the bitmaps + the list (include/mmlrec.h: row_marks)
        self.marks = torch.zeros(ops.marks_bytes(vocab), dtype=torch.uint8, device=device)
        self.touched = torch.zeros(max(int(cap), 1), dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)


class ParamStore:
    """Gradient buffers and optimizer state for one model on one device.

    Dense (MLP) parameter gradients live in ONE flat arena (a single buffer to all-reduce under data parallelism);
    every table gets a dense [V,E] accumulator that is kept all-zero between steps (the optimizer kernels re-zero
    what they consume), so the scatter can add into it without a per-step 400 MB memset."""

    def __init__(self, model, device):
        self.device = device
        self.model = model
        tables, dense = [], []
        for name, p in model.named_parameters():
            (tables if name.startswith("embedding_dict.") else dense).append((name, p))
        self.sig = tuple(p.data_ptr() for _, p in tables + dense)
        total = sum(p.numel() for _, p in dense)
        self.arena = torch.zeros(max(total, 1), dtype=torch.float32, device=device)
        self.pvals = {}
        off = 0
        for name, p in dense:
            g = self.arena[off:off + p.numel()].view(p.shape)
            off += p.numel()
            self.pvals[name] = PVal(p.data, g, name)
            self.pvals[name].stable = True
        self.table_names = [n for n, _ in tables]
        for name, p in tables:
            self.pvals[name] = PVal(p.data, None, name, is_table=True)
        par = getattr(model, "_parallel", None)
        if par is not None and par.mode == "row_sharded":
            # the trained rows of this rank live in ONE flat buffer (parallel.RowSharding); the full per-field tables
            # stay registered (state_dict / predict contract) but are never written by a step
            self.pvals["embedding_shard"] = PVal(par.shard, None, "embedding_shard", is_table=True)
            self.table_names = ["embedding_shard"]
        self.table_grads_ready = False
        self.opt = None
        self.rows = None
        self.extra = {}  # derived / frozen tensors registered by models (STAR)

    def ensure_table_grads(self):
        if not self.table_grads_ready:
            for n in self.table_names:
                pv = self.pvals[n]
                pv.grad = torch.zeros_like(pv.data)
                pv.needs_grad = True
            self.table_grads_ready = True

    def ensure_rows(self, cap, names=None):
        """Touched-row bookkeeping over the tables this rank updates (all of them unless `names` is given)."""
        names = list(self.table_names if names is None else names)
        if self.rows is None or self.rows.touched.numel() < cap or self.rows_names != names:
            self.rows = TableRows([self.pvals[n].data.shape[0] for n in names], self.device, cap)
            self.rows_names = names
        return self.rows

    def ensure_grad_marks(self, tables):
        """Byte map over the rows of `tables` (a gather's field order; ops.marks_bytes layout) for the marked-gradient
        dense update.  Returns (map, byte offset of every table)."""
        vocab = [int(t.data.shape[0]) for t in tables]
        key = tuple(t.data.data_ptr() for t in tables)
        if getattr(self, "_grad_marks_key", None) != key:
            self.grad_marks = torch.zeros(ops.marks_bytes(vocab), dtype=torch.uint8, device=self.device)
            self._grad_marks_key = key
        base, off = [], 0
        for v in vocab:
            base.append(off)
            off += (v + 31) // 32 * 32
        return self.grad_marks, base

    def ensure_det(self, tables):
        """Buffers of the deterministic scatter for `tables` (a gather's field order): int64 [V, E] totals per table (kept
        all zero between steps by the scatter's second launch), a mark map of its own and a magnitude slot."""
        key = tuple(t.data.data_ptr() for t in tables)
        if getattr(self, "_det_key", None) != key:
            uniq = {}
            for t in tables:
                uniq.setdefault(t.data.data_ptr(), torch.zeros(t.data.shape, dtype=torch.int64, device=self.device))
            self._det = dict(acc64=[uniq[t.data.data_ptr()] for t in tables],
                             marks=torch.zeros(ops.marks_bytes([int(t.data.shape[0]) for t in tables]), dtype=torch.uint8,
                                               device=self.device),
                             slot=ops.amax_slots(1, self.device)[0])
            self._det_key = key
        return self._det

    def stale(self):
        return self.sig != tuple(p.data_ptr() for _, p in self.model.named_parameters())

    def reset_written(self):
        for pv in self.pvals.values():
            pv.written = 0
        for pv in self.extra.values():
            pv.written = 0


class _OptState(dict):
    """name -> (state1, state2), zero-initialised on first use (a row-sharded run never touches the full tables)."""

    def __init__(self, store, kind):
        super().__init__()
        self.store, self.kind = store, kind

    def __missing__(self, name):
        pv = self.store.pvals[name]
        s1 = torch.zeros_like(pv.data) if self.kind != "sgd" else None
        s2 = torch.zeros_like(pv.data) if self.kind == "adam" else None
        self[name] = (s1, s2)
        return self[name]


class Optimizer:
    """K8 front end: dense update for MLP parameters; for the tables one of
      dense_exact : every row every step, like the reference's torch.optim over dense gradients;
      sparse_rows : rows of the batch only -- exactly the dense result for SGD / Adagrad, "lazy Adam" otherwise;
      lazy_exact  : rows of the batch only, but the zero-gradient steps a row skipped are replayed before it is next
                    read (mml_opt_catchup_rows) and for all rows before evaluation (flush): the dense Adam / RMSprop
                    trajectory at sparse cost (SURVEY.md A14 "hard part" solved without changing results).
    'auto' = sparse_rows for SGD / Adagrad (exactly the dense result); for Adam / RMSprop lazy_exact where it is
    available -- embedding width 4, 8 or 16, one table per field, no regulariser on the tables (round 4: the same
    dense trajectory, tests at 2e-6, at 51 M instead of 37 M samples/s on AE-30 incl. the flush of a 500-step epoch) --
    else dense_exact.  MMLREC_AUTO_TABLE_UPDATE=dense_exact keeps the reference's literal schedule under 'auto'."""

    def __init__(self, store, kind, lr, table_update="auto"):
        self.store, self.kind, self.lr = store, kind, float(lr)
        if kind not in L.OPT_KINDS:
            raise NotImplementedError(kind)  # model/basemodel.py:581
        self.auto = table_update == "auto"
        if table_update == "auto":
            if kind in ("sgd", "adagrad"):
                table_update = "sparse_rows"
            else:
                import os
                tabs = [p for n, p in store.model.named_parameters() if n.startswith("embedding_dict.")]
                widths = {int(p.shape[1]) for p in tabs}
                cols = store.model._sparse_cols() if hasattr(store.model, "_sparse_cols") else []
                one_per_field = len({f.embedding_name for f in cols}) == len(cols)
                # (small tables: the literal dense update of a few hundred thousand parameters is one short launch, the
                # row bookkeeping of lazy_exact -- mark + compact, catch-up, row update: four launches -- costs more than it
                # saves; KuaiRec-32's 24 k rows x 16: 90 us of bookkeeping against ~5 us, round 5)
                big = sum(p.numel() for p in tabs) > int(os.environ.get("MMLREC_LAZY_MIN_PARAMS", str(1 << 22)))
                ok = (bool(tabs) and widths <= {4, 8, 16} and len(widths) == 1 and one_per_field and big and
                      os.environ.get("MMLREC_AUTO_TABLE_UPDATE", "lazy_exact") == "lazy_exact")
                table_update = "lazy_exact" if ok else "dense_exact"
            if self._table_reg(self._reg_map()):  # a regulariser on the tables moves every row every step
                table_update = "dense_exact"
        if table_update == "lazy_exact" and kind in ("sgd", "adagrad"):
            table_update = "sparse_rows"  # nothing to replay: zero gradients do not move these optimizers
        if table_update not in ("dense_exact", "sparse_rows", "lazy_exact"):
            raise ValueError("table_update must be auto, dense_exact, sparse_rows or lazy_exact")
        self.table_update = table_update
        self.last = None   # lazy_exact: per-table int32 [V] "row is current as of step"
        self.dirty = False
        dev = store.device
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.state = _OptState(store, kind)  # moments are allocated when a tensor is first updated
        self.steps_done = 0

    def calls(self, plan):
        """Optimizer call list for one step (appended after a plan's backward)."""
        c = self.calls_split(plan)
        return c["pre"] + c["early"] + c["mlp"] + c["tables"]

    def can_split_dense(self, plan):
        """The dense table update may run as (untouched rows early, next to the forward) + (touched rows after the
        scatter): needs the batch's row set before the forward (mml_index_unique: E <= 16, indices on this rank) and
        an update that is the same function of (p, g, state) in both kernels (no regulariser on the tables)."""
        if self.table_update != "dense_exact" or self._table_reg(self._reg_map()):
            return False
        gop = plan.ops[0] if plan.ops else None
        if not isinstance(gop, GatherOp):
            return False
        return all(t.data.shape[1] <= 16 and t.data.shape[1] % 4 == 0 for t in gop.tables)

    def calls_split(self, plan, split_dense=False):
        """{'pre': step-counter bump (+ the index pre-pass), 'early': the untouched-rows half of a split dense table
        update (may run beside the forward / backward), 'mlp': dense MLP update, 'tables': table update (+ touched-list
        reset)} so a trainer can put them on different streams."""
        lib = L.load()
        st = self.store
        pre = [(lib.mml_counter_update, (self.step_dev.data_ptr(), 1, 0))]
        if self.table_update == "lazy_exact":
            pre += self._lazy_pre_calls(plan)
        split_dense = bool(split_dense) and self.can_split_dense(plan)
        if split_dense and getattr(plan.ops[0], "mark_rows", None) is None:  # (else the gather itself lists the rows)
            pre += self._unique_pre_calls(plan)
        calls = []
        reg = self._reg_map()
        # (a regularised parameter is updated even when no gradient reaches it -- the reference's dead PLE tensors,
        # its arena slice stays zero, the update sees the regulariser's gradient alone)
        dense = [(pv, n) for n, pv in st.pvals.items()
                 if not pv.is_table and pv.grad is not None and (pv.written or pv.data.data_ptr() in reg)]
        entries = [(pv.data, pv.grad) + self.state[n] + (reg.get(pv.data.data_ptr()),) for pv, n in dense]
        tabs = [st.pvals[n] for n in st.table_names if st.pvals[n].written]
        tnames = [n for n in st.table_names if st.pvals[n].written]
        treg = self._table_reg(reg)
        if treg and self.table_update != "dense_exact":
            raise NotImplementedError("l2_reg_embedding / l1 on the tables makes every row's gradient non-zero: use "
                                      "table_update='dense_exact' (the reference's own dense optimizer)")
        hyper = ops.make_hyper(self.kind, self.lr, step=0, step_dev=self.step_dev, zero_grad=False)
        plan.keep.append(hyper)
        if entries:
            arr = ops.make_opt_tensors(entries)
            plan.keep.append(arr)
            per = {"sgd": 12, "adam": 28, "adagrad": 20, "rmsprop": 20}[self.kind]
            calls.append((lib.mml_opt_step_dense, (arr, len(entries), C.byref(hyper)),
                          dict(kernel=_opt_dense_symbol(sum(e[0].numel() for e in entries), len(entries)),
                               bytes=float(per) * sum(e[0].numel() for e in entries))))
        mlp_calls, calls, early = calls, [], []
        if tabs:
            if self.table_update == "dense_exact":
                # the early half of a split update shares the chip with the forward / backward; its grid can be capped
                # (mml_opt_hyper.max_blocks, MMLREC_EARLY_BLOCKS) so that it leaves them wave slots.  Same-box A/B runs
                # (B = 65 536 and 4 096, caps 512 .. 2048) stayed inside the run-to-run noise, so the default is the
                # full grid, at which the stream runs at its stand-alone bandwidth.
                # Round 3: the single marked launch runs beside the weight-gradient GEMMs of the side stream.  Every loop
                # form is its own kernel (csrc/optim_ew.hip: the plain loop at 54 VGPRs / 8 waves per SIMD, two chunks per
                # thread at 100 / 4, four at 172 / 2, eight at 256 / 1).  Alone (same box, ms): two chunks 0.39-0.44, plain
                # 0.43-0.49, four chunks 0.47-0.50, eight 0.85.  In the step (three interleaved repetitions on a quiet box):
                # plain 1.928, two chunks 1.912, four chunks under a 3072-workgroup cap 1.883 -- the two-wave form leaves
                # the GEMMs their registers -- but on other boxes the three are level within the +-3 % drift of a run.
                # Default: two chunks on the full grid, the best stream by itself (0.66-0.74 of 8 TB/s) and level in the
                # step; MMLREC_TAIL_BLOCKS (workgroup cap, 0 = plain loop) / MMLREC_OPT_U override.
                cap = int(os.environ.get("MMLREC_EARLY_BLOCKS", "0")) if split_dense else \
                    int(os.environ.get("MMLREC_TAIL_BLOCKS", str(1 << 20)))
                hz = ops.make_hyper(self.kind, self.lr, step=0, step_dev=self.step_dev, zero_grad=not split_dense,
                                    max_blocks=cap)
                plan.keep.append(hz)
                # p, g, m, v read + p, m, v written; the split form never reads g
                per = {"sgd": 12, "adam": 28, "adagrad": 20, "rmsprop": 20}[self.kind] - (4 if split_dense else 0)
                # one C call = one launch (the same size rule mml_opt_step_dense applies inside a call), so that a
                # call's label is the kernel symbol a profiler reports
                # the huge tables stream through opt_dense_kernel<true> (one call), every other table shares one
                # balanced opt_flat_kernel launch (the same split mml_opt_step_dense makes for a mixed call)
                big = [i for i in range(len(tabs)) if tabs[i].data.numel() >= (1 << 22)]
                if len(big) > 4 or sum(tabs[i].data.numel() for i in big) < (1 << 24):
                    big = []
                small = [i for i in range(len(tabs)) if i not in big]
                groups = [g_ for g_ in (big, small) if g_]
                seen_of = {}
                if split_dense:
                    seen_of = dict(zip(st.rows_names, st.rows.seen))
                # marked gradients (the scatter of this plan marked every row it added to): the streaming launch does not
                # read the gradient of unmarked rows -- 24 instead of 28 bytes per Adam parameter
                marks_of = {}
                gop = plan.ops[0] if plan.ops else None
                gm = getattr(gop, "grad_marks", None)
                if gm is not None and not split_dense:
                    _, base = st.ensure_grad_marks(gop.tables)
                    for f, t in enumerate(gop.tables):
                        marks_of[id(t)] = gm[base[f]:base[f] + t.data.shape[0]]
                # Round 6, built and measured, NOT the default: every table in ONE marked streaming launch (workgroups dealt
                # in proportion to the tables' sizes, csrc/optim_ew.hip) instead of the streaming launch of the huge tables
                # + a flat launch of the small ones (AE-30: 26 tables, 31-34 us).  Two call lists over ONE model replayed in
                # alternating blocks, both orders (tools/lab/ab_inproc.py --shared, profiles/r06_tail_lab.txt): the single
                # launch is 13 us per step SLOWER (workgroups dealt in proportion to the sizes) or 8 us slower (every tensor
                # the grid a launch of its own would get, small tables first: the form kept).  MMLREC_OPT_ONE_LAUNCH=1: on.
                one_launch = (big and small and cap > 0 and len(tabs) <= L.MAX_OPT_TENSORS and not split_dense and
                              all(id(t) in marks_of for t in tabs) and
                              os.environ.get("MMLREC_OPT_ONE_LAUNCH", "0") == "1")
                if one_launch:
                    big = small + big   # (the small tables' workgroups first: they start with the launch)
                    groups = [big]
                for grp in groups:
                    marked = grp is big and all(id(tabs[i]) in marks_of for i in grp)
                    arr = ops.make_opt_tensors([(tabs[i].data, tabs[i].grad) + self.state[tnames[i]] +
                                                (treg, seen_of.get(tnames[i]), marks_of[id(tabs[i])] if marked else None)
                                                for i in grp])
                    plan.keep.append(arr)
                    numel = sum(tabs[i].data.numel() for i in grp)
                    nbytes = float(per) * numel
                    v = int(os.environ.get("MMLREC_OPT_VARIANT", "0"))
                    form = (3 if (cap > 0 and marked) else 2 if (cap > 0 and split_dense) else
                            1 if (v & 2 and not marked and not split_dense) else 0)
                    if marked:  # g is read for the touched rows only (~1 %): count the mark bytes instead
                        nbytes += sum(tabs[i].data.shape[0] - 4.0 * tabs[i].data.numel() for i in grp)
                    (early if split_dense else calls).append(
                        (lib.mml_opt_step_dense, (arr, len(grp), C.byref(hz)),
                         dict(kernel=_opt_dense_symbol(numel, len(grp), form), bytes=nbytes)))
            if self.table_update != "dense_exact" or split_dense:
                rows = st.rows
                lazy = self.table_update == "lazy_exact"
                F = len(tabs)
                E = tabs[0].data.shape[1]
                pt = ops._ptr_array([pv.data for pv in tabs])
                pg = ops._ptr_array([pv.grad for pv in tabs])
                p1 = ops._ptr_array([self.state[n][0] for n in tnames]) if self.kind != "sgd" else None
                p2 = ops._ptr_array([self.state[n][1] for n in tnames]) if self.kind == "adam" else None
                ps = ops._ptr_array(rows.seen)
                rb = (L.i64 * (F + 1))(*rows.rowbase)
                plan.keep += [pt, pg, p1, p2, ps, rb]
                pl = ops._ptr_array([self.last[n] for n in tnames]) if lazy else None
                plan.keep.append(pl)
                calls.append((lib.mml_opt_step_rows, (pt, pg, p1, p2, ps, rb, F, E, rows.touched.data_ptr(),
                                                      rows.count.data_ptr(), rows.touched.numel(), pl,
                                                      C.byref(hyper)), dict(kernel="opt_rows_kernel")))
                # the list is REBUILT every step by the compaction that follows the marking kernels (E in 4, 8, 16: it
                # resets the counter itself); only the appending atomic path needs the reset here
                if E not in (4, 8, 16) or os.environ.get("MMLREC_SCATTER_OLD"):
                    calls.append((lib.mml_counter_update, (rows.count.data_ptr(), 0, 1)))
        return {"pre": pre, "early": early, "mlp": mlp_calls, "tables": calls}

    def _unique_pre_calls(self, plan):
        """Index pre-pass of the split dense update: the batch's distinct rows -> `seen` bitmaps + touched list."""
        lib, st = L.load(), self.store
        gop = plan.ops[0]
        names, rows = st.table_names, st.rows
        if rows is None or list(st.rows_names) != list(names):
            raise L.MMLError("split dense update needs ParamStore.ensure_rows over every table")
        F = len(names)
        E = st.pvals[names[0]].data.shape[1]
        vocab = (L.i64 * F)(*[st.pvals[n].data.shape[0] for n in names])
        ps = ops._ptr_array(rows.seen)
        rb = (L.i64 * (F + 1))(*rows.rowbase)
        X, nrows = gop.index_view(plan)
        col = (L.i32 * F)(*gop.cols)
        plan.keep += [vocab, ps, rb, col]
        return gop.pre_index_calls(plan) + [
            (lib.mml_index_unique, (vocab, col, F, E, X.data_ptr(), ops._ld(X), nrows, ps, rb,
                                    rows.touched.data_ptr(), rows.count.data_ptr(), rows.touched.numel(),
                                    rows.marks.data_ptr(), plan.status.data_ptr()),
             dict(kernel="mark_rows_kernel+rows_compact_kernel", bytes=float(nrows) * F * 5))]

    # ---- regulariser (model/basemodel.py:524-540) ---------------------------------------------------------
    def _reg_map(self):
        """data_ptr -> (l1, l2) summed over the groups the model registered with add_regularization_weight."""
        out = {}
        for weight_list, l1, l2 in getattr(self.store.model, "regularization_weight", []):
            if not (l1 > 0 or l2 > 0):
                continue
            for w in weight_list:
                p = w[1] if isinstance(w, tuple) else w
                a, b = out.get(p.data_ptr(), (0.0, 0.0))
                out[p.data_ptr()] = (a + float(l1), b + float(l2))
        return out

    def _table_reg(self, reg):
        """(l1, l2) of the embedding tables (one setting for all of them: l2_reg_embedding), or None."""
        vals = {reg[p.data_ptr()] for n, p in self.store.model.named_parameters()
                if n.startswith("embedding_dict.") and p.data_ptr() in reg}
        if not vals:
            return None
        if len(vals) > 1:
            raise NotImplementedError("different regularisers on different embedding tables")
        return vals.pop()

    # ---- lazy_exact ----------------------------------------------------------------------------------
    def _lazy_pre_calls(self, plan):
        """Before the gather: unique rows of the batch (LDS dedup on the indices) -> replay their skipped steps."""
        lib, st = L.load(), self.store
        gop = plan.ops[0]
        names = st.table_names
        rows = st.rows
        if self.last is None:
            self.last = {n: torch.zeros(st.pvals[n].data.shape[0], dtype=torch.int32, device=st.device) for n in names}
        F = len(names)
        E = st.pvals[names[0]].data.shape[1]
        vocab = (L.i64 * F)(*[st.pvals[n].data.shape[0] for n in names])
        ps = ops._ptr_array(rows.seen)
        rb = (L.i64 * (F + 1))(*rows.rowbase)
        pt = ops._ptr_array([st.pvals[n].data for n in names])
        p1 = ops._ptr_array([self.state[n][0] for n in names])
        p2 = ops._ptr_array([self.state[n][1] for n in names]) if self.kind == "adam" else None
        pl = ops._ptr_array([self.last[n] for n in names])
        hyper = ops.make_hyper(self.kind, self.lr, step=0, step_dev=self.step_dev)
        plan.keep += [vocab, ps, rb, pt, p1, p2, pl, hyper]
        catchup = (lib.mml_opt_catchup_rows, (pt, p1, p2, pl, rb, F, E, rows.touched.data_ptr(), rows.count.data_ptr(),
                                              rows.touched.numel(), C.byref(hyper)), dict(kernel="opt_catchup_kernel"))
        if getattr(gop, "owns_lazy", False):
            # row-sharded tables: the keys to bring up to date only exist on the owner after the index exchange, so
            # the gather op launches (unique -> catch-up) itself, on the flat shard (F == 1)
            if F != 1:
                raise L.MMLError("row-sharded lazy_exact expects the single flat shard")

            def launch(keys_ptr, n, stream):
                if n:
                    L.check(lib.mml_index_unique_idx32(vocab, 1, E, keys_ptr, 1, n, ps, rb, rows.touched.data_ptr(),
                                                       rows.count.data_ptr(), rows.touched.numel(),
                                                       rows.marks.data_ptr(), plan.status.data_ptr(), stream),
                            "mml_index_unique_idx32")
                    L.check(catchup[0](*catchup[1], stream), "mml_opt_catchup_rows")
            gop.lazy_launch = launch
            return []
        if not isinstance(gop, GatherOp):
            raise L.MMLError("lazy_exact table updates are not available on the table-wise sharded path")
        X, nrows = gop.index_view(plan)
        col = (L.i32 * F)(*gop.cols)
        plan.keep.append(col)
        return gop.pre_index_calls(plan) + [
            (lib.mml_index_unique, (vocab, col, F, E, X.data_ptr(), ops._ld(X), nrows, ps, rb,
                                    rows.touched.data_ptr(), rows.count.data_ptr(), rows.touched.numel(),
                                    rows.marks.data_ptr(), plan.status.data_ptr()),
             dict(kernel="mark_rows_kernel+rows_compact_kernel", bytes=float(nrows) * F * 5)),
            catchup,
        ]

    def flush(self):
        """Bring EVERY table row to the current step (needed before anything outside the fused step reads a table)."""
        if self.table_update != "lazy_exact" or not self.dirty or self.last is None:
            return
        hyper = ops.make_hyper(self.kind, self.lr, step=0, step_dev=self.step_dev)
        for n in self.store.table_names:
            s1, s2 = self.state[n]
            ops.opt_catchup_dense(self.store.pvals[n].data, s1, s2 if self.kind == "adam" else None, self.last[n], hyper)
        self.dirty = False
