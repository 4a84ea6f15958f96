"""Feature schema and parameter containers (counterpart of the reference's model/utils.py).

Names, constructor signatures, parameter registration order, initialisers and state_dict keys follow the reference
so seeds and checkpoints carry over; the arithmetic does not live here -- modules only HOLD parameters and describe
themselves to the step-plan builder (engine.py), which emits the HIP kernels.
"""
import math
from collections import OrderedDict, namedtuple

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import engine as E

DEFAULT_GROUP_NAME = "default_group"


# ---- feature columns (reference model/utils.py:328-395) ------------------------------------------
class SparseFeat(namedtuple("SparseFeat", ["name", "vocabulary_size", "embedding_dim", "use_hash", "dtype",
                                           "embedding_name", "group_name"])):
    __slots__ = ()

    def __new__(cls, name, vocabulary_size, embedding_dim=4, use_hash=False, dtype="int32", embedding_name=None,
                group_name=DEFAULT_GROUP_NAME):
        if embedding_dim == "auto":
            embedding_dim = 6 * int(pow(vocabulary_size, 0.25))
        if use_hash:
            print("Notice! Feature Hashing on the fly is not supported")
        return super().__new__(cls, name, vocabulary_size, embedding_dim, use_hash, dtype,
                               name if embedding_name is None else embedding_name, group_name)

    def __hash__(self):
        return hash(self.name)


class VarLenSparseFeat(namedtuple("VarLenSparseFeat", ["sparsefeat", "maxlen", "combiner", "length_name"])):
    """Accepted for API compatibility; the hot path never builds one (utils/data_utils.py:73-75), and the
    step plans reject it."""
    __slots__ = ()

    def __new__(cls, sparsefeat, maxlen, combiner="mean", length_name=None):
        return super().__new__(cls, sparsefeat, maxlen, combiner, length_name)

    name = property(lambda self: self.sparsefeat.name)
    vocabulary_size = property(lambda self: self.sparsefeat.vocabulary_size)
    embedding_dim = property(lambda self: self.sparsefeat.embedding_dim)
    embedding_name = property(lambda self: self.sparsefeat.embedding_name)
    group_name = property(lambda self: self.sparsefeat.group_name)

    def __hash__(self):
        return hash(self.name)


class DenseFeat(namedtuple("DenseFeat", ["name", "dimension", "dtype"])):
    __slots__ = ()

    def __new__(cls, name, dimension=1, dtype="float32"):
        return super().__new__(cls, name, dimension, dtype)

    def __hash__(self):
        return hash(self.name)


def build_input_features(feature_columns):
    """{feature name: (first column, one-past-last column)} in X (reference model/utils.py:407-431)."""
    features = OrderedDict()
    start = 0
    for feat in feature_columns:
        if feat.name in features:
            continue
        if isinstance(feat, SparseFeat):
            width = 1
        elif isinstance(feat, DenseFeat):
            width = feat.dimension
        elif isinstance(feat, VarLenSparseFeat):
            width = feat.maxlen
        else:
            raise TypeError("Invalid feature column type,got", type(feat))
        features[feat.name] = (start, start + width)
        start += width
        if isinstance(feat, VarLenSparseFeat) and feat.length_name is not None and feat.length_name not in features:
            features[feat.length_name] = (start, start + 1)
            start += 1
    return features


def get_feature_names(feature_columns):
    return list(build_input_features(feature_columns).keys())


def create_embedding_matrix(feature_columns, init_std=0.0001, linear=False, sparse=False, device="cpu"):
    """One nn.Embedding per sparse field, N(0, init_std) (reference model/utils.py:466-488).  Only the weight is
    used: lookups go through the fused gather kernel."""
    cols = [f for f in feature_columns if isinstance(f, (SparseFeat, VarLenSparseFeat))] if len(feature_columns) else []
    cols = [f for f in cols if isinstance(f, SparseFeat)] + [f for f in cols if isinstance(f, VarLenSparseFeat)]
    d = nn.ModuleDict({f.embedding_name: nn.Embedding(f.vocabulary_size, f.embedding_dim if not linear else 1,
                                                      sparse=sparse) for f in cols})
    for emb in d.values():
        nn.init.normal_(emb.weight, mean=0, std=init_std)
    return d.to(device)


def get_mask(domain_values, mask_values, num_domains):
    """One-hot [N, D] int mask of domain_values == mask_values[d] (reference model/utils.py:639-645)."""
    dv = torch.as_tensor(np.asarray(domain_values), dtype=torch.float32).reshape(-1, 1).repeat(1, num_domains)
    mv = torch.as_tensor(np.asarray(mask_values), dtype=torch.float32).reshape(1, -1).repeat(dv.shape[0], 1)
    return (dv == mv).int()


def activation_code(name):
    if isinstance(name, str):
        key = name.lower()
        if key in E.ACT:
            return E.ACT[key]
    if name is None:
        return L.ACT_NONE
    raise NotImplementedError(f"activation {name!r} (the hot path supports relu / sigmoid / linear)")


def activation_layer(act_name, hidden_size=None, dice_dim=2):
    """Kept for API compatibility (modules built from it are descriptive only)."""
    code = activation_code(act_name)
    return {L.ACT_RELU: nn.ReLU(inplace=True), L.ACT_SIGMOID: nn.Sigmoid(), L.ACT_NONE: nn.Identity()}[code]


# ---- parameter containers -------------------------------------------------------------------------
class DNN(nn.Module):
    """[Linear -> act] x L parameter stack (reference model/utils.py:92-161).  Weights N(0, init_std), biases keep
    nn.Linear's default init.  use_bn adds a BatchNorm1d after every Linear (engine.BNOp); dropout_rate > 0 adds an
    engine.DropoutOp after every activation while the model is in training mode (reference :159)."""

    def __init__(self, inputs_dim, hidden_units, activation="relu", l2_reg=0, dropout_rate=0, use_bn=False,
                 init_std=0.0001, dice_dim=3, device="cpu"):
        super().__init__()
        if len(hidden_units) == 0:
            raise ValueError("hidden_units is empty!!")
        if not 0 <= dropout_rate < 1:
            raise ValueError("dropout_rate must be in [0, 1)")  # (reference docstring, model/utils.py:109)
        self.l2_reg, self.use_bn, self.dropout_rate, self.activation = l2_reg, use_bn, dropout_rate, activation
        self.act_code = activation_code(activation)
        dims = [inputs_dim] + list(hidden_units)
        self.linears = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        if use_bn:  # fc -> bn -> activation (reference :132-134, :153-157); registered right after the linears
            self.bn = nn.ModuleList([nn.BatchNorm1d(dims[i + 1]) for i in range(len(dims) - 1)])
        for name, p in self.linears.named_parameters():
            if "weight" in name:
                nn.init.normal_(p, mean=0, std=init_std)
        self.to(device)

    @property
    def out_dim(self):
        return self.linears[-1].out_features

    def layer_problems(self, plan, store, prefix, x, last16=False):
        """One LinearGroup problem per layer (the output value of layer l is the input of layer l+1); the caller
        launches layer l of sibling stacks together (emit_dnn_stacks)."""
        h = x
        out = []
        plain = not self.use_bn and not (self.dropout_rate and plan.dropout_on)
        for l, lin in enumerate(self.linears):
            q = dict(x=h, W=store.pvals[f"{prefix}.linears.{l}.weight"], b=store.pvals[f"{prefix}.linears.{l}.bias"])
            # bf16-storage path (engine.Plan.bf16): the value between two layers of the stack is read by the next layer's
            # GEMM only -- stored as bf16 when this layer and the next both run on the bf16-storage kernels
            nxt = self.linears[l + 1] if l + 1 < len(self.linears) else None
            # (last16: the caller vouches that whatever reads the stack's OUTPUT takes bf16 -- the fast gate kernels)
            store16 = (plan.bf16 and plain and h.is16 and
                       E.g16_layer_ok(plan.B, lin.in_features, lin.out_features, plan.training) and
                       (E.g16_layer_ok(plan.B, nxt.in_features, nxt.out_features, plan.training) if nxt is not None
                        else bool(last16)))
            if self.use_bn:  # the GEMM writes the pre-normalisation value; emit_dnn_stacks adds the BatchNorm op
                q["out"] = plan.val(lin.out_features, name=f"{prefix}.{l}.z")
                q["bn"] = dict(y=plan.val(lin.out_features, act=self.act_code, name=f"{prefix}.{l}"),
                               gamma=store.pvals[f"{prefix}.bn.{l}.weight"], beta=store.pvals[f"{prefix}.bn.{l}.bias"],
                               module=self.bn[l])
                h = q["bn"]["y"]
            else:
                q["out"] = h = plan.val(lin.out_features, act=self.act_code, name=f"{prefix}.{l}", store16=store16)
            if self.dropout_rate and plan.dropout_on:  # act -> dropout (reference :156-159); emit_dnn_stacks adds the op
                q["drop"] = dict(x=h, y=plan.val(lin.out_features, name=f"{prefix}.{l}.drop"), p=self.dropout_rate,
                                 site=E.dropout_site(f"{prefix}.{l}"))
                h = q["drop"]["y"]
            out.append(q)
        return out

    def forward(self, inputs):
        from ..functional import linear_act
        if self.use_bn:
            raise NotImplementedError("stand-alone DNN.forward with BatchNorm: use the model's forward")
        if self.dropout_rate and self.training:
            raise NotImplementedError("stand-alone DNN.forward with dropout in training mode: use the model's forward")
        h = inputs
        for lin in self.linears:
            h = linear_act(h, lin.weight, lin.bias, self.act_code)
        return h


def emit_dnn_stacks(plan, stacks):
    """stacks: list of per-stack problem lists (from DNN.layer_problems).  Layer l of every stack is launched as ONE
    grouped GEMM (experts + gate DNNs of an MMoE share the input and the launch)."""
    depth = max(len(s) for s in stacks)
    for l in range(depth):
        probs = [s[l] for s in stacks if len(s) > l]
        plan.add(E.LinearGroupOp(probs))
        for q in probs:
            if "bn" in q:
                b = q["bn"]
                plan.add(E.BNOp(q["out"], b["y"], b["gamma"], b["beta"], b["module"]))
            if "drop" in q:
                d = q["drop"]
                plan.add(E.DropoutOp(d["x"], d["y"], d["p"], plan.dropout_seed, d["site"]))
    return [s[-1]["drop"]["y"] if "drop" in s[-1] else (s[-1]["bn"]["y"] if "bn" in s[-1] else s[-1]["out"])
            for s in stacks]


def blocks_out_act(plan, blocks):
    """Activation code the GIVEN outputs of emit_blocks_into must carry: the blocks' own, or none when dropout sits
    between the activation and the output (training mode)."""
    if plan.dropout_on and any(b.dropout_rate for b in blocks):
        return L.ACT_NONE
    return L.ACT_RELU


def emit_blocks_into(plan, store, blocks, prefixes, ins, outs):
    """One grouped launch for single-layer DNN blocks whose outputs are GIVEN values (column slices of a concatenation
    buffer: cross-stitch / SNR-trans / MSSM levels).  A block with BatchNorm writes its Linear output to a scratch value
    and the BatchNorm op produces the given one."""
    probs = []
    for blk, pfx, x, o in zip(blocks, prefixes, ins, outs):
        q = dict(x=x, W=store.pvals[f"{pfx}.linears.0.weight"], b=store.pvals[f"{pfx}.linears.0.bias"])
        drop = bool(blk.dropout_rate) and plan.dropout_on
        # with dropout the activation's output is a scratch value and the dropout op produces the given one
        if drop and o.act != L.ACT_NONE:
            raise L.MMLError("emit_blocks_into: the given output of a block with dropout must carry no activation "
                             "(blocks_out_act)")
        a = plan.val(o.n, act=blk.act_code, name=o.name + ".a") if drop else o
        if blk.use_bn:
            q["out"] = plan.val(o.n, name=o.name + ".z")
            q["bn"] = dict(y=a, gamma=store.pvals[f"{pfx}.bn.0.weight"], beta=store.pvals[f"{pfx}.bn.0.bias"],
                           module=blk.bn[0])
        else:
            q["out"] = a
        if drop:
            q["drop"] = dict(x=a, y=o, p=blk.dropout_rate, site=E.dropout_site(f"{pfx}.0"))
        probs.append(q)
    plan.add(E.LinearGroupOp(probs))
    for q in probs:
        if "bn" in q:
            b = q["bn"]
            plan.add(E.BNOp(q["out"], b["y"], b["gamma"], b["beta"], b["module"]))
        if "drop" in q:
            d = q["drop"]
            plan.add(E.DropoutOp(d["x"], d["y"], d["p"], plan.dropout_seed, d["site"]))


class DomainBatchNorm(nn.Module):
    """STAR's per-domain BatchNorm (reference model/utils.py:553-636).  Like the reference's, its tensors are plain
    attributes -- gamma / beta frozen at (1, 0), the population statistics (zeros / ones) outside state_dict and not
    moved by .to(): they are created on the device of the first plan that uses them.  What it computes:
    engine.DomainBNOp / csrc/bn.hip."""

    def __init__(self, num_features, num_domains, decay=0.99, epsilon=1e-3, device=None, type=None):
        super().__init__()
        self.num_features, self.num_domains, self.decay, self.epsilon = num_features, num_domains, decay, epsilon
        self._pop = None

    def population(self, device):
        """(pop_means, pop_vars) as [num_domains, num_features] device tensors."""
        if self._pop is None or self._pop[0].device != device:
            self._pop = (torch.zeros(self.num_domains, self.num_features, device=device),
                         torch.ones(self.num_domains, self.num_features, device=device))
        return self._pop

    def __deepcopy__(self, memo):
        new = DomainBatchNorm(self.num_features, self.num_domains, self.decay, self.epsilon)
        if self._pop is not None:
            new._pop = tuple(t.clone() for t in self._pop)
        return new


class PredictionLayer(nn.Module):
    """bias [1] + sigmoid for task == 'binary' (reference model/utils.py:225-248); fused into the head kernel."""

    def __init__(self, task="binary", use_bias=True, **kwargs):
        if task not in ["binary", "multiclass", "regression"]:
            raise ValueError("task must be binary,multiclass or regression")
        super().__init__()
        self.use_bias, self.task = use_bias, task
        if use_bias:
            self.bias = nn.Parameter(torch.zeros((1,)))


class SharedSpecificLinear(nn.Module):
    """STAR layer: y = x @ (W_spec[d] * W_shared) + b_spec[d] + b_shared, weights stored [in, out]
    (reference model/utils.py:163-223).  Reference quirk kept: the per-domain tensors live in plain
    Python lists and the attribute `specific_weight` is overwritten in the loop, so ONLY the last domain's specific
    weight/bias are registered parameters (state_dict keys `specific_weight` / `specific_bias`); the others are
    frozen at their initial draw."""

    def __init__(self, in_features, out_features, num_domains, use_shared=True, use_bias=True, device="cpu"):
        super().__init__()
        if not (use_shared and use_bias):
            raise NotImplementedError("STAR hot path covers use_shared=True, use_bias=True (every shipped config)")
        self.in_features, self.out_features, self.use_shared = in_features, out_features, use_shared
        # drawn on the host generator whatever `device` is, so a seed gives the same weights as the CPU reference
        # (the reference draws on `device`; its CUDA stream of random numbers is a different one anyway)
        self.shared_weight = nn.Parameter(torch.empty((in_features, out_features)))
        self.shared_bias = nn.Parameter(torch.empty(out_features))
        self._reset(self.shared_weight, self.shared_bias)
        self.specific_weights, self.specific_biases = [], []
        for _ in range(num_domains):
            self.specific_weight = nn.Parameter(torch.empty((in_features, out_features)))
            self.specific_bias = nn.Parameter(torch.empty(out_features))
            self._reset(self.specific_weight, self.specific_bias)
            self.specific_weights.append(self.specific_weight)
            self.specific_biases.append(self.specific_bias)
        self.to(device)

    def _reset(self, w, b):
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.shared_weight)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        nn.init.uniform_(b, -bound, bound)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        # the frozen per-domain tensors are invisible to nn.Module; keep them on the module's device
        n = len(self.specific_weights)
        for d in range(n - 1):
            self.specific_weights[d] = nn.Parameter(fn(self.specific_weights[d].data), requires_grad=False)
            self.specific_biases[d] = nn.Parameter(fn(self.specific_biases[d].data), requires_grad=False)
        self.specific_weights[n - 1] = self.specific_weight
        self.specific_biases[n - 1] = self.specific_bias
        return self

    def extra_repr(self):
        return "in_features={}, out_features={}".format(self.in_features, self.out_features)


def l2_on_weights(model, modules, l2):
    """Puts every non-BN `weight` of the given modules (or bare weight tensors) on the model's L2 list -- the
    add_regularization_weight(filter(...)) idiom the reference repeats after each sub-network (e.g. model/mmoe.py:59-62)."""
    for m in modules:
        if m is None:
            continue
        if isinstance(m, nn.Parameter):
            model.add_regularization_weight(m, l2=l2)
        else:
            model.add_regularization_weight(
                [(n, p) for n, p in m.named_parameters() if "weight" in n and "bn" not in n], l2=l2)


def dnn_options(mc, init_std, device):
    """(activation, keyword arguments) every DNN of a model is built with, from model_config."""
    return dict(activation=mc.get("dnn_activation", "relu"), dropout_rate=mc.get("dnn_dropout", 0),
                use_bn=mc.get("dnn_use_bn", False), init_std=init_std, device=device)
