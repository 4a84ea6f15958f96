"""PepNet (reference model/pepnet.py:8-157): EPNet feature gate 2*sigmoid(MLP(cat(sg(dnn_input), scene_emb))) on the
input, then per task a PPNet block whose every layer input is re-weighted by its own gate network driven by
cat(sg(gated input), scene_emb).  Gate networks of all tasks/layers read the same input and launch as grouped GEMMs;
the stop-gradients are explicit copies whose values carry no gradient buffer."""
import torch.nn as nn

from .. import _lib as L
from .. import engine as E
from .basemodel import BaseModel
from .utils import PredictionLayer, activation_layer


class GateNN(nn.Module):
    """Linear -> ReLU -> Linear -> Sigmoid, output doubled (reference model/pepnet.py:8-32)."""

    def __init__(self, input_dim, hidden_dim=None, output_dim=None, hidden_activation="relu", dropout_rate=0.0,
                 batch_norm=False, device="cpu"):
        super().__init__()
        if batch_norm or dropout_rate > 0:
            raise NotImplementedError("GateNN batch_norm / dropout are outside the hot path")
        if hidden_dim is None:
            hidden_dim = output_dim
        self.gate = nn.Sequential(nn.Linear(input_dim, hidden_dim), activation_layer(hidden_activation),
                                  nn.Linear(hidden_dim, output_dim), nn.Sigmoid())
        self.to(device)


class PPNetBlock(nn.Module):
    """Gated MLP parameter container (reference model/pepnet.py:34-78)."""

    def __init__(self, input_dim, output_dim=1, gate_input_dim=32, gate_hidden_dim=None, hidden_units=[],
                 hidden_activations="relu", dropout_rates=0.0, batch_norm=False, use_bias=True, device="cpu"):
        super().__init__()
        if batch_norm or dropout_rates or not use_bias or output_dim != 1:
            raise NotImplementedError("PPNetBlock options outside the shipped PepNet configuration")
        self.gate_layers = nn.ModuleList()
        self.mlp_layers = nn.ModuleList()
        dims = [input_dim] + list(hidden_units)
        for i in range(len(dims) - 1):
            dense = nn.Linear(dims[i], dims[i + 1], bias=use_bias)  # created before the gate: same RNG order as the reference
            self.gate_layers.append(GateNN(gate_input_dim, gate_hidden_dim, output_dim=dims[i]))
            self.mlp_layers.append(nn.Sequential(dense, activation_layer(hidden_activations)))
        self.gate_layers.append(GateNN(gate_input_dim, gate_hidden_dim, output_dim=dims[-1]))
        self.mlp_layers.append(nn.Linear(dims[-1], output_dim, bias=use_bias))
        self.to(device)


class PepNet(BaseModel):
    def __init__(self, dnn_feature_columns, init_std=0.0001, device="cpu", gpus=None, config=None):
        super().__init__(linear_feature_columns=[], dnn_feature_columns=dnn_feature_columns, init_std=init_std,
                         device=device, gpus=gpus, config=config)
        mc, dc = self.model_config, self.data_config
        self.dnn_use_bn = mc.get("dnn_use_bn", False)
        self.dnn_hidden_units = mc.get("dnn_hidden_units", [256, 128])
        scene_emb_dim = mc.get("emb", 8)
        scene_feature = dc.get("scene_feature", "")
        self.user_sf, self.item_sf = dc.get("user_sf", ""), dc.get("item_sf", "")
        if self.user_sf != "" or self.item_sf != "":
            raise NotImplementedError("PepNet user_sf / item_sf side features are not planned yet (empty in every "
                                      "shipped config)")
        if scene_feature == "":
            raise NotImplementedError("PepNet needs data_config.scene_feature")
        self.scene_index = self.feature_index[scene_feature]
        task_dim = scene_emb_dim
        input_dim = self.compute_input_dim(dnn_feature_columns)
        self.feature_gate = GateNN(input_dim=input_dim + scene_emb_dim, hidden_dim=128, output_dim=input_dim,
                                   device=device)
        self.ppn = nn.ModuleList([PPNetBlock(input_dim=input_dim, output_dim=1, gate_input_dim=input_dim + task_dim,
                                             gate_hidden_dim=None, hidden_units=self.dnn_hidden_units, device=device)
                                  for _ in range(self.num_tasks)])
        self.out = nn.ModuleList([PredictionLayer(task) for task in self.task_types])
        self.to(device)

    def _gate_hidden(self, plan, store, gin_vals):
        """First layer (Linear + ReLU) of a list of (prefix, input value) gate networks as ONE grouped launch."""
        l1 = []
        for pfx, gin in gin_vals:
            w0 = store.pvals[f"{pfx}.gate.0.weight"]
            h = plan.val(w0.data.shape[0], act=L.ACT_RELU, name=pfx + ".h", pad_k=True)
            l1.append(dict(x=gin, W=w0, b=store.pvals[f"{pfx}.gate.0.bias"], out=h))
        plan.add(E.LinearGroupOp(l1))
        return [q["out"] for q in l1]

    def _gate_out(self, plan, store, items, fuse):
        """Second layer (Linear + 2 * sigmoid) of gate networks and the products x (.) gate (reference pepnet.py:31-32,
        :72-78): items = (prefix, gate hidden value, x value, product value).  fuse: K7 -- "full": the product leaves the
        GEMM's epilogue (mml_pep_gate_fwd) and its backward rides on the input-gradient launch of the layer that reads it
        (mml_pep_gate_bwd); "fwd": the forward alone, the backward one batched element-wise launch (products no Linear
        layer reads: the gated input, the products in front of the heads); False: one batched element-wise launch each
        way."""
        l2 = []
        for pfx, hg, x, prod in items:
            w2 = store.pvals[f"{pfx}.gate.2.weight"]
            g = plan.val(w2.data.shape[0], act=L.ACT_SIGMOID2, name=pfx + ".g", pad_k=True)
            q = dict(x=hg, W=w2, b=store.pvals[f"{pfx}.gate.2.bias"], out=g)
            if fuse:
                q.update(mul=x, prod=prod)
                if fuse == "fwd":
                    q["prod_bwd"] = "ext"
            l2.append(q)
        plan.add(E.LinearGroupOp(l2))
        if fuse != "full":
            plan.add(E.MulBatchOp([(x, q["out"], prod) for (pfx, hg, x, prod), q in zip(items, l2)],
                                  fwd_fused=(fuse == "fwd")))

    def _build_graph(self, plan, store, x0):
        import os
        Edim = self.embedding_size
        K0 = x0.n
        T, nl = self.num_tasks, len(self.dnn_hidden_units)
        p = self.scene_index[0]  # column offset used as list position (reference pepnet.py:97, :126)
        scene = x0.buf[:, p * Edim:(p + 1) * Edim]
        # K7 fusion needs operands the LDS-DMA GEMM can take as 16-byte vectors (MMLREC_PEP_FUSE=0: the batched
        # element-wise products of round 2)
        # MMLREC_PEP_FUSE=1: K7 -- the products of the hidden PPNet layers leave the gate GEMM's epilogue and their backward
        # rides on the input-gradient launch of the layer that reads them.  OFF by default: measured on Amazon-8 at
        # B = 65 536 it is level with the batched element-wise products (2.33 against 2.34 ms): what the
        # products lose (0.58 -> 0.28 ms) the GEMMs give back -- their epilogues load the extra operands (the LDS-DMA
        # pipeline drains at every tile) and the gate mode only fits 128 x 64 tiles.  Fusing the forward alone is a loss
        # (2.58 ms: the activation derivatives no longer fold into a single consumer).
        # Round 6: ON.  From 8 192 samples the weight-stationary kernel (csrc/gemm_ws.hip) serves these layers: its turn
        # stores 16 bytes per lane row-major, so the extra operands of the product (forward) and of the gate mode
        # (backward) are read and written in the same whole-line pieces as the outputs, with no pipeline to drain; the
        # forward of the gated input rides in its gate GEMM's epilogue as well ("fwd": its backward stays one element-wise
        # launch that also sums the four tasks' gradient parts), and the products in front of the heads are formed inside
        # the head kernel (gated heads).
        # Measured (same box, tools/lab/pep_parts.sh): B = 65 536 1.976 -> 1.757 ms, 8 192 0.592 -> 0.574, 4 096 0.480 -> 0.472
        # (there the tile kernel's K7 epilogue and the gated heads carry it): on at every batch, MMLREC_PEP_FUSE=0 = off.
        env = os.environ.get("MMLREC_PEP_FUSE")
        fuse_on = plan.device.type == "cuda" and env != "0"
        fuse_fwd = fuse_on and os.environ.get("MMLREC_PEP_FUSE_FWD", "1") != "0"

        def can_fuse(n):
            return fuse_on and n % 16 == 0

        def gate_input(src_val, name):
            # (zero-padded rows: K0 + E is rarely a multiple of the GEMM's 16-wide k-step, e.g. 64 + 8)
            v = plan.val(K0 + Edim, needs_grad=False, name=name, pad_k=True)
            # (round 6, built and measured, not the default: with MMLREC_PEP_COPY_AMAX=1 the two copies raise the operand's
            #  magnitude slot themselves -- mml_copy2d_desc.amax_out; the padding columns are zero -- instead of a magnitude
            #  pass over the assembled [B, K0 + E] buffer.  Two 17-20 us launches leave the step and the step does not move
            #  (-3 us, both orders of tools/lab/ab_inproc.py): a copy that ends in 2 048 looks at one slot line costs 25 us
            #  more, one walked by 256 workgroups streams at a third of the rate.  profiles/r06_ab_pep_copy_amax.txt)
            slot = plan.new_amax() if os.environ.get("MMLREC_PEP_COPY_AMAX", "0") == "1" else None
            plan.add(E.CopyColsOp(src_val.buf, v.buf[:, :K0], amax_out=slot))
            plan.add(E.CopyColsOp(scene, v.buf[:, K0:], amax_out=slot))
            if slot is not None:
                v.amax = slot
            return v

        # EPNet: the feature gate on the input
        fh = self._gate_hidden(plan, store, [("feature_gate", gate_input(x0, "epnet_in"))])[0]
        x2 = plan.val(K0, name="gated_input", pad_k=True)
        # (not fused: the gated input feeds the tasks' first PRODUCTS, not a Linear layer whose input-gradient launch
        # could carry its backward)
        self._gate_out(plan, store, [("feature_gate", fh, x0, x2)], fuse="fwd" if (fuse_fwd and can_fuse(K0)) else False)
        gin = gate_input(x2, "ppnet_in")
        # PPNet: the hidden layers of ALL gate networks read the same input: one grouped launch
        ghs = self._gate_hidden(plan, store, [(f"ppn.{t}.gate_layers.{l}", gin) for t in range(T) for l in range(nl + 1)])
        hidden = [x2] * T
        heads = []
        for l in range(nl + 1):
            # the gate products of all tasks of this layer: fused into the gates' output GEMM where the product feeds a
            # Linear layer (l < nl), one batched launch each way in front of the heads
            ok = all(can_fuse(hidden[t].n) for t in range(T))
            fuse = ("full" if l < nl else ("fwd" if fuse_fwd else False)) if ok else False  # (the last product feeds the heads)
            # the last layer: gated heads (round 6) -- the head kernel reads h and its gate, forms the product in registers
            # and writes both gradients; no product buffer, no element-wise launch either way
            gated = (l == nl and fuse_on and os.environ.get("MMLREC_PEP_GATED_HEAD", "1") != "0" and
                     all(E._fast_row_width_ok(hidden[t].n) for t in range(T)))
            if gated:
                l2 = []
                for t in range(T):
                    pfx = f"ppn.{t}.gate_layers.{l}"
                    w2 = store.pvals[f"{pfx}.gate.2.weight"]
                    g = plan.val(w2.data.shape[0], act=L.ACT_SIGMOID2, name=pfx + ".g", pad_k=True)
                    l2.append(dict(x=ghs[t * (nl + 1) + l], W=w2, b=store.pvals[f"{pfx}.gate.2.bias"], out=g))
                plan.add(E.LinearGroupOp(l2))
                heads = [dict(Hin=hidden[t], gate=l2[t]["out"], w=store.pvals[f"ppn.{t}.mlp_layers.{l}.weight"],
                              bias=store.pvals[f"out.{t}.bias"], bias2=store.pvals[f"ppn.{t}.mlp_layers.{l}.bias"])
                         for t in range(T)]
                continue
            hins = [plan.val(hidden[t].n, name=f"ppn.{t}.hin.{l}", pad_k=True) for t in range(T)]
            self._gate_out(plan, store, [(f"ppn.{t}.gate_layers.{l}", ghs[t * (nl + 1) + l], hidden[t], hins[t])
                                         for t in range(T)], fuse=fuse)
            if l < nl:
                probs = [dict(x=hins[t], W=store.pvals[f"ppn.{t}.mlp_layers.{l}.0.weight"],
                              b=store.pvals[f"ppn.{t}.mlp_layers.{l}.0.bias"],
                              out=plan.val(self.dnn_hidden_units[l], act=L.ACT_RELU, name=f"ppn.{t}.h.{l}", pad_k=True))
                         for t in range(T)]
                plan.add(E.LinearGroupOp(probs))
                hidden = [q["out"] for q in probs]
            else:
                heads = [dict(Hin=hins[t], w=store.pvals[f"ppn.{t}.mlp_layers.{l}.weight"],
                              bias=store.pvals[f"out.{t}.bias"], bias2=store.pvals[f"ppn.{t}.mlp_layers.{l}.bias"])
                         for t in range(T)]
        return E.HeadOp(heads)
