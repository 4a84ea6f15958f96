"""Stand-alone differentiable wrappers over single C-ABI ops, for code that calls a parameter container directly
(e.g. `DNN.forward`) instead of going through a model's step plan."""
import torch

from . import _lib as L
from . import ops


class _LinearAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act):
        x = x.contiguous()
        out = torch.empty(x.shape[0], W.shape[0], dtype=torch.float32, device=x.device)
        ops.gemm_fwd([dict(A=x, W=W, bias=b, C=out, act=act)])
        ctx.save_for_backward(x, W, out)
        ctx.act, ctx.has_b = act, b is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, W, out = ctx.saved_tensors
        dz = torch.empty_like(out)
        ops.act_bwd(out, dy.contiguous(), dz, ctx.act)
        dx = torch.empty_like(x)
        ops.gemm_dgrad([dict(dA=dx, Y=None, srcs=[(dz, W, 0)])])
        dW = torch.empty_like(W)
        db = torch.empty(W.shape[0], device=W.device) if ctx.has_b else None
        ops.gemm_wgrad([dict(dC=dz, A=x, dW=dW, dbias=db)])
        return dx, dW, db, None


def linear_act(x, W, b=None, act=L.ACT_NONE):
    """act(x @ W^T + b) on the fp32 MFMA GEMM kernels (K3), differentiable."""
    if not x.is_cuda:
        raise L.MMLError("mmlrec_amd.functional needs CUDA(HIP) tensors; there is no CPU fallback")
    return _LinearAct.apply(x.float(), W, b, int(act))
