"""Classifier + cross entropy as ONE autograd node on the fused C-ABI operator (``avt_linear_softmax_xent_fwd / _bwd``).

The reference keeps them apart -- ``torch.nn.Linear`` inside ``BaseModel`` (models/base_model.py:203-216), ``MultiDimCrossEntropy``
inside ``BasicLossAccuracy`` (func/train_eval_ops.py:27-44, loss_fn/multidim_xentropy.py:11-25) -- and the drop-in model of this
package mirrors that module boundary, because the logits are a model output.  ``LinearCrossEntropy`` is the same maths for callers
that own both ends (feature rows -> un-reduced loss, target rank, logits): the backward writes (softmax - onehot) * gloss once, in
bf16, in the class-padded layout its three consumers read, instead of an fp32 dlogits tensor that the Linear's backward re-casts.
Parameters are named and shaped like ``torch.nn.Linear``'s (``weight (out, in)``, ``bias``)."""
import torch
import torch.nn as nn

from .. import ops


class _LinearXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, target, ignore_index):
        C, K = weight.shape
        Cpad = (C + 63) // 64 * 64
        xb = x.reshape(-1, K).to(torch.bfloat16).contiguous()
        wb = torch.zeros((Cpad, K), device=x.device, dtype=torch.bfloat16)
        wb[:C] = weight.detach().to(torch.bfloat16)
        bp = None
        if bias is not None:
            bp = torch.zeros(Cpad, device=x.device, dtype=torch.float32)
            bp[:C] = bias.detach()
        tg = target.reshape(-1).contiguous().long()
        logits, loss, lse, rank = ops.linear_softmax_xent_fwd(xb, wb, bp, tg, C, ignore_index)
        ctx.save_for_backward(logits, tg, lse, xb, wb)
        ctx.meta = (C, Cpad, K, ignore_index, bias is not None, x.shape)
        out_logits = logits[:, :C]
        ctx.mark_non_differentiable(rank, out_logits)
        return loss.reshape(target.shape), rank.reshape(target.shape), out_logits.reshape(target.shape + (C,))

    @staticmethod
    def backward(ctx, gloss, _grank, _glogits):
        logits, tg, lse, xb, wb = ctx.saved_tensors
        C, Cpad, K, ignore_index, has_bias, xshape = ctx.meta
        dw = torch.zeros((Cpad, K), device=xb.device, dtype=torch.float32)
        db = torch.zeros(Cpad, device=xb.device, dtype=torch.float32) if has_bias else None
        dx = ops.linear_softmax_xent_bwd(logits, tg, lse, gloss.reshape(-1).float().contiguous(), xb, wb, C, dw=dw, dbias=db,
                                         want_dx=ctx.needs_input_grad[0], ignore_index=ignore_index)
        return (dx.reshape(xshape) if dx is not None else None), dw[:C], (db[:C] if has_bias else None), None, None


class LinearCrossEntropy(nn.Module):
    def __init__(self, in_features, out_features, bias=True, ignore_index=-1):
        super().__init__()
        assert in_features % 8 == 0
        self.weight = nn.Parameter(torch.empty(out_features, in_features).normal_(0, 0.01))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        self.ignore_index = ignore_index

    def forward(self, feats, target):
        """feats (..., in_features), target (...) int64 -> (loss (...), rank (...), logits (..., out_features)); loss is un-reduced,
        0 where target == ignore_index; rank = number of logits above the target's (-1 where ignored)."""
        return _LinearXentFn.apply(feats, self.weight, self.bias, target, self.ignore_index)
