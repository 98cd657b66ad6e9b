"""Cross entropy over arbitrary leading dims (reference loss_fn/multidim_xentropy.py:10-25) on the fused HIP
softmax-cross-entropy kernels.  ``MultiDimCrossEntropy(ignore_index=-1, reduction='none')`` is what
func/train_eval_ops.py:27-44 builds; class weights (``balance_classes``) are not on the AVT path."""
import torch
import torch.nn as nn

from .. import ops


class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        C = logits.size(-1)
        lg = logits.float().contiguous()
        tg = target.contiguous().long()
        loss, lse, rank = ops.xent_fwd(lg, tg, C, ignore_index)
        ctx.save_for_backward(lg, tg, lse)
        ctx.ignore_index, ctx.C = ignore_index, C
        ctx.mark_non_differentiable(rank)
        return loss, rank

    @staticmethod
    def backward(ctx, gloss, _grank):
        lg, tg, lse = ctx.saved_tensors
        ldd = (ctx.C + 7) // 8 * 8
        d = ops.xent_bwd(lg, tg, lse, gloss.float().contiguous(), ctx.C, ldd, ctx.ignore_index)
        return d[:, :ctx.C].float(), None, None


def cross_entropy_with_rank(logits2d, target1d, ignore_index=-1):
    """(loss[R], rank[R]); rank = number of logits strictly above the target's (-1 for ignored rows)."""
    return _XentFn.apply(logits2d, target1d, ignore_index)


class MultiDimCrossEntropy(nn.Module):
    def __init__(self, ignore_index=-100, reduction='mean', weight=None):
        super().__init__()
        if weight is not None:
            raise NotImplementedError('class-weighted cross entropy is outside the accelerated path')
        self.ignore_index, self.reduction = ignore_index, reduction

    def forward_with_rank(self, inp, tgt):
        """(loss, rank): rank[...] = number of logits strictly above the target's (-1 where the target is ignored) -- what
        top-k accuracy needs, from the same pass over the logits (common/utils.py:17-44 would run a second top-k)."""
        assert inp.ndim == tgt.ndim + 1
        assert inp.shape[:-1] == tgt.shape
        loss, rank = cross_entropy_with_rank(inp.reshape(-1, inp.size(-1)), tgt.reshape(-1), self.ignore_index)
        if self.reduction == 'none':
            return loss.reshape(tgt.shape), rank.reshape(tgt.shape)
        valid = (tgt.reshape(-1) != self.ignore_index).sum().clamp(min=1)
        return (loss.sum() / valid if self.reduction == 'mean' else loss.sum()), rank.reshape(tgt.shape)

    def forward(self, inp, tgt):
        return self.forward_with_rank(inp, tgt)[0]
