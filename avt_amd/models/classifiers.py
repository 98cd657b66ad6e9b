"""Classifier layer of the AVT model on the HIP GEMM.  ``HipLinear`` has ``torch.nn.Linear``'s constructor and
state_dict (``weight (out,in)``, ``bias``); it replaces the ``_target_: torch.nn.Linear`` of
conf/model/classifier/linear.yaml:3 (models/base_model.py:87-97).  The class dimension is padded to a multiple of 64
inside the arena (zero rows), logits come back as an fp32 view of the valid columns.

Two ways in:
  * ``forward(x)``: the plain layer (eval, roll-out, any caller that only wants logits).
  * ``forward_with_loss(x, target, ignore_index)``: classifier AND softmax cross-entropy as ONE autograd node on the C ABI's
    ``avt_linear_softmax_xent_fwd / _bwd`` (the reference runs ``nn.Linear`` inside ``BaseModel`` and ``MultiDimCrossEntropy`` inside
    ``BasicLossAccuracy`` -- models/base_model.py:203-216, loss_fn/multidim_xentropy.py:11-25; the training operator of this
    package hands the targets to the model so that both run here, func/train_eval_ops.py).  The backward writes
    (softmax - onehot) * gloss once, in bf16, in the class-padded layout its three consumers read; the logits stay a
    differentiable output (a gradient that reaches them from elsewhere is added into dlogits before the bf16 cut).
``DeviceDropout`` is the ``nn.Dropout`` in front of the classifiers (models/base_model.py:87-97) on the counter-based device
RNG of the kernels (common.hpp ``drop_keep``): the mask is a pure function of (seed, element index), so backward re-derives it
and a test can restate it on the host.
"""
import itertools
import math

import torch
import torch.nn as nn

from .. import ops
from .. import seeds as _seeds
from ..arena import get_arena

_seed_counter = itertools.count(1)


def _draw_seed():
    return (next(_seed_counter) * 2000003 + 7919 * torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF


def fresh_seed():
    """Seed of one dropout application: mixes the process's torch seed with a call counter (a seeded run repeats).  Inside a captured step
    (avt_amd/seeds.py) an indirect seed whose device slot is refilled from the same counter before every replay."""
    return _seeds.fresh(_draw_seed)


class _DropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed, ctx.dtype = p, seed, x.dtype
        return ops.dropout(x.to(torch.bfloat16).contiguous(), p, seed)

    @staticmethod
    def backward(ctx, g):
        return ops.dropout(g.to(torch.bfloat16).contiguous(), ctx.p, ctx.seed).to(ctx.dtype), None, None


class DeviceDropout(nn.Module):
    """``nn.Dropout(p)`` for GPU rows headed into a bf16 GEMM: returns bf16 (the GEMM's input dtype) in training, the input itself in eval."""
    def __init__(self, p=0.5):
        super().__init__()
        if not 0.0 <= p < 1.0:
            raise ValueError(f'dropout probability has to be in [0, 1), got {p}')
        self.p = float(p)
        self.last_seed = 0                 # seed of the most recent training-mode call (tests restate the mask from it)

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        self.last_seed = fresh_seed()
        return _DropFn.apply(x, self.p, self.last_seed)

    def extra_repr(self):
        return f'p={self.p} (counter-based device RNG)'


class HipLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        assert in_features % 8 == 0
        self.in_features, self.out_features = in_features, out_features
        self.out_padded = (out_features + 63) // 64 * 64
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def avt_padded_numel(self):
        d = {'weight': self.out_padded * self.in_features}
        if self.bias is not None:
            d['bias'] = self.out_padded
        return d

    def _prep(self):
        arena = get_arena(self)
        arena.refresh_shadow()
        if torch.is_grad_enabled():
            arena.attach_grads()
        return arena

    def forward(self, x):
        return _LinearFn.apply(self, self._prep(), x, self.weight)

    def forward_with_loss(self, x, target, ignore_index=-1, drop_p=0.0, drop_seed=0):
        """x (..., in_features), target (...) int64 -> (logits (..., out_features) fp32, loss (...), rank (...)): un-reduced
        cross entropy (0 where target == ignore_index) and the number of logits above the target's (-1 where ignored).
        drop_p > 0: the dropout in front of the classifier runs inside the node too (mask = keep(drop_seed, row * in_features + k);
        backward applies it in the data-gradient GEMM's epilogue)."""
        if x.shape[:-1] != target.shape:
            raise ValueError(f'rows {tuple(x.shape[:-1])} do not line up with targets {tuple(target.shape)}')
        return _LinearXentFn.apply(self, self._prep(), x, target, ignore_index, float(drop_p), int(drop_seed), self.weight)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, arena, x, anchor):
        lead = x.shape[:-1]
        xb = x.reshape(-1, m.in_features).to(torch.bfloat16).contiguous()
        w = arena.sh(m.weight, rows=m.out_padded)
        bias = arena.master_padded(arena.name_of[id(m.bias)]) if m.bias is not None else None
        out = ops.linear_fwd(xb, w, bias=bias, out_mode=ops.OUT_F32)          # [R, out_padded] fp32
        ctx.m, ctx.arena, ctx.xb, ctx.lead, ctx.xdtype = m, arena, xb, lead, x.dtype
        return out[:, :m.out_features].reshape(lead + (m.out_features,))

    @staticmethod
    def backward(ctx, dout):
        ctx.arena.attach_grads()          # .grad views dropped between forward and backward (optimizer.zero_grad())
        m, arena, xb = ctx.m, ctx.arena, ctx.xb
        R = xb.size(0)
        d = ops.pad_cast_to_bf16(dout.reshape(R, m.out_features).float().contiguous(), m.out_padded)   # zero padding columns
        w = arena.sh(m.weight, rows=m.out_padded)
        ops.linear_wgrad(d, xb, arena.gr(m.weight, rows=m.out_padded), rows=m.out_features)
        if m.bias is not None:
            ops.colsum(d, arena.gr(m.bias, rows=m.out_padded))
        dx = ops.linear_dgrad(d, w, out_mode=ops.OUT_F32)
        return None, None, dx.reshape(ctx.lead + (m.in_features,)).to(ctx.xdtype), None


class _LinearXentFn(torch.autograd.Function):
    """logits = x W^T + b -> (loss, rank) in forward; dlogits -> {dW, db, dx} in backward: avt_linear_softmax_xent_fwd / _bwd."""
    @staticmethod
    def forward(ctx, m, arena, x, target, ignore_index, drop_p, drop_seed, anchor):
        ctx.set_materialize_grads(False)
        lead = x.shape[:-1]
        C = m.out_features
        xb = x.reshape(-1, m.in_features).to(torch.bfloat16).contiguous()
        if drop_p > 0.0:
            xb = ops.dropout(xb, drop_p, drop_seed)
        ctx.drop = (drop_p, drop_seed)
        w = arena.sh(m.weight, rows=m.out_padded)
        bias = arena.master_padded(arena.name_of[id(m.bias)]) if m.bias is not None else None
        tg = target.reshape(-1).contiguous().long()
        logits, loss, lse, rank = ops.linear_softmax_xent_fwd(xb, w, bias, tg, C, ignore_index)
        ctx.m, ctx.arena, ctx.lead, ctx.ignore_index, ctx.xdtype = m, arena, lead, ignore_index, x.dtype
        ctx.save_for_backward(logits, tg, lse, xb)
        rank = rank.reshape(lead)
        ctx.mark_non_differentiable(rank)                    # (on the tensor that is actually returned)
        return logits[:, :C].reshape(lead + (C,)), loss.reshape(lead), rank

    @staticmethod
    def backward(ctx, glogits, gloss, _grank):
        logits, tg, lse, xb = ctx.saved_tensors
        m, arena = ctx.m, ctx.arena
        arena.attach_grads()
        R = xb.size(0)
        if gloss is None:
            gloss = torch.zeros(R, device=xb.device, dtype=torch.float32)
        if glogits is not None:
            glogits = glogits.reshape(R, m.out_features).float().contiguous()
        w = arena.sh(m.weight, rows=m.out_padded)
        dx = ops.linear_softmax_xent_bwd(logits, tg, lse, gloss.reshape(-1).float().contiguous(), xb, w, m.out_features,
                                         dw=arena.gr(m.weight, rows=m.out_padded),
                                         dbias=arena.gr(m.bias, rows=m.out_padded) if m.bias is not None else None,
                                         want_dx=ctx.needs_input_grad[2], ignore_index=ctx.ignore_index, glogits=glogits,
                                         dx_drop_p=ctx.drop[0], dx_drop_seed=ctx.drop[1])
        if dx is not None:
            dx = dx.reshape(ctx.lead + (m.in_features,)).to(ctx.xdtype)
        return None, None, dx, None, None, None, None, None
