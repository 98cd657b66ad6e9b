"""Classifier layer of the AVT model on the HIP GEMM.  ``HipLinear`` has ``torch.nn.Linear``'s constructor and
state_dict (``weight (out,in)``, ``bias``); it replaces the ``_target_: torch.nn.Linear`` of
conf/model/classifier/linear.yaml:3 (models/base_model.py:87-97).  The class dimension is padded to a multiple of 64
inside the arena (zero rows), logits come back as an fp32 view of the valid columns."""
import math

import torch
import torch.nn as nn

from .. import ops
from ..arena import get_arena


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

    def forward(self, x):
        arena = get_arena(self)
        arena.refresh_shadow()
        if torch.is_grad_enabled():
            arena.attach_grads()
        return _LinearFn.apply(self, arena, x, self.weight)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, arena, x, anchor):
        lead = x.shape[:-1]
        xb = x.reshape(-1, m.in_features).to(torch.bfloat16).contiguous()
        w = arena.sh(m.weight, rows=m.out_padded)
        bias = arena.master_padded(arena.name_of[id(m.bias)]) if m.bias is not None else None
        out = ops.linear_fwd(xb, w, bias=bias, out_mode=ops.OUT_F32)          # [R, out_padded] fp32
        ctx.m, ctx.arena, ctx.xb, ctx.lead = m, arena, xb, lead
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
        return None, None, dx.reshape(ctx.lead + (m.in_features,)), None
