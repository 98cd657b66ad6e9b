"""Temporal aggregators (reference models/temporal_aggregation.py).  ``Identity`` (:21-31) is what every AVT experiment selects
(expts/01_ek100_avt.txt:12); ``Mean`` (:33-47) and the ``Transformer`` encoder aggregator (:73-147) share the head's
kernels; ``RULSTMAggregation`` needs the external RULSTM code and is out of scope."""
import torch
import torch.nn as nn


class Identity(nn.Identity):
    def __init__(self, in_features):
        super().__init__()
        self.in_features = in_features

    def forward(self, *args, **kwargs):
        return super().forward(*args, **kwargs), {}

    @property
    def output_dim(self):
        return self.in_features


class Mean(nn.Module):
    """reference models/temporal_aggregation.py:33-47 (the default group option of conf/config.yaml; a T-way mean)."""
    def __init__(self, in_features):
        super().__init__()
        self.in_features = in_features

    def forward(self, feats):
        return torch.mean(feats, dim=1), {}

    @property
    def output_dim(self):
        return self.in_features


# ---- Transformer-encoder aggregator (reference :73-147) on the head kernels ------------------------------------
class _MHAParams(nn.Module):
    """torch.nn.MultiheadAttention's parameter layout: in_proj_weight (3E, E), in_proj_bias (3E), out_proj Linear."""
    def __init__(self, e):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * e, e))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * e))
        self.out_proj = nn.Linear(e, e)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)


class _Affine(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _EncLayerParams(nn.Module):
    """torch.nn.TransformerEncoderLayer's state_dict layout (self_attn, linear1, linear2, norm1, norm2)."""
    def __init__(self, e, ff):
        super().__init__()
        self.self_attn = _MHAParams(e)
        self.linear1 = nn.Linear(e, ff)
        self.linear2 = nn.Linear(ff, e)
        self.norm1 = _Affine(e)
        self.norm2 = _Affine(e)


class _EncoderParams(nn.Module):
    def __init__(self, e, ff, nlayers):
        super().__init__()
        self.layers = nn.ModuleList([_EncLayerParams(e, ff) for _ in range(nlayers)])
        self.norm = _Affine(e)


class _PosEnc(nn.Module):
    """reference :50-70: sinusoid table as the buffer ``pe`` (max_len, 1, d_model)."""
    def __init__(self, d_model, max_len=1000):
        super().__init__()
        import math
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0).transpose(0, 1).contiguous())


class Transformer(nn.Module):
    """``nn.Linear`` down-projection + sinusoid positions + ``nn.TransformerEncoder`` (post-norm layers, ReLU, dim_feedforward
    2048, dropout 0.1, final LayerNorm) + mean / last over time, with torch's parameter names (``downproject``,
    ``transformer_encoder.layers.{i}.self_attn.in_proj_weight`` ...), as ONE autograd node over the HIP kernels: bf16 MFMA
    GEMMs with fused bias / dropout / residual, the head attention kernel with the causal mask switched off, fused LayerNorm.
    The cloze (masked-feature) auxiliary loss is not on any AVT experiment's path and raises."""
    FF, EPS, PDROP = 2048, 1e-5, 0.1

    def __init__(self, in_features, inter_rep=512, nheads=8, nlayers=6, agg_style='mean', cloze_loss_ratio=0.0, cloze_loss_wt=0.0):
        super().__init__()
        if cloze_loss_ratio > 0:
            raise NotImplementedError('cloze_loss_ratio > 0 (masked-feature auxiliary loss) is outside the accelerated path')
        if agg_style not in ('mean', 'last'):
            raise NotImplementedError(f'Unknown agg style {agg_style}')
        assert in_features % 8 == 0 and inter_rep % nheads == 0 and (inter_rep // nheads) % 8 == 0
        self.in_features, self.inter_rep, self.nheads, self.agg_style = in_features, inter_rep, nheads, agg_style
        self.downproject = nn.Linear(in_features, inter_rep)
        self.pos_encoder = _PosEnc(inter_rep, max_len=1000)
        self.transformer_encoder = _EncoderParams(inter_rep, self.FF, nlayers)
        self.grad_ready_hook = None

    @property
    def output_dim(self):
        return self.inter_rep

    def forward(self, feats):
        from ..arena import get_arena
        arena = get_arena(self)
        arena.refresh_shadow()
        if torch.is_grad_enabled():
            arena.attach_grads()
        seed = _seeds.fresh(Transformer._draw_seed) if self.training else 0
        enc = _TxFn.apply(self, arena, torch.is_grad_enabled(), self.training, seed, feats, self.downproject.weight)
        return (enc.mean(dim=1) if self.agg_style == 'mean' else enc[:, -1]), {}


import itertools as _it          # noqa: E402
Transformer._seed_counter = _it.count(1)
Transformer._draw_seed = staticmethod(lambda: (next(Transformer._seed_counter) * 1000033 + torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF)
from .. import seeds as _seeds   # noqa: E402


def _tx_forward(m, arena, feats, keep, training, seed):
    from .. import ops
    B, T, C = feats.shape
    E, H = m.inter_rep, m.nheads
    hd = E // H
    sh = arena.sh
    p = m.PDROP if training else 0.0
    xb = feats.reshape(B * T, C).to(torch.bfloat16).contiguous()
    x0 = ops.linear_fwd(xb, sh(m.downproject.weight), bias=m.downproject.bias)
    x = ops.embed_pos_fwd(x0, m.pos_encoder.pe.view(-1, E), B, T, E, p, seed)
    saved = {'xb': xb, 'layers': [], 'p': p, 'seed': seed}
    for li, lay in enumerate(m.transformer_encoder.layers):
        s0 = seed + 16 * (li + 1)
        a = lay.self_attn
        qkv = ops.linear_fwd(x, sh(a.in_proj_weight), bias=a.in_proj_bias)
        att, probs = ops.causal_attn_fwd(qkv, B, T, H, hd, p, s0 + 1, causal=False)
        s1 = ops.linear_fwd(att, sh(a.out_proj.weight), bias=a.out_proj.bias, drop_p=p, seed=s0 + 2, res=x)
        x1, m1, r1 = ops.layernorm_fwd(s1, lay.norm1.weight, lay.norm1.bias, m.EPS)
        h = ops.linear_fwd(x1, sh(lay.linear1.weight), bias=lay.linear1.bias)
        act, mask = ops.relu(h)
        ad = ops.dropout(act, p, s0 + 3) if p > 0 else act
        s2 = ops.linear_fwd(ad, sh(lay.linear2.weight), bias=lay.linear2.bias, drop_p=p, seed=s0 + 4, res=x1)
        x2, m2, r2 = ops.layernorm_fwd(s2, lay.norm2.weight, lay.norm2.bias, m.EPS)
        if keep:
            saved['layers'].append((x, qkv, att, probs, s1, m1, r1, x1, mask, ad, s2, m2, r2))
        x = x2
    nf = m.transformer_encoder.norm
    y, mf, rf = ops.layernorm_fwd(x, nf.weight, nf.bias, m.EPS)
    saved['final'] = (x, mf, rf)
    return y.float().view(B, T, E), (saved if keep else None)


def _tx_backward(m, arena, saved, dy):
    from .. import ops
    B, T, E = dy.shape
    H = m.nheads
    hd = E // H
    sh, gr = arena.sh, arena.gr
    p, seed = saved['p'], saved['seed']
    x, mf, rf = saved['final']
    nf = m.transformer_encoder.norm
    dx = ops.layernorm_bwd(dy.reshape(B * T, E).to(torch.bfloat16).contiguous(), x, mf, rf, nf.weight, gr(nf.weight), gr(nf.bias))
    for li in range(len(m.transformer_encoder.layers) - 1, -1, -1):
        lay = m.transformer_encoder.layers[li]
        a = lay.self_attn
        s0 = seed + 16 * (li + 1)
        (x, qkv, att, probs, s1, m1, r1, x1, mask, ad, s2, m2, r2) = saved['layers'][li]
        saved['layers'][li] = None
        ds2 = ops.layernorm_bwd(dx, s2, m2, r2, lay.norm2.weight, gr(lay.norm2.weight), gr(lay.norm2.bias))
        dz = ops.dropout(ds2, p, s0 + 4) if p > 0 else ds2
        ops.colsum(dz, gr(lay.linear2.bias))
        ops.linear_wgrad(dz, ad, gr(lay.linear2.weight))
        dh = ops.linear_dgrad(dz, sh(lay.linear2.weight), act=ops.ACT_MUL_AUX, aux=mask)        # x ReLU'(h)
        if p > 0:
            dh = ops.dropout(dh, p, s0 + 3)                                                     # the inner dropout's mask (commutes)
        ops.colsum(dh, gr(lay.linear1.bias))
        ops.linear_wgrad(dh, x1, gr(lay.linear1.weight))
        dx1 = ops.linear_dgrad(dh, sh(lay.linear1.weight), res=ds2)                             # + the residual branch
        ds1 = ops.layernorm_bwd(dx1, s1, m1, r1, lay.norm1.weight, gr(lay.norm1.weight), gr(lay.norm1.bias))
        dz = ops.dropout(ds1, p, s0 + 2) if p > 0 else ds1
        ops.colsum(dz, gr(a.out_proj.bias))
        ops.linear_wgrad(dz, att, gr(a.out_proj.weight))
        datt = ops.linear_dgrad(dz, sh(a.out_proj.weight))
        dqkv = ops.causal_attn_bwd(qkv, probs, datt, B, T, H, hd, p, s0 + 1, causal=False)
        ops.colsum(dqkv, gr(a.in_proj_bias))
        ops.linear_wgrad(dqkv, x, gr(a.in_proj_weight))
        dx = ops.linear_dgrad(dqkv, sh(a.in_proj_weight), res=ds1)
    scratch = torch.zeros((T, E), device=dy.device, dtype=torch.float32)        # the sinusoid table is a buffer, not a parameter
    dx0 = ops.embed_pos_bwd(dx, scratch, B, T, E, p, seed)
    ops.colsum(dx0, gr(m.downproject.bias))
    ops.linear_wgrad(dx0, saved['xb'], gr(m.downproject.weight))
    dfe = ops.linear_dgrad(dx0, sh(m.downproject.weight), out_mode=ops.OUT_F32)
    if m.grad_ready_hook:
        m.grad_ready_hook(m.downproject.weight, nf.bias)
    return dfe.view(B, T, -1)


class _TxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, arena, keep, training, seed, feats, anchor):
        y, saved = _tx_forward(module, arena, feats.float(), keep, training, seed)
        ctx.module, ctx.arena, ctx.saved = module, arena, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        ctx.arena.attach_grads()
        dfe = _tx_backward(ctx.module, ctx.arena, ctx.saved, dy)
        ctx.saved = None
        return None, None, None, None, None, dfe, None
