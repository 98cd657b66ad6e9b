"""AVT-h: the causal GPT-2 temporal head (reference models/future_prediction.py:51-258) on the HIP kernels.

Same constructor arguments, ``forward(feats, target_shape) -> (past, future, losses, endpoints)`` contract and
state_dict layout as the reference (``encoder.weight (Dh,in)``, ``decoder.weight (in,Dh)``, ``gpt_model.wpe.weight``,
``gpt_model.h.{i}.{ln_1,ln_2}``, ``attn.{c_attn,c_proj}``, ``mlp.{c_fc,c_proj}`` with HF Conv1D (in,out) weights,
``gpt_model.ln_f``).  Scope: the non-quantised path.  Training uses ``output_len == 1`` (the configuration of every AVT
experiment); ``output_len > 1`` is the forward-only roll-out with a KV cache (reference :168-202, eval / no_grad only);
the k-means variants raise NotImplementedError.

encoder -> +wpe, embd-dropout -> n_layer x {LN, c_attn, causal attention (+attn dropout), c_proj (+resid dropout)
+ residual, LN, c_fc + gelu_new, c_proj (+dropout) + residual} -> ln_f -> decoder is ONE autograd node; the
slicing / concatenation that builds ``past`` / ``future`` / the ``feat`` loss stays in torch on (B,T,C) tensors.
"""
import itertools

import torch
import torch.nn as nn

from .. import ops
from .. import seeds as _seeds
from ..arena import get_arena
from ..config import instantiate


class Identity(nn.Module):
    """reference models/future_prediction.py:17-29"""
    def __init__(self, in_features):
        super().__init__()
        self.in_features = in_features

    def forward(self, feats, target_shape=None):
        del target_shape
        return feats, feats, {}, {}

    @property
    def output_dim(self):
        return self.in_features


class _Conv1D(nn.Module):
    """Parameter holder with HF Conv1D's layout: weight (in, out), bias (out)."""
    def __init__(self, nf, nx):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(nx, nf).normal_(std=0.02))
        self.bias = nn.Parameter(torch.zeros(nf))


class _LN(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _GPTAttn(nn.Module):
    def __init__(self, e):
        super().__init__()
        self.c_attn = _Conv1D(3 * e, e)
        self.c_proj = _Conv1D(e, e)


class _GPTMlp(nn.Module):
    def __init__(self, e, inner):
        super().__init__()
        self.c_fc = _Conv1D(inner, e)
        self.c_proj = _Conv1D(e, inner)


class _GPTBlock(nn.Module):
    def __init__(self, e, inner):
        super().__init__()
        self.ln_1 = _LN(e)
        self.attn = _GPTAttn(e)
        self.ln_2 = _LN(e)
        self.mlp = _GPTMlp(e, inner)


class _GPT2(nn.Module):
    def __init__(self, n_embd, n_layer, n_positions, n_inner):
        super().__init__()
        self.wpe = nn.Embedding(n_positions, n_embd)
        nn.init.normal_(self.wpe.weight, std=0.02)
        self.h = nn.ModuleList([_GPTBlock(n_embd, n_inner) for _ in range(n_layer)])
        self.ln_f = _LN(n_embd)


class AVTh(nn.Module):
    _seed_counter = itertools.count(1)

    @staticmethod
    def _draw_seed():
        return (next(AVTh._seed_counter) * 1000003 + torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF

    def __init__(self, in_features: int, output_len: int = -1, output_len_eval: int = -1, avg_last_n: int = -1,
                 inter_dim: int = 768, future_pred_loss=None, return_past_too: bool = False, drop_last_n: int = 0,
                 quantize_before_rollout: bool = False, assign_to_centroids: str = None,
                 num_cluster_centers: int = 50000, freeze_encoder_decoder: bool = False, **kwargs):
        super().__init__()
        if in_features == 1 or assign_to_centroids or quantize_before_rollout:
            raise NotImplementedError('quantised / k-means AVT-h variants are outside the accelerated path')
        if drop_last_n != 0:
            raise NotImplementedError('drop_last_n is a reference debugging switch; not supported')
        kwargs = dict(kwargs)
        kwargs.pop('future_pred_loss_wt', None)           # rides along in the HF config upstream
        self.n_head = kwargs.pop('n_head', 12)
        n_layer = kwargs.pop('n_layer', 12)
        n_positions = kwargs.pop('n_positions', 1024)
        n_inner = kwargs.pop('n_inner', None) or 4 * inter_dim
        self.ln_eps = kwargs.pop('layer_norm_epsilon', 1e-5)
        self.embd_pdrop = kwargs.pop('embd_pdrop', 0.1)
        self.attn_pdrop = kwargs.pop('attn_pdrop', 0.1)
        self.resid_pdrop = kwargs.pop('resid_pdrop', 0.1)
        act = kwargs.pop('activation_function', 'gelu_new')
        assert act == 'gelu_new', 'only the GPT-2 default activation is implemented'
        assert inter_dim % self.n_head == 0 and (inter_dim // self.n_head) % 8 == 0
        self.encoder = nn.Linear(in_features, inter_dim, bias=False)
        self.decoder = nn.Linear(inter_dim, in_features, bias=False)
        if freeze_encoder_decoder:
            raise NotImplementedError('freeze_encoder_decoder is not supported')
        self.gpt_model = _GPT2(inter_dim, n_layer, n_positions, n_inner)
        self.output_len, self.output_len_eval = output_len, output_len_eval
        self.avg_last_n, self.inter_dim, self.in_features = avg_last_n, inter_dim, in_features
        self.future_pred_loss = instantiate(future_pred_loss, reduction='none') if future_pred_loss is not None else None
        self.return_past_too = return_past_too
        self.grad_ready_hook = None

    @property
    def output_dim(self):
        return self.in_features

    def _decode_all(self, feats, extra=None, group=None):
        """(B, T, C) fp32 -> decoder(GPT-2(encoder(feats))) (B, T, C) fp32 through one fused autograd node.
        extra (B, k, E): input embeddings appended behind the encoded frames (the fed-back hidden states of a roll-out WITH gradients);
        then returns (decoded (B, T + k, C), last hidden state of the newest token (B, E))."""
        arena = get_arena(self)
        arena.refresh_shadow()
        if torch.is_grad_enabled():
            arena.attach_grads()
        keep = torch.is_grad_enabled()
        # dropout masks are a pure function of (seed, element index): the seed mixes the process's torch seed (torch.manual_seed)
        # with a call counter, so a seeded run repeats and differently seeded runs differ
        seed = _seeds.fresh(AVTh._draw_seed) if self.training else 0          # (an indirect seed inside a captured step: avt_amd/seeds.py)
        dec, last = _HeadFn.apply(self, arena, keep, self.training, seed, feats, self.encoder.weight, extra, group or _NodeGroup(1))
        return dec if (extra is None and group is None) else (dec, last)

    def _rollout(self, feats, output_len):
        arena = get_arena(self)
        arena.refresh_shadow()
        return _head_rollout(self, arena, feats.float(), output_len)

    def _rollout_with_grad(self, feats, output_len):
        """Roll-out WITH gradients (reference :168-202 in training mode, e.g. a (B, Z, C) target_shape or output_len > 1 in the config): step i
        re-runs the differentiable head node on the T observed frames + the i fed-back hidden states instead of attending over a key / value
        cache -- the same maths for a causal model (oracle/avt_oracle.py restates it the same way), every step's gradients accumulate into the
        same arena ranges, and the segment is reported to the gradient exchange once, by the LAST of the nodes to run backward (_NodeGroup).
        With dropout on, each step draws fresh masks for every position (the reference's cache keeps the masks of the step that first computed
        a position): the same distribution, not the same stream -- parity tests run it with dropout off, as everywhere."""
        group = _NodeGroup(output_len)
        extra, decs = None, []
        for step in range(output_len):
            dec, last = self._decode_all(feats, extra=extra, group=group)                # dec (B, T + step, C)
            decs.append(dec if step == 0 else dec[:, -1:])
            if step + 1 < output_len:
                extra = last.unsqueeze(1) if extra is None else torch.cat([extra, last.unsqueeze(1)], dim=1)
        return torch.cat(decs, dim=1)

    def forward(self, feats, target_shape):
        addl_endpoints = {}
        if feats.ndim == 2:
            feats = feats.unsqueeze(1)
        if len(target_shape) == 3:
            output_len = target_shape[1]
        elif self.training or self.output_len_eval < 0:
            output_len = self.output_len
        else:
            output_len = self.output_len_eval
        orig_len = feats.size(1)
        if output_len == 1:
            all_outputs = self._decode_all(feats)                                # reference :163-203, one GPT-2 call
        elif output_len > 1:
            if torch.is_grad_enabled():
                if output_len + orig_len - 1 > self.gpt_model.wpe.weight.size(0):
                    raise ValueError('roll-out longer than n_positions')
                all_outputs = self._rollout_with_grad(feats, output_len)         # reference :168-202 with gradients: no cache, same maths
            else:
                all_outputs = self._rollout(feats, output_len)                   # reference :168-202 with the KV cache
        else:
            raise NotImplementedError('output_len <= 0 (no GPT-2 call) is not a configuration of the AVT experiments')
        losses = {}
        if self.future_pred_loss is not None:                                    # reference :205-215
            n = min(feats.size(1), all_outputs.size(1))
            if isinstance(self.future_pred_loss, nn.MSELoss) and feats.is_cuda:
                losses = {'feat': _MseShiftFn.apply(all_outputs[:, :n].contiguous(), feats[:, :n].float().contiguous())}
            else:
                losses = {'feat': self.future_pred_loss(all_outputs[:, :n - 1], feats[:, 1:n])}
        prev = feats
        if self.return_past_too:                                                 # reference :232-240
            final = torch.cat((prev, all_outputs[:, orig_len - 1:, :]), dim=1)
        elif output_len > 0:
            final = all_outputs[:, -output_len:]
        else:
            final = all_outputs
        if self.avg_last_n > 0:                                                  # reference :241-242
            final = torch.mean(final[:, -self.avg_last_n:, :], dim=1)
        updated_past = torch.cat([prev[:, :1, :], all_outputs[:, :(orig_len - 1)]], dim=1)   # reference :249-250
        return updated_past, final, losses, addl_endpoints


class _NodeGroup:
    """The head nodes of ONE AVTh.forward (1, or ``output_len`` for a roll-out with gradients): the backward that brings ``left`` to zero
    reports the finished gradient segment."""
    def __init__(self, n):
        self.left = n


class _MseShiftFn(torch.autograd.Function):
    """``MSELoss(reduction='none')(decoded[:, :T-1], feats[:, 1:T])`` (reference :207-215) as one kernel each way."""
    @staticmethod
    def forward(ctx, dec, x):
        ctx.save_for_backward(dec, x)
        return ops.mse_shift_fwd(dec, x)

    @staticmethod
    def backward(ctx, g):
        dec, x = ctx.saved_tensors
        return ops.mse_shift_bwd(dec, x, g.float().contiguous())


@torch.no_grad()
def _head_rollout(m: AVTh, arena, x, output_len):
    """Forward-only roll-out (reference :168-202): the first GPT-2 call sees the T observed frames; every further call feeds
    the last hidden state (ln_f output) of the newest token back as the next input embedding at position T + step - 1 and
    attends over the cached keys / values of all earlier tokens.  Returns decoder(all hidden states) (B, T+output_len-1, C)."""
    B, T, C = x.shape
    E, H = m.inter_dim, m.n_head
    hd = E // H
    sh = arena.sh
    g = m.gpt_model
    tmax = T + output_len - 1
    assert tmax <= g.wpe.weight.size(0), 'roll-out longer than n_positions'
    xb = x.reshape(B * T, C).to(torch.bfloat16).contiguous()
    enc = ops.linear_fwd(xb, sh(m.encoder.weight))
    h = ops.embed_pos_fwd(enc, g.wpe.weight, B, T, E, 0.0, 0)
    caches = []
    for blk in g.h:
        l1, _, _ = ops.layernorm_fwd(h, blk.ln_1.weight, blk.ln_1.bias, m.ln_eps, save_stats=False)
        qkv = ops.conv1d_fwd(l1, sh(blk.attn.c_attn.weight), bias=blk.attn.c_attn.bias)
        kc = torch.zeros((B, tmax, E), device=x.device, dtype=torch.bfloat16)
        vc = torch.zeros((B, tmax, E), device=x.device, dtype=torch.bfloat16)
        q3 = qkv.view(B, T, 3 * E)
        kc[:, :T].copy_(q3[:, :, E:2 * E])
        vc[:, :T].copy_(q3[:, :, 2 * E:])
        caches.append((kc, vc))
        att, _ = ops.causal_attn_fwd(qkv, B, T, H, hd, 0.0, 0)
        h1 = ops.conv1d_fwd(att, sh(blk.attn.c_proj.weight), bias=blk.attn.c_proj.bias, res=h)
        l2, _, _ = ops.layernorm_fwd(h1, blk.ln_2.weight, blk.ln_2.bias, m.ln_eps, save_stats=False)
        a = ops.conv1d_fwd(l2, sh(blk.mlp.c_fc.weight), bias=blk.mlp.c_fc.bias, act=ops.ACT_GELU_TANH)
        h = ops.conv1d_fwd(a, sh(blk.mlp.c_proj.weight), bias=blk.mlp.c_proj.bias, res=h1)
    lf, _, _ = ops.layernorm_fwd(h, g.ln_f.weight, g.ln_f.bias, m.ln_eps, save_stats=False)
    outs = [ops.linear_fwd(lf, sh(m.decoder.weight), out_mode=ops.OUT_F32).view(B, T, C)]
    feats = lf.view(B, T, E)[:, -1].contiguous()                     # last hidden state of the newest token
    for step in range(1, output_len):
        pos = T + step - 1
        h = ops.embed_pos_fwd(feats, g.wpe.weight[pos:pos + 1], B, 1, E, 0.0, 0)
        for blk, (kc, vc) in zip(g.h, caches):
            l1, _, _ = ops.layernorm_fwd(h, blk.ln_1.weight, blk.ln_1.bias, m.ln_eps, save_stats=False)
            qkv = ops.conv1d_fwd(l1, sh(blk.attn.c_attn.weight), bias=blk.attn.c_attn.bias)
            att = ops.causal_attn_decode(qkv, kc, vc, B, H, hd, pos)
            h1 = ops.conv1d_fwd(att, sh(blk.attn.c_proj.weight), bias=blk.attn.c_proj.bias, res=h)
            l2, _, _ = ops.layernorm_fwd(h1, blk.ln_2.weight, blk.ln_2.bias, m.ln_eps, save_stats=False)
            a = ops.conv1d_fwd(l2, sh(blk.mlp.c_fc.weight), bias=blk.mlp.c_fc.bias, act=ops.ACT_GELU_TANH)
            h = ops.conv1d_fwd(a, sh(blk.mlp.c_proj.weight), bias=blk.mlp.c_proj.bias, res=h1)
        feats, _, _ = ops.layernorm_fwd(h, g.ln_f.weight, g.ln_f.bias, m.ln_eps, save_stats=False)
        outs.append(ops.linear_fwd(feats, sh(m.decoder.weight), out_mode=ops.OUT_F32).view(B, 1, C))
    return torch.cat(outs, dim=1)


def _head_forward(m: AVTh, arena, x, keep, training, seed, extra=None):
    B, T0, C = x.shape
    E, H = m.inter_dim, m.n_head
    hd = E // H
    sh = arena.sh
    pe = m.embd_pdrop if training else 0.0
    pa = m.attn_pdrop if training else 0.0
    pr = m.resid_pdrop if training else 0.0
    xb = x.reshape(B * T0, C).to(torch.bfloat16).contiguous()
    enc = ops.linear_fwd(xb, sh(m.encoder.weight))
    T = T0
    if extra is not None:                                      # fed-back hidden states behind the encoded frames (roll-out with gradients)
        T = T0 + extra.size(1)
        emb = torch.empty((B, T, E), device=x.device, dtype=torch.bfloat16)
        emb[:, :T0].copy_(enc.view(B, T0, E))
        emb[:, T0:].copy_(extra)
        enc = emb.view(B * T, E)
    h = ops.embed_pos_fwd(enc, m.gpt_model.wpe.weight, B, T, E, pe, seed)
    saved = {'xb': xb, 'layers': [], 'p': (pe, pa, pr), 'seed': seed, 'T0': T0}
    for li, blk in enumerate(m.gpt_model.h):
        s0 = seed + 16 * (li + 1)
        l1, m1, r1 = ops.layernorm_fwd(h, blk.ln_1.weight, blk.ln_1.bias, m.ln_eps)
        qkv = ops.conv1d_fwd(l1, sh(blk.attn.c_attn.weight), bias=blk.attn.c_attn.bias)
        att, probs = ops.causal_attn_fwd(qkv, B, T, H, hd, pa, s0 + 1)
        h1 = ops.conv1d_fwd(att, sh(blk.attn.c_proj.weight), bias=blk.attn.c_proj.bias, drop_p=pr, seed=s0 + 2, res=h)
        l2, m2, r2 = ops.layernorm_fwd(h1, blk.ln_2.weight, blk.ln_2.bias, m.ln_eps)
        pre = torch.empty((B * T, blk.mlp.c_fc.weight.size(1)), device=x.device, dtype=torch.bfloat16)
        a = ops.conv1d_fwd(l2, sh(blk.mlp.c_fc.weight), bias=blk.mlp.c_fc.bias, act=ops.ACT_GELU_TANH, c2=pre)
        h2 = ops.conv1d_fwd(a, sh(blk.mlp.c_proj.weight), bias=blk.mlp.c_proj.bias, drop_p=pr, seed=s0 + 3, res=h1)
        if keep:
            saved['layers'].append((h, m1, r1, l1, qkv, att, probs, h1, m2, r2, l2, pre, a))
        h = h2
    lf, mf, rf = ops.layernorm_fwd(h, m.gpt_model.ln_f.weight, m.gpt_model.ln_f.bias, m.ln_eps)
    dec = ops.linear_fwd(lf, sh(m.decoder.weight), out_mode=ops.OUT_F32)
    saved['final'] = (h, mf, rf, lf)
    return dec.view(B, T, C), lf.view(B, T, E)[:, -1].float(), (saved if keep else None)


def _head_backward(m: AVTh, arena, saved, ddec, dlast=None, fire_hook=True):
    """ddec (B, T, C): gradient of the decoded features; dlast (B, E) | None: gradient of the newest token's last hidden state (roll-out with
    gradients).  Returns (d feats (B, T0, C), d extra (B, T - T0, E) | None)."""
    B, T, C = ddec.shape
    E, H = m.inter_dim, m.n_head
    hd = E // H
    sh, gr = arena.sh, arena.gr
    pe, pa, pr = saved['p']
    seed = saved['seed']
    hook = m.grad_ready_hook
    dd = ddec.reshape(B * T, C).to(torch.bfloat16).contiguous()
    h, mf, rf, lf = saved['final']
    ops.linear_wgrad(dd, lf, gr(m.decoder.weight))
    dlf = ops.linear_dgrad(dd, sh(m.decoder.weight))
    if dlast is not None:                                      # + the gradient that arrives through the next roll-out step's input
        ops.add_rows(dlf.view(B, T * E)[:, (T - 1) * E:], dlast.to(torch.bfloat16).contiguous())
    g = m.gpt_model
    dh = ops.layernorm_bwd(dlf, h, mf, rf, g.ln_f.weight, gr(g.ln_f.weight), gr(g.ln_f.bias))
    for li in range(len(g.h) - 1, -1, -1):
        blk = g.h[li]
        s0 = seed + 16 * (li + 1)
        (h, m1, r1, l1, qkv, att, probs, h1, m2, r2, l2, pre, a) = saved['layers'][li]
        saved['layers'][li] = None
        dy = ops.dropout(dh, pr, s0 + 3) if pr > 0 else dh
        ops.colsum(dy, gr(blk.mlp.c_proj.bias))
        ops.conv1d_wgrad(a, dy, gr(blk.mlp.c_proj.weight))
        dpre = ops.conv1d_dgrad(dy, sh(blk.mlp.c_proj.weight), act=ops.ACT_MUL_AUX, aux=pre, colsum=gr(blk.mlp.c_fc.bias))
        ops.conv1d_wgrad(l2, dpre, gr(blk.mlp.c_fc.weight))
        dl2 = ops.conv1d_dgrad(dpre, sh(blk.mlp.c_fc.weight))
        dh1 = ops.layernorm_bwd(dl2, h1, m2, r2, blk.ln_2.weight, gr(blk.ln_2.weight), gr(blk.ln_2.bias), dres=dh)
        dy = ops.dropout(dh1, pr, s0 + 2) if pr > 0 else dh1
        ops.colsum(dy, gr(blk.attn.c_proj.bias))
        ops.conv1d_wgrad(att, dy, gr(blk.attn.c_proj.weight))
        datt = ops.conv1d_dgrad(dy, sh(blk.attn.c_proj.weight))
        dqkv = ops.causal_attn_bwd(qkv, probs, datt, B, T, H, hd, pa, s0 + 1)
        ops.colsum(dqkv, gr(blk.attn.c_attn.bias))
        ops.conv1d_wgrad(l1, dqkv, gr(blk.attn.c_attn.weight))
        dl1 = ops.conv1d_dgrad(dqkv, sh(blk.attn.c_attn.weight))
        dh = ops.layernorm_bwd(dl1, h, m1, r1, blk.ln_1.weight, gr(blk.ln_1.weight), gr(blk.ln_1.bias), dres=dh1)
    denc = ops.embed_pos_bwd(dh, gr(g.wpe.weight), B, T, E, pe, seed)
    T0, dextra = saved['T0'], None
    if T0 != T:
        d3 = denc.view(B, T, E)
        dextra = d3[:, T0:].float()
        denc = d3[:, :T0].contiguous().view(B * T0, E)
    ops.linear_wgrad(denc, saved['xb'], gr(m.encoder.weight))
    dx = ops.linear_dgrad(denc, sh(m.encoder.weight), out_mode=ops.OUT_F32)
    if hook and fire_hook:
        hook(m.encoder.weight, g.ln_f.bias)
    return dx.view(B, T0, C), dextra


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, arena, keep, training, seed, feats, anchor, extra, group):
        dec, last, saved = _head_forward(module, arena, feats.float(), keep, training, seed, extra)
        ctx.module, ctx.arena, ctx.saved, ctx.group = module, arena, saved, group
        ctx.set_materialize_grads(False)  # an unused output (the last hidden state outside a roll-out) hands backward None, not zeros
        return dec, last

    @staticmethod
    def backward(ctx, ddec, dlast):
        ctx.arena.attach_grads()          # .grad views dropped between forward and backward (optimizer.zero_grad())
        saved = ctx.saved
        if ddec is None:                  # (only the fed-back hidden state of this roll-out step was used)
            h = saved['final'][0]
            B = dlast.size(0)
            ddec = torch.zeros((B, h.size(0) // B, ctx.module.in_features), device=dlast.device, dtype=torch.float32)
        ctx.group.left -= 1
        dx, dextra = _head_backward(ctx.module, ctx.arena, saved, ddec, dlast, fire_hook=ctx.group.left <= 0)
        ctx.saved = None
        return None, None, None, None, None, dx, None, dextra, None
