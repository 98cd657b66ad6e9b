"""Per-frame ViT-*/16 backbone on the HIP kernels, with timm 0.4.12's ``VisionTransformer(num_classes=0)`` parameter
names / shapes (``cls_token, pos_embed, patch_embed.proj, blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2},
norm``) so a timm / reference checkpoint loads unchanged (reference call sites: models/video_classification.py:224,
255-256; checkpoint prefix ``backbone.model`` from expts/01_ek100_avt.txt:3).

The whole backbone is ONE autograd node: forward enqueues the kernel sequence and keeps the activations backward
needs; backward enqueues the reverse sequence and writes parameter gradients straight into the arena's fp32
gradient buffer (fp32 atomics accumulate), so autograd never materialises per-parameter gradient tensors.
Activations are bf16, LayerNorm/softmax statistics fp32.
"""
import torch
import torch.nn as nn

from .. import ops
from ..arena import get_arena


class _Affine(nn.Module):
    """Parameter holder with nn.LayerNorm's state_dict layout."""
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, mlp_ratio=4):
        super().__init__()
        self.norm1 = _Affine(dim)
        self.attn = _Attn(dim)
        self.norm2 = _Affine(dim)
        self.mlp = _Mlp(dim, dim * mlp_ratio)


class _PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=16, stride=16)


class HipViT(nn.Module):
    EPS = 1e-6

    def __init__(self, embed_dim=768, depth=12, num_heads=12, img_size=224):
        super().__init__()
        assert embed_dim == num_heads * 64, 'the attention kernel is specialised for head_dim 64'
        assert img_size % 16 == 0
        self.embed_dim, self.depth, self.num_heads, self.img_size = embed_dim, depth, num_heads, img_size
        self.seq = (img_size // 16) ** 2 + 1
        self.patch_embed = _PatchEmbed(embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.seq, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim) for _ in range(depth)])
        self.norm = _Affine(embed_dim)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.grad_ready_hook = None          # callable(first_param, last_param) fired as backward finishes a segment

    def forward(self, frames):
        """frames fp32 (N, 3, H, W) -> CLS features fp32 (N, D)."""
        arena = get_arena(self)
        arena.refresh_shadow()
        if torch.is_grad_enabled():
            arena.attach_grads()
        keep = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return _ViTFn.apply(self, arena, keep, frames, self.cls_token)   # one parameter stands in for all of them


def _vit_forward(m: HipViT, arena, frames, keep):
    D, H, S = m.embed_dim, m.num_heads, m.seq
    N = frames.size(0)
    M = N * S
    sh = arena.sh
    patches = ops.im2col_patch16(frames)
    R = ops.posres_prep(m.pos_embed, m.cls_token, m.patch_embed.proj.bias, S, D)
    x = ops.gemm(patches, sh(m.patch_embed.proj.weight).view(D, 768), M, D, 768, res=R, res_period=S)
    saved = {'patches': patches if keep else None, 'blocks': []}
    for blk in m.blocks:
        ln1, mean1, rstd1 = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, m.EPS)
        qkv = ops.linear_fwd(ln1, sh(blk.attn.qkv.weight), bias=blk.attn.qkv.bias)
        att, lse = ops.vit_attn_fwd(qkv, N, S, H)
        x1 = ops.linear_fwd(att, sh(blk.attn.proj.weight), bias=blk.attn.proj.bias, res=x)
        ln2, mean2, rstd2 = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, m.EPS)
        pre = torch.empty((M, 4 * D), device=x.device, dtype=torch.bfloat16) if keep else None
        act = ops.linear_fwd(ln2, sh(blk.mlp.fc1.weight), bias=blk.mlp.fc1.bias, act=ops.ACT_GELU_ERF, c2=pre)
        x2 = ops.linear_fwd(act, sh(blk.mlp.fc2.weight), bias=blk.mlp.fc2.bias, res=x1)
        if keep:
            saved['blocks'].append((x, mean1, rstd1, ln1, qkv, att, lse, x1, mean2, rstd2, ln2, pre, act))
        x = x2
    feat, meanf, rstdf = ops.layernorm_fwd(x, m.norm.weight, m.norm.bias, m.EPS, rows=N, ldx=S * D)
    saved['final'] = (x, meanf, rstdf) if keep else None
    return feat, saved


def _vit_backward(m: HipViT, arena, saved, dfeat):
    D, H, S = m.embed_dim, m.num_heads, m.seq
    N = dfeat.size(0)
    M = N * S
    sh, gr = arena.sh, arena.gr
    hook = m.grad_ready_hook
    x, meanf, rstdf = saved['final']
    dx = torch.zeros((M, D), device=dfeat.device, dtype=torch.bfloat16)     # only the CLS rows receive gradient
    last = m.blocks[-1]
    ops.layernorm_bwd(dfeat, x, meanf, rstdf, m.norm.weight, gr(m.norm.weight), gr(m.norm.bias),
                      colsum=gr(last.mlp.fc2.bias), rows=N, ldx=S * D, dx=dx, lddx=S * D)
    if hook:
        hook(m.norm.weight, m.norm.bias)
    for i in range(m.depth - 1, -1, -1):
        blk = m.blocks[i]
        (x, mean1, rstd1, ln1, qkv, att, lse, x1, mean2, rstd2, ln2, pre, act) = saved['blocks'][i]
        saved['blocks'][i] = None
        # x2 = act @ W2^T + b2 + x1          (db2 was accumulated by the producer of dx)
        ops.linear_wgrad(dx, act, gr(blk.mlp.fc2.weight))
        dh = ops.linear_dgrad(dx, sh(blk.mlp.fc2.weight), act=ops.ACT_MUL_AUX, aux=pre, colsum=gr(blk.mlp.fc1.bias))
        del act, pre
        ops.linear_wgrad(dh, ln2, gr(blk.mlp.fc1.weight))
        dln2 = ops.linear_dgrad(dh, sh(blk.mlp.fc1.weight))
        del dh, ln2
        dx1 = ops.layernorm_bwd(dln2, x1, mean2, rstd2, blk.norm2.weight, gr(blk.norm2.weight), gr(blk.norm2.bias),
                                dres=dx, colsum=gr(blk.attn.proj.bias))
        del dln2, x1, dx
        ops.linear_wgrad(dx1, att, gr(blk.attn.proj.weight))
        datt = ops.linear_dgrad(dx1, sh(blk.attn.proj.weight))
        dqkv = ops.vit_attn_bwd(qkv, att, datt, lse, N, S, H, dbias=gr(blk.attn.qkv.bias))
        del datt, att, qkv
        ops.linear_wgrad(dqkv, ln1, gr(blk.attn.qkv.weight))
        dln1 = ops.linear_dgrad(dqkv, sh(blk.attn.qkv.weight))
        del dqkv, ln1
        prev_bias = gr(m.blocks[i - 1].mlp.fc2.bias) if i > 0 else None
        dx = ops.layernorm_bwd(dln1, x, mean1, rstd1, blk.norm1.weight, gr(blk.norm1.weight), gr(blk.norm1.bias),
                               dres=dx1, colsum=prev_bias)
        del dln1, dx1, x
        if hook:
            hook(blk.norm1.weight, blk.mlp.fc2.bias)
    ops.linear_wgrad(dx, saved['patches'], gr(m.patch_embed.proj.weight).view(D, 768))
    ops.patch_embed_bwd_reduce(dx, gr(m.pos_embed).view(-1), gr(m.cls_token).view(-1), gr(m.patch_embed.proj.bias), N, S, D)
    if hook:
        hook(m.patch_embed.proj.weight, m.pos_embed)


class _ViTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, arena, keep, frames, anchor):
        feat, saved = _vit_forward(module, arena, frames.float().contiguous(), keep=keep)
        ctx.module, ctx.arena, ctx.saved = module, arena, saved
        return feat.float()

    @staticmethod
    def backward(ctx, dfeat):
        _vit_backward(ctx.module, ctx.arena, ctx.saved, dfeat.to(torch.bfloat16).contiguous())
        ctx.saved = None
        return None, None, None, None, None
