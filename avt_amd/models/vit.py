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
from ..common.patch_video import PatchVideo


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
    # timm's forward_features returns x[:, 0] after the final norm, so in the LAST block only token 0 of the attention /
    # proj / MLP outputs is ever consumed, in both directions: that block runs its k|v projection on all
    # tokens and everything else on the CLS rows only (10/12 of one block's GEMM work and its attention disappear; results
    # are identical).  False = the generic all-token path for every block (kept for the parity tests).
    cls_only_last_block = True
    # LayerNorm folded into the GEMMs around it (round 5; csrc/lnfold.hip, include/avt_hip.h "LayerNorm folded into the GEMMs"): norm1 -> qkv and
    # norm2 -> fc1 of every all-token block read the residual stream itself; the rows' statistics come out of the epilogue of the GEMM that wrote
    # the stream.  No normalised copy is written or kept: one read + one write of [tokens, D] per LayerNorm less (ln_fwd2_kernel: 6.3 ms of a
    # 256-clip step).  False = a LayerNorm kernel in front of every projection (kept for the parity tests; the CLS-only last block always does that).
    fold_layernorm = True
    # ... from this many token rows on.  Below, the fold loses (late round 5, profiles/r05zh_fold_small_batch.txt, same-box A/B of the whole step: without it +8.4 % at the
    # reference's own 3 clips per GPU, +7 % at 5, +3.8 % at 8, +1 % at 16, equal at 32 = 63040 rows, -0.7 ... -0.9 % from 48 clips on): its own kernels (statistics,
    # weight folding, the folded weight gradients' second pass: 0.7 ms per step) are batch-independent, and under the persistent GEMM's range the folded GELU epilogue
    # of the one-tile kernel reads its table from global memory (100 against 65 us per fc1 forward at 5910 rows).  0 = fold at every size (what the parity tests run).
    fold_min_rows = 60000
    # The GELU derivative saved by fc1 forward is read once, by the fc2 data gradient on the same kernel: where the persistent GEMM takes the shape it is
    # kept in that kernel's fragment-major order (ops.FragTensor; include/avt_hip.h ABI 7) -- no LDS patch round trip for the writer, contiguous requests
    # for the reader: fc1 forward -0.7 %, fc2 data gradient -2.5 % per launch, the step +0.3 % (profiles/r05s_auxfrag.txt).  Same values bit for bit.
    # False = the row-major tensor everywhere (kept for the parity tests).
    frag_gelu_derivative = True

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
        """frames fp32 (N, 3, H, W) -> CLS features fp32 (N, D).  A ``PatchVideo`` (the input pipeline's patch rows) is taken as it is."""
        patch_rows = isinstance(frames, PatchVideo)
        if tuple(frames.shape[-2:]) != (self.img_size, self.img_size) or frames.size(-4 if patch_rows else 1) != 3:
            raise ValueError(f'HipViT was built for 3x{self.img_size}x{self.img_size} frames (pos_embed has {self.seq} '
                             f'positions), got {tuple(frames.shape[1:])}')
        arena = get_arena(self)
        arena.refresh_shadow()
        if torch.is_grad_enabled():
            arena.attach_grads()
        keep = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        return _ViTFn.apply(self, arena, keep, frames.patches if patch_rows else frames, self.cls_token)   # one parameter stands in for all of them


def use_fold(m, M, D, n_full_blocks):
    """Is the LayerNorm fold taken for ``M`` token rows of width ``D``?  (HipViT.fold_layernorm / fold_min_rows)"""
    return bool(m.fold_layernorm) and n_full_blocks > 0 and D % 32 == 0 and D <= 2048 and M >= int(m.fold_min_rows)


def _deriv_buffer(m, M, N, K, device):
    """Where fc1 forward saves GELU'(pre): fragment-major when the persistent kernel takes both the writer and the reader (same M, N, K), else row-major."""
    if m.frag_gelu_derivative and not ops.FORCE_TILE and ops.gemm_frag_ok(M, N, K):
        return ops.FragTensor(M, N, device)
    return torch.empty((M, N), device=device, dtype=torch.bfloat16)


def _vit_forward(m: HipViT, arena, frames, keep):
    D, H, S = m.embed_dim, m.num_heads, m.seq
    # bf16 [frames * S, 768] = the input pipeline's patch rows (PatchVideo); fp32 (N, 3, H, W) = frames, cut into rows here
    patches = frames if (frames.dim() == 2 and frames.dtype == torch.bfloat16) else ops.im2col_patch16(frames)
    M = patches.size(0)
    N = M // S
    sh = arena.sh
    full_blocks = m.blocks[:-1] if m.cls_only_last_block else m.blocks
    fold = use_fold(m, M, D, len(full_blocks))
    R = ops.posres_prep(m.pos_embed, m.cls_token, m.patch_embed.proj.bias, S, D)
    part = ops.ln_stat_part(M, D, frames.device) if fold else None          # the next LayerNorm's statistics, emitted by the GEMM that writes its input
    x = ops.gemm(patches, sh(m.patch_embed.proj.weight).view(D, 768), M, D, 768, res=R, res_period=S, stat_part=part)
    saved = {'patches': patches if keep else None, 'blocks': [], 'cls_last': bool(m.cls_only_last_block), 'fold': fold}
    for bi, blk in enumerate(full_blocks):
        if fold:
            f1 = arena.fold(blk.attn.qkv.weight, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.bias)
            f2 = arena.fold(blk.mlp.fc1.weight, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.bias)
            sf1, sb1 = ops.ln_stats_finalize(part, D, m.EPS, want_bwd=keep)
            qkv = ops.linear_fwd(x, f1.G, bias=f1.b2, ln_stat=sf1, ln_c=f1.c)
            att, lse = ops.vit_attn_fwd(qkv, N, S, H)
            x1 = ops.linear_fwd(att, sh(blk.attn.proj.weight), bias=blk.attn.proj.bias, res=x, stat_part=part)
            sf2, sb2 = ops.ln_stats_finalize(part, D, m.EPS, want_bwd=keep)
            pre = _deriv_buffer(m, M, 4 * D, D, x.device) if keep else None
            act = ops.linear_fwd(x1, f2.G, bias=f2.b2, act=ops.ACT_GELU_ERF, c2=pre, ln_stat=sf2, ln_c=f2.c)
            more = bi + 1 < len(full_blocks)                                  # another folded block follows: it needs the statistics of x2
            x2 = ops.linear_fwd(act, sh(blk.mlp.fc2.weight), bias=blk.mlp.fc2.bias, res=x1, stat_part=part if more else None)
            if keep:
                saved['blocks'].append((x, sf1, sb1, qkv, att, lse, x1, sf2, sb2, pre, act))
            x = x2
            continue
        ln1, mean1, rstd1 = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, m.EPS)
        qkv = ops.linear_fwd(ln1, sh(blk.attn.qkv.weight), bias=blk.attn.qkv.bias)
        att, lse = ops.vit_attn_fwd(qkv, N, S, H)
        x1 = ops.linear_fwd(att, sh(blk.attn.proj.weight), bias=blk.attn.proj.bias, res=x)
        ln2, mean2, rstd2 = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, m.EPS)
        pre = _deriv_buffer(m, M, 4 * D, D, x.device) if keep else None
        act = ops.linear_fwd(ln2, sh(blk.mlp.fc1.weight), bias=blk.mlp.fc1.bias, act=ops.ACT_GELU_ERF, c2=pre)
        x2 = ops.linear_fwd(act, sh(blk.mlp.fc2.weight), bias=blk.mlp.fc2.bias, res=x1)
        if keep:
            saved['blocks'].append((x, mean1, rstd1, ln1, qkv, att, lse, x1, mean2, rstd2, ln2, pre, act))
        x = x2
    del part
    if m.cls_only_last_block:
        x, last = _last_block_forward(m, arena, m.blocks[-1], x, N, keep)          # x: [N, D], the CLS rows only
        if keep:
            saved['blocks'].append(last)
        feat, meanf, rstdf = ops.layernorm_fwd(x, m.norm.weight, m.norm.bias, m.EPS)
    else:
        feat, meanf, rstdf = ops.layernorm_fwd(x, m.norm.weight, m.norm.bias, m.EPS, rows=N, ldx=S * D)
    saved['final'] = (x, meanf, rstdf) if keep else None
    return feat, saved


def _cls_rows(t, N, S, D):
    """The CLS rows of a [N*S, D] token tensor as a strided [N, D] view (row stride S*D)."""
    return t.view(N, S * D)[:, :D]


def _last_block_forward(m: HipViT, arena, blk, x, N, keep):
    """Last block, CLS-only: k|v for every token, q / attention / proj / MLP for token 0 of each frame."""
    D, H, S = m.embed_dim, m.num_heads, m.seq
    sh = arena.sh
    ln1, mean1, rstd1 = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, m.EPS)
    wqkv, bqkv = sh(blk.attn.qkv.weight), blk.attn.qkv.bias.detach()
    kv = ops.linear_fwd(ln1, wqkv[D:], bias=bqkv[D:])                                   # [N*S, 2D]
    q = ops.linear_fwd(_cls_rows(ln1, N, S, D), wqkv[:D], bias=bqkv[:D])               # [N, D]
    att, probs = ops.cls_attn_fwd(q, kv, N, S, H)
    x1 = ops.linear_fwd(att, sh(blk.attn.proj.weight), bias=blk.attn.proj.bias, res=_cls_rows(x, N, S, D))
    ln2, mean2, rstd2 = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, m.EPS)
    pre = torch.empty((N, 4 * D), device=x.device, dtype=torch.bfloat16) if keep else None
    act = ops.linear_fwd(ln2, sh(blk.mlp.fc1.weight), bias=blk.mlp.fc1.bias, act=ops.ACT_GELU_ERF, c2=pre)
    x2 = ops.linear_fwd(act, sh(blk.mlp.fc2.weight), bias=blk.mlp.fc2.bias, res=x1)
    return x2, ((x, mean1, rstd1, ln1, kv, q, probs, att, x1, mean2, rstd2, ln2, pre, act) if keep else None)


def _last_block_backward(m: HipViT, arena, blk, saved, dx2, N, prev_bias):
    """dx2 [N, D] = gradient of the block's CLS-row output; returns the gradient of the block's (all-token) input."""
    D, H, S = m.embed_dim, m.num_heads, m.seq
    sh, gr = arena.sh, arena.gr
    (x, mean1, rstd1, ln1, kv, q, probs, att, x1, mean2, rstd2, ln2, pre, act) = saved
    ops.linear_wgrad(dx2, act, gr(blk.mlp.fc2.weight))
    dh = ops.linear_dgrad(dx2, sh(blk.mlp.fc2.weight), act=ops.ACT_MUL_AUX, aux=pre, colsum=gr(blk.mlp.fc1.bias))
    ops.linear_wgrad(dh, ln2, gr(blk.mlp.fc1.weight))
    dln2 = ops.linear_dgrad(dh, sh(blk.mlp.fc1.weight))
    dx1 = ops.layernorm_bwd(dln2, x1, mean2, rstd2, blk.norm2.weight, gr(blk.norm2.weight), gr(blk.norm2.bias),
                            dres=dx2, colsum=gr(blk.attn.proj.bias))
    ops.linear_wgrad(dx1, att, gr(blk.attn.proj.weight))
    datt = ops.linear_dgrad(dx1, sh(blk.attn.proj.weight))
    dq, dkv = ops.cls_attn_bwd(q, kv, probs, datt, N, S, H)
    gb, gw, wqkv = gr(blk.attn.qkv.bias), gr(blk.attn.qkv.weight), sh(blk.attn.qkv.weight)
    ops.colsum(dq, gb[:D])                       # colsum(dk) == 0 and colsum(dv) == colsum(datt): see avt_cls_attn_bwd
    ops.colsum(datt, gb[2 * D:])
    ops.linear_wgrad(dq, _cls_rows(ln1, N, S, D), gw[:D])
    ops.linear_wgrad(dkv, ln1, gw[D:])
    dln1 = ops.linear_fwd(dkv, arena.sh_t(blk.attn.qkv.weight)[:, D:])                  # [N*S, D] = dkv @ Wkv, W^T columns D..3D
    dln1_cls = _cls_rows(dln1, N, S, D)
    ops.linear_dgrad(dq, wqkv[:D], res=dln1_cls, out=dln1_cls)                          # CLS rows += dq @ Wq, in place
    dx = ops.layernorm_bwd(dln1, x, mean1, rstd1, blk.norm1.weight, gr(blk.norm1.weight), gr(blk.norm1.bias), colsum=prev_bias)
    ops.add_rows(_cls_rows(dx, N, S, D), dx1)                                           # residual path of the CLS rows
    if prev_bias is not None:
        ops.colsum(dx1, prev_bias)
    return dx


def _vit_backward(m: HipViT, arena, saved, dfeat):
    D, H, S = m.embed_dim, m.num_heads, m.seq
    N = dfeat.size(0)
    M = N * S
    sh, gr, sh_t = arena.sh, arena.gr, arena.sh_t
    hook = m.grad_ready_hook
    x, meanf, rstdf = saved['final']
    last = m.blocks[-1]
    first_full = m.depth - 1
    if saved['fold']:
        arena.fold_scratch_guard()
    if saved['cls_last']:
        dx2 = ops.layernorm_bwd(dfeat, x, meanf, rstdf, m.norm.weight, gr(m.norm.weight), gr(m.norm.bias),
                                colsum=gr(last.mlp.fc2.bias))
        if hook:
            hook(m.norm.weight, m.norm.bias)
        prev_bias = gr(m.blocks[-2].mlp.fc2.bias) if m.depth > 1 else None
        dx = _last_block_backward(m, arena, last, saved['blocks'][-1], dx2, N, prev_bias)
        saved['blocks'][-1] = None
        del dx2
        if hook:
            hook(last.norm1.weight, last.mlp.fc2.bias)
        first_full = m.depth - 2
    else:
        dx = torch.zeros((M, D), device=dfeat.device, dtype=torch.bfloat16)     # only the CLS rows receive gradient
        ops.layernorm_bwd(dfeat, x, meanf, rstdf, m.norm.weight, gr(m.norm.weight), gr(m.norm.bias),
                          colsum=gr(last.mlp.fc2.bias), rows=N, ldx=S * D, dx=dx, lddx=S * D)
        if hook:
            hook(m.norm.weight, m.norm.bias)
    for i in range(first_full, -1, -1):
        blk = m.blocks[i]
        prev_bias = gr(m.blocks[i - 1].mlp.fc2.bias) if i > 0 else None
        if saved['fold']:
            # LayerNorm folded into qkv / fc1 (csrc/lnfold.hip): the gradients of their outputs travel multiplied by the rows' rstd (dY'), the raw
            # weight gradients dY'^T x are centred, scaled by gamma and split into dW / dgamma / dbeta / dbias by FoldedLinear.backward_weights
            (x, sf1, sb1, qkv, att, lse, x1, sf2, sb2, pre, act) = saved['blocks'][i]
            saved['blocks'][i] = None
            f1 = arena.fold(blk.attn.qkv.weight, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.bias)
            f2 = arena.fold(blk.mlp.fc1.weight, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.bias)
            ops.linear_wgrad(dx, act, gr(blk.mlp.fc2.weight))
            arena.fold_scratch_busy = True                                   # (cleared when the block's last shared scratch has been consumed)
            dh = ops.linear_fwd(dx, sh_t(blk.mlp.fc2.weight), act=ops.ACT_MUL_AUX, aux=pre, colsum=f2.dbt, ln_stat=sb2)     # = rstd2 o dh; dbt = colsum(dh)
            del act, pre
            ops.linear_wgrad(dh, x1, f2.T)
            f2.backward_weights()
            dln2 = ops.linear_fwd(dh, f2.Gt)
            del dh
            dx1 = ops.layernorm_bwd_folded(dln2, x1, sf2, dres=dx, colsum=gr(blk.attn.proj.bias))
            del dln2, x1, dx
            ops.linear_wgrad(dx1, att, gr(blk.attn.proj.weight))
            datt = ops.linear_fwd(dx1, sh_t(blk.attn.proj.weight))
            dqkv = ops.vit_attn_bwd(qkv, att, datt, lse, N, S, H, dbias=f1.dbt, row_stat=sb1)                          # = rstd1 o dqkv
            del datt, att, qkv
            ops.linear_wgrad(dqkv, x, f1.T)
            f1.backward_weights()
            arena.fold_scratch_busy = False
            dln1 = ops.linear_fwd(dqkv, f1.Gt)
            del dqkv
            dx = ops.layernorm_bwd_folded(dln1, x, sf1, dres=dx1, colsum=prev_bias)
            del dln1, dx1, x
            if hook:
                hook(blk.norm1.weight, blk.mlp.fc2.bias)
            continue
        (x, mean1, rstd1, ln1, qkv, att, lse, x1, mean2, rstd2, ln2, pre, act) = saved['blocks'][i]
        saved['blocks'][i] = None
        # x2 = act @ W2^T + b2 + x1          (db2 was accumulated by the producer of dx)
        ops.linear_wgrad(dx, act, gr(blk.mlp.fc2.weight))
        # data gradients read the transposed bf16 shadow W^T k-major (arena.transposed_of): both operands k-major = the persistent 8-phase kernel
        dh = ops.linear_fwd(dx, sh_t(blk.mlp.fc2.weight), act=ops.ACT_MUL_AUX, aux=pre, colsum=gr(blk.mlp.fc1.bias))
        del act, pre
        ops.linear_wgrad(dh, ln2, gr(blk.mlp.fc1.weight))
        dln2 = ops.linear_fwd(dh, sh_t(blk.mlp.fc1.weight))
        del dh, ln2
        dx1 = ops.layernorm_bwd(dln2, x1, mean2, rstd2, blk.norm2.weight, gr(blk.norm2.weight), gr(blk.norm2.bias),
                                dres=dx, colsum=gr(blk.attn.proj.bias))
        del dln2, x1, dx
        ops.linear_wgrad(dx1, att, gr(blk.attn.proj.weight))
        datt = ops.linear_fwd(dx1, sh_t(blk.attn.proj.weight))
        dqkv = ops.vit_attn_bwd(qkv, att, datt, lse, N, S, H, dbias=gr(blk.attn.qkv.bias))
        del datt, att, qkv
        ops.linear_wgrad(dqkv, ln1, gr(blk.attn.qkv.weight))
        dln1 = ops.linear_fwd(dqkv, sh_t(blk.attn.qkv.weight))
        del dqkv, ln1
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
        feat, saved = _vit_forward(module, arena, frames.contiguous() if (frames.dim() == 2 and frames.dtype == torch.bfloat16) else frames.float().contiguous(), keep=keep)
        ctx.module, ctx.arena, ctx.saved = module, arena, saved
        return feat.float()

    @staticmethod
    def backward(ctx, dfeat):
        ctx.arena.attach_grads()          # .grad views dropped between forward and backward (optimizer.zero_grad())
        _vit_backward(ctx.module, ctx.arena, ctx.saved, dfeat.to(torch.bfloat16).contiguous())
        ctx.saved = None
        return None, None, None, None, None
