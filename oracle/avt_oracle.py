"""Pure-PyTorch fp32 CPU restatement of the reference's ViT-B/16 + AVT-h training path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function cites the reference location
(paths are relative to the upstream facebookresearch/AVT checkout) whose arithmetic it restates.  The
ViT lives in the un-vendored dependency ``timm==0.4.12`` (env.yaml:302) and GPT-2 in
``transformers==4.2.2`` (env.yaml:183); their published algorithms are restated here and pinned by
``oracle/make_golden.py`` against (a) the reference's own modules imported in the build container
(BaseModel / AVTh / Basic / BasicLossAccuracy / MultiDimCrossEntropy / Warmup / CosineLR, with the
installed HF ``GPT2Model`` underneath) and (b) HF ``ViTModel`` as an independent implementation of the
same ViT architecture.  Pin status: head + losses + schedulers pinned to the imported reference
(tests/golden/*.npz); ViT arithmetic "parity pinned to an independent implementation, not to timm
itself" (timm is not installable here).

State-dict names and shapes are identical to the reference's (timm naming under ``backbone.model.``,
HF naming under ``future_predictor.gpt_model.`` with Conv1D weights stored (in, out)).
"""
import math

import numpy as np
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# ViT (timm 0.4.12 vision_transformer.py semantics; call sites models/video_classification.py:224,255)
# --------------------------------------------------------------------------------------------------
class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch=16, in_chans=3):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)  # (N, 196, D)


class _ViTAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        n, s, d = x.shape
        qkv = self.qkv(x).reshape(n, s, 3, self.num_heads, d // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q @ k.transpose(-2, -1)) * self.scale
        att = att.softmax(dim=-1)
        out = (att @ v).transpose(1, 2).reshape(n, s, d)
        return self.proj(out)


class _ViTMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))  # exact (erf) GELU


class _ViTBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _ViTAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _ViTMlp(dim, dim * mlp_ratio)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class OracleViT(nn.Module):
    """timm ``VisionTransformer(num_classes=0)``: returns the CLS token after the final LayerNorm."""
    def __init__(self, embed_dim=768, depth=12, num_heads=12, img=224, patch=16):
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = _PatchEmbed(embed_dim, patch)
        num_patches = (img // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([_ViTBlock(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)[:, 0]


VIT_CONFIGS = {
    # name -> (embed_dim, depth, heads); conf/model/backbone/avt_b.yaml:3, avt_b_in21k.yaml:3
    'vit_base_patch16_224': (768, 12, 12),
    'vit_base_patch16_224_in21k': (768, 12, 12),
    'vit_large_patch16_224': (1024, 24, 16),
    'vit_large_patch16_224_in21k': (1024, 24, 16),
}


class OracleTIMMModel(nn.Module):
    """models/video_classification.py:249-257 (TIMMModel) + :213-238 (process_each_frame)."""
    def __init__(self, num_classes=None, model_type='vit_base_patch16_224', vit=None):
        super().__init__()
        del num_classes
        self.model = vit if vit is not None else OracleViT(*VIT_CONFIGS[model_type])

    @property
    def output_dim(self):
        return self.model.embed_dim

    def forward(self, video):  # (N, C, T, H, W) -> (N, D, T, 1, 1)
        n, t = video.size(0), video.size(2)
        flat = video.transpose(1, 2).flatten(0, 1)
        feats = self.model(flat)
        return feats.view((n, t) + feats.shape[1:]).transpose(1, 2).unsqueeze(-1).unsqueeze(-1)


class OracleIdentityBackbone(nn.Module):
    """Config 1 (pre-extracted TSN features, expts/02_ek100_avt_tsn): features pass straight through."""
    def forward(self, video):
        return video


# --------------------------------------------------------------------------------------------------
# GPT-2 (HF transformers modeling_gpt2.py semantics; call sites models/future_prediction.py:89-93,178-181)
# --------------------------------------------------------------------------------------------------
class _Conv1D(nn.Module):
    """HF Conv1D: y = x @ W + b with W stored (in, out)."""
    def __init__(self, nf, nx):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(nx, nf).normal_(std=0.02))
        self.bias = nn.Parameter(torch.zeros(nf))

    def forward(self, x):
        return x @ self.weight + self.bias


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


class _GPT2Attention(nn.Module):
    def __init__(self, n_embd, n_head, attn_pdrop, resid_pdrop):
        super().__init__()
        self.n_head = n_head
        self.c_attn = _Conv1D(3 * n_embd, n_embd)
        self.c_proj = _Conv1D(n_embd, n_embd)
        self.attn_dropout = nn.Dropout(attn_pdrop)
        self.resid_dropout = nn.Dropout(resid_pdrop)

    def forward(self, x):
        b, t, c = x.shape
        hd = c // self.n_head
        q, k, v = self.c_attn(x).split(c, dim=2)
        q = q.view(b, t, self.n_head, hd).transpose(1, 2)
        k = k.view(b, t, self.n_head, hd).transpose(1, 2)
        v = v.view(b, t, self.n_head, hd).transpose(1, 2)
        att = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        causal = torch.tril(torch.ones(t, t, dtype=torch.bool, device=x.device))
        att = att.masked_fill(~causal, torch.finfo(att.dtype).min)
        att = self.attn_dropout(att.softmax(dim=-1))
        out = (att @ v).transpose(1, 2).reshape(b, t, c)
        return self.resid_dropout(self.c_proj(out))


class _GPT2MLP(nn.Module):
    def __init__(self, n_embd, n_inner, resid_pdrop):
        super().__init__()
        self.c_fc = _Conv1D(n_inner, n_embd)
        self.c_proj = _Conv1D(n_embd, n_inner)
        self.dropout = nn.Dropout(resid_pdrop)

    def forward(self, x):
        return self.dropout(self.c_proj(gelu_new(self.c_fc(x))))


class _GPT2Block(nn.Module):
    def __init__(self, n_embd, n_head, eps, attn_pdrop, resid_pdrop):
        super().__init__()
        self.ln_1 = nn.LayerNorm(n_embd, eps=eps)
        self.attn = _GPT2Attention(n_embd, n_head, attn_pdrop, resid_pdrop)
        self.ln_2 = nn.LayerNorm(n_embd, eps=eps)
        self.mlp = _GPT2MLP(n_embd, 4 * n_embd, resid_pdrop)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class OracleGPT2(nn.Module):
    """GPT2Model(inputs_embeds, position_ids) without wte (deleted at models/future_prediction.py:95)."""
    def __init__(self, n_embd, n_layer=12, n_head=12, n_positions=1024, layer_norm_epsilon=1e-5,
                 embd_pdrop=0.1, attn_pdrop=0.1, resid_pdrop=0.1):
        super().__init__()
        self.wpe = nn.Embedding(n_positions, n_embd)
        self.drop = nn.Dropout(embd_pdrop)
        self.h = nn.ModuleList([
            _GPT2Block(n_embd, n_head, layer_norm_epsilon, attn_pdrop, resid_pdrop) for _ in range(n_layer)])
        self.ln_f = nn.LayerNorm(n_embd, eps=layer_norm_epsilon)

    def forward(self, inputs_embeds, position_ids):
        h = self.drop(inputs_embeds + self.wpe(position_ids))
        for blk in self.h:
            h = blk(h)
        return self.ln_f(h)


# --------------------------------------------------------------------------------------------------
# AVT-h (models/future_prediction.py:51-258, non-quantised path; output_len > 1 = the roll-out of :168-202)
# --------------------------------------------------------------------------------------------------
class OracleAVTh(nn.Module):
    def __init__(self, in_features, output_len=1, avg_last_n=1, inter_dim=2048, future_pred_loss=True,
                 return_past_too=True, n_head=4, n_layer=6, output_len_eval=-1, **gpt_kwargs):
        super().__init__()
        gpt_kwargs.pop('future_pred_loss_wt', None)   # rides along in the HF config upstream (:21 of expt 01)
        self.output_len_eval = output_len_eval
        self.encoder = nn.Linear(in_features, inter_dim, bias=False)      # :80
        self.decoder = nn.Linear(inter_dim, in_features, bias=False)      # :81
        self.gpt_model = OracleGPT2(inter_dim, n_layer=n_layer, n_head=n_head, **gpt_kwargs)  # :89-93
        self.in_features, self.inter_dim = in_features, inter_dim
        self.output_len, self.avg_last_n = output_len, avg_last_n
        self.return_past_too, self.use_feat_loss = return_past_too, future_pred_loss

    @property
    def output_dim(self):
        return self.in_features

    def forward(self, feats, target_shape=None):
        t = feats.size(1)
        if target_shape is not None and len(target_shape) == 3:                   # :123-130
            output_len = target_shape[1]
        elif self.training or self.output_len_eval < 0:
            output_len = self.output_len
        else:
            output_len = self.output_len_eval
        inputs = self.encoder(feats)                                              # :163
        pos = torch.arange(0, t, dtype=torch.long, device=feats.device)          # :170-173
        hidden = self.gpt_model(inputs, pos)                                      # :178-181
        for _ in range(1, output_len):
            # :168-202: the newest token's last hidden state is the next input embedding, at the next position.  The
            # reference keeps HF's past_key_values; re-running the whole (causal, dropout-free) sequence is the same maths.
            # (with dropout off it is the same maths in training mode too -- pinned with gradients by golden G12, oracle/make_golden_r6.py)
            assert not self.training or all(d.p == 0.0 for d in self.gpt_model.modules() if isinstance(d, nn.Dropout)), \
                'the oracle restates the roll-out for a dropout-free model only'
            inputs = torch.cat([inputs, hidden[:, -1:]], dim=1)
            pos = torch.arange(0, inputs.size(1), dtype=torch.long, device=feats.device)
            hidden = torch.cat([hidden, self.gpt_model(inputs, pos)[:, -1:]], dim=1)
        decoded = self.decoder(hidden)                                            # :190
        losses = {}
        if self.use_feat_loss:                                                    # :207-215
            n = min(t, decoded.size(1))
            losses['feat'] = (decoded[:, :n - 1] - feats[:, 1:n]) ** 2
        if self.return_past_too:                                                  # :232-235
            final = torch.cat((feats, decoded[:, t - 1:]), dim=1)
        elif output_len > 0:
            final = decoded[:, -output_len:]
        else:
            final = decoded
        if self.avg_last_n > 0:                                                   # :241-242
            final = final[:, -self.avg_last_n:].mean(dim=1)
        past = torch.cat([feats[:, :1], decoded[:, :t - 1]], dim=1)               # :249-250
        return past, final, losses, {}


# --------------------------------------------------------------------------------------------------
# BaseModel glue (models/base_model.py:140-273) for the AVT configs: Identity aggregators, Linear classifier
# --------------------------------------------------------------------------------------------------
class OracleBaseModel(nn.Module):
    def __init__(self, backbone: nn.Module, future_predictor: nn.Module, feat_dim: int,
                 num_classes: Dict[str, int], dropout=0.2, classifier_on_past=True):
        super().__init__()
        self.backbone = backbone
        self.future_predictor = future_predictor
        self.dropout = nn.Dropout(dropout)
        self.classifiers = nn.ModuleDict({k: nn.Linear(feat_dim, c) for k, c in num_classes.items()})
        self.classifier_on_past = classifier_on_past
        for m in self.modules():                      # base_model.py:110-127: every nn.Linear <- N(0, 0.01), bias 0
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward_singlecrop(self, video, target_shape=None):
        out = {}
        b, t = video.size(0), video.size(1)
        feats = self.backbone(video.flatten(0, 1))                    # :153-154
        out['backbone'] = feats
        feats = feats.mean(dim=[-1, -2])                              # :157
        out['backbone_mean'] = feats.mean(dim=-1)                     # :159
        feats = feats.permute(0, 2, 1)                                # :166
        out['temp_agg'] = feats                                       # Identity aggregator (:175-177)
        out['temp_agg_projected'] = feats                             # empty project_mlp (:179)
        agg = feats.reshape((b, t) + feats.shape[1:]).flatten(1, 2)   # :183-191
        past, future, aux, _ = self.future_predictor(agg, target_shape)
        out['future'], out['past'] = future, past
        if self.classifier_on_past:                                   # :203-207
            drop = self.dropout(past)
            for k, cls in self.classifiers.items():
                out[f'past_logits/{k}'] = cls(drop)
        out['future_projected'] = agg                                 # :209
        out['future_agg'] = future                                    # Identity aggregator after future pred
        drop = self.dropout(future)                                   # :215-216
        for k, cls in self.classifiers.items():
            out[f'logits/{k}'] = cls(drop)
        return out, dict(aux)

    def forward(self, video, target_shape=None):                      # :240-273
        if video.ndim == 6:
            crops = [video]
        elif video.ndim == 7:
            crops = list(torch.unbind(video, dim=2))
        else:
            raise NotImplementedError(f'Unsupported size {tuple(video.shape)}')
        res = [self.forward_singlecrop(c, target_shape) for c in crops]
        outs = {k: torch.stack([r[0][k] for r in res]).mean(0) for k in res[0][0]}
        losses = {k: torch.stack([r[1][k] for r in res]).mean(0) for k in res[0][1]}
        return outs, losses


# --------------------------------------------------------------------------------------------------
# Losses / accuracy (func/train_eval_ops.py:45-85, loss_fn/multidim_xentropy.py:11-25, common/utils.py:17-44)
# --------------------------------------------------------------------------------------------------
def multidim_cross_entropy(logits, target, ignore_index=-1):
    flat = F.cross_entropy(logits.reshape(-1, logits.size(-1)), target.reshape(-1),
                           ignore_index=ignore_index, reduction='none')
    return flat.reshape(target.shape)


def topk_accuracy(logits, target, topk=(1, 5)):
    if bool(torch.all(target < 0)):
        return [torch.zeros([]) for _ in topk]
    logits, target = logits.flatten(0, -2), target.flatten()
    _, pred = logits.topk(max(topk), 1, True, True)
    correct = pred.t().eq(target[None])
    return [correct[:k].flatten().sum(dtype=torch.float32) * (100.0 / target.size(0)) for k in topk]


def basic_loss_accuracy(outputs, target, target_subclips):
    losses, accs = {}, {}
    for key, tgt in target.items():
        logits = outputs[f'logits/{key}']
        losses[f'cls_{key}'] = multidim_cross_entropy(logits, tgt)
        a1, a5 = topk_accuracy(logits, tgt, (1, min(5, logits.size(-1))))
        accs[f'acc1/{key}'], accs[f'acc5/{key}'] = a1, a5
        pk = f'past_logits/{key}'
        if pk in outputs and target_subclips is not None:
            past_tgt = torch.mode(target_subclips[key], -1)[0]
            losses[f'past_cls_{key}'] = multidim_cross_entropy(outputs[pk], past_tgt)
    return losses, accs


def total_loss(losses, loss_wts):
    """func/train.py:207-217: mean every loss tensor, weighted sum over keys whose weight is > 0."""
    tot = None
    for k, v in losses.items():
        w = loss_wts.get(k, 0.0)
        if w > 0:
            term = w * v.mean()
            tot = term if tot is None else tot + term
    return tot


# --------------------------------------------------------------------------------------------------
# Transformer-encoder temporal aggregator (models/temporal_aggregation.py:50-147; cloze loss off)
# --------------------------------------------------------------------------------------------------
class _OraclePosEnc(nn.Module):
    def __init__(self, d_model, max_len=1000):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)                             # :61
        pe[:, 1::2] = torch.cos(position * div_term)                             # :62
        self.register_buffer('pe', pe.unsqueeze(0).transpose(0, 1))              # (max_len, 1, d)


class OracleTransformerAgg(nn.Module):
    """Post-norm encoder layers written out (torch.nn.TransformerEncoderLayer: x = LN1(x + SA(x)); x = LN2(x + W2 relu(W1 x)),
    dim_feedforward 2048, eps 1e-5; dropout is identity here -- parity runs are dropout-free) with torch's parameter names."""
    def __init__(self, in_features, inter_rep=512, nheads=8, nlayers=6, agg_style='mean'):
        super().__init__()
        self.nheads, self.agg_style, self.inter_rep = nheads, agg_style, inter_rep
        self.downproject = nn.Linear(in_features, inter_rep)                     # :84
        self.pos_encoder = _OraclePosEnc(inter_rep)                              # :87
        layer = nn.TransformerEncoderLayer(d_model=inter_rep, nhead=nheads)      # :85 (parameter container only)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=nlayers, norm=nn.LayerNorm(inter_rep), enable_nested_tensor=False)

    @property
    def output_dim(self):
        return self.inter_rep

    def forward(self, feats):
        b, t, _ = feats.shape
        e, h = self.inter_rep, self.nheads
        x = self.downproject(feats) + self.pos_encoder.pe[:t, 0][None]           # :121 (B, T, E)
        for lay in self.transformer_encoder.layers:
            a = lay.self_attn
            q, k, v = F.linear(x, a.in_proj_weight, a.in_proj_bias).split(e, dim=-1)
            q = q.view(b, t, h, e // h).transpose(1, 2) * (e // h) ** -0.5
            k = k.view(b, t, h, e // h).transpose(1, 2)
            v = v.view(b, t, h, e // h).transpose(1, 2)
            att = (q @ k.transpose(-1, -2)).softmax(-1) @ v
            x = lay.norm1(x + a.out_proj(att.transpose(1, 2).reshape(b, t, e)))
            x = lay.norm2(x + lay.linear2(F.relu(lay.linear1(x))))
        x = self.transformer_encoder.norm(x)
        return (x.mean(dim=1) if self.agg_style == 'mean' else x[:, -1]), {}     # :138-141


# --------------------------------------------------------------------------------------------------
# Input pipeline (func/train.py:550-569 transform list; common/transforms.py functional forms)
# --------------------------------------------------------------------------------------------------
def resize_shape(clip_h, clip_w, target):
    """common/transforms.py:78-87: shorter side -> target, the other side scaled, never below target."""
    scale = target * 1.0 / min(clip_h, clip_w)
    return max(int(clip_h * scale), target), max(int(clip_w * scale), target)


# ---- ColorJitterVideo with non-zero strengths (common/transforms.py:399-421) --------------------------------------------------
# The reference hands the flipped, resized clip -- all frames stacked into ONE tall 8-bit image -- to torchvision's ColorJitter
# (torchvision 0.8.2, env.yaml:305: absent from /root/reference and from this image).  On a PIL image that class applies, in a random
# order (``torch.randperm(4)``) and with factors drawn by ``torch.tensor(1.0).uniform_(lo, hi)``, four Pillow operations
# (torchvision/transforms/functional_pil.py of 0.8.2): ImageEnhance.Brightness / Contrast / Color (= Image.blend(degenerate, image, factor)
# with degenerate = black | the mean grey of the WHOLE stacked image | the pixel's own grey) and a hue shift on Pillow's 8-bit HSV form.
# They are restated below on uint8 arrays, following Pillow's C code (Blend.c, Convert.c: rgb2l, rgb2hsv_row, hsv2rgb_row) and pinned
# against the installed Pillow itself (oracle/make_golden_r3.py: max difference 0 on random images and factors; Pillow 12.2.0).
JITTER_OPS = ('brightness', 'contrast', 'saturation', 'hue')


def pil_luma(rgb):
    """Pillow RGB -> L (ITU-R 601-2, integer form): (19595 R + 38470 G + 7471 B + 0x8000) >> 16."""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def pil_blend(degenerate, image, factor):
    """Image.blend(degenerate, image, factor) (Blend.c): float32 arithmetic; truncation inside [0, 1], clipping outside."""
    d, im = degenerate.astype(np.int32), image.astype(np.int32)
    t = (d.astype(np.float32) + np.float32(factor) * (im - d).astype(np.float32)).astype(np.float32)
    if 0.0 <= factor <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def pil_rgb2hsv(rgb):
    """Convert.c rgb2hsv_row: float32 ratios, the hue expression in double (its literals are doubles), 8-bit truncation."""
    r, g, b = (rgb[..., i].astype(np.float32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = maxc - minc
    with np.errstate(divide='ignore', invalid='ignore'):
        s = cr / maxc
        rc, gc, bc = ((maxc - c) / cr for c in (r, g, b))
        rc64, gc64, bc64 = rc.astype(np.float64), gc.astype(np.float64), bc.astype(np.float64)
        h = np.where(r == maxc, (bc - gc).astype(np.float64), np.where(g == maxc, 2.0 + rc64 - bc64, 4.0 + gc64 - rc64)).astype(np.float32)
        h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip(np.nan_to_num(h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
        us = np.clip(np.nan_to_num(s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    grey = cr == 0
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc.astype(np.int32)], -1).astype(np.uint8)


def pil_hsv2rgb(hsv):
    """Convert.c hsv2rgb_row, with C's promotions spelled out: ``(float)h * 6.0 / 255.0`` is double arithmetic, ``f`` and ``fs`` are
    stored as floats, ``fs * f`` is a float product, everything inside ``round()`` is double, and ``round`` is half away from zero.
    Pinned exhaustively (all 2^24 triples) against Pillow 12.2.0 by tests/test_oracle_cpu.py."""
    hd = hsv[..., 0].astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hd).astype(np.int32)
    f = (hd - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (hsv[..., 1].astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    v = hsv[..., 2].astype(np.int32)
    fv = v.astype(np.float64)
    rnd = lambda x: np.clip(np.floor(x + 0.5).astype(np.int32), 0, 255)           # round(): half away from zero (arguments are >= 0)
    pp = rnd(fv * (1.0 - fs.astype(np.float64)))
    q = rnd(fv * (1.0 - (fs * f).astype(np.float64)))                              # fs * f: float * float
    t = rnd(fv * (1.0 - fs.astype(np.float64) * (1.0 - f.astype(np.float64))))
    i = i % 6
    out = np.stack([np.choose(i, [v, q, pp, pp, t, v]), np.choose(i, [t, v, v, q, pp, pp]), np.choose(i, [pp, pp, t, v, v, q])], -1)
    return np.where((hsv[..., 1] == 0)[..., None], np.repeat(v[..., None], 3, -1), out).astype(np.uint8)


def pil_color_jitter(img_u8, ops):
    """img_u8: uint8 (H, W, 3), the stacked clip; ops: [(name, factor)] in application order (names of JITTER_OPS)."""
    img = np.ascontiguousarray(img_u8)
    for name, factor in ops:
        factor = float(factor)
        if name == 'brightness':
            img = pil_blend(np.zeros_like(img), img, factor)
        elif name == 'contrast':
            mean = int(pil_luma(img).astype(np.float64).mean() + 0.5)            # ImageStat.Stat(image.convert('L')).mean[0], rounded
            img = pil_blend(np.full_like(img, mean), img, factor)
        elif name == 'saturation':
            img = pil_blend(np.repeat(pil_luma(img)[..., None], 3, -1), img, factor)
        elif name == 'hue':                                                      # functional_pil.adjust_hue: np_h += np.uint8(hue_factor * 255)
            hsv = pil_rgb2hsv(img)
            hsv[..., 0] = (hsv[..., 0].astype(np.int32) + (int(factor * 255) & 255)) & 255
            img = pil_hsv2rgb(hsv)
        else:
            raise ValueError(name)
    return img


def video_preproc(clip_u8, new_hw, flip, crop_ij, crop_hw, scale_pix=1.0, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5),
                  reverse_channels=False, color_jitter_roundtrip=False, color_jitter_ops=None):
    """One clip through the reference's training / eval transform chain with the random draws given explicitly.
    clip_u8: uint8 (T, H, W, 3) -> float (3, T, h, w).
    color_jitter_roundtrip: the training chain's ColorJitterVideo (func/train.py:554-557, common/transforms.py:399-421) with all
    strengths 0 (conf/data/default.yaml:37-40): torchvision's ColorJitter (0.8.2, env.yaml:305 -- absent from /root/reference and
    from this image) leaves the image unchanged, but the wrapper's ToPILImage()/ToTensor() round trip does
    ``pic.mul(255).byte()`` (torchvision.transforms.functional.to_pil_image) and ``/ 255`` (to_tensor)."""
    x = clip_u8.float().permute(3, 0, 1, 2) / 255.0                              # to_tensor, common/transforms.py:124-146
    x = F.interpolate(x, size=tuple(new_hw), mode='bilinear')                    # resize, :60-91 (align_corners = None -> False)
    if flip:
        x = x.flip((-1,))                                                        # hflip, :167-175
    if color_jitter_ops:                                                         # ColorJitterVideo, :399-421: frames stacked on the height axis
        c, t, h, w = x.shape
        stacked = x.mul(255).byte().permute(1, 2, 3, 0).reshape(t * h, w, c).numpy()          # to_pil_image: mul(255).byte()
        x = torch.from_numpy(pil_color_jitter(stacked, color_jitter_ops)).view(t, h, w, c).permute(3, 0, 1, 2).float().div(255)
    elif color_jitter_roundtrip:
        x = x.mul(255).byte().float().div(255)                                   # ColorJitterVideo with zero strengths, :417-421
    x = x * scale_pix                                                            # func/train.py:559-560
    if reverse_channels:
        x = x[[2, 1, 0], ...]                                                    # func/train.py:561-564
    m = torch.as_tensor(mean, dtype=x.dtype)[:, None, None, None]
    sd = torch.as_tensor(std, dtype=x.dtype)[:, None, None, None]
    x = (x - m) / sd                                                             # normalize, :149-164
    i, j = crop_ij
    return x[..., i:i + crop_hw[0], j:j + crop_hw[1]]                            # crop, :36-42


# --------------------------------------------------------------------------------------------------
# Optimiser + LR schedule (torch.optim.SGD nesterov; common/scheduler.py:57-75, 88-135)
# --------------------------------------------------------------------------------------------------
def sgd_nesterov_step(p, g, buf, lr, momentum=0.9, weight_decay=0.0, first=False):
    """In-place on fp32 tensors; ``buf`` is the momentum buffer (initialised to g on the first step)."""
    g = g + weight_decay * p
    if first:
        buf.copy_(g)
    else:
        buf.mul_(momentum).add_(g)
    p.sub_(lr * (g + momentum * buf))


def lr_schedule(base_lr, warmup_iters, cosine_iters, n_steps, init_lr_ratio=0.0, eta_min=0.0):
    """LR seen by optimizer.step() number i (i = 0..n_steps-1) under Warmup(CosineLR), stepped per iteration.

    Reproduces the reference quirk: warm-up yields base*i/W for i < W, then CosineAnnealingLR's
    *recursive* update continues from base*(W-1)/W, so the peak LR is never reached.
    """
    W = max(warmup_iters, 1)
    ratio0 = init_lr_ratio if W > 1 else 1.0
    lrs, lr, last, cos_epoch = [], None, 0, 0
    lr = base_lr * (ratio0 + (1 - ratio0) * 0.0)
    for _ in range(n_steps):
        lrs.append(lr)
        if last < W - 1:
            last += 1
            lr = base_lr * (ratio0 + (1 - ratio0) * (last / W))
        else:
            cos_epoch += 1
            T = cosine_iters
            if cos_epoch >= T:
                lr = 0.0
            elif (cos_epoch - 1 - T) % (2 * T) == 0:
                lr = lr + (base_lr - eta_min) * (1 - math.cos(math.pi / T)) / 2
            else:
                lr = ((1 + math.cos(math.pi * cos_epoch / T)) / (1 + math.cos(math.pi * (cos_epoch - 1) / T))
                      * (lr - eta_min) + eta_min)
    return lrs


# --------------------------------------------------------------------------------------------------
# Deterministic closed-form parameter fill shared by the golden generator and the tests
# --------------------------------------------------------------------------------------------------
def _hash_uniform(n, seed):
    """u[i] in [-1, 1): a pure integer hash of (i, seed) -- bit-identical on every box, no RNG state involved."""
    m32 = 0xFFFFFFFF
    x = (torch.arange(n, dtype=torch.int64) * 2654435761 + (seed * 40503 + 12345)) & m32
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & m32
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & m32
    x = x ^ (x >> 16)
    return x.to(torch.float64) / 2147483648.0 - 1.0


def closed_form_fill_(named_tensors, scale_overrides: Optional[Dict[str, float]] = None):
    """Fill every tensor in-place with w.flatten()[i] = s * sqrt(3) * u(i, hash(name)), u uniform in [-1, 1).

    s = 1/sqrt(fan_in) for matrices (so activations stay O(1) and attention is far from uniform), 0.05 for biases,
    position/CLS embeddings 0.05; LayerNorm weights are 1 + 0.1*u.  Weights are full-rank (unlike a sinusoid fill)
    and reproducible on any box without shipping 1.5 GB of parameters.
    """
    for name, t in sorted(named_tensors, key=lambda kv: kv[0]):
        h = 0
        for ch in name:
            h = (h * 131 + ord(ch)) % 1000003
        fan_in = t.shape[-1] if t.ndim >= 2 else 1
        if t.ndim == 4:
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
        s = math.sqrt(3.0 / fan_in) if t.ndim >= 2 else 0.05
        is_norm_w = (('norm' in name or 'ln_' in name) and name.endswith('weight'))
        if 'c_attn.weight' in name or 'c_fc.weight' in name or ('c_proj.weight' in name):
            s = math.sqrt(3.0 / t.shape[0])           # Conv1D is (in, out)
        if 'pos_embed' in name or 'cls_token' in name or 'wpe' in name:
            s = 0.05
        if is_norm_w:
            s = 0.1
        if scale_overrides:
            for key, val in scale_overrides.items():
                if key in name:
                    s = val
        vals = s * _hash_uniform(t.numel(), h)
        if is_norm_w:
            vals = 1.0 + vals
        with torch.no_grad():
            t.copy_(vals.to(t.dtype).reshape(t.shape))
