"""Shared builders for the parity tests: the same architecture as an oracle (CPU, fp32) and as the HIP model."""
import numpy as np
import torch

from avt_amd.config import Cfg
from oracle import avt_oracle as O

LOSS_WTS = {'cls_action': 1.0, 'past_cls_action': 1.0, 'feat': 1.0}


def load_golden(path):
    z = np.load(path)
    return {k.replace('__', '/'): torch.from_numpy(np.asarray(z[k])) for k in z.files}


def model_cfg(backbone, in_dim, inter_dim, n_layer, n_head, dropout=0.0, head_drop=0.0, **head_kw):
    fp = Cfg(_target_='models.future_prediction.AVTh', n_head=n_head, n_layer=n_layer, output_len=1,
             inter_dim=inter_dim, return_past_too=True, avg_last_n=1, future_pred_loss_wt=1.0,
             embd_pdrop=head_drop, attn_pdrop=head_drop, resid_pdrop=head_drop,
             future_pred_loss=Cfg(_target_='torch.nn.MSELoss'), **head_kw)
    return Cfg(backbone=backbone, backbone_last_n_modules_to_drop=0, backbone_dim=in_dim, intermediate_featdim=None,
               temporal_aggregator=Cfg(_target_='models.temporal_aggregation.Identity'),
               temporal_aggregator_after_future_pred=Cfg(_target_='models.temporal_aggregation.Identity'),
               future_predictor=fp, classifier=Cfg(_target_='torch.nn.Linear', bias=True),
               same_temp_agg_dim=False, project_dim_for_nce=None, dropout=dropout, use_cls_mappings=False,
               classifier_on_past=True, add_regression_head=False, bn=Cfg(eps=0.001, mom=0.1))


def build_hip_model(kind, in_dim, inter_dim, n_layer, n_head, C, vit=None, device='cuda', **head_kw):
    from avt_amd.models.base_model import BaseModel
    if kind == 'feat':
        bb = Cfg(_target_='models.video_classification.IdentityFeatures')
    else:
        dim, depth, heads, img = vit
        bb = Cfg(_target_='models.video_classification.TIMMModel', model_type='custom', embed_dim=dim, depth=depth,
                 num_heads=heads, img_size=img)
    cfg = model_cfg(bb, in_dim, inter_dim, n_layer, n_head, **head_kw)
    return BaseModel(cfg, {'action': C}, {}).to(device)


def build_oracle_model(kind, in_dim, inter_dim, n_layer, n_head, C, vit=None, **head_kw):
    if kind == 'feat':
        bb = O.OracleIdentityBackbone()
    else:
        dim, depth, heads, img = vit
        bb = O.OracleTIMMModel(vit=O.OracleViT(dim, depth, heads, img=img))
    head = O.OracleAVTh(in_dim, inter_dim=inter_dim, n_layer=n_layer, n_head=n_head, embd_pdrop=0., attn_pdrop=0., resid_pdrop=0., **head_kw)
    return O.OracleBaseModel(bb, head, in_dim, {'action': C}, dropout=0.0)


def oracle_step(orc, video, target, sub):
    out, aux = orc(video, target_shape=target.shape)
    losses, accs = O.basic_loss_accuracy(out, {'action': target}, {'action': sub})
    losses.update(aux)
    tot = O.total_loss(losses, LOSS_WTS)
    orc.zero_grad()
    tot.backward()
    return out, losses, accs, tot


def hip_step(model, video, target, sub):
    from avt_amd.func.train_eval_ops import Basic
    op = Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
    data = {'video': video, 'target': {'action': target}, 'target_subclips': {'action': sub}}
    model.zero_grad()
    _, out, losses, accs = op(data, train_mode=True)
    tot = None
    for k, v in losses.items():
        if LOSS_WTS.get(k, 0) > 0:
            t = LOSS_WTS[k] * v.mean()
            tot = t if tot is None else tot + t
    tot.backward()
    torch.cuda.synchronize()
    return out, losses, accs, tot


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 -- sees a systematically wrong small-magnitude region that max-abs normalisation hides."""
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


# ---- the product's dropout masks, restated on the host -----------------------------------------------------------------------
def rng_keep_mask(seed, numel, p):
    """keep[i] of the HIP kernels' counter-based dropout (avt_amd/csrc/common.hpp: rng_u32 / drop_keep): a splitmix64-style hash
    of (seed, element index), kept when its bits 16..47 are >= p * 2^32.  uint64 arithmetic wraps, as on the device."""
    with np.errstate(over='ignore'):
        idx = np.arange(numel, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        u = (z >> np.uint64(16)) & np.uint64(0xFFFFFFFF)
    thresh = min(max(int(float(np.float32(p)) * 4294967296.0), 0), 4294967295)
    return torch.from_numpy((u >= np.uint64(thresh)).astype(np.float32))


class FixedMaskDropout(torch.nn.Module):
    """nn.Dropout with the mask a HIP kernel would draw for (seed, flat element index)."""
    def __init__(self, p, seed):
        super().__init__()
        self.p, self.seed = p, seed

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = rng_keep_mask(self.seed, x.numel(), self.p).view(x.shape)
        return x * keep * float(np.float32(1.0) / (np.float32(1.0) - np.float32(self.p)))


def give_oracle_the_hip_masks(orc, seed, p):
    """Replace the oracle head's dropouts by the masks the HIP head draws from `seed` (avt_amd/models/future_prediction.py:
    embedding dropout = seed; layer l: attention probabilities seed + 16 (l + 1) + 1, the two residual dropouts + 2 / + 3)."""
    g = orc.future_predictor.gpt_model
    g.drop = FixedMaskDropout(p, seed)
    for li, blk in enumerate(g.h):
        s0 = seed + 16 * (li + 1)
        blk.attn.attn_dropout = FixedMaskDropout(p, s0 + 1)
        blk.attn.resid_dropout = FixedMaskDropout(p, s0 + 2)
        blk.mlp.dropout = FixedMaskDropout(p, s0 + 3)


class SequencedMaskDropout(torch.nn.Module):
    """The classifier dropout of the HIP model restated for the oracle.  The HIP model draws ONE mask over the concatenated
    [past rows; future rows] x D matrix (avt_amd/models/base_model.py::_logits: element index = row * D + column); the
    oracle applies its dropout to the past rows first and to the future rows second (models/base_model.py:203-216), so the
    n-th call of a forward takes the next slice of that mask.  ``reset()`` before every oracle forward."""
    def __init__(self, p, seed):
        super().__init__()
        self.p, self.seed, self.offset = p, seed, 0

    def reset(self):
        self.offset = 0

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = rng_keep_mask(self.seed, self.offset + x.numel(), self.p)[self.offset:].view(x.shape)
        self.offset += x.numel()
        return x * keep * float(np.float32(1.0) / (np.float32(1.0) - np.float32(self.p)))
