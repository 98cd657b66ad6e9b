"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference and HF transformers); the GPU box never sees
/root/reference -- it sees only the .npz vectors this script wrote.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

What is imported unmodified from the reference (with tiny stand-ins for the absent hydra / omegaconf /
submitit / cv2 packages, which only provide `instantiate` plumbing, never arithmetic):
  models.base_model.BaseModel, models.future_prediction.AVTh, models.temporal_aggregation.Identity,
  func/train_eval_ops.py::{Basic, BasicLossAccuracy}, loss_fn.multidim_xentropy.MultiDimCrossEntropy,
  common.utils.accuracy, common.scheduler.{Warmup, CosineLR}.
The ViT (timm, absent) is supplied by HF ``ViTModel`` -- an independent implementation of the same
architecture -- wrapped to look like timm's ``model(frames) -> (N, D)``.

Weights are a closed-form deterministic fill (oracle.avt_oracle.closed_form_fill_), so only inputs and
outputs are stored.  Fixtures:
  G1 tiny end-to-end (feature backbone, in=32, Dh=64, L=2, heads=4, T=10, C=17, B=2): all outputs, losses,
     accuracies, total loss, selected grads, one SGD-nesterov step.
  G2 full-size AVT-h config 1 (in=1024, Dh=2048, L=6, heads=4, T=10, B=2, C=3806): logits, losses, grad norms.
  G3 tiny ViT (D=128, depth=2, heads=2 -> head_dim 64, 32x32 images, patch 16) + head: HF ViT vs restatement, outputs.
  G3b one full-size ViT-B/16 forward on 2 frames (HF ViT): CLS features.
  G4 LR schedule vectors from the reference Warmup(CosineLR).
  G5 per-op known answers (LayerNorm, gelu erf/tanh, causal softmax, CE w/ ignore_index, top-k, shifted MSE).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


# ---------------------------------------------------------------------------------------------------
# Stand-ins for packages the container lacks (plumbing only)
# ---------------------------------------------------------------------------------------------------
class Cfg(dict):
    """Attribute-access dict standing in for an OmegaConf node."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _locate(path):
    mod, _, attr = path.rpartition('.')
    return getattr(importlib.import_module(mod), attr)


def _instantiate(cfg, *args, **kwargs):
    kwargs.pop('_recursive_', None)
    conf = {k: v for k, v in dict(cfg).items() if k != '_target_'}
    conf.update(kwargs)
    return _locate(cfg['_target_'])(*args, **conf)


def install_stubs():
    hydra = types.ModuleType('hydra')
    hydra.utils = types.ModuleType('hydra.utils')
    hydra.utils.instantiate = _instantiate
    hydra.utils.call = _instantiate
    hydra.types = types.ModuleType('hydra.types')
    hydra.types.TargetConf = dict
    omegaconf = types.ModuleType('omegaconf')
    omegaconf.OmegaConf = Cfg
    sys.modules.update({'hydra': hydra, 'hydra.utils': hydra.utils, 'hydra.types': hydra.types,
                        'omegaconf': omegaconf, 'submitit': types.ModuleType('submitit'),
                        'cv2': types.ModuleType('cv2')})
    ds = types.ModuleType('datasets')
    ds.__path__ = []
    bvd = types.ModuleType('datasets.base_video_dataset')
    bvd.FUTURE_PREFIX = 'future'       # the only thing func/train_eval_ops.py:14 imports it for
    sys.modules['datasets'] = ds
    sys.modules['datasets.base_video_dataset'] = bvd
    sys.path.insert(0, REF)


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class FeatBackbone(torch.nn.Module):
    """Config-1 'backbone': pre-extracted features pass straight through (N, C, 1, 1, 1)."""
    def __init__(self, num_classes=None):
        super().__init__()

    def forward(self, x):
        return x


class HFViTAsTimm(torch.nn.Module):
    """HF ViTModel behind timm's call signature (frames -> CLS feature after the final LayerNorm)."""
    def __init__(self, dim, depth, heads, img):
        super().__init__()
        import transformers
        cfg = transformers.ViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                                     intermediate_size=4 * dim, hidden_act='gelu', layer_norm_eps=1e-6,
                                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                     image_size=img, patch_size=16, qkv_bias=True)
        self.vit = transformers.ViTModel(cfg, add_pooling_layer=False)

    def forward(self, frames):
        return self.vit(pixel_values=frames).last_hidden_state[:, 0]


def push_timm_into_hf(osd, hf_vit, dim, depth):
    """Load timm-named ViT tensors (oracle state_dict) into HF ViTModel (transformers 5.x key names)."""
    hsd = hf_vit.state_dict()
    hsd['embeddings.cls_token'] = osd['cls_token']
    hsd['embeddings.position_embeddings'] = osd['pos_embed']
    hsd['embeddings.patch_embeddings.projection.weight'] = osd['patch_embed.proj.weight']
    hsd['embeddings.patch_embeddings.projection.bias'] = osd['patch_embed.proj.bias']
    hsd['layernorm.weight'], hsd['layernorm.bias'] = osd['norm.weight'], osd['norm.bias']
    for i in range(depth):
        p, q = f'layers.{i}.', f'blocks.{i}.'
        for j, n in enumerate(('q_proj', 'k_proj', 'v_proj')):
            hsd[p + f'attention.{n}.weight'] = osd[q + 'attn.qkv.weight'][j * dim:(j + 1) * dim]
            hsd[p + f'attention.{n}.bias'] = osd[q + 'attn.qkv.bias'][j * dim:(j + 1) * dim]
        hsd[p + 'attention.o_proj.weight'] = osd[q + 'attn.proj.weight']
        hsd[p + 'attention.o_proj.bias'] = osd[q + 'attn.proj.bias']
        hsd[p + 'layernorm_before.weight'], hsd[p + 'layernorm_before.bias'] = osd[q + 'norm1.weight'], osd[q + 'norm1.bias']
        hsd[p + 'layernorm_after.weight'], hsd[p + 'layernorm_after.bias'] = osd[q + 'norm2.weight'], osd[q + 'norm2.bias']
        hsd[p + 'mlp.fc1.weight'], hsd[p + 'mlp.fc1.bias'] = osd[q + 'mlp.fc1.weight'], osd[q + 'mlp.fc1.bias']
        hsd[p + 'mlp.fc2.weight'], hsd[p + 'mlp.fc2.bias'] = osd[q + 'mlp.fc2.weight'], osd[q + 'mlp.fc2.bias']
    hf_vit.load_state_dict(hsd)


def model_cfg(backbone, in_dim, inter_dim, n_layer, n_head, dropout=0.0, feat_loss=True):
    fp = Cfg(_target_='models.future_prediction.AVTh', n_head=n_head, n_layer=n_layer, output_len=1,
             inter_dim=inter_dim, return_past_too=True, avg_last_n=1, future_pred_loss_wt=1.0,
             # parity runs have dropout disabled (stated with the tolerance in the tests)
             embd_pdrop=0.0, attn_pdrop=0.0, resid_pdrop=0.0)
    if feat_loss:
        fp['future_pred_loss'] = Cfg(_target_='torch.nn.MSELoss')
    return Cfg(backbone=backbone, backbone_last_n_modules_to_drop=0, backbone_dim=in_dim,
               intermediate_featdim=None,
               temporal_aggregator=Cfg(_target_='models.temporal_aggregation.Identity'),
               temporal_aggregator_after_future_pred=Cfg(_target_='models.temporal_aggregation.Identity'),
               future_predictor=fp, classifier=Cfg(_target_='torch.nn.Linear', bias=True),
               same_temp_agg_dim=False, project_dim_for_nce=None, dropout=dropout, use_cls_mappings=False,
               classifier_on_past=True, add_regression_head=False, bn=Cfg(eps=0.001, mom=0.1))


def synth_batch(b, t, c, feat_shape, seed):
    g = torch.Generator().manual_seed(seed)
    video = torch.rand((b, t) + feat_shape, generator=g) * 2 - 1
    target = torch.randint(0, c, (b,), generator=g)
    sub = torch.randint(-1, c, (b, t, 1), generator=g)
    return video, target, sub


def run_reference(model, ops_mod, video, target, sub, loss_wts, do_step=None):
    """One reference training step through func/train_eval_ops.Basic (+ the step maths of func/train.py:207-233)."""
    from oracle import avt_oracle as O
    op = ops_mod.Basic(model, torch.device('cpu'), None,
                       Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
    data = {'video': video, 'target': {'action': target}, 'target_subclips': {'action': sub}}
    _, outputs, losses, accs = op(data, train_mode=True)
    total = O.total_loss(losses, loss_wts)
    model.zero_grad()
    total.backward()
    res = {'total_loss': total.detach()}
    res.update({f'out/{k}': v.detach() for k, v in outputs.items()})
    res.update({f'loss/{k}': v.detach() for k, v in losses.items()})
    res.update({f'acc/{k}': v.detach() for k, v in accs.items()})
    return res


def to_np(d):
    return {k.replace('/', '__'): (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_stubs()
    from oracle import avt_oracle as O
    import models.base_model as ref_bm                       # noqa: reference module
    ops_mod = load_by_path('func.train_eval_ops', os.path.join(REF, 'func', 'train_eval_ops.py'))
    sys.modules.setdefault('func', types.ModuleType('func'))
    import common.scheduler as ref_sched
    this = sys.modules[__name__]
    sys.modules['golden_helpers'] = this
    loss_wts = {'cls_action': 1.0, 'past_cls_action': 1.0, 'feat': 1.0}
    report = []

    # ---------------- G1: tiny end-to-end on features --------------------------------------------
    IN, DH, L, H, T, C, B = 32, 64, 2, 4, 10, 17, 2
    cfg = model_cfg(Cfg(_target_='golden_helpers.FeatBackbone'), IN, DH, L, H)
    ref = ref_bm.BaseModel(cfg, {'action': C}, {})
    O.closed_form_fill_(list(ref.named_parameters()))
    video, target, sub = synth_batch(B, T, C, (IN, 1, 1, 1), seed=1)
    sub[0, 3, 0] = -1
    res = run_reference(ref, ops_mod, video, target, sub, loss_wts)
    grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters()}
    for n in ['classifiers.action.weight', 'classifiers.action.bias', 'future_predictor.encoder.weight',
              'future_predictor.decoder.weight', 'future_predictor.gpt_model.h.0.attn.c_attn.weight',
              'future_predictor.gpt_model.h.0.attn.c_attn.bias', 'future_predictor.gpt_model.h.1.mlp.c_proj.weight',
              'future_predictor.gpt_model.wpe.weight', 'future_predictor.gpt_model.ln_f.weight',
              'future_predictor.gpt_model.h.0.ln_1.bias']:
        res[f'grad/{n}'] = grads[n]
    # one SGD-nesterov step exactly as conf/opt/optimizer/sgd.yaml + expts/01:26-28
    opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-6)
    opt.step()
    res2 = run_reference(ref, ops_mod, video, target, sub, loss_wts)
    opt.step()                                               # second step exercises the momentum buffer
    res['step2/total_loss'] = res2['total_loss']
    res['post2/classifiers.action.weight'] = ref.classifiers.action.weight.detach().clone()
    res['post2/future_predictor.encoder.weight'] = ref.future_predictor.encoder.weight.detach().clone()
    res['in/video'], res['in/target'], res['in/sub'] = video, target, sub
    # restatement check
    orc = O.OracleBaseModel(O.OracleIdentityBackbone(), O.OracleAVTh(IN, inter_dim=DH, n_layer=L, n_head=H,
                            embd_pdrop=0., attn_pdrop=0., resid_pdrop=0.), IN, {'action': C}, dropout=0.0)
    O.closed_form_fill_(list(orc.named_parameters()))
    oo, ol = orc(video, target_shape=target.shape)
    lo, _ = O.basic_loss_accuracy(oo, {'action': target}, {'action': sub})
    lo.update(ol)
    d = float((oo['logits/action'] - res['out/logits/action']).abs().max())
    dt = float((O.total_loss(lo, loss_wts) - res['total_loss']).abs())
    report.append(f'G1 restatement vs reference: max|dlogits|={d:.3e} |dtotal|={dt:.3e}')
    assert d < 1e-5 and dt < 1e-5
    np.savez_compressed(os.path.join(OUT, 'g1_tiny_head.npz'), **to_np(res))

    # ---------------- G2: full-size AVT-h (config 1) ---------------------------------------------
    IN, DH, L, H, T, C, B = 1024, 2048, 6, 4, 10, 3806, 2
    cfg = model_cfg(Cfg(_target_='golden_helpers.FeatBackbone'), IN, DH, L, H)
    ref = ref_bm.BaseModel(cfg, {'action': C}, {})
    O.closed_form_fill_(list(ref.named_parameters()))
    video, target, sub = synth_batch(B, T, C, (IN, 1, 1, 1), seed=2)
    res = run_reference(ref, ops_mod, video, target, sub, loss_wts)
    keep = {k: v for k, v in res.items() if k.startswith(('loss/', 'acc/', 'total'))}
    keep['out/logits/action'] = res['out/logits/action']
    keep['out/past_logits/action_sub'] = res['out/past_logits/action'][:, :, ::16].contiguous()
    keep['out/future'] = res['out/future']
    keep['out/past'] = res['out/past']
    for n, p in ref.named_parameters():
        keep[f'gradnorm/{n}'] = p.grad.detach().norm()
    keep['grad/classifiers.action.bias'] = ref.classifiers.action.bias.grad.detach().clone()
    keep['grad/future_predictor.gpt_model.h.5.ln_2.weight'] = ref.future_predictor.gpt_model.h[5].ln_2.weight.grad.detach().clone()
    keep['grad/future_predictor.encoder.weight_sub'] = ref.future_predictor.encoder.weight.grad.detach()[::64, ::32].contiguous()
    np.savez_compressed(os.path.join(OUT, 'g2_full_head.npz'), **to_np(keep))
    orc = O.OracleBaseModel(O.OracleIdentityBackbone(), O.OracleAVTh(IN, inter_dim=DH, n_layer=L, n_head=H,
                            embd_pdrop=0., attn_pdrop=0., resid_pdrop=0.), IN, {'action': C}, dropout=0.0)
    O.closed_form_fill_(list(orc.named_parameters()))
    oo, ol = orc(video, target_shape=target.shape)
    d = float((oo['logits/action'] - res['out/logits/action']).abs().max())
    report.append(f'G2 restatement vs reference (full-size head): max|dlogits|={d:.3e}')
    assert d < 2e-4

    # ---------------- G3: tiny ViT + head end-to-end (HF ViT under the reference BaseModel) ------
    D, DEPTH, HEADS, IMG, DH, L, H, T, C, B = 128, 2, 2, 32, 64, 2, 4, 4, 17, 2   # head_dim 64 like ViT-B/L
    hf = HFViTAsTimm(D, DEPTH, HEADS, IMG)

    class RefFrameModel(torch.nn.Module):       # reference FrameLevelModel semantics with the HF ViT inside
        def __init__(self, num_classes=None):
            super().__init__()
            self.model = hf

        def forward(self, video):
            n, t = video.size(0), video.size(2)
            f = self.model(video.transpose(1, 2).flatten(0, 1))
            return f.view((n, t) + f.shape[1:]).transpose(1, 2).unsqueeze(-1).unsqueeze(-1)
    this.RefFrameModel = RefFrameModel
    cfg = model_cfg(Cfg(_target_='golden_helpers.RefFrameModel'), D, DH, L, H)
    ref = ref_bm.BaseModel(cfg, {'action': C}, {})
    orc = O.OracleBaseModel(O.OracleTIMMModel(vit=O.OracleViT(D, DEPTH, HEADS, img=IMG)),
                            O.OracleAVTh(D, inter_dim=DH, n_layer=L, n_head=H, embd_pdrop=0., attn_pdrop=0.,
                                         resid_pdrop=0.), D, {'action': C}, dropout=0.0)
    O.closed_form_fill_(list(orc.named_parameters()))
    push_timm_into_hf(orc.backbone.model.state_dict(), hf.vit, D, DEPTH)
    ref_sd = ref.state_dict()
    for k, v in orc.state_dict().items():
        if not k.startswith('backbone.'):
            ref_sd[k] = v
    ref.load_state_dict(ref_sd)
    video, target, sub = synth_batch(B, T, C, (3, 1, IMG, IMG), seed=3)
    res = run_reference(ref, ops_mod, video, target, sub, loss_wts)
    oo, ol = orc(video, target_shape=target.shape)
    lo, _ = O.basic_loss_accuracy(oo, {'action': target}, {'action': sub})
    lo.update(ol)
    tot = O.total_loss(lo, loss_wts)
    orc.zero_grad()
    tot.backward()
    d = float((oo['logits/action'] - res['out/logits/action']).abs().max())
    dt = float((tot - res['total_loss']).abs())
    gq = hf.vit.layers[0].attention.q_proj.weight.grad
    gv = hf.vit.layers[0].attention.v_proj.weight.grad
    dg = float((orc.backbone.model.blocks[0].attn.qkv.weight.grad[2 * D:] - gv).abs().max() / gv.abs().max())
    print('|gq|max', float(gq.abs().max()), '|gv|max', float(gv.abs().max()),
          'abs dq', float((orc.backbone.model.blocks[0].attn.qkv.weight.grad[:D] - gq).abs().max()))
    report.append(f'G3 tiny ViT+head: restatement vs reference(HF ViT inside): max|dlogits|={d:.3e} |dtotal|={dt:.3e} rel dgrad(v)={dg:.3e}')
    assert d < 1e-4 and dt < 1e-4 and dg < 1e-3, report[-1]
    res['grad/backbone.model.blocks.0.attn.qkv.weight'] = orc.backbone.model.blocks[0].attn.qkv.weight.grad.detach().clone()
    res['grad/backbone.model.patch_embed.proj.weight'] = orc.backbone.model.patch_embed.proj.weight.grad.detach().clone()
    res['grad/backbone.model.pos_embed'] = orc.backbone.model.pos_embed.grad.detach().clone()
    res['grad/backbone.model.cls_token'] = orc.backbone.model.cls_token.grad.detach().clone()
    res['grad/hf_query0'] = gq.detach().clone()
    res['grad/future_predictor.encoder.weight'] = ref.future_predictor.encoder.weight.grad.detach().clone()
    res['in/video'], res['in/target'], res['in/sub'] = video, target, sub
    np.savez_compressed(os.path.join(OUT, 'g3_tiny_vit.npz'), **to_np(res))

    # ---------------- G3b: full-size ViT-B/16 CLS features on 2 frames ---------------------------
    hfb = HFViTAsTimm(768, 12, 12, 224)
    vit = O.OracleViT(768, 12, 12)
    O.closed_form_fill_(list(vit.named_parameters()))
    push_timm_into_hf(vit.state_dict(), hfb.vit, 768, 12)
    g = torch.Generator().manual_seed(4)
    frames = torch.rand((2, 3, 224, 224), generator=g) * 2 - 1
    with torch.no_grad():
        f_hf = hfb(frames)
        f_or = vit(frames)
    d = float((f_hf - f_or).abs().max())
    report.append(f'G3b ViT-B/16 restatement vs HF ViT: max|dCLS|={d:.3e} (|CLS|max={float(f_hf.abs().max()):.3f}) params={sum(p.numel() for p in vit.parameters())}')
    assert d < 1e-4
    np.savez_compressed(os.path.join(OUT, 'g3b_vitb_cls.npz'), frames_seed=np.int64(4), cls_hf=f_hf.numpy(), cls_oracle=f_or.numpy())

    # ---------------- G4: LR schedules from the reference schedulers -----------------------------
    sched = {}
    for (W_ep, cos_ep, ipe, base, world) in [(2, 3, 4, 1e-4, 8), (20, 30, 3, 1e-4, 1), (0, 5, 2, 0.1, 2), (1, 2, 1, 0.01, 1)]:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=base * world, momentum=0.9, nesterov=True)
        cos = ref_sched.CosineLR(opt, num_epochs=cos_ep, iters_per_epoch=ipe, world_size=world, eta_min=0.0)
        wu = ref_sched.Warmup(opt, cos, init_lr_ratio=0.0, num_epochs=W_ep, iters_per_epoch=ipe, world_size=world)
        n = (W_ep + cos_ep) * ipe + 3
        lrs = []
        for _ in range(n):
            lrs.append(opt.param_groups[0]['lr'])
            opt.step()
            wu.step()
        key = f'W{W_ep}_C{cos_ep}_I{ipe}_B{base}_N{world}'
        sched[key] = np.asarray(lrs, dtype=np.float64)
        mine = O.lr_schedule(base * world, W_ep * ipe, cos_ep * ipe, n)
        dd = float(np.abs(np.asarray(mine) - sched[key]).max())
        report.append(f'G4 {key}: restatement max|dLR|={dd:.3e}')
        assert dd < 1e-12, (mine, lrs)
    np.savez_compressed(os.path.join(OUT, 'g4_lr_schedules.npz'), **sched)

    # ---------------- G5: per-op known answers ----------------------------------------------------
    import transformers.activations as hf_act
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 48, generator=g) * 2
    w, b = torch.randn(48, generator=g), torch.randn(48, generator=g)
    ops = {'x': x, 'ln_w': w, 'ln_b': b,
           'ln_eps1e-6': torch.nn.functional.layer_norm(x, (48,), w, b, 1e-6),
           'ln_eps1e-5': torch.nn.functional.layer_norm(x, (48,), w, b, 1e-5),
           'gelu_erf': torch.nn.functional.gelu(x), 'gelu_new': hf_act.NewGELUActivation()(x)}
    logits = torch.randn(3, 5, 11, generator=g) * 3
    tgt = torch.randint(-1, 11, (3, 5), generator=g)
    tgt[0, 0] = -1
    from loss_fn.multidim_xentropy import MultiDimCrossEntropy
    from common.utils import accuracy as ref_acc
    ops['ce_logits'], ops['ce_target'] = logits, tgt
    ops['ce_loss'] = MultiDimCrossEntropy(ignore_index=-1, reduction='none')(logits, tgt)
    a1, a5 = ref_acc(logits, tgt, topk=(1, 5))
    ops['acc1'], ops['acc5'] = a1, a5
    a1n, _ = ref_acc(logits, torch.full_like(tgt, -1), topk=(1, 5))
    ops['acc1_all_ignored'] = a1n
    np.savez_compressed(os.path.join(OUT, 'g5_ops.npz'), **to_np(ops))

    with open(os.path.join(OUT, 'REPORT.txt'), 'w') as f:
        f.write('Golden generation report (oracle/make_golden.py), torch %s transformers %s\n' %
                (torch.__version__, __import__('transformers').__version__))
        f.write('\n'.join(report) + '\n')
    print('\n'.join(report))


if __name__ == '__main__':
    main()
