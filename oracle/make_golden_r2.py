"""Round-2 golden fixtures (tests/golden/g2b_*, g6_*, g7_*, g8_*), produced by running the REFERENCE's own modules.

Same rules as oracle/make_golden.py (which this script imports its plumbing from, leaving the round-1 fixtures untouched):
runs only in the build container (needs /root/reference + HF transformers), commits only inputs / outputs as .npz.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r2.py

  G2b  BASELINE config 4's head (expts/07_ek100_avt_longer.txt:30,56-58): full-size AVT-h on ViT-B-sized features,
       in=768, Dh=2048, 6 layers, 4 heads, **T=15**, B=2, C=3806 -- reference BaseModel + AVTh + Basic op, one training step.
  G6   eval path: reference BaseModel in eval mode, **7-D multi-crop video** (3 crops averaged,
       models/base_model.py:251-273) and **roll-out** ``output_len_eval=3`` (models/future_prediction.py:168-202, HF KV cache):
       (a) tiny ViT (HF ViT inside the reference FrameLevelModel) + head; (b) full-size head on features, T=10, output_len_eval=4.
  G7   BASELINE config 5's backbone: full-depth ViT-L/16 (D=1024, L=24, H=16) CLS features on 1 frame from HF ViTModel.
  G9   input pipeline: the reference's own common/transforms.py functions (to_tensor, resize, hflip, normalize, crop)
       on random uint8 clips with explicit draws -- pins the fused GPU preprocessing kernel.
  G10  the Transformer-encoder temporal aggregator (models/temporal_aggregation.py:73-147), eval mode, fwd + grads.
  G8   other head shapes sharing the kernels (expts/13_50s_avt.txt:16-17 and expts/04*): n_head=2/n_layer=8 and
       n_head=8/n_layer=8 tiny heads, one training step each.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as G                      # noqa: E402  (plumbing stand-ins, HF-ViT wrapper, synth_batch, run_reference)

OUT = G.OUT
Cfg = G.Cfg


def main():
    import types
    torch.manual_seed(0)
    torch.set_num_threads(8)
    G.install_stubs()
    from oracle import avt_oracle as O
    import models.base_model as ref_bm
    ops_mod = G.load_by_path('func.train_eval_ops', os.path.join(G.REF, 'func', 'train_eval_ops.py'))
    sys.modules.setdefault('func', types.ModuleType('func'))
    sys.modules['golden_helpers'] = G
    loss_wts = {'cls_action': 1.0, 'past_cls_action': 1.0, 'feat': 1.0}
    report = []

    # ---------------- G2b: full-size AVT-h, T = 15 (config 4) ------------------------------------------------------
    IN, DH, L, H, T, C, B = 768, 2048, 6, 4, 15, 3806, 2
    cfg = G.model_cfg(Cfg(_target_='golden_helpers.FeatBackbone'), IN, DH, L, H)
    ref = ref_bm.BaseModel(cfg, {'action': C}, {})
    O.closed_form_fill_(list(ref.named_parameters()))
    video, target, sub = G.synth_batch(B, T, C, (IN, 1, 1, 1), seed=12)
    res = G.run_reference(ref, ops_mod, video, target, sub, loss_wts)
    keep = {k: v for k, v in res.items() if k.startswith(('loss/', 'acc/', 'total'))}
    keep['out/logits/action'] = res['out/logits/action']
    keep['out/past_logits/action_sub'] = res['out/past_logits/action'][:, :, ::16].contiguous()
    keep['out/future'], keep['out/past'] = res['out/future'], res['out/past']
    for n, p in ref.named_parameters():
        keep[f'gradnorm/{n}'] = p.grad.detach().norm()
    keep['grad/future_predictor.gpt_model.wpe.weight_rows0_16'] = ref.future_predictor.gpt_model.wpe.weight.grad.detach()[:16, ::8].contiguous()
    keep['grad/future_predictor.encoder.weight_sub'] = ref.future_predictor.encoder.weight.grad.detach()[::64, ::32].contiguous()
    np.savez_compressed(os.path.join(OUT, 'g2b_full_head_T15.npz'), **G.to_np(keep))
    orc = O.OracleBaseModel(O.OracleIdentityBackbone(), O.OracleAVTh(IN, inter_dim=DH, n_layer=L, n_head=H, embd_pdrop=0.,
                            attn_pdrop=0., resid_pdrop=0.), IN, {'action': C}, dropout=0.0)
    O.closed_form_fill_(list(orc.named_parameters()))
    oo, _ = orc(video, target_shape=target.shape)
    d = float((oo['logits/action'] - res['out/logits/action']).abs().max())
    report.append(f'G2b restatement vs reference (full-size head, T=15): max|dlogits|={d:.3e}')
    assert d < 2e-4

    # ---------------- G6a: tiny ViT, 3 crops, roll-out output_len_eval = 3, eval mode -------------------------------
    D, DEPTH, HEADS, IMG, DH, L, H, T, C, B, NCROP, OLEN = 128, 2, 2, 32, 64, 2, 4, 4, 17, 2, 3, 3
    hf = G.HFViTAsTimm(D, DEPTH, HEADS, IMG)

    class RefFrameModel(torch.nn.Module):
        def __init__(self, num_classes=None):
            super().__init__()
            self.model = hf

        def forward(self, video):
            n, t = video.size(0), video.size(2)
            f = self.model(video.transpose(1, 2).flatten(0, 1))
            return f.view((n, t) + f.shape[1:]).transpose(1, 2).unsqueeze(-1).unsqueeze(-1)
    G.RefFrameModel = RefFrameModel
    cfg = G.model_cfg(Cfg(_target_='golden_helpers.RefFrameModel'), D, DH, L, H)
    cfg.future_predictor['output_len_eval'] = OLEN
    ref = ref_bm.BaseModel(cfg, {'action': C}, {})
    orc = O.OracleBaseModel(O.OracleTIMMModel(vit=O.OracleViT(D, DEPTH, HEADS, img=IMG)),
                            O.OracleAVTh(D, inter_dim=DH, n_layer=L, n_head=H, output_len_eval=OLEN, embd_pdrop=0.,
                                         attn_pdrop=0., resid_pdrop=0.), D, {'action': C}, dropout=0.0)
    O.closed_form_fill_(list(orc.named_parameters()))
    G.push_timm_into_hf(orc.backbone.model.state_dict(), hf.vit, D, DEPTH)
    ref_sd = ref.state_dict()
    for k, v in orc.state_dict().items():
        if not k.startswith('backbone.'):
            ref_sd[k] = v
    ref.load_state_dict(ref_sd)
    g = torch.Generator().manual_seed(13)
    video = torch.rand((B, T, NCROP, 3, 1, IMG, IMG), generator=g) * 2 - 1      # (B, #clips, #crops, C, T, H, W)
    target = torch.randint(0, C, (B,), generator=g)
    ref.eval(); orc.eval()
    with torch.no_grad():
        r_out, r_loss = ref(video, target_shape=target.shape)
        o_out, o_loss = orc(video, target_shape=target.shape)
    d = max(float((o_out[k] - r_out[k]).abs().max()) for k in ['logits/action', 'past_logits/action', 'future', 'past'])
    dl = float((o_loss['feat'] - r_loss['feat']).abs().max())
    report.append(f'G6a tiny ViT, 3 crops, roll-out {OLEN}: restatement vs reference max|d|={d:.3e} |dfeat|={dl:.3e}')
    assert d < 1e-4 and dl < 1e-4, report[-1]
    res = {f'out/{k}': r_out[k] for k in ['logits/action', 'past_logits/action', 'future', 'past', 'future_agg', 'backbone_mean']}
    res['loss/feat'] = r_loss['feat']
    res['in/video'], res['in/target'] = video, target
    # the same clips, single crop 0 only and no roll-out: shows that both switches change the answer (guards a vacuous test)
    with torch.no_grad():
        ref.future_predictor.output_len_eval = -1
        s_out, _ = ref(video[:, :, 0], target_shape=target.shape)
        ref.future_predictor.output_len_eval = OLEN
    res['out_single_crop_no_rollout/logits/action'] = s_out['logits/action']
    np.savez_compressed(os.path.join(OUT, 'g6a_rollout_multicrop_tiny.npz'), **G.to_np(res))

    # ---------------- G6b: full-size head on features, T = 10, roll-out 4, 2 crops ----------------------------------
    IN, DH, L, H, T, C, B, NCROP, OLEN = 768, 2048, 6, 4, 10, 3806, 2, 2, 4
    cfg = G.model_cfg(Cfg(_target_='golden_helpers.FeatBackbone'), IN, DH, L, H)
    cfg.future_predictor['output_len_eval'] = OLEN
    ref = ref_bm.BaseModel(cfg, {'action': C}, {})
    O.closed_form_fill_(list(ref.named_parameters()))
    g = torch.Generator().manual_seed(14)
    video = torch.rand((B, T, NCROP, IN, 1, 1, 1), generator=g) * 2 - 1
    target = torch.randint(0, C, (B,), generator=g)
    ref.eval()
    with torch.no_grad():
        r_out, r_loss = ref(video, target_shape=target.shape)
    orc = O.OracleBaseModel(O.OracleIdentityBackbone(), O.OracleAVTh(IN, inter_dim=DH, n_layer=L, n_head=H, output_len_eval=OLEN,
                            embd_pdrop=0., attn_pdrop=0., resid_pdrop=0.), IN, {'action': C}, dropout=0.0)
    O.closed_form_fill_(list(orc.named_parameters()))
    orc.eval()
    with torch.no_grad():
        o_out, _ = orc(video, target_shape=target.shape)
    d = float((o_out['logits/action'] - r_out['logits/action']).abs().max())
    report.append(f'G6b full-size head, 2 crops, roll-out {OLEN}: restatement vs reference max|dlogits|={d:.3e}')
    assert d < 2e-4
    res = {'out/logits/action': r_out['logits/action'], 'out/future': r_out['future'], 'out/past': r_out['past'],
           'out/past_logits/action_sub': r_out['past_logits/action'][:, :, ::16].contiguous(), 'loss/feat_sub': r_loss['feat'][:, :, ::8].contiguous()}
    np.savez_compressed(os.path.join(OUT, 'g6b_rollout_full_head.npz'), **G.to_np(res))

    # ---------------- G7: full-depth ViT-L/16 CLS features on 1 frame (config 5 backbone) --------------------------
    hfl = G.HFViTAsTimm(1024, 24, 16, 224)
    vit = O.OracleViT(1024, 24, 16)
    O.closed_form_fill_(list(vit.named_parameters()))
    G.push_timm_into_hf(vit.state_dict(), hfl.vit, 1024, 24)
    g = torch.Generator().manual_seed(15)
    frames = torch.rand((1, 3, 224, 224), generator=g) * 2 - 1
    with torch.no_grad():
        f_hf, f_or = hfl(frames), vit(frames)
    d = float((f_hf - f_or).abs().max())
    report.append(f'G7 ViT-L/16 restatement vs HF ViT: max|dCLS|={d:.3e} (|CLS|max={float(f_hf.abs().max()):.3f}) '
                  f'params={sum(p.numel() for p in vit.parameters())}')
    assert d < 2e-4
    np.savez_compressed(os.path.join(OUT, 'g7_vitl_cls.npz'), frames_seed=np.int64(15), cls_hf=f_hf.numpy(), cls_oracle=f_or.numpy())
    del hfl, vit

    # ---------------- G8: other head shapes (n_head 2 / 8, n_layer 8), tiny, one training step ---------------------
    for tag, (IN, DH, L, H, T, C, B) in {'h2_l8': (32, 64, 8, 2, 6, 13, 2), 'h8_l8': (32, 128, 8, 8, 6, 13, 2)}.items():
        cfg = G.model_cfg(Cfg(_target_='golden_helpers.FeatBackbone'), IN, DH, L, H)
        ref = ref_bm.BaseModel(cfg, {'action': C}, {})
        O.closed_form_fill_(list(ref.named_parameters()))
        video, target, sub = G.synth_batch(B, T, C, (IN, 1, 1, 1), seed=16)
        res = G.run_reference(ref, ops_mod, video, target, sub, loss_wts)
        keep = {k: v for k, v in res.items() if k.startswith(('loss/', 'total')) or k in ('out/logits/action', 'out/past_logits/action', 'out/future')}
        for n in ['classifiers.action.weight', 'future_predictor.encoder.weight', 'future_predictor.gpt_model.h.0.attn.c_attn.weight',
                  f'future_predictor.gpt_model.h.{L - 1}.mlp.c_fc.weight', 'future_predictor.gpt_model.wpe.weight']:
            keep[f'grad/{n}'] = dict(ref.named_parameters())[n].grad.detach().clone()
        keep['in/video'], keep['in/target'], keep['in/sub'] = video, target, sub
        np.savez_compressed(os.path.join(OUT, f'g8_head_{tag}.npz'), **G.to_np(keep))
        report.append(f'G8 {tag}: total loss {float(res["total_loss"]):.6f}')

    # ---------------- G9: input pipeline -- the reference's own common/transforms.py functions ----------------------
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    for nm in ('RandomCrop', 'RandomResizedCrop', 'ColorJitter', 'ToPILImage', 'ToTensor'):
        setattr(tvt, nm, type(nm, (), {}))                   # class-level stand-ins: only the functional forms are exercised
    tv.transforms = tvt
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.transforms', tvt)
    import common.transforms as RT
    g = torch.Generator().manual_seed(17)
    B, T, H, W, CROP = 3, 2, 72, 128, 48
    clips = torch.randint(0, 256, (B, T, H, W, 3), generator=g, dtype=torch.uint8)
    draws = [dict(target=56, flip=0, crop=(3, 17), reverse=False, scale=1.0), dict(target=61, flip=1, crop=(13, 0), reverse=False, scale=1.0),
             dict(target=50, flip=1, crop=(2, 40), reverse=True, scale=255.0)]
    mean, std = (0.5, 0.45, 0.4), (0.5, 0.25, 0.2)
    outs, params = [], []
    for b, d in enumerate(draws):
        x = RT.to_tensor(clips[b])
        x = RT.resize(x, d['target'], 'bilinear')
        nh, nw = x.shape[-2:]
        assert (nh, nw) == O.resize_shape(H, W, d['target'])
        if d['flip']:
            x = RT.hflip(x)
        x = x * d['scale']
        if d['reverse']:
            x = x[[2, 1, 0], ...]
        x = RT.normalize(x, mean, std)
        x = RT.crop(x, d['crop'][0], d['crop'][1], CROP, CROP)
        outs.append(x)
        mine = O.video_preproc(clips[b], (nh, nw), d['flip'], d['crop'], (CROP, CROP), d['scale'], mean, std, d['reverse'])
        dd = float((mine - x).abs().max())
        report.append(f'G9 clip {b}: restatement vs reference transforms max|d|={dd:.3e}')
        assert dd < 1e-6
        params.append([nh, nw, d['flip'], d['crop'][0], d['crop'][1], int(d['reverse']), d['scale']])
    # evaluation chain: Resize(int) -> scale -> normalize -> MultiCropVideo(crop, 3 crops, flips) with the reference's multi_crop
    MC_T, MC_C = 56, 48
    mc = []
    for b in range(2):
        x = RT.normalize(RT.resize(RT.to_tensor(clips[b]), MC_T, 'bilinear') * 1.0, mean, std)
        mc.append(torch.stack(RT.multi_crop(x, (MC_C, MC_C), 3, True), 0))                 # (6, C, T, h, w)
    np.savez_compressed(os.path.join(OUT, 'g9_preproc.npz'), clips=clips.numpy(), params=np.asarray(params, dtype=np.float64),
                        mean=np.asarray(mean), std=np.asarray(std), out=torch.stack(outs).numpy(),
                        mc_target=np.int64(MC_T), mc_crop=np.int64(MC_C), mc_out=torch.stack(mc).numpy())

    # ---------------- G10: Transformer-encoder temporal aggregator (reference module, eval mode = dropout off) ---------
    import models.temporal_aggregation as ref_ta
    IN, E, NH, NL, T, B = 32, 64, 4, 2, 6, 3
    ref = ref_ta.Transformer(IN, inter_rep=E, nheads=NH, nlayers=NL)
    O.closed_form_fill_([(n, p) for n, p in ref.named_parameters()])
    ref.eval()
    g = torch.Generator().manual_seed(18)
    feats = (torch.rand((B, T, IN), generator=g) * 2 - 1).requires_grad_()
    wout = torch.rand((B, E), generator=g) * 2 - 1
    agg, _ = ref(feats)
    (agg * wout).sum().backward()
    orc = O.OracleTransformerAgg(IN, inter_rep=E, nheads=NH, nlayers=NL)
    orc.load_state_dict(ref.state_dict())
    orc.eval()
    f2 = feats.detach().clone().requires_grad_()
    o_agg, _ = orc(f2)
    (o_agg * wout).sum().backward()
    d = float((o_agg - agg).abs().max())
    dg = float((f2.grad - feats.grad).abs().max() / feats.grad.abs().max())
    report.append(f'G10 Transformer aggregator: restatement vs reference max|d|={d:.3e} rel dgrad(input)={dg:.3e}')
    assert d < 1e-5 and dg < 1e-4, report[-1]
    keep = {'in/feats': feats.detach(), 'in/wout': wout, 'out/agg': agg.detach(), 'grad/feats': feats.grad.detach()}
    for n in ['downproject.weight', 'downproject.bias', 'transformer_encoder.layers.0.self_attn.in_proj_weight',
              'transformer_encoder.layers.0.self_attn.in_proj_bias', 'transformer_encoder.layers.1.self_attn.out_proj.weight',
              'transformer_encoder.layers.1.linear1.weight', 'transformer_encoder.layers.0.linear2.bias',
              'transformer_encoder.layers.0.norm1.weight', 'transformer_encoder.norm.bias']:
        keep[f'grad/{n}'] = dict(ref.named_parameters())[n].grad.detach().clone()
    np.savez_compressed(os.path.join(OUT, 'g10_transformer_agg.npz'), **G.to_np(keep))

    with open(os.path.join(OUT, 'REPORT_r2.txt'), 'w') as f:
        f.write('Golden generation report (oracle/make_golden_r2.py), torch %s transformers %s\n' %
                (torch.__version__, __import__('transformers').__version__))
        f.write('\n'.join(report) + '\n')
    print('\n'.join(report))


if __name__ == '__main__':
    main()
