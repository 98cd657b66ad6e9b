"""Round-3 golden fixtures (tests/golden/g8b_*), produced by running the REFERENCE's own modules.

Same rules as oracle/make_golden.py / make_golden_r2.py: runs only in the build container (needs /root/reference + HF
transformers), commits only inputs / outputs as .npz; weights are the closed-form hash fill, so nothing else travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r3.py

  G8b  SURVEY 8f-4 at the widths the reference's experiments name: AVT-h with inter_dim = 2048, n_layer = 8 and
       n_head = 2  (head_dim 1024, expts/04_ek100_avt_ig65m.txt:13-16)  /  n_head = 8 (head_dim 256, expts/13_50s_avt.txt:15-18),
       on ViT-B-sized features (in = 768), T = 10, B = 2, C = 3806 -- reference BaseModel + AVTh + Basic op, one training step
       (outputs, the three losses, every parameter's gradient norm and sub-sampled gradients).
  G11  input pipeline with NON-ZERO colour jitter: the reference's own ``ColorJitterVideo`` wrapper
       (common/transforms.py:399-421: frames stacked into one tall image -> ToPILImage -> ColorJitter -> ToTensor) inside its transform chain
       to_tensor -> resize -> hflip -> jitter -> scale -> normalize -> crop.  torchvision is absent from this image, so its three classes are
       stood in for by what torchvision 0.8.2 does on PIL images, executed by the REAL Pillow: ColorJitter = ImageEnhance.Brightness /
       Contrast / Color and the HSV hue shift of functional_pil.adjust_hue, applied in an explicit order with explicit factors (the draws
       are inputs); ToPILImage = pic.mul(255).byte(); ToTensor = / 255.  Also pins oracle.pil_* against Pillow on random images.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as G                      # noqa: E402

OUT = G.OUT
Cfg = G.Cfg


def main():
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
    IN, DH, L, T, C, B = 768, 2048, 8, 10, 3806, 2
    for tag, H in {'h2': 2, 'h8': 8}.items():
        cfg = G.model_cfg(Cfg(_target_='golden_helpers.FeatBackbone'), IN, DH, L, H)
        ref = ref_bm.BaseModel(cfg, {'action': C}, {})
        O.closed_form_fill_(list(ref.named_parameters()))
        video, target, sub = G.synth_batch(B, T, C, (IN, 1, 1, 1), seed=31)
        res = G.run_reference(ref, ops_mod, video, target, sub, loss_wts)
        keep = {k: v for k, v in res.items() if k.startswith(('loss/', 'acc/', 'total'))}
        keep['out/logits/action'] = res['out/logits/action']
        keep['out/past_logits/action_sub'] = res['out/past_logits/action'][:, :, ::16].contiguous()
        keep['out/future'], keep['out/past'] = res['out/future'], res['out/past']
        params = dict(ref.named_parameters())
        for n, p in params.items():
            keep[f'gradnorm/{n}'] = p.grad.detach().norm()
        keep['grad/future_predictor.gpt_model.wpe.weight_rows0_16'] = params['future_predictor.gpt_model.wpe.weight'].grad.detach()[:16, ::8].contiguous()
        keep['grad/future_predictor.encoder.weight_sub'] = params['future_predictor.encoder.weight'].grad.detach()[::64, ::32].contiguous()
        keep[f'grad/future_predictor.gpt_model.h.{L - 1}.attn.c_attn.weight_sub'] = params[f'future_predictor.gpt_model.h.{L - 1}.attn.c_attn.weight'].grad.detach()[::64, ::96].contiguous()
        keep['grad/future_predictor.gpt_model.h.0.attn.c_attn.bias'] = params['future_predictor.gpt_model.h.0.attn.c_attn.bias'].grad.detach().clone()
        np.savez_compressed(os.path.join(OUT, f'g8b_head_2048x8_{tag}.npz'), **G.to_np(keep))
        orc = O.OracleBaseModel(O.OracleIdentityBackbone(), O.OracleAVTh(IN, inter_dim=DH, n_layer=L, n_head=H, embd_pdrop=0.,
                                attn_pdrop=0., resid_pdrop=0.), IN, {'action': C}, dropout=0.0)
        O.closed_form_fill_(list(orc.named_parameters()))
        oo, _ = orc(video, target_shape=target.shape)
        d = float((oo['logits/action'] - res['out/logits/action']).abs().max())
        report.append(f'G8b {tag} (inter_dim 2048, n_layer 8, n_head {H}: head_dim {DH // H}): total loss {float(res["total_loss"]):.6f}, '
                      f'restatement vs reference max|dlogits| = {d:.3e}')
        assert d < 2e-4
        del ref, orc
    # ---------------- G11: ColorJitterVideo with non-zero strengths ---------------------------------------------------------------
    from PIL import Image, ImageEnhance
    import PIL

    class PILColorJitter:                       # torchvision 0.8.2 ColorJitter on a PIL image, with the random draws made explicit
        ops = []

        def __init__(self, *a, **k):
            pass

        def __call__(self, img):
            for name, factor in PILColorJitter.ops:
                if name == 'brightness':
                    img = ImageEnhance.Brightness(img).enhance(factor)
                elif name == 'contrast':
                    img = ImageEnhance.Contrast(img).enhance(factor)
                elif name == 'saturation':
                    img = ImageEnhance.Color(img).enhance(factor)
                else:                            # functional_pil.adjust_hue
                    h, s_, v = img.convert('HSV').split()
                    np_h = np.array(h, dtype=np.uint8)
                    with np.errstate(over='ignore'):
                        np_h += np.uint8(int(factor * 255) & 255)
                    img = Image.merge('HSV', (Image.fromarray(np_h, 'L'), s_, v)).convert('RGB')
            return img

    class ToPILImage:                           # functional.to_pil_image of a float CHW tensor: pic.mul(255).byte(), HWC
        def __call__(self, pic):
            return Image.fromarray(pic.mul(255).byte().permute(1, 2, 0).contiguous().numpy(), 'RGB')

    class ToTensor:                             # functional.to_tensor of an 8-bit PIL image: CHW float / 255
        def __call__(self, img):
            return torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).float().div(255)

    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    for nm in ('RandomCrop', 'RandomResizedCrop'):
        setattr(tvt, nm, type(nm, (), {}))
    tvt.ColorJitter, tvt.ToPILImage, tvt.ToTensor = PILColorJitter, ToPILImage, ToTensor
    tv.transforms = tvt
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tvt
    import common.transforms as RT
    # (a) the restated Pillow operations against Pillow itself
    rng = np.random.default_rng(3)
    worst = 0
    for trial in range(6):
        img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
        order = rng.permutation(4)
        ops = [(O.JITTER_OPS[i], float(rng.uniform(0.6, 1.4)) if i < 3 else float(rng.uniform(-0.1, 0.1))) for i in order]
        PILColorJitter.ops = ops
        ref_img = np.array(PILColorJitter()(Image.fromarray(img, 'RGB')))
        worst = max(worst, int(np.abs(ref_img.astype(int) - O.pil_color_jitter(img, ops).astype(int)).max()))
    report.append(f'G11 oracle.pil_color_jitter vs Pillow {PIL.__version__} (6 random images, random order and factors): max |diff| = {worst}')
    assert worst == 0
    # (b) the reference's chain with ColorJitterVideo on uint8 clips
    g = torch.Generator().manual_seed(41)
    B, T, H, W, CROP = 4, 3, 72, 128, 48
    clips = torch.randint(0, 256, (B, T, H, W, 3), generator=g, dtype=torch.uint8)
    draws = [dict(target=56, flip=0, crop=(3, 17), ops=[('brightness', 1.31), ('saturation', 0.72), ('contrast', 1.18), ('hue', 0.07)]),
             dict(target=61, flip=1, crop=(13, 0), ops=[('hue', -0.09), ('contrast', 0.66), ('brightness', 0.8), ('saturation', 1.37)]),
             dict(target=50, flip=1, crop=(2, 40), ops=[('contrast', 1.4), ('saturation', 1.0), ('brightness', 1.0)]),
             # identity geometry (target = the shorter side): the resize returns the 8-bit pixels themselves, so this clip pins the four
             # operations EXACTLY on the device as well (the resized clips differ from torch's bilinear in a last bit here and there)
             dict(target=72, flip=1, crop=(11, 31), ops=[('saturation', 1.33), ('hue', 0.1), ('brightness', 0.77), ('contrast', 1.27)])]
    mean, std = (0.5, 0.45, 0.4), (0.5, 0.25, 0.2)
    outs, params, op_ids, op_fac = [], [], [], []
    cj = RT.ColorJitterVideo()
    for b, d in enumerate(draws):
        x = RT.resize(RT.to_tensor(clips[b]), d['target'], 'bilinear')
        nh, nw = x.shape[-2:]
        if d['flip']:
            x = RT.hflip(x)
        PILColorJitter.ops = d['ops']
        x = cj(x)
        x = RT.crop(RT.normalize(x * 1.0, mean, std), d['crop'][0], d['crop'][1], CROP, CROP)
        outs.append(x)
        mine = O.video_preproc(clips[b], (nh, nw), d['flip'], d['crop'], (CROP, CROP), 1.0, mean, std, False, color_jitter_ops=d['ops'])
        dd = float((mine - x).abs().max())
        report.append(f'G11 clip {b} ({" > ".join(n for n, _ in d["ops"])}): restatement vs the reference chain max|d| = {dd:.3e}')
        assert dd < 1e-6
        params.append([nh, nw, d['flip'], d['crop'][0], d['crop'][1]])
        ids = [O.JITTER_OPS.index(n) for n, _ in d['ops']] + [-1] * (4 - len(d['ops']))
        op_ids.append(ids); op_fac.append([f for _, f in d['ops']] + [0.0] * (4 - len(d['ops'])))
    np.savez_compressed(os.path.join(OUT, 'g11_color_jitter.npz'), clips=clips.numpy(), params=np.asarray(params, dtype=np.int64),
                        op_ids=np.asarray(op_ids, dtype=np.int64), op_factors=np.asarray(op_fac, dtype=np.float64),
                        mean=np.asarray(mean), std=np.asarray(std), out=torch.stack(outs).numpy())

    with open(os.path.join(OUT, 'REPORT_r3.txt'), 'w') as f:
        f.write('\n'.join(report) + '\n')
    print('\n'.join(report))


if __name__ == '__main__':
    main()
