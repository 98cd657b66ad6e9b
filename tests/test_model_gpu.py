"""End-to-end parity (-m gpu): the HIP model behind the reference's plugin surface vs (a) golden vectors produced by
the reference's own modules (tests/golden, oracle/make_golden.py) and (b) the fp32 CPU oracle on the same seeded
inputs and weights.  Dropout is disabled for parity (p = 0 in the head config, model.dropout = 0): the mask stream of
the HIP path is its own counter-based RNG and cannot match torch's.

Stated tolerances (bf16 activations/weights in the GEMMs, fp32 accumulation and statistics, vs an fp32 reference):
  logits / features: max-abs error <= 3e-2 of the tensor's max-abs (bf16 carries 8 mantissa bits: ~0.2-0.4 % per
  rounding, ~20 roundings deep, worst element of thousands);  scalar total loss: <= 3e-2 relative;  parameter
  gradients: max-abs error <= 5e-2 of the gradient's max-abs.
"""
import os

import pytest
import torch

from helpers import LOSS_WTS, build_hip_model, build_oracle_model, hip_step, load_golden, oracle_step, rel

pytestmark = pytest.mark.gpu
TOL_OUT, TOL_GRAD = 3e-2, 5e-2


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')


def _fill(model):
    from oracle.avt_oracle import closed_form_fill_
    closed_form_fill_(list(model.named_parameters()))


def test_g1_tiny_head_vs_reference_golden(golden_dir):
    g = load_golden(os.path.join(golden_dir, 'g1_tiny_head.npz'))
    model = build_hip_model('feat', 32, 64, 2, 4, 17)
    _fill(model)
    video, target, sub = g['in/video'].cuda(), g['in/target'].cuda(), g['in/sub'].cuda()
    out, losses, accs, tot = hip_step(model, video, target, sub)
    for k in ['logits/action', 'past_logits/action', 'future', 'past', 'backbone_mean', 'temp_agg', 'future_agg']:
        assert out[k].shape == g[f'out/{k}'].shape, k
        assert rel(out[k], g[f'out/{k}']) < TOL_OUT, (k, rel(out[k], g[f'out/{k}']))
    for k in ['cls_action', 'past_cls_action', 'feat']:
        assert losses[k].shape == g[f'loss/{k}'].shape
        assert rel(losses[k], g[f'loss/{k}']) < TOL_OUT, (k, rel(losses[k], g[f'loss/{k}']))
    assert abs(float(tot) - float(g['total_loss'])) / abs(float(g['total_loss'])) < 3e-2
    assert float(accs['acc1/action']) == float(g['acc/acc1/action'])
    assert float(accs['acc5/action']) == float(g['acc/acc5/action'])
    params = dict(model.named_parameters())
    for k in [k for k in g if k.startswith('grad/')]:
        name = k[len('grad/'):]
        e = rel(params[name].grad, g[k])
        assert e < TOL_GRAD, (name, e)
    # two SGD-nesterov steps (momentum buffer exercised on the second)
    from avt_amd.optim import FusedSGD
    opt = FusedSGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-6, arena=model.arena)
    opt.step()
    _, _, _, tot2 = hip_step(model, video, target, sub)
    opt.step()
    torch.cuda.synchronize()
    assert abs(float(tot2) - float(g['step2/total_loss'])) / abs(float(g['step2/total_loss'])) < 5e-2
    for name in ['classifiers.action.weight', 'future_predictor.encoder.weight']:
        assert rel(params[name], g[f'post2/{name}']) < 2e-2, name


def test_g2_full_head_vs_reference_golden(golden_dir):
    """Config 1 of BASELINE.json at full size: in=1024, Dh=2048, 6 layers, 4 heads, T=10, B=2, C=3806."""
    g = load_golden(os.path.join(golden_dir, 'g2_full_head.npz'))
    from oracle.make_golden import synth_batch
    model = build_hip_model('feat', 1024, 2048, 6, 4, 3806)
    _fill(model)
    video, target, sub = synth_batch(2, 10, 3806, (1024, 1, 1, 1), seed=2)
    out, losses, accs, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    assert rel(out['logits/action'], g['out/logits/action']) < TOL_OUT
    assert rel(out['past_logits/action'][:, :, ::16], g['out/past_logits/action_sub']) < TOL_OUT
    assert rel(out['future'], g['out/future']) < TOL_OUT
    assert rel(out['past'], g['out/past']) < TOL_OUT
    assert abs(float(tot) - float(g['total_loss'])) / abs(float(g['total_loss'])) < 3e-2
    params = dict(model.named_parameters())
    worst = 0.0
    for name, p in params.items():
        gn, ref = float(p.grad.float().norm()), float(g[f'gradnorm/{name}'])
        worst = max(worst, abs(gn - ref) / (ref + 1e-12))
        assert abs(gn - ref) / (ref + 1e-12) < 6e-2, (name, gn, ref)
    assert rel(params['classifiers.action.bias'].grad, g['grad/classifiers.action.bias']) < TOL_GRAD
    assert rel(params['future_predictor.gpt_model.h.5.ln_2.weight'].grad, g['grad/future_predictor.gpt_model.h.5.ln_2.weight']) < TOL_GRAD
    assert rel(params['future_predictor.encoder.weight'].grad[::64, ::32], g['grad/future_predictor.encoder.weight_sub']) < TOL_GRAD


def test_g3_tiny_vit_vs_reference_golden(golden_dir):
    g = load_golden(os.path.join(golden_dir, 'g3_tiny_vit.npz'))
    model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    _fill(model)
    out, losses, accs, tot = hip_step(model, g['in/video'].cuda(), g['in/target'].cuda(), g['in/sub'].cuda())
    for k in ['logits/action', 'past_logits/action', 'future', 'past', 'backbone']:
        assert out[k].shape == g[f'out/{k}'].shape, k
        assert rel(out[k], g[f'out/{k}']) < TOL_OUT, (k, rel(out[k], g[f'out/{k}']))
    assert abs(float(tot) - float(g['total_loss'])) / abs(float(g['total_loss'])) < 3e-2
    params = dict(model.named_parameters())
    for k in ['grad/backbone.model.patch_embed.proj.weight', 'grad/backbone.model.pos_embed', 'grad/backbone.model.cls_token',
              'grad/future_predictor.encoder.weight']:
        name = k[len('grad/'):]
        assert rel(params[name].grad, g[k]) < TOL_GRAD, (name, rel(params[name].grad, g[k]))
    gq = params['backbone.model.blocks.0.attn.qkv.weight'].grad
    for j, nm in enumerate('qkv'):
        e = rel(gq[j * 128:(j + 1) * 128], g['grad/backbone.model.blocks.0.attn.qkv.weight'][j * 128:(j + 1) * 128])
        assert e < TOL_GRAD, (nm, e)


def test_g3b_vitb_cls_features_vs_hf_golden(golden_dir):
    """Full-size ViT-B/16 forward on 2 frames vs HF ViTModel (independent implementation of the timm architecture)."""
    g = load_golden(os.path.join(golden_dir, 'g3b_vitb_cls.npz'))
    from avt_amd.models.vit import HipViT
    from oracle.avt_oracle import closed_form_fill_
    vit = HipViT(768, 12, 12).cuda()
    closed_form_fill_(list(vit.named_parameters()))
    gen = torch.Generator().manual_seed(4)
    frames = torch.rand((2, 3, 224, 224), generator=gen) * 2 - 1
    with torch.no_grad():
        f = vit(frames.cuda())
    torch.cuda.synchronize()
    assert rel(f, g['cls_hf']) < 3e-2, rel(f, g['cls_hf'])


@pytest.mark.parametrize('cfg', [dict(vit=(128, 2, 2, 32), T=4, B=3, Dh=64, L=2, H=4, C=17, std=0.5),
                                 dict(vit=(192, 3, 3, 48), T=5, B=2, Dh=128, L=2, H=4, C=50, std=0.4),
                                 dict(vit=(256, 2, 4, 32), T=3, B=2, Dh=128, L=1, H=4, C=24, std=0.4, tile=808),
                                 dict(vit=(128, 2, 2, 32), T=4, B=3, Dh=64, L=2, H=4, C=17, std=0.5, tile=256)])
def test_random_weights_vs_oracle(cfg):
    """Random-normal weights large enough that attention is far from uniform (exercises q/k/softmax gradients).
    The `tile` cases push every GEMM of the model through one big-tile kernel (the one a full-size batch selects)."""
    from avt_amd import ops as _ops
    _ops.FORCE_TILE = cfg.get('tile', 0)
    try:
        _random_weights_vs_oracle(cfg)
    finally:
        _ops.FORCE_TILE = 0


def _random_weights_vs_oracle(cfg):
    dim, depth, heads, img = cfg['vit']
    torch.manual_seed(0)
    orc = build_oracle_model('vit', dim, cfg['Dh'], cfg['L'], cfg['H'], cfg['C'], vit=cfg['vit'])
    for n, p in orc.named_parameters():
        with torch.no_grad():
            if p.ndim >= 2:
                fan = p.shape[0] if 'c_' in n and 'weight' in n else p[0].numel()
                p.normal_(0, cfg['std'] * (2.0 / fan) ** 0.5 * 2)
            elif 'bias' in n:
                p.normal_(0, 0.1)
            else:
                p.normal_(1.0, 0.1)
    model = build_hip_model('vit', dim, cfg['Dh'], cfg['L'], cfg['H'], cfg['C'], vit=cfg['vit'])
    model.load_state_dict(orc.state_dict())
    g = torch.Generator().manual_seed(7)
    B, T, C = cfg['B'], cfg['T'], cfg['C']
    video = torch.rand((B, T, 3, 1, img, img), generator=g) * 2 - 1
    target = torch.randint(0, C, (B,), generator=g)
    sub = torch.randint(-1, C, (B, T, 1), generator=g)
    o_out, o_losses, _, o_tot = oracle_step(orc, video, target, sub)
    out, losses, _, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    assert rel(out['logits/action'], o_out['logits/action']) < TOL_OUT
    assert rel(out['past_logits/action'], o_out['past_logits/action']) < TOL_OUT
    assert abs(float(tot) - float(o_tot)) / abs(float(o_tot)) < 3e-2
    op = dict(orc.named_parameters())
    bad = []
    for n, p in model.named_parameters():
        e = rel(p.grad, op[n].grad)
        if e > TOL_GRAD and float(op[n].grad.abs().max()) > 1e-6:
            bad.append((n, e))
    assert not bad, bad


def test_full_size_step_vs_oracle():
    """BASELINE config 2 architecture (ViT-B/16 + AVT-h 2048x6x4, C=3806) at B=1, T=3: logits / loss / sampled grads."""
    torch.manual_seed(1)
    vitc = (768, 12, 12, 224)
    orc = build_oracle_model('vit', 768, 2048, 6, 4, 3806, vit=vitc)
    for n, p in orc.backbone.named_parameters():
        with torch.no_grad():
            if p.ndim >= 2:
                p.normal_(0, 0.03)
    model = build_hip_model('vit', 768, 2048, 6, 4, 3806, vit=vitc)
    model.load_state_dict(orc.state_dict())
    g = torch.Generator().manual_seed(11)
    B, T, C = 1, 3, 3806
    video = torch.rand((B, T, 3, 1, 224, 224), generator=g) * 2 - 1
    target = torch.randint(0, C, (B,), generator=g)
    sub = torch.randint(-1, C, (B, T, 1), generator=g)
    o_out, o_losses, _, o_tot = oracle_step(orc, video, target, sub)
    out, losses, _, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    assert rel(out['logits/action'], o_out['logits/action']) < 3e-2
    assert rel(out['backbone_mean'], o_out['backbone_mean']) < 3e-2
    assert abs(float(tot) - float(o_tot)) / abs(float(o_tot)) < 3e-2
    op = dict(orc.named_parameters())
    for n in ['classifiers.action.weight', 'future_predictor.decoder.weight', 'future_predictor.gpt_model.h.0.mlp.c_fc.weight',
              'backbone.model.norm.weight', 'backbone.model.blocks.11.mlp.fc2.weight', 'backbone.model.blocks.6.attn.qkv.weight',
              'backbone.model.blocks.0.attn.proj.bias', 'backbone.model.blocks.0.mlp.fc1.bias', 'backbone.model.patch_embed.proj.weight',
              'backbone.model.pos_embed']:
        e = rel(dict(model.named_parameters())[n].grad, op[n].grad)
        assert e < 6e-2, (n, e)
