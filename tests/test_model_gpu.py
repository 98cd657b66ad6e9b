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

from helpers import LOSS_WTS, build_hip_model, build_oracle_model, cosine, hip_step, load_golden, oracle_step, rel

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
def test_random_weights_vs_oracle(cfg, route):
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


def test_full_size_step_vs_oracle(route):
    """BASELINE config 2 architecture (ViT-B/16 + AVT-h 2048x6x4, C=3806) at B=1, T=3: logits / loss / sampled grads (both routes: conftest.ROUTES)."""
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


# ---- round 2: every BASELINE config at its own architecture ---------------------------------------------------------------
def _grad_report(model, orc, names=None, min_abs=1e-7):
    """Per-parameter gradient errors in three metrics: max-abs normalised, relative L2, cosine."""
    from helpers import cosine, rel_l2
    op = dict(orc.named_parameters())
    rows = []
    for n, p in model.named_parameters():
        if names is not None and n not in names:
            continue
        ref = op[n].grad
        if ref is None or float(ref.abs().max()) < min_abs:
            continue
        rows.append((n, rel(p.grad, ref), rel_l2(p.grad, ref), cosine(p.grad, ref)))
    return rows


def test_g2b_full_head_T15_vs_reference_golden(golden_dir):
    """BASELINE config 4's head (expts/07_ek100_avt_longer: 15 frames): in=768, Dh=2048, 6 layers, 4 heads (hd 512), T=15."""
    g = load_golden(os.path.join(golden_dir, 'g2b_full_head_T15.npz'))
    from oracle.make_golden import synth_batch
    model = build_hip_model('feat', 768, 2048, 6, 4, 3806)
    _fill(model)
    video, target, sub = synth_batch(2, 15, 3806, (768, 1, 1, 1), seed=12)
    out, losses, accs, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    assert rel(out['logits/action'], g['out/logits/action']) < TOL_OUT
    assert rel(out['past_logits/action'][:, :, ::16], g['out/past_logits/action_sub']) < TOL_OUT
    assert rel(out['future'], g['out/future']) < TOL_OUT and rel(out['past'], g['out/past']) < TOL_OUT
    assert abs(float(tot) - float(g['total_loss'])) / abs(float(g['total_loss'])) < 3e-2
    params = dict(model.named_parameters())
    for name, p in params.items():
        gn, ref = float(p.grad.float().norm()), float(g[f'gradnorm/{name}'])
        assert abs(gn - ref) / (ref + 1e-12) < 6e-2, (name, gn, ref)
    wpe = params['future_predictor.gpt_model.wpe.weight'].grad
    assert rel(wpe[:16, ::8], g['grad/future_predictor.gpt_model.wpe.weight_rows0_16']) < TOL_GRAD
    assert float(wpe[15:].abs().max()) == 0.0
    assert rel(params['future_predictor.encoder.weight'].grad[::64, ::32], g['grad/future_predictor.encoder.weight_sub']) < TOL_GRAD


def _route_is_really_taken(model, route, rows):
    """The route fixture must select what it says: 'product' at these sizes = LayerNorm kernels (no fold), 'fold' = the folded GEMMs."""
    from avt_amd.models.vit import HipViT, use_fold
    vit = next(m for m in model.modules() if isinstance(m, HipViT))
    assert use_fold(vit, rows, vit.embed_dim, vit.depth - 1) == (route == 'fold'), (route, rows, HipViT.fold_min_rows)


def _print_margins(tag, route, rows):
    w = (max(rows, key=lambda r: r[1]), max(rows, key=lambda r: r[2]), min(rows, key=lambda r: r[3]))
    print('MARGINS %s route=%s tensors=%d\n  worst max-abs %s\n  worst rel-L2 %s\n  worst cosine %s' % ((tag, route, len(rows)) + w))


def _full_arch_vs_oracle(vitc, T, B, seed, std=0.03, tol=3e-2, route=None):
    torch.manual_seed(seed)
    D = vitc[0]
    orc = build_oracle_model('vit', D, 2048, 6, 4, 3806, vit=vitc)
    for n, p in orc.backbone.named_parameters():
        with torch.no_grad():
            if p.ndim >= 2:
                p.normal_(0, std)
    model = build_hip_model('vit', D, 2048, 6, 4, 3806, vit=vitc)
    model.load_state_dict(orc.state_dict())
    if route is not None:
        _route_is_really_taken(model, route, B * T * ((vitc[3] // 16) ** 2 + 1))
    g = torch.Generator().manual_seed(seed + 10)
    C = 3806
    video = torch.rand((B, T, 3, 1, 224, 224), generator=g) * 2 - 1
    target = torch.randint(0, C, (B,), generator=g)
    sub = torch.randint(-1, C, (B, T, 1), generator=g)
    o_out, o_losses, _, o_tot = oracle_step(orc, video, target, sub)
    out, losses, _, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    print('OUTPUTS vit=%s T=%d B=%d route=%s  logits %.3e  past logits %.3e  backbone_mean %.3e  total loss %.3e' % (
        vitc[:3], T, B, route, rel(out['logits/action'], o_out['logits/action']), rel(out['past_logits/action'], o_out['past_logits/action']),
        rel(out['backbone_mean'], o_out['backbone_mean']), abs(float(tot) - float(o_tot)) / abs(float(o_tot))))
    assert rel(out['logits/action'], o_out['logits/action']) < tol
    assert rel(out['past_logits/action'], o_out['past_logits/action']) < tol
    assert rel(out['backbone_mean'], o_out['backbone_mean']) < tol
    for k in ['cls_action', 'past_cls_action', 'feat']:
        assert rel(losses[k], o_losses[k]) < tol, k
    assert abs(float(tot) - float(o_tot)) / abs(float(o_tot)) < tol
    return model, orc


def test_config2_full_arch_T10_every_gradient_vs_oracle(route):
    """BASELINE config 2 at its full architecture and T = 10 (ViT-B/16 + AVT-h 2048x6x4, C = 3806), B = 1: outputs, the three
    losses, and EVERY parameter gradient in three metrics, on BOTH routes (conftest.ROUTES: the folded GEMMs the bench's batches take and the LayerNorm-kernel
    route the product takes at this size).  Stated tolerance (bf16 activations / weights in the GEMMs, fp32
    accumulate, vs the fp32 oracle; tightened at the end of round 5): max-abs <= 3e-2 of the gradient's max-abs, relative L2 <= 2.5e-2, cosine >= 0.9995 (measured worst,
    profiles/r05y_parity_margins.txt: 1.96e-2 / 1.72e-2 / 0.99986; round 4:
    2.7e-2 / 1.9e-2 / 0.99984)."""
    model, orc = _full_arch_vs_oracle((768, 12, 12, 224), T=10, B=1, seed=21, route=route)
    rows = _grad_report(model, orc)
    assert len(rows) > 200
    _print_margins('config2 B=1 T=10', route, rows)
    bad = [r for r in rows if r[1] > 3e-2 or r[2] > 2.5e-2 or r[3] < 0.9995]
    assert not bad, bad[:10]


def test_config2_three_clips_per_gpu_every_gradient_vs_oracle(route):
    """The reference's OWN batch (expts/01_ek100_avt.txt:5: 3 clips per GPU) at the full architecture, T = 10: 5910 token rows -- the small-M tile rules
    (3-deep 64 x 64 ring / one partial round of the 8-phase kernel), and on the 'product' route the LayerNorm kernels + the unfolded backward, i.e. exactly
    what ``train_net.py`` runs with the reference's experiment file.  Outputs, the three losses and EVERY parameter gradient vs the fp32 oracle, same limits
    as the B = 1 test."""
    model, orc = _full_arch_vs_oracle((768, 12, 12, 224), T=10, B=3, seed=31, route=route)
    rows = _grad_report(model, orc)
    assert len(rows) > 200
    _print_margins('config2 B=3 T=10', route, rows)
    bad = [r for r in rows if r[1] > 3e-2 or r[2] > 2.5e-2 or r[3] < 0.9995]
    assert not bad, bad[:10]


def test_config4_full_arch_T15_vs_oracle(route):
    """BASELINE config 4 (expts/07: 15 frames per clip) at full architecture, B = 1, both routes."""
    model, orc = _full_arch_vs_oracle((768, 12, 12, 224), T=15, B=1, seed=22, route=route)
    names = {'classifiers.action.weight', 'future_predictor.decoder.weight', 'future_predictor.gpt_model.wpe.weight',
             'future_predictor.gpt_model.h.3.attn.c_attn.weight', 'backbone.model.blocks.11.attn.qkv.weight',
             'backbone.model.blocks.11.attn.qkv.bias', 'backbone.model.blocks.10.mlp.fc2.bias', 'backbone.model.blocks.5.mlp.fc1.weight',
             'backbone.model.blocks.0.norm1.weight', 'backbone.model.patch_embed.proj.weight', 'backbone.model.cls_token'}
    rows = _grad_report(model, orc, names)
    assert len(rows) == len(names)
    _print_margins('config4 B=1 T=15', route, rows)
    bad = [r for r in rows if r[1] > 3.5e-2 or r[2] > 3e-2 or r[3] < 0.9995]      # (measured worst: 2.2e-2 / 1.8e-2 / 0.99985, profiles/r05y_parity_margins.txt; 6e-2 / 6e-2 / 0.998 until round 5)
    assert not bad, bad


def test_config5_vitl_full_depth_cls_features_vs_hf_golden(golden_dir):
    """BASELINE config 5's backbone at full size (ViT-L/16: D = 1024, 24 layers, 16 heads) vs HF ViTModel."""
    g = load_golden(os.path.join(golden_dir, 'g7_vitl_cls.npz'))
    from avt_amd.models.vit import HipViT
    from oracle.avt_oracle import closed_form_fill_
    vit = HipViT(1024, 24, 16).cuda()
    closed_form_fill_(list(vit.named_parameters()))
    assert sum(p.numel() for p in vit.parameters()) == 303301632
    gen = torch.Generator().manual_seed(15)
    frames = torch.rand((1, 3, 224, 224), generator=gen) * 2 - 1
    with torch.no_grad():
        f = vit(frames.cuda())
    torch.cuda.synchronize()
    print('OUTPUTS config5 full-depth ViT-L CLS features vs HF golden: %.3e' % rel(f, g['cls_hf']))
    assert rel(f, g['cls_hf']) < 4e-2, rel(f, g['cls_hf'])        # 24 layers deep: twice the roundings of ViT-B


def test_config5_vitl_arch_step_vs_oracle(route):
    """ViT-L/16 block shapes (D = 1024, H = 16, MLP 4096) x 3 layers + the full-size head, one training step, B = 1, T = 3, both routes."""
    model, orc = _full_arch_vs_oracle((1024, 3, 16, 224), T=3, B=1, seed=23, route=route)
    rows = _grad_report(model, orc)
    _print_margins('config5 3-layer ViT-L B=1 T=3', route, rows)
    bad = [r for r in rows if r[1] > 3.5e-2 or r[2] > 3e-2 or r[3] < 0.9995]      # (measured worst: 2.2e-2 / 1.8e-2 / 0.99985, profiles/r05y_parity_margins.txt; 6e-2 / 6e-2 / 0.998 until round 5)
    assert not bad, bad[:10]


@pytest.mark.parametrize('tag,shape', [('h2_l8', (32, 64, 8, 2, 13)), ('h8_l8', (32, 128, 8, 8, 13))])
def test_g8_other_head_shapes_vs_reference_golden(golden_dir, tag, shape):
    """n_head = 2 / 8, n_layer = 8 heads (expts/13_50s_avt.txt:16-17) on the same kernels."""
    g = load_golden(os.path.join(golden_dir, f'g8_head_{tag}.npz'))
    IN, DH, L, H, C = shape
    model = build_hip_model('feat', IN, DH, L, H, C)
    _fill(model)
    out, losses, accs, tot = hip_step(model, g['in/video'].cuda(), g['in/target'].cuda(), g['in/sub'].cuda())
    assert rel(out['logits/action'], g['out/logits/action']) < TOL_OUT
    assert rel(out['past_logits/action'], g['out/past_logits/action']) < TOL_OUT
    assert abs(float(tot) - float(g['total_loss'])) / abs(float(g['total_loss'])) < 3e-2
    params = dict(model.named_parameters())
    for k in [k for k in g if k.startswith('grad/')]:
        assert rel(params[k[5:]].grad, g[k]) < TOL_GRAD, (k, rel(params[k[5:]].grad, g[k]))


@pytest.mark.parametrize('tag,H', [('h2', 2), ('h8', 8)])
def test_g8b_real_width_heads_vs_reference_golden(golden_dir, tag, H):
    """SURVEY 8f-4 at the widths the reference's experiments name: inter_dim = 2048, n_layer = 8, n_head = 2 (head_dim 1024,
    expts/04_ek100_avt_ig65m.txt:13-16) and n_head = 8 (head_dim 256, expts/13_50s_avt.txt:15-18), T = 10, C = 3806, one training
    step against the reference's BaseModel + AVTh + Basic op: outputs, total loss, EVERY parameter's gradient norm, sampled gradients."""
    g = load_golden(os.path.join(golden_dir, f'g8b_head_2048x8_{tag}.npz'))
    from oracle.make_golden import synth_batch
    L = 8
    model = build_hip_model('feat', 768, 2048, L, H, 3806)
    _fill(model)
    video, target, sub = synth_batch(2, 10, 3806, (768, 1, 1, 1), seed=31)
    out, losses, accs, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    assert rel(out['logits/action'], g['out/logits/action']) < TOL_OUT
    assert rel(out['past_logits/action'][:, :, ::16], g['out/past_logits/action_sub']) < TOL_OUT
    assert rel(out['future'], g['out/future']) < TOL_OUT and rel(out['past'], g['out/past']) < TOL_OUT
    assert abs(float(tot) - float(g['total_loss'])) / abs(float(g['total_loss'])) < 3e-2
    params = dict(model.named_parameters())
    for name, p in params.items():
        gn, ref = float(p.grad.float().norm()), float(g[f'gradnorm/{name}'])
        assert abs(gn - ref) / (ref + 1e-12) < 6e-2, (name, gn, ref)
    assert rel(params['future_predictor.gpt_model.wpe.weight'].grad[:16, ::8], g['grad/future_predictor.gpt_model.wpe.weight_rows0_16']) < TOL_GRAD
    assert rel(params['future_predictor.encoder.weight'].grad[::64, ::32], g['grad/future_predictor.encoder.weight_sub']) < TOL_GRAD
    k = f'future_predictor.gpt_model.h.{L - 1}.attn.c_attn.weight'
    assert rel(params[k].grad[::64, ::96], g[f'grad/{k}_sub']) < TOL_GRAD
    k = 'future_predictor.gpt_model.h.0.attn.c_attn.bias'
    assert rel(params[k].grad, g[f'grad/{k}']) < TOL_GRAD


@pytest.mark.parametrize('tag', ['full', 'tiny'])
def test_g12_rollout_with_gradients_vs_reference_golden(golden_dir, tag):
    """Roll-out WITH gradients (models/future_prediction.py:168-202 in training mode; output_len 3 at the full head size, 4 at a tiny one) against the
    reference's BaseModel + AVTh + Basic op run with HF's key / value cache: outputs, total loss, EVERY parameter's gradient norm, sampled gradients
    (AVTh._rollout_with_grad re-runs the differentiable head node per step instead of caching: the same maths)."""
    g = load_golden(os.path.join(golden_dir, f'g12_rollout_train_{tag}.npz'))
    from oracle.make_golden import synth_batch
    IN, DH, L, H, T, C, B, OL, seed = {'full': (768, 2048, 6, 4, 10, 3806, 2, 3, 61), 'tiny': (32, 64, 2, 4, 6, 17, 3, 4, 62)}[tag]
    model = build_hip_model('feat', IN, DH, L, H, C)
    model.future_predictor.output_len = OL
    _fill(model)
    video, target, sub = synth_batch(B, T, C, (IN, 1, 1, 1), seed=seed)
    out, losses, accs, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    step = 16 if C > 64 else 1
    assert rel(out['logits/action'], g['out/logits/action']) < TOL_OUT
    assert rel(out['past_logits/action'][:, :, ::step], g['out/past_logits/action_sub']) < TOL_OUT
    assert rel(out['future'], g['out/future']) < TOL_OUT and rel(out['past'], g['out/past']) < TOL_OUT
    assert abs(float(tot) - float(g['total_loss'])) / abs(float(g['total_loss'])) < 3e-2
    params = dict(model.named_parameters())
    for name, p in params.items():
        gn, ref = float(p.grad.float().norm()), float(g[f'gradnorm/{name}'])
        assert abs(gn - ref) / (ref + 1e-12) < 6e-2, (name, gn, ref)
    rows = T + OL - 1
    wpe = params['future_predictor.gpt_model.wpe.weight'].grad
    assert rel(wpe[:rows + 2, ::(8 if DH > 64 else 1)], g['grad/future_predictor.gpt_model.wpe.weight_rows']) < TOL_GRAD
    assert float(wpe[rows:].abs().max()) == 0.0 and float(wpe[rows - 1].abs().max()) > 0.0      # the last fed-back token sits at position T + OL - 2
    se, sc = (64, 32) if DH > 64 else (1, 1)
    assert rel(params['future_predictor.encoder.weight'].grad[::se, ::sc], g['grad/future_predictor.encoder.weight_sub']) < TOL_GRAD
    assert rel(params['future_predictor.decoder.weight'].grad[::sc, ::se], g['grad/future_predictor.decoder.weight_sub']) < TOL_GRAD
    k = f'future_predictor.gpt_model.h.{L - 1}.attn.c_attn.weight'
    assert rel(params[k].grad[::se, ::(96 if DH > 64 else 1)], g[f'grad/{k}_sub']) < TOL_GRAD
    k = 'future_predictor.gpt_model.h.0.attn.c_attn.bias'
    assert rel(params[k].grad, g[f'grad/{k}']) < TOL_GRAD


def test_config5_vitl_full_depth_backward_vs_oracle(route):
    """BASELINE config 5 at its FULL depth (ViT-L/16: D = 1024, 24 layers, 16 heads + the full-size head), B = 1, T = 2: one
    training step against the fp32 oracle with gradients sampled over the whole depth (first / middle / last blocks, every kind
    of parameter), both routes.  Limits = 1.5 x the measured worst (profiles/r06a_parity_margins.txt: 2.3e-2 / 1.8e-2 / 0.99983; 8e-2 / 6e-2 / 0.997 through
    round 5): max-abs <= 3.5e-2, relative L2 <= 2.8e-2, cosine >= 0.9995."""
    model, orc = _full_arch_vs_oracle((1024, 24, 16, 224), T=2, B=1, seed=24, std=0.02, tol=5e-2, route=route)
    names = {'classifiers.action.weight', 'future_predictor.encoder.weight', 'backbone.model.norm.weight', 'backbone.model.cls_token',
             'backbone.model.pos_embed', 'backbone.model.patch_embed.proj.weight', 'backbone.model.patch_embed.proj.bias'}
    for i in (0, 1, 11, 12, 22, 23):
        names |= {f'backbone.model.blocks.{i}.{n}' for n in ('attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight', 'mlp.fc1.weight',
                                                            'mlp.fc1.bias', 'mlp.fc2.weight', 'mlp.fc2.bias', 'norm1.weight', 'norm2.bias')}
    rows = _grad_report(model, orc, names)
    assert len(rows) >= len(names) - 2, len(rows)           # blocks.23 q-rows of non-CLS tokens etc. may be exactly zero
    _print_margins('config5 full-depth ViT-L B=1 T=2', route, rows)
    bad = [r for r in rows if r[1] > 3.5e-2 or r[2] > 2.8e-2 or r[3] < 0.9995]
    assert not bad, bad[:10]


def test_dropout_on_training_step_matches_the_oracle_given_the_same_masks():
    """The step the bench times has dropout ON (GPT-2 head: embedding, attention-probability and two residual dropouts per layer,
    p = 0.1).  The HIP kernels draw their masks from a counter-based hash of (seed, element index); tests/helpers.py restates
    that hash on the host and hands the SAME masks to the fp32 oracle, so the whole training step -- outputs, the three losses
    and every parameter gradient -- can be compared end to end with dropout active.  (The classifier's nn.Dropout uses torch's
    own generator and is switched off here; its mask statistics are covered by the per-op tests.)"""
    import itertools
    from helpers import give_oracle_the_hip_masks
    from avt_amd.models.future_prediction import AVTh
    IN, DH, L, H, C, B, T, P = 64, 128, 3, 4, 17, 3, 6, 0.1
    torch.manual_seed(1234)
    orc = build_oracle_model('feat', IN, DH, L, H, C)
    with torch.no_grad():
        for n, p in orc.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.15)
    model = build_hip_model('feat', IN, DH, L, H, C, head_drop=P)
    model.load_state_dict(orc.state_dict())
    AVTh._seed_counter = itertools.count(7)                     # the next forward draws seed = 7 * 1000003 + torch.initial_seed()
    seed = (7 * 1000003 + torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
    give_oracle_the_hip_masks(orc, seed, P)
    orc.train()
    g = torch.Generator().manual_seed(5)
    video = torch.rand((B, T, IN, 1, 1, 1), generator=g) * 2 - 1
    target = torch.randint(0, C, (B,), generator=g)
    sub = torch.randint(-1, C, (B, T, 1), generator=g)
    o_out, o_losses, _, o_tot = oracle_step(orc, video, target, sub)
    out, losses, _, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    # the masks bite: the same oracle without them is far away
    orc_nodrop = build_oracle_model('feat', IN, DH, L, H, C)
    orc_nodrop.load_state_dict(orc.state_dict(), strict=False)
    n_out, _, _, _ = oracle_step(orc_nodrop, video, target, sub)
    assert rel(n_out['logits/action'], o_out['logits/action']) > 5e-2
    assert rel(out['logits/action'], o_out['logits/action']) < TOL_OUT
    assert rel(out['past_logits/action'], o_out['past_logits/action']) < TOL_OUT
    for k in ['cls_action', 'past_cls_action', 'feat']:
        assert rel(losses[k], o_losses[k]) < 3e-2, k
    assert abs(float(tot) - float(o_tot)) / abs(float(o_tot)) < 3e-2
    rows = _grad_report(model, orc)
    assert len(rows) > 30
    # a 128-wide, 3-layer toy head amplifies bf16 rounding more than the full-size one (measured worst: 4.3e-2 / 4.3e-2 / 0.9991 on the
    # encoder weight, three layers below the loss); a wrong mask anywhere in forward or backward gives errors of order 1
    # (max-abs 7.8e-2 on the position-embedding gradient: a sum over only B = 3 bf16-rounded rows)
    bad = [r for r in rows if r[1] > 1e-1 or r[2] > 6e-2 or r[3] < 0.998]
    assert not bad, bad[:10]


def test_dropout_on_step_at_the_bench_head_matches_the_oracle_given_the_same_masks():
    """Round 4: the step bench.py times, at its real head -- 768 -> AVT-h 2048 x 6 layers x 4 heads -> 3806 classes, T = 10, the
    head's dropouts (p = 0.1) AND the classifier's dropout (p = 0.2, since round 4 drawn from the same counter-based device RNG,
    one mask over the concatenated past + future rows) all ON -- against the fp32 oracle given exactly those masks
    (tests/helpers.py restates the hash on the host), at the standard limits of the dropout-free tests."""
    import itertools
    from helpers import SequencedMaskDropout, give_oracle_the_hip_masks
    from avt_amd.models.future_prediction import AVTh
    IN, DH, L, H, C, B, T, P, PC = 768, 2048, 6, 4, 3806, 4, 10, 0.1, 0.2
    torch.manual_seed(4321)
    orc = build_oracle_model('feat', IN, DH, L, H, C)
    _fill(orc)
    model = build_hip_model('feat', IN, DH, L, H, C, head_drop=P, dropout=PC)
    model.load_state_dict(orc.state_dict())
    AVTh._seed_counter = itertools.count(11)
    seed = (11 * 1000003 + torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
    from oracle.make_golden import synth_batch
    video, target, sub = synth_batch(B, T, C, (IN, 1, 1, 1), seed=21)
    out, losses, _, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    assert model.dropout.last_seed != 0
    give_oracle_the_hip_masks(orc, seed, P)
    orc.dropout = SequencedMaskDropout(PC, model.dropout.last_seed)
    orc.train()
    # the classifier mask bites: with the head's masks only, the logits are far away
    orc.dropout.p = 0.0
    n_out, _, _, _ = oracle_step(orc, video, target, sub)
    orc.dropout.p = PC; orc.dropout.reset()
    o_out, o_losses, _, o_tot = oracle_step(orc, video, target, sub)
    assert rel(n_out['logits/action'], o_out['logits/action']) > 5e-2
    assert rel(out['logits/action'], o_out['logits/action']) < TOL_OUT
    assert rel(out['past_logits/action'], o_out['past_logits/action']) < TOL_OUT
    assert rel(out['future'], o_out['future']) < TOL_OUT and rel(out['past'], o_out['past']) < TOL_OUT
    for k in ['cls_action', 'past_cls_action', 'feat']:
        assert rel(losses[k], o_losses[k]) < 3e-2, k
    assert abs(float(tot) - float(o_tot)) / abs(float(o_tot)) < 3e-2
    rows = _grad_report(model, orc)
    assert len(rows) > 70
    print('worst max-abs %s\nworst rel-L2 %s\nworst cosine %s' % (max(rows, key=lambda r: r[1]), max(rows, key=lambda r: r[2]), min(rows, key=lambda r: r[3])))
    bad = [r for r in rows if r[1] > 4e-2 or r[2] > 3e-2 or r[3] < 0.999]
    assert not bad, bad[:10]


def test_one_node_classifier_loss_equals_the_two_node_path():
    """Training through ``Basic`` runs dropout -> classifier -> cross entropy as ONE autograd node (the labels travel to the model,
    HipLinear.forward_with_loss -> avt_linear_softmax_xent_fwd / _bwd); a caller that takes the logits from ``BaseModel`` and scores
    them with ``BasicLossAccuracy`` itself gets two nodes.  Same kernels on the same values: identical losses and accuracies,
    gradients equal up to the fp32 summation order of the deterministic split-K."""
    from avt_amd.config import Cfg
    from avt_amd.func.train_eval_ops import Basic, BasicLossAccuracy
    from oracle.make_golden import synth_batch
    IN, DH, L, H, C, B, T = 64, 128, 2, 4, 37, 3, 5
    torch.manual_seed(7)
    model = build_hip_model('feat', IN, DH, L, H, C)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.2)
    video, target, sub = synth_batch(B, T, C, (IN, 1, 1, 1), seed=3)
    video, target, sub = video.cuda(), target.cuda(), sub.cuda()
    out1, losses1, accs1, tot1 = hip_step(model, video, target, sub)
    assert not model.take_scored()                              # consumed by the loss module
    g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    model.train()
    out2, aux2 = model(video, target_shape=target.shape)
    losses2, accs2 = BasicLossAccuracy()(out2, {'action': target}, {'action': sub})
    losses2.update(aux2)
    tot2 = sum(LOSS_WTS[k] * v.mean() for k, v in losses2.items())
    tot2.backward()
    torch.cuda.synchronize()
    assert torch.equal(out1['logits/action'], out2['logits/action']) and torch.equal(out1['past_logits/action'], out2['past_logits/action'])
    for k in losses2:
        assert torch.equal(losses1[k], losses2[k]), k
    for k in accs2:
        assert float(accs1[k]) == float(accs2[k]), k
    for n, p in model.named_parameters():
        assert rel(p.grad, g1[n]) < 1e-5, n
    # a second loss on the logits themselves reaches the node as a gradient of its logits output and is added in
    model.zero_grad()
    op = Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
    _, out3, losses3, _ = op({'video': video, 'target': {'action': target}, 'target_subclips': {'action': sub}}, train_mode=True)
    extra = lambda o: 0.3 * (o['logits/action'] ** 2).mean() + 0.1 * o['past_logits/action'].sum(-1).mean()
    (sum(LOSS_WTS[k] * v.mean() for k, v in losses3.items()) + extra(out3)).backward()
    g3 = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    out4, aux4 = model(video, target_shape=target.shape)
    losses4, _ = BasicLossAccuracy()(out4, {'action': target}, {'action': sub})
    losses4.update(aux4)
    (sum(LOSS_WTS[k] * v.mean() for k, v in losses4.items()) + extra(out4)).backward()
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        assert rel(p.grad, g3[n]) < 2e-2, n                      # (the two-node path rounds the summed fp32 gradient to bf16 once more)


# ---- round 2: eval path ----------------------------------------------------------------------------------------
def test_g6a_multicrop_rollout_tiny_vit_vs_reference_golden(golden_dir):
    """7-D multi-crop video (3 crops averaged, models/base_model.py:251-273) + KV-cache roll-out (output_len_eval = 3,
    models/future_prediction.py:168-202), eval mode, against the reference's BaseModel / AVTh."""
    g = load_golden(os.path.join(golden_dir, 'g6a_rollout_multicrop_tiny.npz'))
    model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32), output_len_eval=3)
    _fill(model)
    model.eval()
    video, target = g['in/video'].cuda(), g['in/target'].cuda()
    with torch.no_grad():
        out, aux = model(video, target_shape=target.shape)
        model.future_predictor.output_len_eval = -1
        single, _ = model(video[:, :, 0], target_shape=target.shape)
    torch.cuda.synchronize()
    for k in ['logits/action', 'past_logits/action', 'future', 'past', 'future_agg', 'backbone_mean']:
        assert out[k].shape == g[f'out/{k}'].shape, k
        assert rel(out[k], g[f'out/{k}']) < TOL_OUT, (k, rel(out[k], g[f'out/{k}']))
    assert rel(aux['feat'], g['loss/feat']) < TOL_OUT
    assert rel(single['logits/action'], g['out_single_crop_no_rollout/logits/action']) < TOL_OUT
    assert rel(out['logits/action'], g['out_single_crop_no_rollout/logits/action']) > 5e-2       # both switches matter


def test_g6b_rollout_full_size_head_vs_reference_golden(golden_dir):
    g = load_golden(os.path.join(golden_dir, 'g6b_rollout_full_head.npz'))
    model = build_hip_model('feat', 768, 2048, 6, 4, 3806, output_len_eval=4)
    _fill(model)
    model.eval()
    gen = torch.Generator().manual_seed(14)
    video = (torch.rand((2, 10, 2, 768, 1, 1, 1), generator=gen) * 2 - 1).cuda()
    with torch.no_grad():
        out, aux = model(video, target_shape=(2,))
    torch.cuda.synchronize()
    assert rel(out['logits/action'], g['out/logits/action']) < TOL_OUT
    assert rel(out['future'], g['out/future']) < TOL_OUT and rel(out['past'], g['out/past']) < TOL_OUT
    assert rel(out['past_logits/action'][:, :, ::16], g['out/past_logits/action_sub']) < TOL_OUT
    assert rel(aux['feat'][:, :, ::8], g['loss/feat_sub']) < TOL_OUT
    # the roll-out WITH gradients (round 6: cache-free, one differentiable head node per step) computes the same thing as the cached eval roll-out:
    # same logits up to the bf16 rounding of a different kernel route for the single-token steps (golden G12 pins it to the reference with gradients)
    model.future_predictor.output_len_eval = -1
    model.future_predictor.output_len = 4
    with torch.enable_grad():
        out_g, _ = model(video[:, :, 0], target_shape=(2,))
    with torch.no_grad():
        out_c, _ = model(video[:, :, 0], target_shape=(2,))
    torch.cuda.synchronize()
    assert out_g['logits/action'].requires_grad and not out_c['logits/action'].requires_grad
    assert rel(out_g['logits/action'], out_c['logits/action']) < 1e-2 and rel(out_g['future'], out_c['future']) < 1e-2


# ---- round 2: structure checks -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cls_only', [True, False])
def test_layernorm_fold_equals_the_layernorm_kernel_path(cls_only):
    """HipViT.fold_layernorm (round 5: norm1 -> qkv and norm2 -> fc1 folded into the GEMMs, statistics from the producing GEMM's epilogue, scaled
    gradients, centred weight gradients -- csrc/lnfold.hip) against a LayerNorm kernel in front of every projection: same outputs, losses and
    EVERY parameter gradient (incl. the folded layers' weights, biases, gamma, beta) within the limits the oracle tests use; gamma / beta away
    from (1, 0) and rows with a common offset so that every term of the fold is exercised.  Then against the fp32 oracle itself."""
    from avt_amd.models.vit import HipViT
    torch.manual_seed(3)
    vit = (256, 3, 4, 32)
    orc = build_oracle_model('vit', 256, 64, 2, 4, 17, vit=vit)
    with torch.no_grad():
        for n, p in orc.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.08)
            elif 'norm' in n and n.endswith('weight'):
                p.uniform_(0.5, 1.5)
            elif n.endswith('bias'):
                p.normal_(0, 0.2)
        orc.backbone.model.pos_embed.add_(0.7)                # a common offset on every token row
    g = torch.Generator().manual_seed(5)
    video = torch.rand((3, 4, 3, 1, 32, 32), generator=g) * 2 - 1
    target = torch.randint(0, 17, (3,), generator=g)
    sub = torch.randint(-1, 17, (3, 4, 1), generator=g)
    res = {}
    old = (HipViT.fold_layernorm, HipViT.cls_only_last_block)
    try:
        HipViT.cls_only_last_block = cls_only
        for fold in (False, True):
            HipViT.fold_layernorm = fold
            model = build_hip_model('vit', 256, 64, 2, 4, 17, vit=vit)
            model.load_state_dict(orc.state_dict())
            out, losses, _, tot = hip_step(model, video.cuda(), target.cuda(), sub.cuda())
            res[fold] = ({k: v.detach().float().cpu() for k, v in out.items() if torch.is_tensor(v)}, float(tot),
                         {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()})
    finally:
        HipViT.fold_layernorm, HipViT.cls_only_last_block = old
    (o0, t0, g0), (o1, t1, g1) = res[False], res[True]
    for k in o0:
        assert rel(o1[k], o0[k]) < 3e-2, k
    assert abs(t1 - t0) / abs(t0) < 3e-2
    worst = 0.0
    for n in g0:
        if float(g0[n].abs().max()) == 0.0:
            assert float(g1[n].abs().max()) == 0.0, n
            continue
        e = float((g1[n] - g0[n]).abs().max() / g0[n].abs().max())
        worst = max(worst, e)
        assert e < 4e-2 and cosine(g1[n], g0[n]) > 0.999, (n, e, cosine(g1[n], g0[n]))
    # ... and the folded path against the fp32 oracle
    o_out, _, _, o_tot = oracle_step(orc, video, target, sub)
    assert rel(o1['logits/action'], o_out['logits/action']) < 3e-2 and abs(t1 - float(o_tot)) / abs(float(o_tot)) < 3e-2
    og = {n: p.grad.detach().float().cpu() for n, p in orc.named_parameters() if p.grad is not None}
    for n, gref in og.items():
        if float(gref.abs().max()) == 0.0:
            continue
        e = float((g1[n] - gref).abs().max() / gref.abs().max())
        assert e < 5e-2 and cosine(g1[n], gref) > 0.998, (n, e)



def test_fragment_major_gelu_derivative_changes_no_bit_of_the_step():
    """HipViT.frag_gelu_derivative (ABI 7: GELU' kept in the persistent GEMM's fragment-major order between fc1 forward and the fc2 data gradient) on a
    batch large enough for the persistent kernel (170 frames x 197 tokens, width 256): the fragment-major tensors are really used, and outputs, loss
    and EVERY parameter gradient equal those of the row-major path bit for bit."""
    from avt_amd.models import vit as vit_mod
    from avt_amd import ops
    torch.manual_seed(11)
    vit = (256, 3, 4, 224)
    g = torch.Generator().manual_seed(12)
    video = (torch.rand((17, 10, 3, 1, 224, 224), generator=g) * 2 - 1).cuda()
    target = torch.randint(0, 17, (17,), generator=g).cuda()
    sub = torch.randint(-1, 17, (17, 10, 1), generator=g).cuda()
    ref_model = build_hip_model('vit', 256, 64, 2, 4, 17, vit=vit)
    state = {k: v.clone() for k, v in ref_model.state_dict().items()}
    res, kinds = {}, {}
    old, real = vit_mod.HipViT.frag_gelu_derivative, vit_mod._deriv_buffer
    try:
        for frag in (False, True):
            vit_mod.HipViT.frag_gelu_derivative = frag
            seen = []
            vit_mod._deriv_buffer = lambda *a, _s=seen: (_s.append(type(real(*a))), real(*a))[1]
            model = build_hip_model('vit', 256, 64, 2, 4, 17, vit=vit)
            model.load_state_dict(state)
            out, losses, _, tot = hip_step(model, video, target, sub)
            res[frag] = ({k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}, float(tot),
                         {n: p.grad.detach().clone() for n, p in model.named_parameters()})
            kinds[frag] = seen
    finally:
        vit_mod.HipViT.frag_gelu_derivative, vit_mod._deriv_buffer = old, real
    assert kinds[True] and all(k is ops.FragTensor for k in kinds[True]), kinds[True]
    assert kinds[False] and all(k is torch.Tensor for k in kinds[False]), kinds[False]
    (o0, t0, g0), (o1, t1, g1) = res[False], res[True]
    assert t0 == t1
    for k in o0:
        assert torch.equal(o0[k], o1[k]), k
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n


def test_cls_only_last_block_equals_all_token_path():
    """The last ViT block computed for the CLS rows only gives the same features and gradients as the all-token path."""
    from avt_amd.models.vit import HipViT
    torch.manual_seed(3)
    res = {}
    frames = (torch.rand((6, 3, 48, 48)) * 2 - 1).cuda()
    dfeat = torch.randn((6, 192)).cuda()
    for flag in (True, False):
        torch.manual_seed(5)
        vit = HipViT(192, 3, 3, img_size=48).cuda()
        with torch.no_grad():
            for n, p in vit.named_parameters():
                if p.ndim >= 2:
                    p.normal_(0, 0.08)
                elif 'bias' in n:
                    p.normal_(0, 0.1)
        vit.cls_only_last_block = flag
        vit.zero_grad()
        f = vit(frames)
        f.backward(dfeat)
        torch.cuda.synchronize()
        res[flag] = (f.detach().clone(), {n: p.grad.detach().clone() for n, p in vit.named_parameters()})
    assert rel(res[True][0], res[False][0]) < 1e-2
    for n in res[True][1]:
        a, b = res[True][1][n], res[False][1][n]
        if float(b.abs().max()) < 1e-6:               # qkv bias, k part: exactly zero in the CLS path, rounding noise in the other
            assert float(a.abs().max()) < 1e-3, n
            continue
        assert rel(a, b) < 2.5e-2, (n, rel(a, b))


def test_torch_optimizer_over_a_middle_slice_of_the_arena_still_gets_its_gradients():
    """A torch optimizer that owns only the future predictor (a MIDDLE slice of the flat arena: backbone before it, classifier after
    it) drops only its own .grad views in zero_grad(set_to_none=True); the first and last parameters of the arena keep theirs.
    The fused backward must notice and re-attach them, otherwise optimizer.step() silently updates nothing (round-2 advisor finding)."""
    from avt_amd.config import Cfg
    from avt_amd.func.train import Trainer
    from avt_amd.func.train_eval_ops import Basic
    g = torch.Generator().manual_seed(9)
    data = {'video': (torch.rand((2, 4, 3, 1, 32, 32), generator=g) * 2 - 1).cuda(), 'target': {'action': torch.randint(0, 17, (2,), generator=g).cuda()},
            'target_subclips': {'action': torch.randint(-1, 17, (2, 4, 1), generator=g).cuda()}}
    torch.manual_seed(0)
    model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.1)
    for mod in (model.backbone, model.classifiers):
        for p in mod.parameters():
            p.requires_grad = False
    head = list(model.future_predictor.parameters())
    opt = torch.optim.AdamW(head, lr=1e-2)
    before = [p.detach().clone() for p in head]
    tr = Trainer(model, Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy')), opt, None, LOSS_WTS)
    for _ in range(2):
        tr.step(data)
        assert all(p.grad is not None for p in head)
        a = model.arena
        assert all(p.grad.data_ptr() == a.grad.data_ptr() + 4 * a.offsets[a.name_of[id(p)]] for p in head)
    torch.cuda.synchronize()
    moved = sum(not torch.equal(b, p.detach()) for b, p in zip(before, head))
    assert moved > 0.9 * len(head), (moved, len(head))


def test_reference_loop_order_with_a_torch_optimizer_trains():
    """forward -> optimizer.zero_grad() (set_to_none=True) -> backward -> step, the reference's order (func/train.py:221-233),
    with torch.optim.SGD on the HIP model: the fused backward re-attaches the gradient views, so the parameters move exactly
    as they do under the fused optimizer."""
    from avt_amd.config import Cfg
    from avt_amd.func.train import Trainer
    from avt_amd.func.train_eval_ops import Basic
    from avt_amd.optim import FusedSGD
    g = torch.Generator().manual_seed(9)
    video = torch.rand((2, 4, 3, 1, 32, 32), generator=g) * 2 - 1
    data = {'video': video.cuda(), 'target': {'action': torch.randint(0, 17, (2,), generator=g).cuda()},
            'target_subclips': {'action': torch.randint(-1, 17, (2, 4, 1), generator=g).cuda()}}
    states = {}
    for kind in ('torch', 'fused'):
        torch.manual_seed(0)
        model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.ndim >= 2:
                    p.normal_(0, 0.1)
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-4)
        opt = torch.optim.SGD(model.parameters(), **kw) if kind == 'torch' else FusedSGD(model.parameters(), arena=model.arena, **kw)
        op = Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
        tr = Trainer(model, op, opt, None, LOSS_WTS)
        for _ in range(3):
            tr.step(data)
        torch.cuda.synchronize()
        states[kind] = {n: p.detach().clone() for n, p in model.named_parameters()}
        moved = [n for n in before if not torch.equal(before[n], states[kind][n])]
        assert len(moved) > 0.9 * len(before), (kind, len(moved), len(before))
    for n in states['torch']:
        assert rel(states['torch'][n], states['fused'][n]) < 1e-3, n


def test_captured_step_equals_the_eager_steps():
    """Round 6 (ABI 9): a training step recorded into a hipGraph and replayed (avt_amd/func/graph.py::CapturedStep) -- dropout ON, so the masks must be
    fresh at every replay (indirect seeds: avt_amd/seeds.py, csrc/common.hpp resolve_seed), a learning rate that changes every step (avt_sgd_step_dev),
    a different batch every step.  Against the same six steps issued eagerly from the same initial state and seed counters: every loss, every parameter
    and the momentum buffer equal bit for bit."""
    import itertools
    from avt_amd.config import Cfg
    from avt_amd.func.graph import CapturedStep
    from avt_amd.func.train import Trainer
    from avt_amd.func.train_eval_ops import Basic
    from avt_amd.models import classifiers
    from avt_amd.models.future_prediction import AVTh
    from avt_amd.optim import FusedSGD
    g = torch.Generator().manual_seed(9)
    batches = []
    for _ in range(6):
        batches.append({'video': (torch.rand((3, 4, 3, 1, 32, 32), generator=g) * 2 - 1).cuda(), 'target': {'action': torch.randint(0, 17, (3,), generator=g).cuda()},
                        'target_subclips': {'action': torch.randint(-1, 17, (3, 4, 1), generator=g).cuda()}})
    order = [0, 0, 2, 3, 4, 5]                    # CapturedStep's two warm-up steps run on the batch it is given

    class Decay:                                   # a scheduler in the reference's sense: stepped once per iteration, rewrites param_groups[i]['lr']
        def __init__(self, opt): self.opt = opt
        def step(self):
            for grp in self.opt.param_groups: grp['lr'] *= 0.9

    def run(captured):
        torch.manual_seed(0)
        classifiers._seed_counter = itertools.count(1)
        AVTh._seed_counter = itertools.count(1)
        model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32), dropout=0.2, head_drop=0.1)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.ndim >= 2:
                    p.normal_(0, 0.1)
        model.train()
        assert model.dropout.p > 0 and model.future_predictor.resid_pdrop > 0
        opt = FusedSGD([{'params': [p for n, p in model.named_parameters() if not n.endswith('bias')], 'lr': 0.05, 'weight_decay': 1e-4},
                        {'params': [p for n, p in model.named_parameters() if n.endswith('bias')], 'lr': 0.02, 'weight_decay': 0.0}],
                       lr=0.05, momentum=0.9, nesterov=True, arena=model.arena)
        op = Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
        tr = Trainer(model, op, opt, Decay(opt), LOSS_WTS)
        losses = []
        if captured:
            cap = CapturedStep(tr, batches[0], warmup=2)
            assert len(cap.seeds.gens) >= 2 and opt.steps == 2
            for k in order[2:]:
                losses.append(cap.step(batches[k])[0].detach().clone())
            assert cap.replays == 4 and opt.steps == 6
        else:
            for k in order:
                losses.append(tr.step(batches[k])[0].detach().clone())
            losses = losses[2:]
        torch.cuda.synchronize()
        return [float(x) for x in losses], {n: p.detach().clone() for n, p in model.named_parameters()}, opt.momentum_buf.clone(), [grp['lr'] for grp in opt.param_groups]

    le, pe, me, lre = run(False)
    lc, pc, mc, lrc = run(True)
    assert le == lc, (le, lc)
    assert len(set(le)) == 4                       # (different batches and masks: the steps really differ)
    assert lre == lrc
    for n in pe:
        assert torch.equal(pe[n], pc[n]), n
    assert torch.equal(me, mc)


def test_grad_clip_and_frozen_groups():
    """opt.grad_clip.max_norm (func/train.py:224-231) and zero-LR groups (func/train.py:735-742) are honoured."""
    from avt_amd.config import Cfg
    from avt_amd.func.train import Trainer, build_optimizer
    from avt_amd.func.train_eval_ops import Basic
    torch.manual_seed(0)
    model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    cfg = Cfg(opt=Cfg(lr_wd=[['backbone', 0.0, 0.0], ['future_predictor', 0.05, 1e-4], ['classifiers', 0.05, 1e-4]],
                      optimizer=Cfg(_target_='torch.optim.SGD', momentum=0.9, nesterov=True), bias_bn_wd_scale=1.0,
                      scale_lr_by_bs=False, classifier_only=False), train=Cfg(batch_size=2))
    opt = build_optimizer(cfg, model, 1)
    assert all(not p.requires_grad for p in model.backbone.parameters())
    op = Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
    tr = Trainer(model, op, opt, None, LOSS_WTS, grad_clip={'max_norm': 1e-3, 'norm_type': 2})
    g = torch.Generator().manual_seed(9)
    data = {'video': (torch.rand((2, 4, 3, 1, 32, 32), generator=g) * 2 - 1).cuda(),
            'target': {'action': torch.randint(0, 17, (2,), generator=g).cuda()},
            'target_subclips': {'action': torch.randint(-1, 17, (2, 4, 1), generator=g).cuda()}}
    bb = {n: p.detach().clone() for n, p in model.backbone.named_parameters()}
    hd = model.future_predictor.encoder.weight.detach().clone()
    tr.step(data)
    torch.cuda.synchronize()
    for n, p in model.backbone.named_parameters():
        assert torch.equal(p, bb[n]), n
    step = (model.future_predictor.encoder.weight - hd).abs().max()
    assert 0 < float(step) < 0.05 * 1e-3 * 1.01 + 1e-4 * 0.05 * float(hd.abs().max()) + 1e-7      # |dp| <= lr * (clipped |g| + wd |p|)
    assert float(model.arena.grad.abs().max()) == 0.0            # consumed ranges re-zeroed, frozen ranges kept clean


def test_train_net_entry_runs_the_composed_experiment(tmp_path):
    """train_net.py composes conf/ + expts/01_ek100_avt.txt (the reference's Hydra surface) and trains on the GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'train_net.py'), '-c', os.path.join(root, 'expts', '01_ek100_avt.txt'),
                        '--steps', '3', '--batch', '2'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('iter ')]
    assert len(lines) == 3, r.stdout[-2000:]
    losses = [float(l.split('loss ')[1].split()[0]) for l in lines]
    assert all(l == l and 0 < l < 100 for l in losses), losses
    # the same experiment fed by uint8 frames through the fused GPU input pipeline (expts/01's 248-280 resize, 224 crop)
    r = subprocess.run([sys.executable, os.path.join(root, 'train_net.py'), '-c', os.path.join(root, 'expts', '01_ek100_avt.txt'),
                        '--steps', '2', '--batch', '2', 'synthetic.uint8_source=[256,456]'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([l for l in r.stdout.splitlines() if l.startswith('iter ')]) == 2, r.stdout[-2000:]
    # ... by default as patch rows straight from the input kernel: no fp32 frames, no im2col pass (round 6); synthetic.emit_patches=false keeps the tensor
    assert 'abi calls: avt_im2col_patch16 0 ' in r.stdout and 'avt_video_preproc_u8 2' in r.stdout, r.stdout[-2000:]
    # ... and with the step replayed from a hipGraph (round 6: train.captured_step=true): two eager steps on the capturing stream, then three replays
    r = subprocess.run([sys.executable, os.path.join(root, 'train_net.py'), '-c', os.path.join(root, 'expts', '01_ek100_avt.txt'),
                        '--steps', '5', '--batch', '2', 'train.captured_step=true'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('iter ')]
    assert len(lines) == 3, r.stdout[-2000:]
    losses = [float(l.split('loss ')[1].split()[0]) for l in lines]
    assert all(l == l and 0 < l < 100 for l in losses) and len(set(losses)) == 3, losses


def test_model_on_patch_rows_equals_model_on_the_fp32_clip():
    """The input pipeline's PatchVideo (bf16 patch rows straight from avt_video_preproc_u8) through BaseModel + Basic gives the outputs, losses and
    gradients of the same step on the fp32 clip tensor, bit for bit, and makes no avt_im2col_patch16 call (timm PatchEmbed via
    models/video_classification.py:213-227)."""
    from avt_amd import lib
    from avt_amd.common.gpu_transforms import GpuClipTransform
    torch.manual_seed(3)
    model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.1)
    g = torch.Generator().manual_seed(8)
    B, T = 3, 4
    u8 = torch.randint(0, 256, (B, T, 40, 56, 3), generator=g, dtype=torch.uint8).cuda()
    target, sub = torch.randint(0, 17, (B,), generator=g).cuda(), torch.randint(-1, 17, (B, T, 1), generator=g).cuda()
    params = [(40, 56, 0, 3, 11), (44, 61, 1, 9, 20), (36, 50, 1, 0, 5)]
    video = GpuClipTransform(40, -1, 32, train=True)(u8, params=params)
    pv = GpuClipTransform(40, -1, 32, train=True, emit_patches=True)(u8, params=params)
    out_a, losses_a, _, tot_a = hip_step(model, video, target, sub)
    grads_a = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    before = lib.CALLS_BY_NAME.get('avt_im2col_patch16', 0)
    out_b, losses_b, _, tot_b = hip_step(model, pv, target, sub)
    assert lib.CALLS_BY_NAME.get('avt_im2col_patch16', 0) == before
    assert torch.equal(out_a['logits/action'], out_b['logits/action']) and torch.equal(out_a['past_logits/action'], out_b['past_logits/action'])
    assert float(tot_a) == float(tot_b)
    for n, p in model.named_parameters():
        assert torch.equal(p.grad, grads_a[n]), n


def test_checkpoint_round_trip_with_the_reference_format(tmp_path):
    """a checkpoint in the reference's format ({'model', 'optimizer' (torch.optim.SGD), 'lr_scheduler', 'epoch'},
    reference parameter names) resumes on the HIP model + fused optimizer; after one step each side the HIP checkpoint loads
    back into the fp32 oracle + torch.optim.SGD with equal parameters and momentum buffers (bf16-GEMM tolerance)."""
    from avt_amd.common.scheduler import CosineLR, Warmup
    from avt_amd.func.train import load_checkpoint, store_checkpoint
    from avt_amd.optim import FusedSGD
    torch.manual_seed(0)
    orc = build_oracle_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    with torch.no_grad():
        for n, p in orc.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.1)
    kw = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-4)
    o_opt = torch.optim.SGD(orc.parameters(), **kw)
    g = torch.Generator().manual_seed(5)
    video = torch.rand((2, 4, 3, 1, 32, 32), generator=g) * 2 - 1
    target, sub = torch.randint(0, 17, (2,), generator=g), torch.randint(-1, 17, (2, 4, 1), generator=g)
    oracle_step(orc, video, target, sub)
    o_opt.step()                                                         # momentum buffers now exist
    ck = tmp_path / 'checkpoint.pth'
    torch.save({'model': orc.state_dict(), 'optimizer': o_opt.state_dict(), 'lr_scheduler': {}, 'epoch': 1.5}, ck)   # reference layout
    # ---- resume on the HIP side ----
    model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    opt = FusedSGD(model.parameters(), arena=model.arena, **kw)
    sched = Warmup(opt, CosineLR(opt, num_epochs=3, iters_per_epoch=4, world_size=1), num_epochs=1, iters_per_epoch=4, world_size=1)
    assert load_checkpoint(str(ck), model, opt, sched) == 1.5
    for n, p in model.named_parameters():
        assert torch.equal(p.detach().cpu(), dict(orc.named_parameters())[n].detach()), n
    for pg in opt.param_groups:
        pg['lr'] = 0.05
    # ---- one more step on both sides ----
    oracle_step(orc, video, target, sub)
    o_opt.step()
    hip_step(model, video.cuda(), target.cuda(), sub.cuda())
    opt.step()
    torch.cuda.synchronize()
    ck2 = tmp_path / 'checkpoint_hip.pth'
    store_checkpoint(str(ck2), model, opt, sched, 1.75)
    # ---- the HIP checkpoint loads into the torch side (reference resume code path, func/train.py:760-769) ----
    torch.manual_seed(9)
    orc2 = build_oracle_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    o_opt2 = torch.optim.SGD(orc2.parameters(), **kw)
    c = torch.load(ck2, map_location='cpu', weights_only=False)
    orc2.load_state_dict(c['model'])
    o_opt2.load_state_dict(c['optimizer'])
    assert c['epoch'] == 1.75 and set(c['lr_scheduler']) == {'base_sched_dict', 'other_stuff'}
    for (n, a), b in zip(orc.named_parameters(), orc2.parameters()):
        assert rel(b, a) < 1e-2, (n, rel(b, a))
    s1, s2 = o_opt.state_dict()['state'], o_opt2.state_dict()['state']
    assert set(s1) == set(s2)
    for i in s1:
        if float(s1[i]['momentum_buffer'].abs().max()) > 1e-6:
            assert rel(s2[i]['momentum_buffer'], s1[i]['momentum_buffer']) < 5e-2, i


def test_every_parameter_gradient_is_bit_reproducible():
    """Two identical steps give bit-identical gradients for EVERY parameter: weight matrices through split-K partial slabs reduced
    in a fixed order (avt_gemm_accum_bf16), bias / LayerNorm-affine / embedding gradients through per-workgroup partial vectors
    reduced in a fixed order (the "partials" workspaces of include/avt_hip.h); activation gradients are atomics-free.  With the
    switch off the column sums fall back to fp32 atomics: equal to rounding only."""
    from avt_amd import ops
    torch.manual_seed(0)
    model = build_hip_model('vit', 192, 128, 2, 4, 50, vit=(192, 3, 3, 48))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.1)
    g = torch.Generator().manual_seed(3)
    video = (torch.rand((6, 5, 3, 1, 48, 48), generator=g) * 2 - 1).cuda()
    target, sub = torch.randint(0, 50, (6,), generator=g).cuda(), torch.randint(-1, 50, (6, 5, 1), generator=g).cuda()
    assert ops.DETERMINISTIC_REDUCTIONS and ops.DETERMINISTIC_WGRAD
    grads = []
    for _ in range(3):
        hip_step(model, video, target, sub)
        grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters()})
    assert len(grads[0]) > 60
    for n in grads[0]:
        assert float(grads[0][n].abs().max()) > 0 or n.endswith('attn.bias') or 'masked_bias' in n, n
        assert torch.equal(grads[0][n], grads[1][n]) and torch.equal(grads[0][n], grads[2][n]), n
    ops.DETERMINISTIC_REDUCTIONS = False
    try:
        hip_step(model, video, target, sub)
        for n, p in model.named_parameters():
            assert rel(p.grad, grads[0][n]) < 1e-4, n
    finally:
        ops.DETERMINISTIC_REDUCTIONS = True


def test_g10_transformer_aggregator_vs_reference_golden(golden_dir):
    """``temporal_aggregation.Transformer`` (nn.TransformerEncoder aggregator, reference :73-147) on the head's
    kernels -- non-causal attention, ReLU + saved mask, post-norm LayerNorms -- forward and every stored gradient, eval mode."""
    from avt_amd.models.temporal_aggregation import Mean, Transformer
    g = load_golden(os.path.join(golden_dir, 'g10_transformer_agg.npz'))
    m = Transformer(32, inter_rep=64, nheads=4, nlayers=2).cuda()
    _fill(m)
    m.eval()
    feats = g['in/feats'].cuda().requires_grad_()
    agg, aux = m(feats)
    (agg * g['in/wout'].cuda()).sum().backward()
    torch.cuda.synchronize()
    assert aux == {} and m.output_dim == 64
    assert rel(agg, g['out/agg']) < TOL_OUT, rel(agg, g['out/agg'])
    assert rel(feats.grad, g['grad/feats']) < TOL_GRAD, rel(feats.grad, g['grad/feats'])
    params = dict(m.named_parameters())
    # parameter gradients: 1e-1 of max-abs -- ReLU is not smooth: a hidden unit whose pre-activation lies within bf16 rounding of
    # zero switches one token's whole contribution on or off, and with 18 tokens per unit one flip moves a row by several per cent
    from helpers import cosine
    for k in [k for k in g if k.startswith('grad/') and k != 'grad/feats']:
        assert rel(params[k[5:]].grad, g[k]) < 1e-1 and cosine(params[k[5:]].grad, g[k]) > 0.995, (k, rel(params[k[5:]].grad, g[k]))
    # train mode (dropout 0.1 on): runs, differs from eval, finite gradients
    m.train()
    m.zero_grad()
    a2, _ = m(feats.detach())
    a2.sum().backward()
    torch.cuda.synchronize()
    assert rel(a2, agg) > 1e-3 and all(torch.isfinite(p.grad).all() for p in m.parameters())
    mean, _ = Mean(32)(feats.detach())
    assert torch.equal(mean, feats.detach().mean(1))


def test_twenty_step_loss_trajectory_matches_the_oracle(route):
    """End-to-end training behaviour: 20 optimisation steps (fused SGD-nesterov + weight decay, Warmup -> CosineLR stepped per
    iteration as in func/train.py:749-758) on a fixed batch -- the HIP model's loss curve follows the fp32 oracle's under
    torch.optim.SGD with the same LR sequence (dropout off).  Tolerance 3 % per step: bf16 rounding differences compound over
    the steps but must not drift."""
    from avt_amd.common.scheduler import CosineLR, Warmup
    from avt_amd.config import Cfg
    from avt_amd.func.train import Trainer
    from avt_amd.func.train_eval_ops import Basic
    from avt_amd.optim import FusedSGD
    from oracle import avt_oracle as O
    torch.manual_seed(0)
    orc = build_oracle_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    with torch.no_grad():
        for n, p in orc.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.08)
    model = build_hip_model('vit', 128, 64, 2, 4, 17, vit=(128, 2, 2, 32))
    model.load_state_dict(orc.state_dict())
    g = torch.Generator().manual_seed(2)
    video = torch.rand((4, 4, 3, 1, 32, 32), generator=g) * 2 - 1
    target, sub = torch.randint(0, 17, (4,), generator=g), torch.randint(-1, 17, (4, 4, 1), generator=g)
    steps, base_lr = 20, 0.05
    lrs = O.lr_schedule(base_lr, 5, 15, steps)
    kw = dict(momentum=0.9, nesterov=True, weight_decay=1e-4)
    o_opt = torch.optim.SGD(orc.parameters(), lr=base_lr, **kw)
    opt = FusedSGD(model.parameters(), lr=base_lr, arena=model.arena, **kw)
    sched = Warmup(opt, CosineLR(opt, num_epochs=15, iters_per_epoch=1, world_size=1), num_epochs=5, iters_per_epoch=1, world_size=1)
    op = Basic(model, torch.device('cuda'), None, Cfg(_target_='func.train_eval_ops.BasicLossAccuracy'))
    tr = Trainer(model, op, opt, sched, LOSS_WTS)
    data = {'video': video.cuda(), 'target': {'action': target.cuda()}, 'target_subclips': {'action': sub.cuda()}}
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.future_predictor.embd_pdrop = model.future_predictor.attn_pdrop = model.future_predictor.resid_pdrop = 0.0
    o_losses, h_losses = [], []
    for i in range(steps):
        for pg in o_opt.param_groups:
            pg['lr'] = lrs[i]
        assert abs(opt.param_groups[0]['lr'] - lrs[i]) < 1e-12, (i, opt.param_groups[0]['lr'], lrs[i])
        _, _, _, tot = oracle_step(orc, video, target, sub)
        o_opt.step()
        o_losses.append(float(tot))
        loss, _, _, _ = tr.step(data, sync_loss=True)
        h_losses.append(loss)
    assert o_losses[-1] < 0.8 * o_losses[0], o_losses                          # the run actually learns
    for i, (a, b) in enumerate(zip(h_losses, o_losses)):
        assert abs(a - b) / abs(b) < 3e-2, (i, a, b)


def test_step_is_faster_than_the_eager_pytorch_restatement_on_the_same_gpu():
    """Context for the bench line (no published MI355X number exists): the fp32 restatement of the reference's model
    (timm-style explicit attention, HF GPT-2 head, torch.optim.SGD) run EAGERLY on this GPU -- fp32 as the reference trains, and
    under bf16 autocast -- against the HIP path on the same 32-clip batch of config 2.  Prints clips/s for the three; the HIP
    step must be at least 2x the autocast one."""
    import time
    torch.manual_seed(5)
    vitc, B, T, C = (768, 12, 12, 224), 32, 10, 3806
    g = torch.Generator().manual_seed(12)
    video = (torch.rand((B, T, 3, 1, 224, 224), generator=g) * 2 - 1).cuda()
    target, sub = torch.randint(0, C, (B,), generator=g).cuda(), torch.randint(-1, C, (B, T, 1), generator=g).cuda()

    def timed(step, n=3):
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return B * n / (time.perf_counter() - t0)

    orc = build_oracle_model('vit', 768, 2048, 6, 4, C, vit=vitc).cuda()
    opt = torch.optim.SGD(orc.parameters(), lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-5)

    def eager(autocast):
        def step():
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
                oracle_step(orc, video, target, sub)
            opt.step()
        return step
    r_fp32 = timed(eager(False))
    r_amp = timed(eager(True))
    del orc, opt
    torch.cuda.empty_cache()
    from avt_amd.optim import FusedSGD
    model = build_hip_model('vit', 768, 2048, 6, 4, C, vit=vitc)
    hopt = FusedSGD(model.parameters(), lr=1e-4, momentum=0.9, nesterov=True, weight_decay=1e-5, arena=model.arena)

    def hip():
        hip_step(model, video, target, sub)
        hopt.step()
    r_hip = timed(hip)
    print(f'clips/s at {B} clips x {T} frames: eager fp32 {r_fp32:.1f}, eager bf16 autocast {r_amp:.1f}, HIP path {r_hip:.1f}')
    assert r_hip > 2.0 * r_amp and r_hip > 2.0 * r_fp32, (r_fp32, r_amp, r_hip)


@pytest.mark.parametrize('vitc,T,REP', [((768, 12, 12, 224), 10, 128), ((768, 12, 12, 224), 15, 64), ((1024, 24, 16, 224), 10, 48)],
                         ids=['config2_256clips', 'config4_T15_128clips', 'config5_vitl_96clips'])
def test_bench_size_batch_matches_two_clips_tiled(vitc, T, REP):
    """The bench's own sizes (config 2: 256 clips x 10 frames = 504 320 token rows per GEMM, ~150 GB of saved activations; config 4:
    128 clips x 15 frames; config 5: ViT-L/16, 96 clips) through a
    size-independent property: a batch made of two distinct clips repeated 128 times gives every clip the logits of the 2-clip
    run (row position in a 256x256 tile, XCD tile order and persistent attention scheduling must not matter), the same mean
    losses, and -- the losses being batch means -- the same parameter gradients up to fp32 summation order."""
    torch.cuda.empty_cache()
    torch.manual_seed(7)
    C, last = 3806, vitc[1] - 1
    model = build_hip_model('vit', vitc[0], 2048, 6, 4, C, vit=vitc)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.03)
    g = torch.Generator().manual_seed(13)
    v2 = (torch.rand((2, T, 3, 1, 224, 224), generator=g) * 2 - 1).cuda()
    t2, s2 = torch.randint(0, C, (2,), generator=g).cuda(), torch.randint(-1, C, (2, T, 1), generator=g).cuda()
    names = ['classifiers.action.weight', 'future_predictor.decoder.weight', 'future_predictor.gpt_model.h.2.mlp.c_fc.weight',
             'backbone.model.norm.weight', f'backbone.model.blocks.{last}.attn.qkv.weight', 'backbone.model.blocks.6.mlp.fc1.weight',
             'backbone.model.blocks.6.mlp.fc1.bias', 'backbone.model.blocks.3.attn.qkv.bias', 'backbone.model.blocks.0.norm1.weight',
             'backbone.model.blocks.0.attn.proj.weight', 'backbone.model.patch_embed.proj.weight', 'backbone.model.pos_embed']
    params = dict(model.named_parameters())
    out2, losses2, _, tot2 = hip_step(model, v2, t2, s2)
    ref_logits = out2['logits/action'].float().clone()
    ref_past = out2['past_logits/action'].float().clone()                     # [2, T, C]: 10/11 (14/15) of the classifier's rows
    ref_grads = {n: params[n].grad.detach().clone() for n in names}
    ref_tot = float(tot2.detach())
    del out2, losses2, tot2
    vb, tb, sb = v2.repeat(REP, 1, 1, 1, 1, 1), t2.repeat(REP), s2.repeat(REP, 1, 1)
    out, losses, _, tot = hip_step(model, vb, tb, sb)
    assert vb.size(0) == 2 * REP and torch.cuda.max_memory_allocated() > 100e9      # the bench's footprint was really exercised
    lg = out['logits/action'].float()
    assert torch.equal(lg.view(REP, 2, -1), ref_logits.unsqueeze(0).expand(REP, -1, -1))
    assert torch.equal(out['past_logits/action'].float().view(REP, *ref_past.shape), ref_past.unsqueeze(0).expand(REP, *ref_past.shape))
    assert abs(float(tot.detach()) - ref_tot) / abs(ref_tot) < 1e-5
    # 1/B is folded into the bf16 loss gradient: exact for B = 256 / 128 (powers of two), one more bf16 rounding for B = 96
    # (every later bf16 rounding of the backward then falls differently: independent bf16 noise, within the stated 4e-2 gradient tolerance)
    from helpers import cosine
    tol = 2e-3 if (2 * REP) & (2 * REP - 1) == 0 else 3e-2      # 24 layers deep: 2.1e-2 measured on blocks.0.norm1, cosine 0.9998
    for n in names:
        e = rel(params[n].grad, ref_grads[n])
        assert e < tol and cosine(params[n].grad, ref_grads[n]) > 0.9995, (n, e, cosine(params[n].grad, ref_grads[n]))
