"""Round-6 golden fixture (tests/golden/g12_*), produced by running the REFERENCE's own modules.

Same rules as oracle/make_golden.py (runs only in the build container: needs /root/reference + HF transformers; commits only inputs /
outputs as .npz; weights are the closed-form hash fill, so nothing else travels).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r6.py

  G12  roll-out WITH gradients (models/future_prediction.py:168-202 in training mode): the reference's BaseModel + AVTh + Basic op with
       ``output_len = 3`` -- three GPT-2 calls chained through HF's ``past_key_values``, each fed the previous call's last hidden state -- one
       training step at the full head size (in = 768, inter_dim = 2048, 6 layers, 4 heads, T = 10, B = 2, C = 3806) and at a tiny size
       (in = 32, inter_dim = 64, 2 layers, output_len = 4): outputs, the three losses, every parameter's gradient norm, sub-sampled gradients.
       Also pins oracle.OracleAVTh's cache-free restatement of the same roll-out, gradients included.
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

OUT = os.environ.get('AVT_GOLDEN_OUT', G.OUT)
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
    for tag, (IN, DH, L, H, T, C, B, OL, seed) in {'full': (768, 2048, 6, 4, 10, 3806, 2, 3, 61), 'tiny': (32, 64, 2, 4, 6, 17, 3, 4, 62)}.items():
        cfg = G.model_cfg(Cfg(_target_='golden_helpers.FeatBackbone'), IN, DH, L, H)
        cfg['future_predictor']['output_len'] = OL
        ref = ref_bm.BaseModel(cfg, {'action': C}, {})
        O.closed_form_fill_(list(ref.named_parameters()))
        video, target, sub = G.synth_batch(B, T, C, (IN, 1, 1, 1), seed=seed)
        res = G.run_reference(ref, ops_mod, video, target, sub, loss_wts)
        keep = {k: v for k, v in res.items() if k.startswith(('loss/', 'acc/', 'total'))}
        step = 16 if C > 64 else 1
        keep['out/logits/action'] = res['out/logits/action']
        keep['out/past_logits/action_sub'] = res['out/past_logits/action'][:, :, ::step].contiguous()
        keep['out/future'], keep['out/past'] = res['out/future'], res['out/past']
        params = dict(ref.named_parameters())
        for n, p in params.items():
            keep[f'gradnorm/{n}'] = p.grad.detach().norm()
        rows = T + OL - 1                                                   # positions the roll-out reaches: wpe rows beyond stay zero
        keep['grad/future_predictor.gpt_model.wpe.weight_rows'] = params['future_predictor.gpt_model.wpe.weight'].grad.detach()[:rows + 2, ::(8 if DH > 64 else 1)].contiguous()
        se, sc = (64, 32) if DH > 64 else (1, 1)
        keep['grad/future_predictor.encoder.weight_sub'] = params['future_predictor.encoder.weight'].grad.detach()[::se, ::sc].contiguous()
        keep['grad/future_predictor.decoder.weight_sub'] = params['future_predictor.decoder.weight'].grad.detach()[::sc, ::se].contiguous()
        k = f'future_predictor.gpt_model.h.{L - 1}.attn.c_attn.weight'
        keep[f'grad/{k}_sub'] = params[k].grad.detach()[::se, ::(96 if DH > 64 else 1)].contiguous()
        keep['grad/future_predictor.gpt_model.h.0.attn.c_attn.bias'] = params['future_predictor.gpt_model.h.0.attn.c_attn.bias'].grad.detach().clone()
        np.savez_compressed(os.path.join(OUT, f'g12_rollout_train_{tag}.npz'), **G.to_np(keep))
        # the oracle's cache-free restatement: same outputs, same gradients
        orc = O.OracleBaseModel(O.OracleIdentityBackbone(), O.OracleAVTh(IN, output_len=OL, inter_dim=DH, n_layer=L, n_head=H, embd_pdrop=0.,
                                attn_pdrop=0., resid_pdrop=0.), IN, {'action': C}, dropout=0.0)
        O.closed_form_fill_(list(orc.named_parameters()))
        orc.train()
        oo, aux = orc(video, target_shape=target.shape)
        ol, _ = O.basic_loss_accuracy(oo, {'action': target}, {'action': sub})
        ol.update(aux)
        tot = O.total_loss(ol, loss_wts)
        orc.zero_grad()
        tot.backward()
        d = float((oo['logits/action'] - res['out/logits/action']).abs().max())
        worst = 0.0
        for n, p in orc.named_parameters():
            gref = params[n].grad
            worst = max(worst, float((p.grad - gref).abs().max() / (gref.abs().max() + 1e-30)))
        report.append(f'G12 {tag} (output_len {OL}, T {T}, inter_dim {DH}, {L} layers): total loss {float(res["total_loss"]):.6f}; cache-free restatement vs the '
                      f'reference (HF past_key_values): max|dlogits| = {d:.3e}, |dtotal| = {abs(float(tot) - float(res["total_loss"])):.3e}, worst gradient (max-abs / max-abs) = {worst:.3e}')
        assert d < 2e-4 and worst < 2e-4, (d, worst)
        del ref, orc
    with open(os.path.join(OUT, 'REPORT_r6.txt'), 'w') as f:
        f.write('\n'.join(report) + '\n')
    print('\n'.join(report))


if __name__ == '__main__':
    main()
