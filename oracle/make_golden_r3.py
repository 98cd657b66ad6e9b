"""Round-3 golden fixtures (tests/golden/g8b_*), produced by running the REFERENCE's own modules.

Same rules as oracle/make_golden.py / make_golden_r2.py: runs only in the build container (needs /root/reference + HF
transformers), commits only inputs / outputs as .npz; weights are the closed-form hash fill, so nothing else travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r3.py

  G8b  SURVEY 8f-4 at the widths the reference's experiments name: AVT-h with inter_dim = 2048, n_layer = 8 and
       n_head = 2  (head_dim 1024, expts/04_ek100_avt_ig65m.txt:13-16)  /  n_head = 8 (head_dim 256, expts/13_50s_avt.txt:15-18),
       on ViT-B-sized features (in = 768), T = 10, B = 2, C = 3806 -- reference BaseModel + AVTh + Basic op, one training step
       (outputs, the three losses, every parameter's gradient norm and sub-sampled gradients).
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
    with open(os.path.join(OUT, 'REPORT_r3.txt'), 'w') as f:
        f.write('\n'.join(report) + '\n')
    print('\n'.join(report))


if __name__ == '__main__':
    main()
