"""Training driver pieces of the hot path (reference func/train.py): model construction (:660-664), parameter groups /
optimizer / schedulers (:696-758), the data-parallel wrap (:771-778) and the per-iteration step of
``train_one_epoch`` (:203-265).  Data loading, evaluation, H5 logging and checkpoint rotation are outside the
accelerated path; inputs here are synthetic clips or tensors handed in by the caller.
"""
import logging
import time
from typing import Dict

import torch

from ..common import scheduler as sched
from ..common import utils
from ..config import Cfg, instantiate
from ..ddp import GradReducer
from ..models.base_model import BaseModel
from ..optim import FusedSGD

__all__ = ['build_model', 'build_optimizer', 'build_schedulers', 'Trainer', 'synthetic_batch', 'main', 'init_model',
           'init_from_model', 'store_checkpoint', 'load_checkpoint', 'CKPT_FNAME']
CKPT_FNAME = 'checkpoint.pth'


def build_model(cfg, num_classes: Dict[str, int], class_mappings=None, device='cuda'):
    """func/train.py:660-690: construct, initialise sub-modules from checkpoints (``train.init_from_model``), move to device."""
    model = BaseModel(cfg.model, num_classes=num_classes, class_mappings=class_mappings or {})
    init_from_model(model, cfg.get('train', Cfg()).get('init_from_model', None))
    return model.to(device)


def init_model(model, ckpt_path, modules_to_keep=None, logger=None):
    """func/train.py:457-497: non-strict initialisation of ``model`` from a checkpoint file.  Accepts the containers the
    reference accepts (``{'model': ...}``, ``{'state_dict': ...}``, VISSL's ``classy_state_dict``, or a bare state_dict --
    e.g. timm's ``jx_vit_base_patch16_224_in21k`` weights for ``backbone.model``), keeps only keys that start with one of the
    comma-separated prefixes in ``modules_to_keep`` (prefix stripped), drops shape-mismatched entries (e.g. a classifier of
    another dataset), ignores unexpected ones (HF 4.2.2's ``attn.bias`` / ``attn.masked_bias`` mask buffers).
    Returns (missing_keys, unexpected_keys)."""
    logger = logger or logging.getLogger(__name__)
    checkpoint = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    if 'model' in checkpoint:
        state_dict = checkpoint['model']
    elif 'state_dict' in checkpoint:
        state_dict = checkpoint['state_dict']
    elif 'classy_state_dict' in checkpoint:
        state_dict = checkpoint['classy_state_dict']['base_model']['model']['trunk']
    else:
        state_dict = checkpoint
    if modules_to_keep:
        prefixes = modules_to_keep.split(',')
        state_dict = {k[len(pre):]: v for k, v in state_dict.items() for pre in prefixes if k.startswith(pre)}
    else:
        state_dict = dict(state_dict)
    own = dict(model.named_parameters())
    own.update(dict(model.named_buffers()))
    for name, t in own.items():
        if name in state_dict and state_dict[name].shape != t.shape:
            logger.warning('Ckpt shape mismatch for %s (%s vs %s). Ignoring.', name, tuple(state_dict[name].shape), tuple(t.shape))
            del state_dict[name]
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    logger.warning('Could not init from %s: %s', ckpt_path, missing)
    logger.warning('Unused keys in %s: %s', ckpt_path, unexpected)
    return missing, unexpected


def init_from_model(model, spec, logger=None):
    """func/train.py:669-688: ``train.init_from_model`` = list of [path] | [module, path] | [module, prefixes, path]."""
    for elts in (spec or []):
        elts = list(elts)
        if len(elts) == 1:
            target, keep, path = model, None, elts[0]
        elif len(elts) == 2:
            target, keep, path = _get_submodule(model, elts[0]), None, elts[1]
        elif len(elts) == 3:
            target, keep, path = _get_submodule(model, elts[0]), elts[1], elts[2]
        else:
            raise ValueError(f'Incorrect formatting {elts}')
        init_model(target, path, keep, logger)


def store_checkpoint(fpaths, model, optimizer, lr_scheduler, epoch):
    """func/train.py:52-74: ``{'model', 'optimizer', 'lr_scheduler', 'epoch'}`` written by rank 0.  The model's state_dict has
    the reference's names, FusedSGD's has torch.optim.SGD's layout and the schedulers' the reference's
    ``{'base_sched_dict', 'other_stuff'}`` one, so the file resumes under the reference's loop (:760-769) and vice versa."""
    checkpoint = {'model': {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                  'optimizer': optimizer.state_dict(),
                  'lr_scheduler': lr_scheduler.state_dict() if lr_scheduler is not None else {},
                  'epoch': epoch}
    if not isinstance(fpaths, (list, tuple)):
        fpaths = [fpaths]
    for fpath in fpaths:
        logging.info('Storing ckpt at epoch %f to %s', epoch, fpath)
        if utils.get_rank() == 0:
            torch.save(checkpoint, fpath)


def load_checkpoint(fpath, model, optimizer=None, lr_scheduler=None):
    """func/train.py:760-769: strict model load, optimizer / scheduler state, returns the stored epoch."""
    checkpoint = torch.load(fpath, map_location='cpu', weights_only=False)
    sd = {k: v for k, v in checkpoint['model'].items() if not k.endswith(('.attn.bias', '.attn.masked_bias'))}   # HF 4.2.2 mask buffers
    model.load_state_dict(sd)
    if optimizer is not None and checkpoint.get('optimizer'):
        optimizer.load_state_dict(checkpoint['optimizer'])
    if lr_scheduler is not None and checkpoint.get('lr_scheduler'):
        lr_scheduler.load_state_dict(checkpoint['lr_scheduler'])
    return checkpoint.get('epoch', 0)


def _param_groups(model, lr_wd, world_size, bias_bn_wd_scale=1.0, lr_mult=1.0):
    """func/train.py:696-742: per ``opt.lr_wd`` entry two groups (names ending in 'bias' or containing '.bn' get the
    weight decay scaled), LR multiplied by the number of replicas (and by the batch size with ``opt.scale_lr_by_bs``),
    zero-LR groups dropped and their parameters frozen (``requires_grad = False``, func/train.py:735-742) -- a frozen
    backbone then skips its backward and its gradient range is never touched."""
    groups = []
    for modules, lr, wd in lr_wd:
        if not isinstance(modules, (list, tuple)):
            modules = [modules]
        named = []
        for mod_name in modules:
            mod = model if mod_name == '__all__' else _get_submodule(model, mod_name)
            named.extend((mod_name + '.' + n, p) for n, p in mod.named_parameters() if p.requires_grad)
        decay = [p for n, p in named if not (n.endswith('bias') or '.bn' in n)]
        no_decay = [p for n, p in named if (n.endswith('bias') or '.bn' in n)]
        this_lr = lr * world_size * lr_mult
        if this_lr == 0:
            for p in decay + no_decay:
                p.requires_grad = False
            continue
        groups.append({'params': decay, 'lr': this_lr, 'weight_decay': wd})
        groups.append({'params': no_decay, 'lr': this_lr, 'weight_decay': wd * bias_bn_wd_scale})
    return [g for g in groups if g['params']]


def _get_submodule(model, dotted):
    cur = model
    for part in dotted.split('.'):
        cur = getattr(cur, part)
    return cur


def build_optimizer(cfg, model, world_size=1):
    """func/train.py:744 ``hydra.utils.instantiate(cfg.opt.optimizer, params)``; ``_target_: torch.optim.SGD`` maps to the
    fused arena optimizer (same hyper-parameters, same update rule)."""
    if cfg.opt.get('classifier_only', False):                    # func/train.py:692-695 (the BN freeze there is a no-op: ViT / GPT-2 have no BN)
        assert len(cfg.opt.lr_wd) == 1 and cfg.opt.lr_wd[0][0] == 'classifier'
    lr_mult = cfg.train.batch_size if cfg.opt.get('scale_lr_by_bs', False) else 1.0      # func/train.py:718-720
    groups = _param_groups(model, cfg.opt.lr_wd, world_size, cfg.opt.get('bias_bn_wd_scale', 1.0), lr_mult)
    oc = dict(cfg.opt.optimizer)
    target = oc.pop('_target_', 'torch.optim.SGD')
    if target == 'torch.optim.SGD':
        return FusedSGD(groups, lr=groups[0]['lr'], arena=model.arena, **oc)
    from ..config import locate
    return locate(target)(groups, lr=groups[0]['lr'], **oc)


def build_schedulers(cfg, optimizer, iters_per_epoch, world_size=1):
    """func/train.py:749-758: main scheduler then the Warmup wrapper, both stepped per iteration."""
    main = instantiate(cfg.opt.scheduler, optimizer, iters_per_epoch=iters_per_epoch, world_size=world_size)
    return instantiate(cfg.opt.warmup, optimizer, main, iters_per_epoch=iters_per_epoch, world_size=world_size)


def synthetic_batch(batch_size, num_frames, num_classes, device, seed=42, feat_shape=(3, 1, 224, 224)):
    """SURVEY 8d synthetic inputs: video ~ U(-1,1) (mean=std=0.5 normalised pixels), random targets, -1 = unlabeled."""
    g = torch.Generator(device=device).manual_seed(seed)
    video = torch.rand((batch_size, num_frames) + tuple(feat_shape), device=device, generator=g) * 2 - 1
    target = torch.randint(0, num_classes, (batch_size,), device=device, generator=g)
    sub = torch.randint(-1, num_classes, (batch_size, num_frames, 1), device=device, generator=g)
    return {'video': video, 'target': {'action': target}, 'target_subclips': {'action': sub}}


class Trainer:
    """One optimisation step = reference train_one_epoch body (func/train.py:203-265):
    op(data) -> mean each loss -> weighted sum over keys with weight > 0 -> zero_grad, backward (+ overlapped gradient
    all-reduce), step -> lr_scheduler.step().  The NaN check / ``loss.item()`` host syncs of the reference are made
    optional (``sync_loss``) because they stall the launch queue."""
    def __init__(self, model, train_eval_op, optimizer, lr_scheduler=None, loss_wts=None, distributed=False,
                 bucket_bytes=64 << 20, grad_clip=None, force_reducer=False, reduce_mode='all_reduce',
                 wire_dtype=torch.float32, tail_bytes=None, reduce_transport='torch'):
        self.model, self.op, self.optimizer, self.lr_scheduler = model, train_eval_op, optimizer, lr_scheduler
        self.loss_wts = dict(loss_wts or {})
        self.fused = isinstance(optimizer, FusedSGD)
        gc = dict(grad_clip or {})
        self.max_norm = gc.get('max_norm', None)                 # conf/config.yaml train_one_epoch_fn.grad_clip_params
        self.norm_type = float(gc.get('norm_type', 2.0))
        self.world = utils.get_world_size() if distributed else 1
        self.reducer = GradReducer(model, bucket_bytes=bucket_bytes, always=force_reducer, mode=reduce_mode, wire_dtype=wire_dtype,
                                   tail_bytes=tail_bytes, transport=reduce_transport) if (self.world > 1 or force_reducer) else None
        if self.reducer is not None:
            if reduce_transport == 'abi':                        # (the C ABI's own communicator: avt_broadcast_bucket)
                self.reducer.broadcast_parameters_abi()
            else:
                GradReducer.broadcast_parameters(model)
        self.last_losses = {}
        # Early optimizer step (round 6): as backward finishes a suffix of the flat gradient buffer (and, data-parallel, its exchange is enqueued), the
        # fused SGD runs on that suffix at once -- single process: on a side stream, under the rest of backward; data-parallel: on the exchange's stream
        # behind the bucket's collective -- so the weight-sized pass (26 B per parameter, 78 % of them the head's) no longer stands behind backward.
        # Needs the step's gradient scale up front: not with gradient clipping (the norm of ALL gradients comes first).
        self.early_step = bool(self.EARLY_STEP and self.fused and self.max_norm is None and model.arena.grad.is_cuda)
        self._side_used = False
        self._tracker = None
        if self.early_step:
            self._tracker = self.reducer if self.reducer is not None else GradReducer(model, bucket_bytes=bucket_bytes)    # (world of one: segment book-keeping only)
            self._tracker.on_final = self._early_sgd

    EARLY_STEP = False          # measured neutral to slightly negative on one GPU (profiles/r06k_early_step.txt): an option, off by default
    EARLY_MIN_ELEMS = 32 << 20          # a suffix goes out once it holds this many new elements (the head at once, the ViT's blocks in two or three lots)

    def _early_sgd(self, lo, stream):
        opt = self.optimizer
        if lo > 0 and opt._early_lo - lo < self.EARLY_MIN_ELEMS:
            return                          # (deferred: a later call, or step(), covers it)
        side = self._tracker.comm_stream
        if stream is None and self.reducer is None and side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            with torch.cuda.stream(side):
                opt.step_suffix(lo)
            self._side_used = True
        else:
            opt.step_suffix(lo)             # on the exchange's stream (behind the bucket's collective), or on the compute stream after finish()

    def total_loss(self, losses):
        final = None
        for key, val in losses.items():
            wt = self.loss_wts.get(key, 0.0)
            if wt > 0:
                term = wt * torch.mean(val)
                final = term if final is None else final + term
        return final

    def _clip_coef(self, pre_scale):
        """torch.nn.utils.clip_grad_norm_ over the parameters being optimised, as one reduction over their arena ranges."""
        a = self.model.arena
        norms = []
        for g in self.optimizer.param_groups:
            for p in g['params']:
                n = a.name_of[id(p)]
                norms.append(torch.linalg.vector_norm(a.grad[a.offsets[n]:a.offsets[n] + a.sizes[n]], self.norm_type))
        total = torch.linalg.vector_norm(torch.stack(norms), self.norm_type) * pre_scale
        return float(torch.clamp(self.max_norm / (total + 1e-6), max=1.0))

    def step(self, data, sync_loss=False):
        if self._tracker is not None:
            self._tracker.start_step()
            if self.fused:
                self.optimizer.grad_scale = 1.0 / self.world
        elif self.reducer is not None:
            self.reducer.start_step()
        data, outputs, losses, accuracies = self.op(data, train_mode=True)
        loss = self.total_loss(losses)
        self.optimizer.zero_grad()          # FusedSGD: no-op (its step re-zeroes); torch optimizers drop / zero the .grad views
        loss.backward()                     # fused nodes re-attach the views and write the flat gradient buffer
        if self.reducer is not None:
            self.reducer.finish()
        scale = 1.0 / self.world
        if self.max_norm is not None:       # func/train.py:224-231 (norm of the averaged gradients)
            scale *= self._clip_coef(scale)
        if self.fused:
            self.optimizer.grad_scale = scale
        elif scale != 1.0:
            self.model.arena.grad.mul_(scale)
        if self._side_used:                 # the early launches of this step ran on the side stream
            torch.cuda.current_stream().wait_stream(self._tracker.comm_stream)
            self._side_used = False
        self.optimizer.step()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        if sync_loss:
            val = loss.item()
            if not (val == val):
                raise ValueError('Overall loss is NaN')
            return val, outputs, losses, accuracies
        return loss, outputs, losses, accuracies


def main(cfg, steps=10, batch_size=None, log_every=1, ckpt=None):
    """Synthetic-data training loop driven by a composed config (see train_net.py).  ``ckpt``: resume from this file when it
    exists and store to it at the end (the reference does both with ``checkpoint.pth`` in the run directory)."""
    dist_on, rank, world, local = utils.init_distributed_mode(cfg.get('dist_backend', None))
    device = torch.device('cuda', local)
    torch.manual_seed(cfg.get('seed', 42) + rank)
    C = cfg.get('synthetic', Cfg()).get('num_classes', 3806)
    T = cfg.data_train.num_frames
    B = batch_size or cfg.train.batch_size
    model = build_model(cfg, {'action': C}, device=device)
    optimizer = build_optimizer(cfg, model, world)
    iters_per_epoch = cfg.get('synthetic', Cfg()).get('iters_per_epoch', 100)
    lr_sched = build_schedulers(cfg, optimizer, iters_per_epoch, world)
    op = instantiate(cfg.train_eval_op, model, device, None, _recursive_=False)
    trainer = Trainer(model, op, optimizer, lr_sched, cfg.train.train_one_epoch_fn.loss_wts, distributed=dist_on,
                      grad_clip=cfg.train.train_one_epoch_fn.get('grad_clip_params', None))
    feat_shape = tuple(cfg.get('synthetic', Cfg()).get('feat_shape', (3, 1, 224, 224)))
    data = synthetic_batch(B, T, C, device, seed=cfg.get('seed', 42) + rank, feat_shape=feat_shape)
    # synthetic.uint8_source: [H, W] -> the clips start as uint8 frames (what a video decoder hands over) and go through the fused
    # GPU input pipeline every step (resize / flip / normalise / crop with per-clip random draws, func/train.py:550-569)
    u8_hw = cfg.get('synthetic', Cfg()).get('uint8_source', None)
    gpu_tf, clips_u8 = None, None
    if u8_hw:
        from ..common.gpu_transforms import GpuClipTransform
        dt = cfg.data_train
        gpu_tf = GpuClipTransform(dt.scale_h, dt.scale_w, dt.crop_size, dt.mean, dt.std, dt.get('flip_p', 0.5),
                                  dt.get('scale_pix_val', 1.0), dt.get('reverse_channels', False), train=True,
                                  emit_patches=cfg.get('synthetic', Cfg()).get('emit_patches', True))     # patch rows straight from the input kernel (no fp32 frames, no im2col)
        g = torch.Generator(device=device).manual_seed(cfg.get('seed', 42) + rank)
        clips_u8 = torch.randint(0, 256, (B, T, int(u8_hw[0]), int(u8_hw[1]), 3), device=device, dtype=torch.uint8, generator=g)
    start = 0
    import os
    if ckpt and os.path.isfile(ckpt):
        start = load_checkpoint(ckpt, model, optimizer, lr_sched)
        logging.warning('Loaded model from %s (ep %f)', ckpt, start)
    # train.captured_step=true: the step is recorded once into a hipGraph and replayed (func/graph.py; single process, fp32 synthetic clips) -- what a
    # launch-bound batch wants: at the reference's 3 clips per GPU (expts/01_ek100_avt.txt:5) the host otherwise needs as long to issue a step as the device to run it
    captured = None
    if cfg.train.get('captured_step', False) and world == 1 and gpu_tf is None and steps > 2:
        from .graph import CapturedStep
        captured = CapturedStep(trainer, data, warmup=2)            # (its two warm-up steps are training steps: they count)
        steps -= 2
    for it in range(steps):
        t0 = time.time()
        if gpu_tf is not None:
            data['video'] = gpu_tf(clips_u8)
        if captured is not None:
            loss = float(captured.step(data)[0])
            if not (loss == loss):
                raise ValueError('Overall loss is NaN')
        else:
            loss, _, _, accs = trainer.step(data, sync_loss=True)
        dt = time.time() - t0
        if rank == 0 and it % log_every == 0:
            logging.info('iter %d loss %.4f clips/s %.1f lr %.3g', it, loss, B * world / dt, optimizer.param_groups[0]['lr'])
            print(f'iter {it} loss {loss:.4f} clips/s {B * world / dt:.1f} lr {optimizer.param_groups[0]["lr"]:.3g}', flush=True)
    if rank == 0 and gpu_tf is not None:                      # (which input kernels the run used: the uint8 pipeline's patch rows leave no im2col call)
        from .. import lib as _abi
        print('abi calls: avt_im2col_patch16 %d avt_video_preproc_u8 %d' % (_abi.CALLS_BY_NAME.get('avt_im2col_patch16', 0),
                                                                            _abi.CALLS_BY_NAME.get('avt_video_preproc_u8', 0)), flush=True)
    if ckpt:
        store_checkpoint(ckpt, model, optimizer, lr_sched, start + steps / float(iters_per_epoch))
    return trainer
