"""Train / eval operators on the HIP path.

Contract kept from the reference (func/train_eval_ops.py:27-145), because ``conf/config.yaml`` names these classes as
``_target_``s and the training loop unpacks their results:
  * ``Basic(model, device, dataset, cls_loss_acc_fn)``; ``op(data, train_mode) -> (data, outputs, losses, accuracies)``,
  * ``BasicLossAccuracy(dataset, device, balance_classes)``; ``fn(outputs, target, target_subclips) -> (losses, accuracies)``
    with UN-reduced losses keyed ``cls_<type>`` / ``past_cls_<type>`` and scalar ``acc1/<type>`` / ``acc5/<type>``,
  * the model's auxiliary losses (``feat``) are merged into ``losses``.
The implementation is this package's own: one fused softmax-cross-entropy launch per logits tensor yields the loss AND the
rank of the target among the logits, from which top-1 / top-5 follow without a second pass (``xent.hip``); the per-frame
labels of a sub-clip reduce to their mode for the "past" loss; inputs are staged on the device once per call.
"""
import torch
import torch.nn as nn

from ..common import utils
from ..config import instantiate
from ..loss_fn.multidim_xentropy import MultiDimCrossEntropy
from ..models.base_model import PAST_LOGITS_PREFIX

IGNORE = -1          # label of un-annotated frames / clips (reference: ignore_index=-1 everywhere on this path)


class NoLossAccuracy(nn.Module):
    """conf option for feature extraction runs: nothing to score."""
    def __init__(self, *_, **__):
        super().__init__()

    def forward(self, *_, **__):
        return {}, {}


def _row_mode(labels):
    """Most frequent value along the last dim, ties -> the smallest value (torch.mode's convention, which the reference uses
    to turn per-frame labels of a sub-clip into one label, func/train_eval_ops.py:66-68)."""
    if labels.size(-1) == 1:
        return labels[..., 0]
    return torch.mode(labels, dim=-1).values


class BasicLossAccuracy(nn.Module):
    def __init__(self, dataset=None, device=None, balance_classes=False):
        super().__init__()
        if balance_classes:
            raise NotImplementedError('balance_classes (class-weighted CE) is outside the accelerated path')
        self.xent = MultiDimCrossEntropy(ignore_index=IGNORE, reduction='none')

    def _score(self, logits, labels, want_accuracy, done=None):
        if logits.shape[:-1] != labels.shape:
            raise ValueError(f'logits {tuple(logits.shape)} do not line up with labels {tuple(labels.shape)}')
        loss, rank = done if done is not None else self.xent.forward_with_rank(logits, labels)
        if not want_accuracy:
            return loss, None
        return loss, utils.accuracy_from_rank(rank, labels, topk=(1, min(5, logits.size(-1))))

    @staticmethod
    def labels_per_group(target, target_subclips):
        """{target type: {output key prefix: labels}}: what every logits tensor of the model will be scored against -- the clip
        label for the future rows, the mode of the per-frame labels for the past rows (func/train_eval_ops.py:46-77).  The
        training operator hands this to the model so that classifier and cross entropy run as one node."""
        out = {}
        for ttype, labels in target.items():
            out[ttype] = {'': labels}
            if target_subclips is not None and ttype in target_subclips:
                out[ttype][PAST_LOGITS_PREFIX] = _row_mode(target_subclips[ttype])
        return out

    def forward(self, outputs, target, target_subclips, scored=None):
        """scored: {(key prefix, type): (un-reduced loss, rank)} already produced by the model's fused classifier + CE nodes for
        exactly the labels of ``labels_per_group`` (BaseModel.take_scored()); anything missing is scored here."""
        scored = scored or {}
        groups = self.labels_per_group(target, target_subclips)
        losses, accuracies = {}, {}
        for ttype, labels in target.items():
            losses[f'cls_{ttype}'], (top1, top5) = self._score(outputs[f'logits/{ttype}'], labels, True, scored.get(('', ttype)))
            accuracies[f'acc1/{ttype}'], accuracies[f'acc5/{ttype}'] = top1, top5
            past = outputs.get(f'{PAST_LOGITS_PREFIX}logits/{ttype}')
            if past is not None and PAST_LOGITS_PREFIX in groups[ttype]:
                losses[f'past_cls_{ttype}'], _ = self._score(past, groups[ttype][PAST_LOGITS_PREFIX], False, scored.get((PAST_LOGITS_PREFIX, ttype)))
        return losses, accuracies


class Basic:
    def __init__(self, model, device, dataset, cls_loss_acc_fn, reg_criterion=None):
        self.model, self.device = model, device
        self.cls_loss_acc_fn = instantiate(cls_loss_acc_fn, dataset, device)
        # reg_criterion belongs to the regression head (dense anticipation), which BaseModel rejects

    def _to_device(self, tensors):
        return {k: v.to(self.device, non_blocking=True) for k, v in tensors.items()}

    def __call__(self, data, train_mode: bool = True):
        if not isinstance(data, dict):                       # (video, target) pairs of plain classification datasets
            video, labels = data
            data = {'video': video, 'target': {'action': labels} if torch.is_tensor(labels) else labels,
                    'idx': torch.full_like(labels if torch.is_tensor(labels) else next(iter(labels.values())), -1)}
        self.model.train(train_mode)
        target = self._to_device(data['target'])
        subclips = self._to_device(data['target_subclips']) if 'target_subclips' in data else None
        some_target = next(iter(target.values()))
        video = data['video'].to(self.device, non_blocking=True)
        # training: the model gets the labels, so its classifier, the dropout in front of it and the cross entropy run as ONE
        # autograd node (HipLinear.forward_with_loss -> avt_linear_softmax_xent_fwd / _bwd); the loss module then only reduces
        fuse = (train_mode and torch.is_grad_enabled() and isinstance(self.cls_loss_acc_fn, BasicLossAccuracy)
                and hasattr(self._bare_model(), 'take_scored'))
        if fuse:
            outputs, aux_losses = self.model(video, target_shape=some_target.shape,
                                             cls_targets=self.cls_loss_acc_fn.labels_per_group(target, subclips))
            losses, accuracies = self.cls_loss_acc_fn(outputs, target, subclips, scored=self._bare_model().take_scored())
        else:
            outputs, aux_losses = self.model(video, target_shape=some_target.shape)
            losses, accuracies = self.cls_loss_acc_fn(outputs, target, subclips)
        return data, outputs, {**losses, **aux_losses}, accuracies

    def _bare_model(self):
        return getattr(self.model, 'module', self.model)          # a DistributedDataParallel-style wrapper keeps the model in .module
