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

    def _score(self, logits, labels, want_accuracy):
        if logits.shape[:-1] != labels.shape:
            raise ValueError(f'logits {tuple(logits.shape)} do not line up with labels {tuple(labels.shape)}')
        loss, rank = self.xent.forward_with_rank(logits, labels)
        if not want_accuracy:
            return loss, None
        return loss, utils.accuracy_from_rank(rank, labels, topk=(1, min(5, logits.size(-1))))

    def forward(self, outputs, target, target_subclips):
        losses, accuracies = {}, {}
        for ttype, labels in target.items():
            losses[f'cls_{ttype}'], (top1, top5) = self._score(outputs[f'logits/{ttype}'], labels, True)
            accuracies[f'acc1/{ttype}'], accuracies[f'acc5/{ttype}'] = top1, top5
            past = outputs.get(f'{PAST_LOGITS_PREFIX}logits/{ttype}')
            if past is not None and target_subclips is not None:
                losses[f'past_cls_{ttype}'], _ = self._score(past, _row_mode(target_subclips[ttype]), False)
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
        outputs, aux_losses = self.model(data['video'].to(self.device, non_blocking=True), target_shape=some_target.shape)
        losses, accuracies = self.cls_loss_acc_fn(outputs, target, subclips)
        return data, outputs, {**losses, **aux_losses}, accuracies
