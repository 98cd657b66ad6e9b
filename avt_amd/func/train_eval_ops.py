"""Train/eval ops with the reference's contract (func/train_eval_ops.py:27-145): ``BasicLossAccuracy`` and
``Basic(model, device, dataset, cls_loss_acc_fn)``; ``__call__(data, train_mode) -> (data, outputs, losses, accuracies)``
with un-reduced losses keyed ``cls_<type>``, ``past_cls_<type>``, plus the model's aux losses (``feat``)."""
from typing import Dict, Tuple, Union

import torch
import torch.nn as nn

from ..common import utils
from ..config import instantiate
from ..loss_fn.multidim_xentropy import MultiDimCrossEntropy
from ..models.base_model import PAST_LOGITS_PREFIX


class NoLossAccuracy(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *args, **kwargs):
        return {}, {}


class BasicLossAccuracy(nn.Module):
    def __init__(self, dataset=None, device=None, balance_classes=False):
        super().__init__()
        if balance_classes:
            raise NotImplementedError('balance_classes (class-weighted CE) is outside the accelerated path')
        self.cls_criterion = MultiDimCrossEntropy(ignore_index=-1, reduction='none')

    def forward(self, outputs, target, target_subclips):
        losses, accuracies = {}, {}
        for tgt_type, tgt_val in target.items():
            logits = outputs[f'logits/{tgt_type}']
            assert logits.ndim == tgt_val.ndim + 1
            losses[f'cls_{tgt_type}'], rank = self.cls_criterion.forward_with_rank(logits, tgt_val)
            acc1, acc5 = utils.accuracy_from_rank(rank, tgt_val, topk=(1, min(5, logits.size(-1))))
            accuracies[f'acc1/{tgt_type}'] = acc1
            accuracies[f'acc5/{tgt_type}'] = acc5
            past_key = f'{PAST_LOGITS_PREFIX}logits/{tgt_type}'
            if past_key in outputs and target_subclips is not None:
                past_logits = outputs[past_key]
                past_target = torch.mode(target_subclips[tgt_type], -1)[0]
                assert past_logits.shape[:-1] == past_target.shape, (
                    f'past logits {past_logits.shape} and past targets {past_target.shape} must match')
                losses[f'past_cls_{tgt_type}'] = self.cls_criterion(past_logits, past_target)
        return losses, accuracies


class Basic:
    def __init__(self, model, device, dataset, cls_loss_acc_fn, reg_criterion=None):
        super().__init__()
        self.model = model
        self.device = device
        self.cls_loss_acc_fn = instantiate(cls_loss_acc_fn, dataset, device)
        del reg_criterion

    def _basic_preproc(self, data, train_mode):
        if not isinstance(data, dict):
            video, target = data
            data = {'video': video, 'target': target, 'idx': -torch.ones_like(target)}
        if train_mode:
            self.model.train()
        else:
            self.model.eval()
        return data

    def __call__(self, data: Union[Dict[str, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]], train_mode: bool = True):
        data = self._basic_preproc(data, train_mode)
        video = data['video'].to(self.device, non_blocking=True)
        target = {k: v.to(self.device, non_blocking=True) for k, v in data['target'].items()}
        outputs, aux_losses = self.model(video, target_shape=next(iter(target.values())).shape)
        if 'target_subclips' in data:
            target_subclips = {k: v.to(self.device, non_blocking=True) for k, v in data['target_subclips'].items()}
        else:
            target_subclips = None
        losses, accuracies = self.cls_loss_acc_fn(outputs, target, target_subclips)
        losses.update(aux_losses)
        return data, outputs, losses, accuracies
