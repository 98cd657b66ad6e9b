"""The overall AVT model (reference models/base_model.py:17-273): backbone -> spatial mean -> temporal aggregator ->
future predictor -> dropout -> classifier(s), with multi-crop averaging.  Same constructor
(``model_cfg, num_classes, class_mappings``), ``forward(video, target_shape=) -> (outputs, aux_losses)`` key set and
state_dict names as the reference; sub-modules are built from the same ``_target_`` config nodes (resolved by
``avt_amd.config.instantiate`` to the HIP-backed mirrors).  All parameters live in one flat arena
(``avt_amd.arena``) shared by the sub-modules, which is what the fused optimizer and the bucketed gradient
all-reduce operate on.
"""
import operator
from typing import Dict, Tuple

import torch
import torch.nn as nn

from ..arena import get_arena
from ..config import instantiate

CLS_MAP_PREFIX = 'cls_map_'
PAST_LOGITS_PREFIX = 'past_'


class BaseModel(nn.Module):
    def __init__(self, model_cfg, num_classes: Dict[str, int], class_mappings: Dict[Tuple[str, str], torch.FloatTensor]):
        super().__init__()
        _backbone_full = instantiate(model_cfg.backbone, num_classes=1)
        if model_cfg.backbone_last_n_modules_to_drop > 0:
            raise NotImplementedError('backbone_last_n_modules_to_drop > 0 applies to the 3D-CNN backbones (out of scope)')
        self.backbone = _backbone_full
        if 'output_dim' in dir(self.backbone):
            backbone_dim = self.backbone.output_dim
        else:
            backbone_dim = model_cfg.backbone_dim
        self.mapper_to_inter = None
        if model_cfg.intermediate_featdim is None:
            model_cfg.intermediate_featdim = backbone_dim
        if backbone_dim != model_cfg.intermediate_featdim:
            raise NotImplementedError('mapper_to_inter (backbone_dim != intermediate_featdim) is not on the AVT path')
        self.temporal_aggregator = instantiate(model_cfg.temporal_aggregator, in_features=model_cfg.intermediate_featdim)
        self.reset_temp_agg_feat_dim = nn.Sequential()
        temp_agg_output_dim = self.temporal_aggregator.output_dim
        if model_cfg.same_temp_agg_dim and temp_agg_output_dim != model_cfg.intermediate_featdim:
            raise NotImplementedError('same_temp_agg_dim projection is not on the AVT path')
        self.future_predictor = instantiate(model_cfg.future_predictor, in_features=temp_agg_output_dim, _recursive_=False)
        self.project_mlp = nn.Sequential()
        if model_cfg.project_dim_for_nce is not None:
            raise NotImplementedError('project_dim_for_nce (contrastive SSL variant) is out of scope')
        self.temporal_aggregator_after_future_pred = instantiate(model_cfg.temporal_aggregator_after_future_pred,
                                                                 self.future_predictor.output_dim)
        self.dropout = nn.Dropout(model_cfg.dropout)
        cls_input_dim = self.temporal_aggregator_after_future_pred.output_dim
        self.classifiers = nn.ModuleDict()
        self.num_classes = num_classes
        for i, (cls_type, cls_dim) in enumerate(num_classes.items()):
            if model_cfg.use_cls_mappings and i > 0:
                break
            self.classifiers.update({cls_type: instantiate(model_cfg.classifier, in_features=cls_input_dim,
                                                           out_features=cls_dim)})
        for (src, dst), mapping in class_mappings.items():
            self.register_buffer(f'{CLS_MAP_PREFIX}{src}_{dst}', mapping)
        self.regression_head = None
        if model_cfg.add_regression_head:
            raise NotImplementedError('regression head (dense anticipation) is out of scope')
        self._initialize_weights()
        self.cfg = model_cfg

    def _initialize_weights(self):
        """reference :110-127 -- every nn.Linear (and the Linear-compatible classifier) <- N(0, 0.01), bias 0;
        GPT-2 Conv1D weights keep their HF init."""
        from .classifiers import HipLinear
        for m in self.modules():
            if isinstance(m, (nn.Linear, HipLinear)):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ---- arena / gradient plumbing ---------------------------------------------------------------------------------
    @property
    def arena(self):
        return get_arena(self)

    def zero_grad(self, set_to_none: bool = False):
        """Gradients accumulate atomically into one flat fp32 buffer: zeroing is a single memset."""
        if next(self.parameters()).is_cuda:
            arena = get_arena(self)
            arena.attach_grads()
            arena.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def forward_singlecrop(self, video, target_shape=None):
        outputs, aux_losses = {}, {}
        batch_size, num_clips = video.size(0), video.size(1)
        video = video.flatten(0, 1)
        feats = self.backbone(video)
        outputs['backbone'] = feats
        feats = torch.mean(feats, [-1, -2])
        outputs['backbone_mean'] = torch.mean(feats, [-1])
        feats = feats.permute((0, 2, 1))
        feats_agg, agg_losses = self.temporal_aggregator(feats)
        aux_losses.update(agg_losses)
        feats_agg = self.reset_temp_agg_feat_dim(feats_agg)
        outputs['temp_agg'] = feats_agg
        outputs['temp_agg_projected'] = self.project_mlp(feats_agg)
        if num_clips > 1:
            assert (feats_agg.ndim == 2) or (feats_agg.ndim == 3 and feats_agg.size(1) == 1), (
                'Should be using some temporal aggregation when using clips')
            feats_agg = feats_agg.reshape((batch_size, num_clips) + feats_agg.shape[1:])
            if feats_agg.ndim == 4:
                feats_agg = torch.flatten(feats_agg, 1, 2)
        feats_past = feats_agg
        feats_past, feats_future, future_losses, endpoints = self.future_predictor(feats_past, target_shape)
        aux_losses.update(future_losses)
        outputs.update(endpoints)
        outputs['future'] = feats_future
        outputs['past'] = feats_past
        if self.cfg.classifier_on_past:
            outputs.update(self._apply_classifier(self.dropout(feats_past), outputs_prefix=PAST_LOGITS_PREFIX))
        outputs['future_projected'] = self.project_mlp(feats_agg)
        feats_future_agg, future_agg_losses = self.temporal_aggregator_after_future_pred(feats_future)
        aux_losses.update(future_agg_losses)
        outputs['future_agg'] = feats_future_agg
        outputs.update(self._apply_classifier(self.dropout(feats_future_agg)))
        return outputs, aux_losses

    def _apply_classifier(self, input_feat, outputs_prefix=''):
        outputs = {}
        for key in self.num_classes.keys():
            if key in self.classifiers:
                outputs[f'{outputs_prefix}logits/{key}'] = self.classifiers[key](input_feat)
            else:
                src_key = next(iter(self.classifiers.keys()))
                mapper = operator.attrgetter(f'{CLS_MAP_PREFIX}{key}_{src_key}')(self)
                outputs[f'{outputs_prefix}logits/{key}'] = torch.mm(outputs[f'{outputs_prefix}logits/{src_key}'], mapper)
        return outputs

    def forward(self, video, *args, **kwargs):
        """video: (B, #clips, C, T, H, W) or (B, #clips, #crops, C, T, H, W); crops are averaged (reference :240-273)."""
        if next(self.parameters()).is_cuda:
            arena = get_arena(self)
            arena.refresh_shadow()
            if torch.is_grad_enabled():
                arena.attach_grads()
        if video.ndim == 6:
            video_crops = [video]
        elif video.ndim == 7 and video.size(2) == 1:
            video_crops = [video.squeeze(2)]
        elif video.ndim == 7:
            video_crops = torch.unbind(video, dim=2)
        else:
            raise NotImplementedError('Unsupported size %s' % (video.shape,))
        feats_losses = [self.forward_singlecrop(el, *args, **kwargs) for el in video_crops]
        if len(feats_losses) == 1:
            return feats_losses[0]
        feats, losses = zip(*feats_losses)
        feats = {k: torch.mean(torch.stack([d[k] for d in feats], dim=0), dim=0) for k in feats[0]}
        losses = {k: torch.mean(torch.stack([d[k] for d in losses], dim=0), dim=0) for k in losses[0]}
        return feats, losses
