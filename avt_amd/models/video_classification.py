"""Per-frame backbones (reference models/video_classification.py:213-257).  ``TIMMModel`` keeps the reference's
constructor (``num_classes, model_type, drop_cls``) and ``forward(video (N,C,T,H,W)) -> (N,C',T,1,1)`` contract; the
network underneath is the HIP ViT (``avt_amd.models.vit.HipViT``) instead of ``timm.create_model``.  The 3D-CNN /
BNInception builders of the reference file are outside the accelerated path."""
import torch.nn as nn

from ..common.patch_video import PatchVideo
from .vit import HipViT

VIT_CONFIGS = {
    # timm model name -> (embed_dim, depth, num_heads)    [conf/model/backbone/avt_b.yaml:3, avt_b_in21k.yaml:3]
    'vit_base_patch16_224': (768, 12, 12),
    'vit_base_patch16_224_in21k': (768, 12, 12),
    'vit_large_patch16_224': (1024, 24, 16),
    'vit_large_patch16_224_in21k': (1024, 24, 16),
}


def process_each_frame(model, video, *args, **kwargs):
    """(B, C, T, H, W) -> run ``model`` on every frame -> (B, C', T, 1, 1)   (reference :213-227)"""
    batch_size, time_dim = video.size(0), video.size(2)
    if isinstance(video, PatchVideo):            # T' == 1: the frames are already in (clip, frame) order, as patch rows
        flat = video
    else:
        flat = video.transpose(1, 2).flatten(0, 1)
    feats = model(flat, *args, **kwargs)
    return feats.view((batch_size, time_dim) + feats.shape[1:]).transpose(1, 2).unsqueeze(-1).unsqueeze(-1)


class FrameLevelModel(nn.Module):
    def __init__(self, num_classes: int, model: nn.Module = None):
        del num_classes
        super().__init__()
        self.model = model

    def forward(self, video, *args, **kwargs):
        return process_each_frame(self.model, video, *args, **kwargs)


class TIMMModel(FrameLevelModel):
    def __init__(self, num_classes, model_type='vit_base_patch16_224', drop_cls=True, img_size=224, **vit_kwargs):
        super().__init__(num_classes)
        if not drop_cls:
            raise NotImplementedError('only the headless (num_classes=0) ViT is on the accelerated path')
        if model_type in VIT_CONFIGS:
            dim, depth, heads = VIT_CONFIGS[model_type]
        else:                                   # e.g. tiny test configurations: embed_dim/depth/num_heads passed explicitly
            dim, depth, heads = vit_kwargs['embed_dim'], vit_kwargs['depth'], vit_kwargs['num_heads']
        self.model = HipViT(dim, depth, heads, img_size=img_size)

    @property
    def output_dim(self):
        return self.model.embed_dim


class IdentityFeatures(FrameLevelModel):
    """Backbone for pre-extracted features (config 1, expts/02_ek100_avt_tsn): (N, C, T, 1, 1) passes through."""
    def __init__(self, num_classes=None):
        super().__init__(num_classes)

    def forward(self, video):
        if isinstance(video, PatchVideo):
            raise TypeError('pre-extracted features cannot arrive as patch rows')
        return video
