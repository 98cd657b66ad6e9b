"""GPU input pipeline: the reference's per-clip CPU transform chain of func/train.py:550-592 --
``ToTensorVideo -> Resize -> RandomHorizontalFlipVideo -> (ColorJitterVideo) -> x scale_pix_val -> (reverse channels) ->
NormalizeVideo -> RandomCropVideo`` for training, ``... -> CenterCropVideo`` for evaluation -- as one fused HIP kernel over a
whole batch of uint8 clips (``avt_video_preproc_u8``).  The random draws are made on the host with the same generators and
distributions the reference's transforms use (``random.randint`` for the ``"248-280"`` size string, common/transforms.py:70-76;
``random.random() < p`` for the flip, torchvision's RandomHorizontalFlipVideo; ``torch.randint`` for the crop corner, torchvision
RandomCrop.get_params), so a seeded run draws the same sequence.

Configured from the same ``data_train`` / ``data_eval`` keys (conf/data/default.yaml: scale_h, scale_w, crop_size, mean, std, flip_p,
scale_pix_val, reverse_channels, eval_num_crops, eval_flip_crops, color_jitter_*).  Colour jitter is 0 in every AVT experiment
(conf/data/default.yaml:37-40): the zero-strength ``ColorJitterVideo`` of the TRAINING chain is reproduced for what it still does -- its
float -> PIL -> float round trip cuts the resized pixels to 8 bits (``quantize_u8`` of the fused kernel).  With non-zero strengths the
three-stage kernel chain of ``avt_video_preproc_jitter_u8`` runs instead (resize + flip -> 8-bit clip -> the four Pillow operations of
torchvision 0.8.2's ColorJitter, bit-exact -> normalise + crop); the draws follow torchvision 0.8.2's ``ColorJitter.forward``:
``torch.randperm(4)`` for the order, ``torch.tensor(1.0).uniform_(lo, hi)`` per active operation when its turn comes, one draw per clip
(the reference jitters all frames of a clip as one stacked image, common/transforms.py:399-421).
"""
import random

import torch

from .. import ops
from .patch_video import PatchVideo


class GpuClipTransform:
    def __init__(self, scale_h, scale_w=-1, crop_size=224, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), flip_p=0.5,
                 scale_pix_val=1.0, reverse_channels=False, train=True, eval_num_crops=1, eval_flip_crops=False, color_jitter_brightness=0.0, color_jitter_contrast=0.0,
                 color_jitter_saturation=0.0, color_jitter_hue=0.0, emit_patches=False, **_unused):
        # emit_patches: hand the model the patch-embedding GEMM's bf16 rows (common/patch_video.py::PatchVideo) instead of the fp32 clip tensor -- the
        # same pixels, neither the fp32 frames nor the im2col pass; single-crop outputs only (multi-crop evaluation keeps the 7-D tensor)
        self.emit_patches = bool(emit_patches)
        # torchvision ColorJitter._check_input: value v -> range [max(0, 1 - v), 1 + v] (hue: [-v, v], v <= 0.5); None when it is a no-op
        def _rng(v, center=1.0, clip0=True):
            if isinstance(v, (tuple, list)):
                lo, hi = float(v[0]), float(v[1])
            else:
                lo, hi = center - float(v), center + float(v)
                if clip0:
                    lo = max(lo, 0.0)
            return None if lo == hi == center else (lo, hi)
        self.jitter = [_rng(color_jitter_brightness), _rng(color_jitter_contrast), _rng(color_jitter_saturation),
                       _rng(color_jitter_hue, center=0.0, clip0=False)] if train else [None] * 4
        if self.jitter[3] is not None and not (-0.5 <= self.jitter[3][0] <= self.jitter[3][1] <= 0.5):
            raise ValueError('hue jitter must lie in [-0.5, 0.5]')
        if crop_size is None:
            raise NotImplementedError('the fused kernel writes a fixed-size batch: crop_size must be set')
        self.scale_h, self.scale_w = scale_h, scale_w
        self.crop = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
        self.mean, self.std = tuple(mean), tuple(std)
        self.flip_p = flip_p if train else 0.0
        self.scale_pix_val, self.reverse_channels, self.train = scale_pix_val, reverse_channels, train
        if eval_num_crops not in (1, 3):
            raise NotImplementedError(f'eval_num_crops = {eval_num_crops}: MultiCropVideo defines 1 or 3 crops (common/transforms.py:254-296)')
        self.num_crops, self.flip_crops = (1, False) if train else (eval_num_crops, bool(eval_flip_crops))
        self.quantize_u8 = bool(train)              # ColorJitterVideo is in transform_train only (func/train.py:554-557)

    @staticmethod
    def _size(v):
        """common/transforms.py:70-76: an int, or '<min>-<max>' drawn with random.randint (inclusive)."""
        if isinstance(v, int):
            return v
        lo, hi = [int(e) for e in str(v).split('-')]
        return random.randint(lo, hi)

    def draw(self, H, W):
        """One clip's (new_h, new_w, flip, crop_i, crop_j), in the order the reference's transform list consumes randomness."""
        if isinstance(self.scale_w, int) and self.scale_w == -1:                 # func/train.py:510-521
            target = self._size(self.scale_h)
            s = target * 1.0 / min(H, W)
            new_h, new_w = max(int(H * s), target), max(int(W * s), target)      # common/transforms.py:78-87
        else:
            new_h, new_w = self._size(self.scale_h), self._size(self.scale_w)
        flip = int(random.random() < self.flip_p) if self.flip_p > 0 else 0
        th, tw = self.crop
        if new_h < th or new_w < tw:
            raise ValueError(f'crop {self.crop} larger than the resized clip {(new_h, new_w)}')
        if self.train:                                                           # torchvision RandomCrop.get_params
            i = 0 if new_h == th else int(torch.randint(0, new_h - th + 1, size=(1,)).item())
            j = 0 if new_w == tw else int(torch.randint(0, new_w - tw + 1, size=(1,)).item())
        else:                                                                    # center_crop, common/transforms.py:112-121
            i, j = int(round((new_h - th) / 2.0)), int(round((new_w - tw) / 2.0))
        return new_h, new_w, flip, i, j

    def draw_jitter(self):
        """One clip's colour-jitter operations [(op id, factor)] in application order, drawn as torchvision 0.8.2's ColorJitter.forward
        does: a random permutation of (brightness, contrast, saturation, hue), each active one drawing its factor when its turn comes."""
        ops = []
        for fn_id in torch.randperm(4).tolist():
            r = self.jitter[fn_id]
            if r is not None:
                ops.append((fn_id, torch.tensor(1.0).uniform_(r[0], r[1]).item()))
        return ops

    def eval_crops(self, H, W):
        """MultiCropVideo (common/transforms.py:254-296): [(new_h, new_w, flip, i, j)] for the 1 or 3 crops, then their mirror
        images when ``eval_flip_crops`` (a mirrored crop at column j = the crop at new_w - tw - j of the mirrored frame)."""
        new_h, new_w, _, ci, cj = self.draw(H, W)
        th, tw = self.crop
        pos = [(ci, cj)] if self.num_crops == 1 else [(0, 0), (ci, cj), (new_h - th, new_w - tw)]
        out = [(new_h, new_w, 0, i, j) for i, j in pos]
        if self.flip_crops:
            out += [(new_h, new_w, 1, i, new_w - tw - j) for i, j in pos]
        return out

    def __call__(self, clips_u8, params=None, jitter=None):
        """clips_u8: uint8 (B, T, H, W, 3) on the GPU -> fp32 (B, T, 3, 1, crop_h, crop_w), the ``video`` entry of the sample
        dict the model consumes.  ``params`` (B x 5 ints) overrides the draws (tests); ``jitter`` (per clip a list of
        (op id | name, factor) in application order) overrides the colour-jitter draws."""
        B, T, H, W, _ = clips_u8.shape
        multi = (not self.train) and (self.num_crops > 1 or self.flip_crops) and params is None
        if params is None:
            if multi:
                per_clip = [self.eval_crops(H, W) for _ in range(B)]
                nc = len(per_clip[0])
                params = [tuple(c) + (b,) for b in range(B) for c in per_clip[b]]           # output clip b * nc + crop
            else:
                params = [tuple(self.draw(H, W)) + (b,) for b in range(B)]
        else:
            params = [tuple(int(v) for v in q) + ((b,) if len(q) == 5 else ()) for b, q in enumerate(params)]
        th, tw = self.crop
        for new_h, new_w, flip, ci, cj, src in params:          # the kernel trusts these: an out-of-range row would read past a frame
            if not (0 <= src < B and new_h > 0 and new_w > 0 and flip in (0, 1) and 0 <= ci and ci + th <= new_h and 0 <= cj and cj + tw <= new_w):
                raise ValueError(f'bad preprocessing parameters {(new_h, new_w, flip, ci, cj, src)} for {B} clips and a {th}x{tw} crop')
        p = torch.tensor(params, dtype=torch.int32).to(clips_u8.device, non_blocking=True)
        if jitter is None and any(r is not None for r in self.jitter):
            jitter = [self.draw_jitter() for _ in params]                    # after the geometric draws, as in the transform list's order
        if jitter is not None and any(len(j) for j in jitter):
            names = ('brightness', 'contrast', 'saturation', 'hue')
            op_ids = torch.full((len(params), 4), -1, dtype=torch.int32)
            fac = torch.zeros((len(params), 4), dtype=torch.float32)
            for b, ops_b in enumerate(jitter):
                if len(ops_b) > 4:
                    raise ValueError('at most four colour-jitter operations per clip')
                for k, (op, f) in enumerate(ops_b):
                    op = names.index(op) if isinstance(op, str) else int(op)
                    if not 0 <= op <= 3 or (op != 3 and float(f) < 0) or (op == 3 and not -0.5 <= float(f) <= 0.5):
                        raise ValueError(f'bad colour-jitter operation {(op, f)}: ids 0..3, factors >= 0, hue in [-0.5, 0.5]')
                    op_ids[b, k] = op
                    fac[b, k] = float(int(float(f) * 255) & 255) if op == 3 else float(f)   # hue: np.uint8(hue_factor * 255), the 8-bit shift
            slot_mask = sum(((1 << s) if bool((op_ids[:, s] >= 0).any()) else 0) | ((16 << s) if bool((op_ids[:, s] == 1).any()) else 0)
                            for s in range(4))                                       # host tensors: nothing is read back from the device
            as_patches = self.emit_patches and not multi
            out = ops.video_preproc_jitter(clips_u8.contiguous(), p, op_ids.to(clips_u8.device, non_blocking=True),
                                           fac.to(clips_u8.device, non_blocking=True), self.crop, self.scale_pix_val, self.mean, self.std,
                                           self.reverse_channels, max_hw=(max(q[0] for q in params), max(q[1] for q in params)),
                                           slot_mask=slot_mask, patches=as_patches)
            return PatchVideo(out, (len(params), T, 3, 1) + self.crop) if as_patches else out
        if self.emit_patches and not multi:
            return PatchVideo(ops.video_preproc(clips_u8.contiguous(), p, self.crop, self.scale_pix_val, self.mean, self.std, self.reverse_channels,
                                                quantize_u8=self.quantize_u8, patches=True), (len(params), T, 3, 1) + self.crop)
        out = ops.video_preproc(clips_u8.contiguous(), p, self.crop, self.scale_pix_val, self.mean, self.std, self.reverse_channels,
                                quantize_u8=self.quantize_u8)
        if multi:                                   # (B * crops, T, 3, 1, h, w) -> the model's 7-D (B, #clips, #crops, C, T', H, W)
            out = out.view(B, nc, T, 3, 1, *self.crop).permute(0, 2, 1, 3, 4, 5, 6)
        return out
