"""``PatchVideo``: the sample dict's ``video`` entry handed over ALREADY cut into the patch-embedding GEMM's rows.

The reference's loader hands the model an fp32 clip tensor and timm's ``PatchEmbed`` (a stride-16 Conv2d, via models/video_classification.py:213-227)
reads it; here the fused GPU input pipeline (``avt_video_preproc_u8`` with a ``patches`` output, common/transforms.py:124-170 restated) can write the
same pixels straight as the bf16 rows ``[frames * 197, 768]`` that GEMM consumes (CLS slot zero), so neither the fp32 frames nor the im2col pass exist.
The object stands in for the (B, #clips, 3, T', H, W) tensor on the few calls the model path makes on it (``to``, ``shape`` / ``size`` / ``ndim``,
merging the leading dimensions); fp32 clip tensors keep working everywhere -- this is an optional, faster form of the same input."""
import torch


class PatchVideo:
    def __init__(self, patches, shape):
        self.patches = patches                      # bf16 [frames * (P + 1), 768], frames in row-major order of the leading dimensions
        self.shape = torch.Size(shape)              # the clip tensor this stands for: (..., 3, T', H, W), T' == 1
        if self.shape[-4] != 3 or self.shape[-3] != 1 or self.shape[-2] % 16 or self.shape[-1] % 16:
            raise ValueError(f'PatchVideo stands for (..., 3, 1, H, W) frames with H, W multiples of 16 (got {tuple(shape)})')
        frames = 1
        for d in self.shape[:-4]:
            frames *= d
        rows = frames * ((self.shape[-2] // 16) * (self.shape[-1] // 16) + 1)
        if tuple(patches.shape) != (rows, 768) or patches.dtype != torch.bfloat16:
            raise ValueError(f'expected bf16 patch rows {(rows, 768)}, got {tuple(patches.shape)} {patches.dtype}')

    # ---- the tensor surface the model path touches --------------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    def dim(self):
        return len(self.shape)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    @property
    def device(self):
        return self.patches.device

    @property
    def is_cuda(self):
        return self.patches.is_cuda

    @property
    def frames(self):
        n = 1
        for d in self.shape[:-4]:
            n *= d
        return n

    def to(self, *args, **kwargs):
        return self                                 # already resident where the kernels read it

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        if tuple(shape[-4:]) != tuple(self.shape[-4:]):
            raise ValueError('PatchVideo.reshape may only regroup the leading (batch / clip) dimensions')
        n = 1
        for d in shape[:-4]:
            n *= d
        if n != self.frames:
            raise ValueError(f'cannot reshape {tuple(self.shape)} into {shape}')
        return PatchVideo(self.patches, shape)
