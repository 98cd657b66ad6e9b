"""avt_amd -- MI355X-native (gfx950) training hot path of the Anticipative Video Transformer.

Hand-written HIP kernels behind a C ABI (``include/avt_hip.h`` -> ``avt_amd/libavt_hip.so``) plus the host-side
mirror of the reference's plugin surface (``avt_amd.models.*``, ``avt_amd.func.*``, ``avt_amd.loss_fn.*``,
``avt_amd.common.*``) so the reference's Hydra ``_target_`` strings map one-to-one.
"""
__version__ = '0.1.0'
