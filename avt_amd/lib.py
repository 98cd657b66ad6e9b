"""ctypes binding of libavt_hip.so (the C ABI declared in include/avt_hip.h).

The product path has NO fallback: if the shared library is missing or an entry point is absent, importing/using
the ops raises.  ``build()`` in ``__graft_entry__.py`` (or ``make -C avt_amd/csrc``) produces the library.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AVT_HIP_LIB') or os.path.join(_HERE, 'libavt_hip.so')      # AVT_HIP_LIB: A/B a differently built library (lab use)
ABI_VERSION = 9

_P, _I, _F, _L, _U64, _SZ = c_void_p, c_int, c_float, c_long, c_uint64, ctypes.c_size_t

# name -> argtypes (restype is int unless noted); must mirror include/avt_hip.h exactly
SIGNATURES = {
    'avt_abi_version': [],
    'avt_gemm_bf16': [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _F, _U64, _P,
                      _I, _I, _I, _P, _SZ, _P],
    'avt_gemm_accum_bf16': [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, ctypes.c_size_t, _P],
    'avt_gemm_assign_bf16': [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, ctypes.c_size_t, _P],      # (ABI 9: C = result, for a C known to hold zeros)
    'avt_gemm_ln_bf16': [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _F, _U64, _P,
                         _I, _I, _I, _P, _SZ, _P, _P, _P, _P],
    'avt_ln_stats_finalize': [_P, _I, _I, _I, _F, _P, _P, _P],
    'avt_ln_fold_weights': [_P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P],
    'avt_layernorm_bwd_folded': [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P, _SZ, _P],
    'avt_ln_fold_wgrad': [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _SZ, _P],
    'avt_vit_attn_bwd_scaled': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _SZ, _P, _P],
    'avt_layernorm_fwd': [_P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _F, _P],
    'avt_layernorm_bwd': [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P, _SZ, _P],
    'avt_vit_attn_fwd': [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    'avt_vit_attn_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _SZ, _P],
    'avt_causal_attn_fwd': [_P, _P, _P, _I, _I, _I, _I, _F, _F, _U64, _P],
    'avt_causal_attn_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _U64, _P],
    'avt_head_attn_fwd': [_P, _P, _P, _I, _I, _I, _I, _F, _F, _U64, _I, _P],
    'avt_head_attn_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _U64, _I, _P],
    'avt_relu_bf16': [_P, _P, _P, _L, _P],
    'avt_transpose_bf16': [_P, _L, _P, _L, _I, _I, _P],
    'avt_im2col_patch16': [_P, _P, _I, _I, _I, _P],
    'avt_posres_prep': [_P, _P, _P, _P, _I, _I, _P],
    'avt_patch_embed_bwd_reduce': [_P, _P, _P, _P, _I, _I, _I, _P, _SZ, _P],
    'avt_cast_f32_to_bf16': [_P, _P, _L, _P],
    'avt_cast_bf16_to_f32': [_P, _P, _L, _P],
    'avt_dropout_bf16': [_P, _P, _L, _F, _U64, _P],
    'avt_embed_pos_fwd': [_P, _P, _P, _I, _I, _I, _F, _U64, _P],
    'avt_embed_pos_bwd': [_P, _P, _P, _I, _I, _I, _F, _U64, _P],
    'avt_transpose_batch_bf16': [_P, _I, _I, _P],
    'avt_colsum_bf16': [_P, _I, _P, _I, _I, _P, _SZ, _P],
    'avt_mse_shift_fwd': [_P, _P, _P, _I, _I, _I, _P],
    'avt_mse_shift_bwd': [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    'avt_pad_cast_f32_to_bf16': [_P, _I, _P, _I, _I, _I, _P],
    'avt_add_rows_bf16': [_P, _L, _P, _L, _I, _I, _P],
    'avt_cls_attn_fwd': [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _P],
    'avt_cls_attn_bwd': [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P],
    'avt_causal_attn_decode': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    'avt_video_preproc_u8': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _I, _I, _P],
    'avt_video_preproc_jitter_u8': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _I, _I, _P, _SZ, _P, _P],
    'avt_xent_fwd': [_P, _I, _P, _P, _P, _P, _I, _I, _L, _P],
    'avt_xent_bwd': [_P, _I, _P, _P, _P, _P, _I, _I, _I, _L, _P],
    'avt_gemm_frag_ok': [_I, _I, _I],              # (returns 1 / 0, not an error code: called through load(), not call())
    'avt_sgd_step': [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _I, _I, _P],
    'avt_sgd_step_dev': [_P, _P, _P, _P, _L, _P, _F, _F, _F, _I, _I, _I, _P],        # (ABI 9: the learning rate in device memory, for captured steps)
    'avt_linear_softmax_xent_fwd': [_P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _L, _P],
    # gradient exchange over RCCL (ABI 8): comm handles are void*, sizes size_t
    'avt_comm_unique_id': [_P],
    'avt_comm_init_rank': [_P, _I, _I, _I, _P],
    'avt_comm_destroy': [_P],
    'avt_comm_size': [_P, _P, _P],
    'avt_allreduce_bucket': [_P, _P, _SZ, _I, _P],
    'avt_reduce_scatter_bucket': [_P, _P, _SZ, _I, _P],
    'avt_allgather_bucket': [_P, _P, _SZ, _I, _P],
    'avt_broadcast_bucket': [_P, _P, _SZ, _I, _I, _P],
    'avt_linear_softmax_xent_bwd': [_P, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _I, _F, _U64, _I, _I, _I, _I, _L, _P, _SZ, _P, _SZ, _P],
}

_lib = None
N_CALLS = 0          # C-ABI calls made by this process (bench.py reports calls per step: every call is one or two kernel launches)
CALLS_BY_NAME = {}    # ... per entry point


# workspace-size queries (return size_t, cannot fail)
SIZE_QUERIES = {
    'avt_gemm_accum_workspace_bytes': [_I, _I, _I],
    'avt_gemm_colsum_workspace_bytes': [_I, _I, _I],
    'avt_gemm_frag_bytes': [_I, _I],
    'avt_layernorm_bwd_workspace_bytes': [_I, _I],
    'avt_layernorm_bwd_folded_workspace_bytes': [_I, _I],
    'avt_ln_fold_wgrad_workspace_bytes': [_I, _I],
    'avt_vit_attn_bwd_workspace_bytes': [_I, _I, _I],
    'avt_patch_embed_bwd_reduce_workspace_bytes': [_I, _I, _I],
    'avt_colsum_workspace_bytes': [_I, _I],
    'avt_video_jitter_scratch_bytes': [_I, _I, _I, _I],
}


class AvtHipError(RuntimeError):
    pass


def load():
    """Load libavt_hip.so once; raise loudly when it is absent (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AvtHipError(
            f'{LIB_PATH} not found: the HIP extension is not built. Run `python -c "import __graft_entry__ as g; '
            f'g.build()"` or `make -C avt_amd/csrc`. There is no fallback path.')
    lib = ctypes.CDLL(LIB_PATH)
    lib.avt_last_error.restype = c_char_p
    lib.avt_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, argtypes in SIZE_QUERIES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_size_t
    v = lib.avt_abi_version()
    if v != ABI_VERSION:
        raise AvtHipError(f'libavt_hip.so ABI version {v} != binding version {ABI_VERSION}')
    _lib = lib
    return lib


def call(name, *args):
    global N_CALLS
    N_CALLS += 1
    CALLS_BY_NAME[name] = CALLS_BY_NAME.get(name, 0) + 1
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.avt_last_error()
        raise AvtHipError(f'{name} failed (rc={rc}): {msg.decode() if msg else "?"}')
