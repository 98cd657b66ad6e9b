"""Tensor-level wrappers over the C ABI (``avt_amd.lib``): torch is used only for device memory and streams.

Every function enqueues HIP kernels on torch's current stream and returns immediately.  Inputs must live on the
GPU; there is no CPU or eager fallback -- a missing library raises ``AvtHipError``.
"""
import bisect

import torch

from . import lib as _lib

BF16 = torch.bfloat16
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_MUL_AUX = 0, 1, 2, 3       # with c2=..., GELU stores its derivative there
OUT_BF16, OUT_F32, OUT_ACCUM_F32 = 0, 1, 2


# When set to a list, every gemm() appends (variant, flops, start_event, end_event, (M, N, K)): bench.py's live roofline probe.
GEMM_TRACE = None
FORCE_TILE = 0          # tests: route every auto-selected (tile=0) GEMM to one kernel, e.g. 808, to validate it inside the whole model


def _stream():
    return torch.cuda.current_stream().cuda_stream


PERSIST_KMAX = 4096      # mirrors PK_KMAX in csrc/gemm_persist.hip


# Every kernel-template prefix the router below can return for an output tile of 128 rows or more: what bench.py's GEMM-family /
# dominant-kernel numbers aggregate over (a CPU test checks that each name gemm_variant produces for such a tile is caught).
LARGE_TILE_KERNELS = ('gemm_kernel<128', 'gemm_kernel<256', 'gemm_8pp_kernel', 'gemm_8p_kernel', 'gemm_w4_kernel')


def persist_epilogue_kind(out_mode, act, has_bias, has_res, has_aux, has_c2, has_colsum, drop_p=0.0, res_period=0):
    """The epilogue kind (EPK 0..3) of csrc/gemm_persist.hip that covers this call, or None (mirror of avt_gemm_persist's dispatch)."""
    if out_mode != OUT_BF16 or drop_p != 0.0 or res_period != 0:
        return None
    if act == ACT_NONE and not has_res and not has_colsum and not has_c2 and not has_aux:
        return 0
    if act == ACT_GELU_ERF and not has_res and not has_colsum:
        return 1
    if act == ACT_NONE and has_res and not has_colsum and not has_c2 and not has_aux:
        return 2
    if act == ACT_MUL_AUX and has_aux and not has_res and not has_bias and not has_c2:
        return 3
    return None


def gemm_variant(M, N, K, a_kmajor, b_kmajor, out_mode, tile, epilogue_ok=False):
    """Mirror of the tile choice in avt_gemm_bf16 (csrc/gemm.hip): names the kernel template a call lands on.
    ``epilogue_ok``: False / None = the persistent kernel does not cover the epilogue; True or an int = it does (an int names the kind)."""
    epk = None if (epilogue_ok is False or epilogue_ok is None) else (epilogue_ok if type(epilogue_ok) is int else -1)
    epilogue_ok = epk is not None
    epi = 1 if out_mode == OUT_ACCUM_F32 else 0
    bm = tile
    if bm == 0:
        t256 = ((M + 255) // 256) * ((N + 255) // 256)
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        if epi == 0:
            kk = bool(a_kmajor) and bool(b_kmajor)
            if t256 >= 200 or (kk and K % 64 == 0 and t256 >= 96):
                bm = 256
            elif t128 >= 192 and not (kk and K <= 3072):
                bm = 128
            else:
                bm = 64
        else:
            sk = min(max(((K + 63) // 64) // 4, 1), 64)
            bm = 256 if (t256 * sk >= 256 and t256 < 4096) else 128
        if bm == 256 and (K % 64 == 0 or (not a_kmajor and not b_kmajor)):
            bm = 808
            t256 = ((M + 255) // 256) * ((N + 255) // 256)
            if (epi == 0 and a_kmajor and b_kmajor and epilogue_ok and N % 256 == 0 and K % 128 == 0 and 256 <= K <= PERSIST_KMAX
                    and 512 <= t256 < 65536):
                bm = 809          # the persistent form (csrc/gemm_persist.hip: avt_gemm_persist)
        if bm == 64 and epi == 0 and a_kmajor and (M <= 32 or (M <= 64 and b_kmajor)):
            bm = 32           # the skinny kernel (csrc/gemm.hip: gemm_skinny_kernel)
        if bm == 64 and epi == 0 and a_kmajor and (b_kmajor or ((M + 63) // 64) * ((N + 63) // 64) <= 256):
            bm = 643
    if bm == 32:
        return f'gemm_skinny_kernel<{int(bool(b_kmajor))}>'
    shape = {64: '64,64,2,2,64,2,0', 128: '128,128,2,2,64,2,0', 256: '256,256,2,4,64,2,1', 2568: '256,256,2,4,64,2,1,l8', 808: '8p', 809: '8pp', 643: '64,64,2,2,64,3,0'}[bm]
    if bm == 808:
        return f'gemm_8p_kernel<{int(bool(a_kmajor))},{int(bool(b_kmajor))},{epi}>'
    if bm == 809:
        return 'gemm_8pp_kernel' if epk is None or epk < 0 else f'gemm_8pp_kernel<{epk}>'
    return f'gemm_kernel<{shape},{int(bool(a_kmajor))},{int(bool(b_kmajor))},{epi}>'


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype, name):
    if t.device.type != 'cuda':
        raise _lib.AvtHipError(f'{name} must be a GPU tensor (no CPU fallback exists)')
    if t.dtype != dtype:
        raise _lib.AvtHipError(f'{name} must be {dtype}, got {t.dtype}')


# ---- run-to-run identical bias / LayerNorm / embedding gradients ---------------------------------------------------------
# True: every kernel that folds many workgroups into one fp32 vector stores per-workgroup partials into a scratch workspace
# and a second kernel adds them in a fixed order (include/avt_hip.h, "partials").  False: fp32 atomics (arrival order).
DETERMINISTIC_REDUCTIONS = True
_PART_WS = {}                    # (device index, stream) -> fp32 scratch shared by all calls on that stream (they are ordered)


def _partials(device, query, *dims):
    """(pointer, bytes) of the partials workspace for one call, or (None, 0) with the atomics path selected."""
    if not DETERMINISTIC_REDUCTIONS:
        return None, 0
    need = getattr(_lib.load(), query)(*dims)
    key = (device.index, _stream())
    ws = _PART_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 16 << 20), device=device, dtype=torch.uint8)
        _PART_WS[key] = ws
    return ws.data_ptr(), ws.numel()


def _ld(t):
    """Leading dimension (elements) of a 2-D row-major view with unit inner stride."""
    assert t.dim() == 2 and t.stride(1) == 1, 'expected a 2-D tensor with unit inner stride'
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


class FragTensor:
    """A [M, N] bf16 tensor in the persistent GEMM's fragment-major private order (include/avt_hip.h, ABI 7: ldc2 == 0 / ldaux == 0): written by
    gemm(act=ACT_GELU_ERF, c2=FragTensor) and read back by gemm(act=ACT_MUL_AUX, aux=FragTensor) of the same M, N -- nothing else reads it
    (``gemm_frag_unpack`` restates the order for the tests)."""
    __slots__ = ('buf', 'M', 'N')

    def __init__(self, M, N, device):
        self.M, self.N = M, N
        self.buf = torch.empty(_lib.load().avt_gemm_frag_bytes(M, N) // 2, device=device, dtype=BF16)


def gemm_frag_ok(M, N, K):
    """Does a contiguous k-major [M, K] x [N, K] call with the automatic tile choice land on the persistent kernel (the only one that knows the
    fragment-major order)?  Asked of the library, not mirrored here."""
    return bool(_lib.load().avt_gemm_frag_ok(M, N, K))


def gemm_frag_unpack(ft):
    """Row-major [M, N] copy of a FragTensor (tests / debugging): per (128-row strip s, 64-column group c) four blocks i of [4 stores st][64 lanes l][8 values e];
    value e of (st, l) is row 128 s + 32 i + l % 32, column 64 c + 32 (st // 2) + 8 (2 (st % 2) + e // 4) + 4 (l // 32) + e % 4."""
    M, N = ft.M, ft.N
    S, C = (M + 127) // 128, N // 64
    v = ft.buf.view(S, C, 4, 4, 64, 8)                                  # s, c, i, st, l, e
    v = v.view(S, C, 4, 2, 2, 2, 32, 2, 4)                              # s, c, i, j, qh (= st % 2), h (= l // 32), ml, ql (= e // 4), e4
    v = v.permute(0, 2, 6, 1, 3, 4, 7, 5, 8)                            # s, i, ml | c, j, qh, ql, h, e4
    return v.reshape(S * 128, N)[:M].contiguous()


def gemm(A, B, M, N, K, *, a_kmajor=True, b_kmajor=True, out=None, out_mode=OUT_BF16, bias=None, act=ACT_NONE,
         aux=None, c2=None, res=None, res_period=0, drop_p=0.0, seed=0, colsum=None, splitk=0, tile=0,
         ln_stat=None, ln_c=None, stat_part=None):
    """C[M,N] = epilogue(sum_k opA[m,k] opB[n,k]); see avt_gemm_bf16 in include/avt_hip.h.
    ln_stat / ln_c / stat_part: the LayerNorm-fold modes of avt_gemm_ln_bf16 (fold: ln_stat [M,2] + ln_c [N]; scale: ln_stat alone with
    act=ACT_MUL_AUX; stat_part [ceil(N/64), M, 2, 2]: row statistics of the output)."""
    _chk(A, BF16, 'A'); _chk(B, BF16, 'B')
    if tile == 0 and FORCE_TILE:
        tile = FORCE_TILE
    if out is None:
        assert out_mode != OUT_ACCUM_F32, 'accumulate mode needs an output buffer'
        out = torch.empty((M, N), device=A.device, dtype=BF16 if out_mode == OUT_BF16 else torch.float32)
    trace = GEMM_TRACE
    if trace is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    part, part_bytes = _partials(A.device, 'avt_gemm_colsum_workspace_bytes', M, N, tile) if colsum is not None else (None, 0)
    aux_frag, c2_frag = type(aux) is FragTensor, type(c2) is FragTensor
    for ft in ((aux,) if aux_frag else ()) + ((c2,) if c2_frag else ()):
        assert (ft.M, ft.N) == (M, N), 'fragment-major tensor of another shape'
    args = (_p(A), int(a_kmajor), _ld(A), _p(B), int(b_kmajor), _ld(B), _p(out), _ld(out), M, N, K,
            _p(bias), act, _p(aux.buf if aux_frag else aux), 0 if aux is None or aux_frag else _ld(aux),
            _p(c2.buf if c2_frag else c2), 0 if c2 is None or c2_frag else _ld(c2),
            _p(res), _ld(res) if res is not None else 0, res_period, float(drop_p), int(seed), _p(colsum),
            out_mode, splitk, tile, part, part_bytes)
    ln = ln_stat is not None or ln_c is not None or stat_part is not None
    if ln:
        for t_, nm in ((ln_stat, 'ln_stat'), (ln_c, 'ln_c'), (stat_part, 'stat_part')):
            if t_ is not None:
                _chk(t_, torch.float32, nm)
        _lib.call('avt_gemm_ln_bf16', *args, _p(ln_stat), _p(ln_c), _p(stat_part), _stream())
    else:
        _lib.call('avt_gemm_bf16', *args, _stream())
    if trace is not None:
        ev1.record()
        # (the epilogues the persistent kernel covers: bias | erf-GELU (+ derivative) | bias + residual | saved derivative (+ column sums))
        ep_ok = persist_epilogue_kind(out_mode, act, bias is not None, res is not None, aux is not None, c2 is not None, colsum is not None,
                                      drop_p, res_period)
        if ep_ok is not None and ln:          # the LayerNorm-fold variants of the persistent kernel: 4 = 2 + statistics, 5 / 6 = 0 / 1 folded, 7 = 3 scaled
            ep_ok = {0: 5, 1: 6}.get(ep_ok) if ln_c is not None else ({3: 7}.get(ep_ok) if ln_stat is not None else ({2: 4}.get(ep_ok) if N % 64 == 0 else None))
        if c2_frag or aux_frag:               # the fragment-major forms: 8 / 9 = 1 / 6 writing it, 10 / 11 = 3 / 7 reading it
            ep_ok = {1: 8, 6: 9, 3: 10, 7: 11}.get(ep_ok)
        trace.append((gemm_variant(M, N, K, a_kmajor, b_kmajor, out_mode, tile, ep_ok), 2.0 * M * N * K, ev0, ev1, (M, N, K)))
    return out


# ---- deterministic weight-gradient accumulate ------------------------------------------------------------------------------
WGRAD_TILE = 0                   # 0 = the library's choice; 2565 = the 4-wave 128x128-wave-tile kernel (A/B and tests)
DETERMINISTIC_WGRAD = True       # False: fp32 atomics straight into the gradient (out_mode 2), order-dependent in the last bits
_WGRAD_WS = {}                   # (device index, stream) -> uint8 workspace; GEMMs on one stream are ordered, so they share it


def _wgrad_workspace(device, nbytes):
    key = (device.index, _stream())
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 64 << 20), device=device, dtype=torch.uint8)
        _WGRAD_WS[key] = ws
    return ws


# ---- gradient regions known to hold zeros (round 6) -------------------------------------------------------------------------------
# FusedSGD re-zeroes the flat gradient buffer as it consumes it (avt_sgd_step's zero_grad) and says so here; the FIRST weight gradient written
# into a region of such a buffer afterwards is stored instead of added (avt_gemm_assign_bf16: no read of 4 bytes of zeros per weight), every
# later one into an overlapping region is added as before (multi-crop clips, a roll-out with gradients, anything that runs a layer twice).
# Only weight gradients go through here and no other kind of kernel writes into a weight's gradient (biases, LayerNorm parameters and embeddings
# are their own tensors; folded layers' weight gradients and the fused classifier + loss node never call gemm_accum), so "not yet written by
# gemm_accum" means "still zero".  Anything else that writes into a registered buffer between two optimizer steps must call forget_zeroed().
_ZEROED = {}                     # first byte of a buffer -> [byte past its end, intervals written since it was zeroed]
ASSIGN_FIRST_WGRAD = True


def mark_zeroed(buf):
    """``buf`` (a flat fp32 gradient buffer) has just been re-zeroed in stream order by the kernel that consumed it.  The entry lives as long as
    the tensor object does (a freed buffer's address may be handed to a tensor nobody zeroed)."""
    import weakref
    p = buf.data_ptr()
    _ZEROED[p] = [p + buf.numel() * buf.element_size(), ([], []), weakref.ref(buf, lambda _r, _p=p: _ZEROED.pop(_p, None))]


def forget_zeroed(buf=None):
    if buf is None:
        _ZEROED.clear()
    else:
        _ZEROED.pop(buf.data_ptr(), None)


def _first_write(C, rows):
    if not ASSIGN_FIRST_WGRAD or not _ZEROED:
        return False
    lo = C.data_ptr()
    hi = lo + ((rows - 1) * _ld(C) + C.size(1)) * 4
    for b0, (b1, (los, his), _alive) in _ZEROED.items():
        if b0 <= lo and hi <= b1:
            # the intervals written since the buffer was zeroed: sorted and disjoint (a write that overlaps some is merged with them), so a look-up
            # is a bisection, not a scan (a step makes ~85 weight-gradient calls)
            i = bisect.bisect_right(los, lo) - 1
            j = i if (i >= 0 and his[i] > lo) else i + 1           # first interval that overlaps [lo, hi), if any
            k = j
            while k < len(los) and los[k] < hi:
                k += 1
            if j == k:
                los.insert(j, lo); his.insert(j, hi)
                return True
            nlo, nhi = min(lo, los[j]), max(hi, his[k - 1])
            los[j:k] = [nlo]; his[j:k] = [nhi]
            return False
    return False


def gemm_accum(A, B, C, M, N, K):
    """C[M,N] (fp32) += sum_k A[k,m] B[k,n], both operands stored reduction-index-major; split-K partials go through a
    workspace and are added in a fixed order (avt_gemm_accum_bf16) -- bit-reproducible weight gradients.  The first write into a region that
    FusedSGD has just re-zeroed is a store (avt_gemm_assign_bf16; same bits)."""
    if not DETERMINISTIC_WGRAD or FORCE_TILE:
        _first_write(C, M)                                  # (the atomic path adds: the region counts as written)
        return gemm(A, B, M, N, K, a_kmajor=False, b_kmajor=False, out=C, out_mode=OUT_ACCUM_F32)
    _chk(A, BF16, 'A'); _chk(B, BF16, 'B'); _chk(C, torch.float32, 'C')
    need = _lib.load().avt_gemm_accum_workspace_bytes(M, N, K)
    ws = _wgrad_workspace(A.device, need)
    trace = GEMM_TRACE
    if trace is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.call('avt_gemm_assign_bf16' if _first_write(C, M) else 'avt_gemm_accum_bf16', _p(A), _ld(A), _p(B), _ld(B), _p(C), _ld(C), M, N, K, 0, WGRAD_TILE,
              _p(ws), ws.numel(), _stream())
    if trace is not None:
        ev1.record()
        name = gemm_variant(M, N, K, False, False, OUT_ACCUM_F32, WGRAD_TILE if WGRAD_TILE != 2565 else 0).replace(',1>', ',2>')   # EPI 2 = slabs + ordered reduce
        if name.startswith('gemm_8p_kernel') and WGRAD_TILE in (0, 2565):
            name = 'gemm_w4_kernel<2>'                    # the library's default for 256x256-tile weight gradients
        trace.append((name, 2.0 * M * N * K, ev0, ev1, (M, N, K)))
    return C


# ---- the six contractions of a Linear (weight (out,in)) / HF Conv1D (weight (in,out)) layer ---------------------
def linear_fwd(x, w, **kw):
    """y[M,N] = x[M,K] @ w[N,K]^T"""
    return gemm(x, w, x.size(0), w.size(0), w.size(1), a_kmajor=True, b_kmajor=True, **kw)


def linear_dgrad(dy, w, **kw):
    """dx[M,K] = dy[M,N] @ w[N,K]"""
    return gemm(dy, w, dy.size(0), w.size(1), w.size(0), a_kmajor=True, b_kmajor=False, **kw)


def linear_wgrad(dy, x, dw, rows=None, **kw):
    """dw[N,K] (fp32) += dy[M,N]^T @ x[M,K]; ``rows`` limits the valid output rows (padded classifier)."""
    n = dw.size(0) if rows is None else rows
    if kw:
        return gemm(dy, x, n, x.size(1), dy.size(0), a_kmajor=False, b_kmajor=False, out=dw, out_mode=OUT_ACCUM_F32, **kw)
    return gemm_accum(dy, x, dw, n, x.size(1), dy.size(0))


def conv1d_fwd(x, w, **kw):
    """y[M,N] = x[M,K] @ w[K,N]"""
    return gemm(x, w, x.size(0), w.size(1), w.size(0), a_kmajor=True, b_kmajor=False, **kw)


def conv1d_dgrad(dy, w, **kw):
    """dx[M,K] = dy[M,N] @ w[K,N]^T"""
    return gemm(dy, w, dy.size(0), w.size(0), w.size(1), a_kmajor=True, b_kmajor=True, **kw)


def conv1d_wgrad(x, dy, dw, **kw):
    """dw[K,N] (fp32) += x[M,K]^T @ dy[M,N]"""
    if kw:
        return gemm(x, dy, x.size(1), dy.size(1), x.size(0), a_kmajor=False, b_kmajor=False, out=dw, out_mode=OUT_ACCUM_F32, **kw)
    return gemm_accum(x, dy, dw, x.size(1), dy.size(1), x.size(0))


# ---- LayerNorm -------------------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, rows=None, ldx=None, save_stats=True):
    """x: 2-D bf16 view; optional (rows, ldx) override lets the caller normalise strided rows (CLS select)."""
    _chk(x, BF16, 'x')
    D = gamma.numel()
    if rows is None:
        rows, ldx = x.size(0), _ld(x)
    y = torch.empty((rows, D), device=x.device, dtype=BF16)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if save_stats else None
    _lib.call('avt_layernorm_fwd', _p(x), ldx, _p(gamma), _p(beta), _p(y), D, _p(mean), _p(rstd), rows, D, float(eps), _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, *, dres=None, colsum=None, rows=None, ldx=None, dx=None, lddx=None):
    _chk(dy, BF16, 'dy'); _chk(x, BF16, 'x')
    D = gamma.numel()
    if rows is None:
        rows, ldx = x.size(0), _ld(x)
    if dx is None:
        dx = torch.empty((rows, D), device=x.device, dtype=BF16)
        lddx = D
    part, part_bytes = _partials(x.device, 'avt_layernorm_bwd_workspace_bytes', rows, D)
    _lib.call('avt_layernorm_bwd', _p(dy), _ld(dy), _p(x), ldx, _p(mean), _p(rstd), _p(gamma), _p(dres),
              _ld(dres) if dres is not None else 0, _p(dx), lddx, _p(dgamma), _p(dbeta), _p(colsum), rows, D,
              part, part_bytes, _stream())
    return dx


# ---- LayerNorm folded into the GEMMs around it (include/avt_hip.h: avt_gemm_ln_bf16 and friends) ----------------------------------
def ln_stat_part(M, N, device):
    """Scratch for the row statistics a GEMM epilogue emits (``gemm(..., stat_part=...)``): [ceil(N / 64), M, 2 slots, 2] fp32."""
    return torch.empty(((N + 63) // 64, M, 2, 2), device=device, dtype=torch.float32)


def ln_stats_finalize(part, D, eps, want_bwd=True):
    """part [ceil(D / 64), rows, 2, 2] -> (stat_fwd [rows, 2] = {rstd, -mean * rstd}, stat_bwd [rows, 2] = {rstd, 1 / rstd} | None)."""
    _chk(part, torch.float32, 'part')
    rows = part.size(1)
    nslots = (D + 31) // 32
    sf = torch.empty((rows, 2), device=part.device, dtype=torch.float32)
    sb = torch.empty((rows, 2), device=part.device, dtype=torch.float32) if want_bwd else None
    _lib.call('avt_ln_stats_finalize', _p(part), nslots, rows, D, float(eps), _p(sf), _p(sb), _stream())
    return sf, sb


def ln_fold_weights(W, gamma, beta, bias, G, c, b2):
    """G (bf16 [N, K]) = gamma o W, c [N] = G 1, b2 [N] = bias + W beta from the fp32 master parameters."""
    _chk(W, torch.float32, 'W'); _chk(G, BF16, 'G')
    N, K = W.shape
    _lib.call('avt_ln_fold_weights', _p(W), _ld(W), _p(gamma), _p(beta), _p(bias), _p(G), _ld(G), _p(c), _p(b2), N, K, _stream())


def layernorm_bwd_folded(dy, x, stat_fwd, *, dres=None, colsum=None):
    """dx of a folded LayerNorm: dy = d xhat' (already carrying the rows' rstd), see avt_layernorm_bwd_folded."""
    _chk(dy, BF16, 'dy'); _chk(x, BF16, 'x'); _chk(stat_fwd, torch.float32, 'stat_fwd')
    rows, D = x.shape
    dx = torch.empty((rows, D), device=x.device, dtype=BF16)
    part, part_bytes = _partials(x.device, 'avt_layernorm_bwd_folded_workspace_bytes', rows, D) if colsum is not None else (None, 0)
    _lib.call('avt_layernorm_bwd_folded', _p(dy), _ld(dy), _p(x), _ld(x), _p(stat_fwd), _p(dres), _ld(dres) if dres is not None else 0,
              _p(dx), D, _p(colsum), rows, D, part, part_bytes, _stream())
    return dx


def ln_fold_wgrad(T, W, gamma, beta, dbias_tmp, dW, dgamma, dbeta, dbias):
    """Weight-side backward of the fold: raw T = dY'^T x (fp32, re-zeroed here) -> dW, dgamma, dbeta, dbias (all accumulate)."""
    N, K = W.shape
    part, part_bytes = _partials(W.device, 'avt_ln_fold_wgrad_workspace_bytes', N, K)
    _lib.call('avt_ln_fold_wgrad', _p(T), _ld(T), _p(W), _ld(W), _p(gamma), _p(beta), _p(dbias_tmp), _p(dW), _ld(dW), _p(dgamma), _p(dbeta),
              _p(dbias), N, K, part, part_bytes, _stream())


# ---- attention cores ---------------------------------------------------------------------------------------------------
def vit_attn_fwd(qkv, frames, S, H):
    _chk(qkv, BF16, 'qkv')
    D = H * 64
    out = torch.empty((frames * S, D), device=qkv.device, dtype=BF16)
    lse = torch.empty((frames, H, S), device=qkv.device, dtype=torch.float32)
    _lib.call('avt_vit_attn_fwd', _p(qkv), _p(out), _p(lse), frames, S, H, 64, 0.125, _stream())
    return out, lse


def vit_attn_bwd(qkv, out, dout, lse, frames, S, H, dbias=None, row_stat=None):
    """row_stat [frames * S, 2] (= stat_bwd of the LayerNorm folded into the qkv projection): the rows of dqkv leave multiplied by row_stat[:, 0]."""
    dqkv = torch.empty_like(qkv)
    part, part_bytes = _partials(qkv.device, 'avt_vit_attn_bwd_workspace_bytes', frames, S, H) if dbias is not None else (None, 0)
    if row_stat is not None:
        _chk(row_stat, torch.float32, 'row_stat')
        _lib.call('avt_vit_attn_bwd_scaled', _p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(dbias), frames, S, H, 64, 0.125,
                  part, part_bytes, _p(row_stat), _stream())
    else:
        _lib.call('avt_vit_attn_bwd', _p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(dbias), frames, S, H, 64, 0.125,
                  part, part_bytes, _stream())
    return dqkv


def cls_attn_fwd(q, kv, frames, S, H):
    """CLS-query attention of the last ViT block: q [frames, H*64], kv [frames*S, 2*H*64] -> (out [frames, H*64], probs)."""
    _chk(q, BF16, 'q'); _chk(kv, BF16, 'kv')
    D = H * 64
    out = torch.empty((frames, D), device=q.device, dtype=BF16)
    probs = torch.empty((frames, H, S), device=q.device, dtype=torch.float32)
    _lib.call('avt_cls_attn_fwd', _p(q), _ld(q), _p(kv), _ld(kv), _p(out), D, _p(probs), frames, S, H, 64, 0.125, _stream())
    return out, probs


def cls_attn_bwd(q, kv, probs, dout, frames, S, H):
    D = H * 64
    dq = torch.empty((frames, D), device=q.device, dtype=BF16)
    dkv = torch.empty((frames * S, 2 * D), device=q.device, dtype=BF16)
    _lib.call('avt_cls_attn_bwd', _p(q), _ld(q), _p(kv), _ld(kv), _p(probs), _p(dout), _ld(dout), _p(dq), D, _p(dkv), 2 * D,
              frames, S, H, 64, 0.125, _stream())
    return dq, dkv


def causal_attn_decode(qkv, kcache, vcache, B, H, hd, pos):
    """KV-cache step of the AVT-h roll-out: appends the token's k / v at row ``pos`` and attends over rows 0..pos."""
    _chk(qkv, BF16, 'qkv'); _chk(kcache, BF16, 'kcache'); _chk(vcache, BF16, 'vcache')
    assert qkv.is_contiguous() and kcache.is_contiguous() and vcache.is_contiguous()
    out = torch.empty((B, H * hd), device=qkv.device, dtype=BF16)
    _lib.call('avt_causal_attn_decode', _p(qkv), _p(kcache), _p(vcache), _p(out), B, H, hd, pos, kcache.size(1), float(hd) ** -0.5, _stream())
    return out


def causal_attn_fwd(qkv, B, T, H, hd, drop_p=0.0, seed=0, causal=True):
    _chk(qkv, BF16, 'qkv')
    out = torch.empty((B * T, H * hd), device=qkv.device, dtype=BF16)
    probs = torch.empty((B, H, T, T), device=qkv.device, dtype=torch.float32)
    _lib.call('avt_head_attn_fwd', _p(qkv), _p(out), _p(probs), B, T, H, hd, float(hd) ** -0.5, float(drop_p), int(seed), int(causal), _stream())
    return out, probs


def causal_attn_bwd(qkv, probs, dout, B, T, H, hd, drop_p=0.0, seed=0, causal=True):
    dqkv = torch.empty_like(qkv)
    _lib.call('avt_head_attn_bwd', _p(qkv), _p(probs), _p(dout), _p(dqkv), B, T, H, hd, float(hd) ** -0.5, float(drop_p), int(seed), int(causal), _stream())
    return dqkv


def transpose_into(src, dst):
    """dst[c, r] = src[r, c]; 2-D bf16 views with unit inner stride."""
    _chk(src, BF16, 'src'); _chk(dst, BF16, 'dst')
    assert dst.shape == (src.size(1), src.size(0))
    _lib.call('avt_transpose_bf16', _p(src), _ld(src), _p(dst), _ld(dst), src.size(0), src.size(1), _stream())


def transpose_jobs(pairs):
    """Device job table for ``transpose_batch``: [(src, dst)] 2-D bf16 views with unit inner stride, dst = src^T.
    Returns (table tensor, njobs, max_tiles); valid while the tensors stay where they are."""
    import struct
    buf, max_tiles = b'', 0
    for src, dst in pairs:
        _chk(src, BF16, 'src'); _chk(dst, BF16, 'dst')
        assert dst.shape == (src.size(1), src.size(0))
        r, c = src.size(0), src.size(1)
        buf += struct.pack('<QQqqii', src.data_ptr(), dst.data_ptr(), _ld(src), _ld(dst), r, c)
        max_tiles = max(max_tiles, ((r + 63) // 64) * ((c + 63) // 64))
    table = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(pairs[0][0].device)
    return table, len(pairs), max_tiles


def transpose_batch(jobs):
    table, n, max_tiles = jobs
    _lib.call('avt_transpose_batch_bf16', _p(table), n, max_tiles, _stream())


def relu(x):
    """(max(x, 0), mask) with mask = bf16 1/0 = the derivative, consumed by gemm(act=ACT_MUL_AUX) in backward."""
    _chk(x, BF16, 'x')
    assert x.is_contiguous() and x.numel() % 8 == 0
    y, mask = torch.empty_like(x), torch.empty_like(x)
    _lib.call('avt_relu_bf16', _p(x), _p(y), _p(mask), x.numel(), _stream())
    return y, mask


# ---- patch embedding helpers ---------------------------------------------------------------------------------------------
def im2col_patch16(frames):
    """frames fp32 [N,3,H,W] -> bf16 [N*(P+1), 768] with a zero CLS row per frame."""
    _chk(frames, torch.float32, 'frames')
    frames = frames.contiguous()
    n, _, h, w = frames.shape
    rows = n * ((h // 16) * (w // 16) + 1)
    out = torch.empty((rows, 768), device=frames.device, dtype=BF16)
    _lib.call('avt_im2col_patch16', _p(frames), _p(out), n, h, w, _stream())
    return out


def posres_prep(pos, cls, bias, S, D):
    R = torch.empty((S, D), device=pos.device, dtype=BF16)
    _lib.call('avt_posres_prep', _p(pos), _p(cls), _p(bias), _p(R), S, D, _stream())
    return R


def patch_embed_bwd_reduce(dx0, dpos, dcls, dbias, N, S, D):
    part, part_bytes = _partials(dx0.device, 'avt_patch_embed_bwd_reduce_workspace_bytes', N, S, D)
    _lib.call('avt_patch_embed_bwd_reduce', _p(dx0), _p(dpos), _p(dcls), _p(dbias), N, S, D, part, part_bytes, _stream())


# ---- elementwise ---------------------------------------------------------------------------------------------------------
def cast_to_bf16(src, dst=None):
    _chk(src, torch.float32, 'src')
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=BF16)
    _lib.call('avt_cast_f32_to_bf16', _p(src), _p(dst), src.numel(), _stream())
    return dst


def cast_to_f32(src):
    _chk(src, BF16, 'src')
    dst = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    _lib.call('avt_cast_bf16_to_f32', _p(src), _p(dst), src.numel(), _stream())
    return dst


def dropout(x, p, seed):
    _chk(x, BF16, 'x')
    y = torch.empty_like(x)
    _lib.call('avt_dropout_bf16', _p(x), _p(y), x.numel(), float(p), int(seed), _stream())
    return y


def embed_pos_fwd(enc, wpe, B, T, E, p, seed):
    h = torch.empty_like(enc)
    _lib.call('avt_embed_pos_fwd', _p(enc), _p(wpe), _p(h), B, T, E, float(p), int(seed), _stream())
    return h


def embed_pos_bwd(dh, dwpe, B, T, E, p, seed):
    denc = torch.empty_like(dh)
    _lib.call('avt_embed_pos_bwd', _p(dh), _p(denc), _p(dwpe), B, T, E, float(p), int(seed), _stream())
    return denc


def colsum(x, out):
    _chk(x, BF16, 'x')
    part, part_bytes = _partials(x.device, 'avt_colsum_workspace_bytes', x.size(0), x.size(1))
    _lib.call('avt_colsum_bf16', _p(x), _ld(x), _p(out), x.size(0), x.size(1), part, part_bytes, _stream())


def mse_shift_fwd(dec, x):
    """loss[b,t,:] = (dec[b,t,:] - x[b,t+1,:])^2 for t < T-1; dec, x fp32 contiguous [B,T,F]."""
    _chk(dec, torch.float32, 'dec'); _chk(x, torch.float32, 'x')
    B, T, F = dec.shape
    loss = torch.empty((B, T - 1, F), device=dec.device, dtype=torch.float32)
    _lib.call('avt_mse_shift_fwd', _p(dec), _p(x), _p(loss), B, T, F, _stream())
    return loss


def mse_shift_bwd(dec, x, gloss):
    B, T, F = dec.shape
    ddec, dx = torch.empty_like(dec), torch.empty_like(x)
    _lib.call('avt_mse_shift_bwd', _p(dec), _p(x), _p(gloss), _p(ddec), _p(dx), B, T, F, _stream())
    return ddec, dx


def pad_cast_to_bf16(src, ldd):
    """fp32 [rows, cols] (unit inner stride) -> bf16 [rows, ldd], columns >= cols zero."""
    _chk(src, torch.float32, 'src')
    rows, cols = src.shape
    dst = torch.empty((rows, ldd), device=src.device, dtype=BF16)
    _lib.call('avt_pad_cast_f32_to_bf16', _p(src), _ld(src), _p(dst), ldd, rows, cols, _stream())
    return dst


def add_rows(dst, src):
    """dst[r, :] += src[r, :]; both 2-D bf16 views with unit inner stride (dst typically the CLS rows of a [frames*S, D] tensor)."""
    _chk(dst, BF16, 'dst'); _chk(src, BF16, 'src')
    assert dst.shape == src.shape
    _lib.call('avt_add_rows_bf16', _p(dst), _ld(dst), _p(src), _ld(src), dst.size(0), dst.size(1), _stream())


# ---- input pipeline ------------------------------------------------------------------------------------------------------------
def _patch_rows(B, T, OH, OW, device):
    return torch.empty((B * T * ((OH // 16) * (OW // 16) + 1), 768), device=device, dtype=BF16)


def video_preproc(src_u8, params, out_hw, scale_pix=1.0, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), reverse_channels=False,
                  quantize_u8=False, patches=False):
    """uint8 (B,T,H,W,3) -> fp32 (B,T,3,1,OH,OW): /255, bilinear resize, flip, scale, normalise, crop in one kernel.
    params: int32 (Bout,6) = new_h, new_w, flip, crop_i, crop_j, source clip per OUTPUT clip (on the device).
    quantize_u8: cut the resized pixels to 8 bits first (the training chain's zero-strength ColorJitterVideo round trip).
    patches=True: the same pixels as the patch-embedding GEMM's bf16 rows [B T 197, 768] instead (= im2col_patch16 of the fp32 result, bit for bit)."""
    import ctypes
    _chk(src_u8, torch.uint8, 'src'); _chk(params, torch.int32, 'params')
    assert src_u8.dim() == 5 and src_u8.size(-1) == 3 and src_u8.is_contiguous() and params.is_contiguous()
    _, T, H, W, _ = src_u8.shape
    OH, OW = out_hw
    assert params.dim() == 2 and params.size(1) == 6
    B = params.size(0)
    out = _patch_rows(B, T, OH, OW, src_u8.device) if patches else torch.empty((B, T, 3, 1, OH, OW), device=src_u8.device, dtype=torch.float32)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _lib.call('avt_video_preproc_u8', _p(src_u8), None if patches else _p(out), _p(out) if patches else None, _p(params), B, T, H, W, OH, OW, float(scale_pix),
              ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), int(reverse_channels), int(quantize_u8), _stream())
    return out


def video_preproc_jitter(src_u8, params, jitter_ops, jitter_factors, out_hw, scale_pix=1.0, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5),
                         reverse_channels=False, max_hw=None, slot_mask=None, patches=False):
    """video_preproc with ColorJitterVideo: jitter_ops int32 (Bout, 4) in application order (0 brightness, 1 contrast, 2 saturation, 3 hue,
    -1 none), jitter_factors fp32 (Bout, 4) (hue: the 8-bit shift).  The resized clips go through an 8-bit scratch buffer.
    max_hw = (max new_h, max new_w) over the clips and slot_mask (bit s: slot s used by some clip, bit 4 + s: by a contrast operation)
    are what the caller that built ``params`` / ``jitter_ops`` on the host already knows (GpuClipTransform passes them: no device ->
    host read, no launches for empty slots); left None they are read back from the device tensors (one small synchronising copy)."""
    import ctypes
    _chk(src_u8, torch.uint8, 'src'); _chk(params, torch.int32, 'params'); _chk(jitter_ops, torch.int32, 'jitter_ops'); _chk(jitter_factors, torch.float32, 'jitter_factors')
    assert src_u8.dim() == 5 and src_u8.size(-1) == 3 and src_u8.is_contiguous() and params.is_contiguous()
    _, T, H, W, _ = src_u8.shape
    OH, OW = out_hw
    B = params.size(0)
    assert params.shape == (B, 6) and jitter_ops.shape == (B, 4) and jitter_factors.shape == (B, 4)
    if max_hw is None:
        max_hw = params[:, :2].max(dim=0).values.tolist()           # device -> host read (callers with host-side params pass max_hw)
    max_h, max_w = int(max_hw[0]), int(max_hw[1])
    if slot_mask is None:
        host_ops = jitter_ops.cpu()
        slot_mask = sum(((1 << s) if bool((host_ops[:, s] >= 0).any()) else 0) | ((16 << s) if bool((host_ops[:, s] == 1).any()) else 0) for s in range(4))
    scratch = torch.empty(_lib.load().avt_video_jitter_scratch_bytes(B, T, max_h, max_w), device=src_u8.device, dtype=torch.uint8)
    sums = torch.zeros(B, device=src_u8.device, dtype=torch.int64)
    out = _patch_rows(B, T, OH, OW, src_u8.device) if patches else torch.empty((B, T, 3, 1, OH, OW), device=src_u8.device, dtype=torch.float32)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _lib.call('avt_video_preproc_jitter_u8', _p(src_u8), None if patches else _p(out), _p(out) if patches else None, _p(params), _p(jitter_ops.contiguous()), _p(jitter_factors.contiguous()), B, T, H, W,
              OH, OW, max_h, max_w, float(scale_pix), ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), int(reverse_channels),
              int(slot_mask), _p(scratch), scratch.numel(), _p(sums), _stream())
    return out


# ---- cross entropy ---------------------------------------------------------------------------------------------------------
def xent_fwd(logits, target, C, ignore_index=-1):
    """logits fp32 [R, ld>=C]; returns (loss[R], lse[R], rank[R])."""
    _chk(logits, torch.float32, 'logits')
    R = logits.size(0)
    loss = torch.empty(R, device=logits.device, dtype=torch.float32)
    lse = torch.empty(R, device=logits.device, dtype=torch.float32)
    rank = torch.empty(R, device=logits.device, dtype=torch.int32)
    _lib.call('avt_xent_fwd', _p(logits), _ld(logits), _p(target), _p(loss), _p(lse), _p(rank), R, C, ignore_index, _stream())
    return loss, lse, rank


def xent_bwd(logits, target, lse, gout, C, ldd, ignore_index=-1):
    R = logits.size(0)
    dlogits = torch.empty((R, ldd), device=logits.device, dtype=BF16)
    _lib.call('avt_xent_bwd', _p(logits), _ld(logits), _p(target), _p(lse), _p(gout), _p(dlogits), ldd, R, C, ignore_index, _stream())
    return dlogits


# ---- classifier + softmax cross-entropy as one operator ----------------------------------------------------------------------
def linear_softmax_xent_fwd(x, w, bias, target, C, ignore_index=-1):
    """x bf16 [R, K], w bf16 [Cpad, K] (rows >= C zero), bias fp32 [Cpad] | None, target int64 [R] -> (logits fp32 [R, Cpad], loss, lse, rank)."""
    _chk(x, BF16, 'x'); _chk(w, BF16, 'w')
    R, K = x.shape
    Cpad = w.size(0)
    logits = torch.empty((R, Cpad), device=x.device, dtype=torch.float32)
    loss = torch.empty(R, device=x.device, dtype=torch.float32)
    lse = torch.empty(R, device=x.device, dtype=torch.float32)
    rank = torch.empty(R, device=x.device, dtype=torch.int32)
    _lib.call('avt_linear_softmax_xent_fwd', _p(x), _ld(x), _p(w), _ld(w), _p(bias), _p(target), _p(logits), _ld(logits), _p(loss), _p(lse),
              _p(rank), R, C, Cpad, K, ignore_index, _stream())
    return logits, loss, lse, rank


def linear_softmax_xent_bwd(logits, target, lse, gloss, x, w, C, dw=None, dbias=None, want_dx=True, dx_f32=True, ignore_index=-1,
                            glogits=None, dx_drop_p=0.0, dx_drop_seed=0):
    """Backward of linear_softmax_xent_fwd: accumulates into dw [Cpad, K] / dbias [Cpad] (fp32) and returns dx [R, K] (or None).
    glogits: fp32 [R, >=C] gradient that reached the logits from another consumer (added into dlogits before its bf16 cut).
    dx_drop_p / dx_drop_seed: mask of the dropout in front of the classifier, applied to dx in the dgrad GEMM's epilogue."""
    R, K = x.shape
    Cpad = w.size(0)
    dlogits = torch.empty((R, Cpad), device=x.device, dtype=BF16)
    dx = torch.empty((R, K), device=x.device, dtype=torch.float32 if dx_f32 else BF16) if want_dx else None
    ws = _wgrad_workspace(x.device, _lib.load().avt_gemm_accum_workspace_bytes(Cpad, K, R)) if dw is not None else None
    part, part_bytes = _partials(x.device, 'avt_colsum_workspace_bytes', R, Cpad) if dbias is not None else (None, 0)
    if glogits is not None:
        _chk(glogits, torch.float32, 'glogits')
    _lib.call('avt_linear_softmax_xent_bwd', _p(logits), _ld(logits), _p(target), _p(lse), _p(gloss), _p(glogits),
              _ld(glogits) if glogits is not None else 0, _p(x), _ld(x), _p(w), _ld(w), _p(dlogits),
              _p(dw), _ld(dw) if dw is not None else 0, _p(dbias), _p(dx), _ld(dx) if dx is not None else 0, int(dx_f32),
              float(dx_drop_p), int(dx_drop_seed), R, C, Cpad, K,
              ignore_index, _p(ws), ws.numel() if ws is not None else 0, part, part_bytes, _stream())
    return dx


# ---- optimizer -------------------------------------------------------------------------------------------------------------
def sgd_step(param, grad, buf, shadow, lr, momentum, weight_decay, grad_scale=1.0, nesterov=True, first_step=False, zero_grad=True):
    """``lr``: a float, or a one-element fp32 device tensor (avt_sgd_step_dev: the rate of a captured step, rewritten between replays)."""
    if torch.is_tensor(lr):
        _chk(lr, torch.float32, 'lr')
        _lib.call('avt_sgd_step_dev', _p(param), _p(grad), _p(buf), _p(shadow), param.numel(), _p(lr), float(momentum),
                  float(weight_decay), float(grad_scale), int(nesterov), int(first_step), int(zero_grad), _stream())
        return
    _lib.call('avt_sgd_step', _p(param), _p(grad), _p(buf), _p(shadow), param.numel(), float(lr), float(momentum),
              float(weight_decay), float(grad_scale), int(nesterov), int(first_step), int(zero_grad), _stream())
