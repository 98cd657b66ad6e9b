"""Flat parameter storage for the HIP path: one fp32 master buffer, one fp32 gradient buffer and one bf16 shadow
buffer, all with identical layout (registration order == forward order, so backward completes from the END of the
buffers towards the start -- that is what the gradient all-reduce buckets rely on).

Every ``nn.Parameter`` of the model becomes a view into the master buffer, its ``.grad`` a view into the gradient
buffer; the GEMM kernels read the bf16 shadow.  Each tensor is padded to a multiple of 64 elements (16-byte
alignment for every dtype, and room for the classifier's row padding); padding is zero and stays zero.

Reference behaviour mirrored: parameters stay ordinary fp32 ``nn.Parameter``s with the reference's state_dict
names, so ``torch.save(model.state_dict())`` / ``load_state_dict`` / ``torch.optim.SGD`` keep working
(func/train.py:52-74, 457-497, 744).  Loading or stepping through torch bumps the tensors' version counters; the
shadow is refreshed lazily from that.
"""
from typing import Dict, Optional

import torch

from . import ops

ALIGN = 64


def _round_up(n, a=ALIGN):
    return (n + a - 1) // a * a


class ParamArena:
    def __init__(self, module: torch.nn.Module, padded_numel: Optional[Dict[str, int]] = None):
        padded_numel = padded_numel or {}
        params = [(n, p) for n, p in module.named_parameters()]
        assert params, 'module has no parameters'
        dev = params[0][1].device
        if dev.type != 'cuda':
            raise RuntimeError('ParamArena needs the model on a GPU (call model.to("cuda") first); there is no CPU path')
        self.device = dev
        self.names, self.offsets, self.sizes, self.shapes = [], {}, {}, {}
        off = 0
        for n, p in params:
            assert p.dtype == torch.float32, f'{n}: parameters must be fp32 master copies'
            size = _round_up(max(padded_numel.get(n, 0), p.numel()))
            self.names.append(n)
            self.offsets[n], self.sizes[n], self.shapes[n] = off, size, tuple(p.shape)
            off += size
        self.total = off
        self.master = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.shadow = torch.zeros(off, device=dev, dtype=torch.bfloat16)
        self.params = {}
        self.name_of = {}
        with torch.no_grad():
            for n, p in params:
                o, k = self.offsets[n], p.numel()
                view = self.master[o:o + k].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + k].view(p.shape)
                self.params[n] = p
                self.name_of[id(p)] = n
        self._plist = [self.params[n] for n in self.names]
        self._version = None
        self._transposed = {}            # name -> bf16 (in, out) copy of a Linear weight's shadow (see transposed_of)
        self._transpose_jobs = None      # device job table of refresh_transposed
        self._folded = {}                # weight name -> FoldedLinear (LayerNorm folded into the Linear behind it, see folded_of)
        self._fold_jobs = None           # device job table of the G -> G^T transposes
        self._fold_scratch = {}          # (N, K) -> zeroed fp32 scratch of the raw weight gradient; N -> zeroed bias-gradient scratch
        self._fold_dirty = False         # the masters changed since the folded forms were made (re-formed at their next use)
        self.fold_scratch_busy = False   # a backward is between filling and consuming a shared scratch (fold_scratch_guard)
        self.refresh_shadow(force=True)

    # ---- views -------------------------------------------------------------------------------------------------
    def shadow_of(self, name, rows=None):
        """bf16 view with the parameter's shape; ``rows`` widens dim 0 into the (zero) padding."""
        o, shp = self.offsets[name], self.shapes[name]
        if rows is not None:
            shp = (rows,) + shp[1:]
        k = 1
        for s in shp:
            k *= s
        assert k <= self.sizes[name]
        return self.shadow[o:o + k].view(shp)

    def grad_of(self, name, rows=None):
        o, shp = self.offsets[name], self.shapes[name]
        if rows is not None:
            shp = (rows,) + shp[1:]
        k = 1
        for s in shp:
            k *= s
        assert k <= self.sizes[name]
        return self.grad[o:o + k].view(shp)

    def master_padded(self, name):
        o = self.offsets[name]
        return self.master[o:o + self.sizes[name]]

    # ---- consistency ---------------------------------------------------------------------------------------------
    def is_valid(self):
        """False once someone re-allocated the parameters (e.g. ``model.to()``): views no longer alias the arena."""
        for n in (self.names[0], self.names[-1]):
            p = self.params[n]
            if p.data_ptr() != self.master.data_ptr() + 4 * self.offsets[n]:
                return False
        return True

    def _current_version(self):
        return sum(self.params[n]._version for n in self.names)

    def refresh_shadow(self, force=False):
        """Re-cast master -> bf16 shadow when a torch-side write (load_state_dict, torch optimizer) touched it."""
        v = self._current_version()
        if force or v != self._version:
            ops.cast_to_bf16(self.master, self.shadow)
            self._version = v
            self.refresh_transposed()

    def mark_shadow_current(self):
        self._version = self._current_version()
        self.refresh_transposed()

    def transposed_of(self, name):
        """bf16 W^T (in, out) of the 2-D parameter ``name``, kept next to the shadow and refreshed whenever the shadow is
        (one batched 64x64-tile transpose launch per optimizer step).  With it the data
        gradient dx = dy W reads the weight k-major like the forward does: qkv / fc1 / fc2 data gradients run 5-9 % faster
        than through transposing LDS reads (measured: 754 vs 796, 953 vs 1036, 1158 vs 1273 us at 128 clips)."""
        t = self._transposed.get(name)
        if t is None:
            out_f, in_f = self.shapes[name]
            t = torch.empty((in_f, out_f), device=self.device, dtype=torch.bfloat16)
            self._transposed[name] = t
            self._transpose_jobs = None
            ops.transpose_into(self.shadow_of(name), t)
        return t

    def refresh_transposed(self):
        """All registered W^T copies in one launch (a device table of (source, destination) records, rebuilt when a matrix is added)."""
        if self._transposed:
            if self._transpose_jobs is None:
                self._transpose_jobs = ops.transpose_jobs([(self.shadow_of(n), t) for n, t in self._transposed.items()])
            ops.transpose_batch(self._transpose_jobs)
        self.refresh_folded()

    # ---- LayerNorm folded into the Linear behind it (csrc/lnfold.hip) ------------------------------------------------------
    def folded_of(self, wname, gname, bname, biasname):
        """The folded form of LayerNorm(gamma, beta) -> Linear(W, bias): G = gamma o W and G^T (bf16), c = G 1, b' = bias + W beta, re-formed
        from the fp32 masters whenever the shadow is refreshed (every optimizer step), plus the scratch its backward accumulates into."""
        f = self._folded.get(wname)
        if f is None:
            f = FoldedLinear(self, wname, gname, bname, biasname)
            self._folded[wname] = f
            self._fold_jobs = None
            f.refresh()
            ops.transpose_into(f.G, f.Gt)
        return f

    def fold(self, weight, gamma, beta, bias):
        if self._fold_dirty:
            self._refresh_folded_now()
        return self.folded_of(self.name_of[id(weight)], self.name_of[id(gamma)], self.name_of[id(beta)], self.name_of[id(bias)])

    def refresh_folded(self):
        """The masters changed: the folded forms are stale.  They are re-formed at their next USE (``fold``), not here -- a step below
        ``HipViT.fold_min_rows`` token rows does not use them, and re-folding 22 weights per optimizer step for nothing cost 0.2 ms (round-5 advisor)."""
        self._fold_dirty = bool(self._folded)

    def _refresh_folded_now(self):
        self._fold_dirty = False
        for f in self._folded.values():
            f.refresh()
        if self._fold_jobs is None:
            self._fold_jobs = ops.transpose_jobs([(f.G, f.Gt) for f in self._folded.values()])
        ops.transpose_batch(self._fold_jobs)

    def fold_scratch_guard(self):
        """The folded layers of one shape share their backward scratch (T, dbt), which every use leaves zeroed (ln_fold_wgrad).  A backward that died
        between filling and consuming it (out of memory, an interrupt) leaves it dirty: the next backward would add that into another layer's gradient.
        ``fold_scratch_busy`` is set while a scratch is in use; a backward that finds it set zeroes every scratch first (round-5 advisor)."""
        if self.fold_scratch_busy:
            for t in self._fold_scratch.values():
                t.zero_()
            self.fold_scratch_busy = False

    def master_of(self, name):
        o, shp = self.offsets[name], self.shapes[name]
        k = 1
        for s_ in shp:
            k *= s_
        return self.master[o:o + k].view(shp)

    def grads_attached(self):
        """True when every parameter still has a ``.grad`` (a torch optimizer's ``zero_grad(set_to_none=True)`` drops the
        gradients of ITS OWN groups only -- possibly a middle slice of the arena, e.g. the future predictor alone with the
        backbone frozen -- so every parameter is looked at; the look is one attribute read each) and the first / last one
        still alias the flat buffer (re-allocation, e.g. ``model.to()``, moves all of them together)."""
        for p in self._plist:
            if p.grad is None:
                return False
        for n in (self.names[0], self.names[-1]):
            if self.params[n].grad.data_ptr() != self.grad.data_ptr() + 4 * self.offsets[n]:
                return False
        return True

    def attach_grads(self, zero=True):
        """Re-attach ``.grad`` views dropped by ``zero_grad(set_to_none=True)``; zero the buffer when that happened
        (``zero=False``: only re-attach -- the buffer already holds this step's gradients).  Called at the start of every
        fused forward AND backward node, so the reference's order forward -> optimizer.zero_grad() -> backward -> step
        (func/train.py:221-233) works with any torch optimizer."""
        if self.grads_attached():
            return
        dropped = False
        for n in self.names:
            p = self.params[n]
            o = self.offsets[n]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                dropped = True
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        if dropped and zero:
            self.grad.zero_()

    def zero_grad(self):
        self.grad.zero_()

    # ---- lookups by parameter object (sub-modules do not know their prefix) -------------------------------------
    def sh(self, param, rows=None):
        return self.shadow_of(self.name_of[id(param)], rows)

    def gr(self, param, rows=None):
        return self.grad_of(self.name_of[id(param)], rows)

    def sh_t(self, param):
        return self.transposed_of(self.name_of[id(param)])


class FoldedLinear:
    """LayerNorm(gamma, beta) followed by Linear(W, bias), folded: see csrc/lnfold.hip and include/avt_hip.h (avt_gemm_ln_bf16)."""
    def __init__(self, arena, wname, gname, bname, biasname):
        self.arena, self.wname, self.gname, self.bname, self.biasname = arena, wname, gname, bname, biasname
        N, K = arena.shapes[wname]
        dev = arena.device
        self.N, self.K = N, K
        self.G = torch.empty((N, K), device=dev, dtype=torch.bfloat16)
        self.Gt = torch.empty((K, N), device=dev, dtype=torch.bfloat16)
        self.c = torch.empty(N, device=dev, dtype=torch.float32)
        self.b2 = torch.empty(N, device=dev, dtype=torch.float32)
        sc = arena._fold_scratch
        if (N, K) not in sc:
            sc[(N, K)] = torch.zeros((N, K), device=dev, dtype=torch.float32)
        if N not in sc:
            sc[N] = torch.zeros(N, device=dev, dtype=torch.float32)
        self.T, self.dbt = sc[(N, K)], sc[N]            # shared by every folded layer of this shape: used and re-zeroed one layer at a time

    def masters(self):
        a = self.arena
        return a.master_of(self.wname), a.master_of(self.gname), a.master_of(self.bname), a.master_of(self.biasname)

    def grads(self):
        a = self.arena
        return a.grad_of(self.wname), a.grad_of(self.gname), a.grad_of(self.bname), a.grad_of(self.biasname)

    def refresh(self):
        W, g, b, bias = self.masters()
        ops.ln_fold_weights(W, g, b, bias, self.G, self.c, self.b2)

    def backward_weights(self):
        """T (= dY'^T x, accumulated by the caller) and dbt (= colsum(dY)) -> dW, dgamma, dbeta, dbias of the arena's gradient buffer."""
        W, g, b, _ = self.masters()
        dW, dg, db, dbias = self.grads()
        ops.ln_fold_wgrad(self.T, W, g, b, self.dbt, dW, dg, db, dbias)


def get_arena(module: torch.nn.Module, padded_numel_fn=None) -> ParamArena:
    """Arena shared by ``module`` and all its children; (re)built when absent or invalidated by ``.to()``."""
    arena = module.__dict__.get('_avt_arena')
    if arena is not None and arena.is_valid() and all(id(p) in arena.name_of for p in module.parameters()):
        return arena
    padded = {}
    for name, m in module.named_modules():
        fn = getattr(m, 'avt_padded_numel', None)
        if fn is not None:
            for pn, k in fn().items():
                padded[(name + '.' if name else '') + pn] = k
    arena = ParamArena(module, padded)
    for m in module.modules():
        m.__dict__['_avt_arena'] = arena
    return arena
