"""Fused SGD over the model's flat parameter arena -- drop-in for ``torch.optim.SGD`` in
conf/opt/optimizer/sgd.yaml (momentum 0.9, nesterov per expts/01_ek100_avt.txt:28, weight decay from ``opt.lr_wd``).

One HIP kernel per parameter group range updates fp32 master weights + momentum, emits the bf16 shadow the GEMMs read
and re-zeroes the gradient buffer for the next step's atomic accumulation (avt_sgd_step).  ``param_groups`` keeps
torch's shape ({'params', 'lr', 'momentum', 'weight_decay', 'nesterov'}) so the reference's schedulers and
checkpoint code (func/train.py:52-74) work on it.  Groups with identical hyper-parameters covering adjacent arena
ranges are merged into a single launch.
"""
import torch

from . import ops
from .arena import ParamArena


class FusedSGD:
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, nesterov=False, dampening=0, arena: ParamArena = None):
        assert dampening == 0, 'dampening is not supported'
        if arena is None:
            raise ValueError('FusedSGD needs the model arena (model.arena)')
        self.arena = arena
        params = list(params)
        if params and not isinstance(params[0], dict):
            params = [{'params': params}]
        self.param_groups = []
        for g in params:
            g = dict(g)
            g['params'] = list(g['params'])
            g.setdefault('lr', lr)
            g.setdefault('momentum', momentum)
            g.setdefault('weight_decay', weight_decay)
            g.setdefault('nesterov', nesterov)
            self.param_groups.append(g)
        self.momentum_buf = torch.zeros_like(arena.master)
        self.steps = 0
        self.grad_scale = 1.0            # set to 1/world_size by the gradient reducer (sum all-reduce)
        self._ranges = None
        self._uncovered = []
        self._early_lo = arena.total     # this step's update has already been applied to [_early_lo, total) (step_suffix)
        self.lr_dev = None               # captured steps (func/graph.py): fp32 device tensor, slot i = the learning rate of param_groups[i]

    def _build_ranges(self):
        """[(start, end, group_index)] covering each group's parameters as maximal contiguous arena ranges."""
        a = self.arena
        spans = []
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                n = a.name_of[id(p)]
                spans.append((a.offsets[n], a.offsets[n] + a.sizes[n], gi))
        spans.sort()
        merged = []
        for s, e, gi in spans:
            if merged and merged[-1][2] == gi and merged[-1][1] == s:
                merged[-1] = (merged[-1][0], e, gi)
            else:
                merged.append((s, e, gi))
        return merged

    def zero_grad(self, set_to_none: bool = False):
        """No-op by design: step() re-zeroes the gradient range it consumed."""

    def _launch_ranges(self, lo, hi):
        """The update on the part of every group range inside [lo, hi): one launch per maximal run of adjacent ranges whose hyper-parameters agree."""
        a = self.arena
        if self._ranges is None:
            self._ranges = self._build_ranges()
            self._uncovered, pos = [], 0
            for s0, e0, _ in self._ranges:
                if s0 > pos:
                    self._uncovered.append((pos, s0))
                pos = max(pos, e0)
            if pos < a.total:
                self._uncovered.append((pos, a.total))
        i = 0
        R = self._ranges
        while i < len(R):
            s, e, gi = R[i]
            g = self.param_groups[gi]
            key = (g['lr'], g['momentum'], g['weight_decay'], g['nesterov'])
            j = i + 1
            while j < len(R):
                g2 = self.param_groups[R[j][2]]
                if R[j][0] == e and (g2['lr'], g2['momentum'], g2['weight_decay'], g2['nesterov']) == key:
                    e = R[j][1]
                    j += 1
                else:
                    break
            s, e = max(s, lo), min(e, hi)
            if s < e:
                ops.sgd_step(a.master[s:e], a.grad[s:e], self.momentum_buf[s:e], a.shadow[s:e], g['lr'] if self.lr_dev is None else self.lr_dev[gi:gi + 1], g['momentum'],
                             g['weight_decay'], grad_scale=self.grad_scale, nesterov=bool(g['nesterov']),
                             first_step=(self.steps == 0), zero_grad=True)
            i = j

    @torch.no_grad()
    def step_suffix(self, lo):
        """Apply THIS step's update to [lo, the part already updated) now, on the current stream: the gradients there are final (backward fills the flat
        buffer from its end; func/train.py::Trainer calls this as segments finish, round 6).  ``step()`` then only has [0, lo) left.  ``grad_scale`` and the
        groups' learning rates must already be the step's.  Ranges must be 64-element aligned only in the sense that the arena's tensors are."""
        lo = max(int(lo), 0)
        if lo < self._early_lo:
            self._launch_ranges(lo, self._early_lo)
            self._early_lo = lo

    @torch.no_grad()
    def step(self):
        a = self.arena
        self._launch_ranges(0, self._early_lo)
        self._early_lo = a.total
        for s0, e0 in self._uncovered:           # frozen parameters: nobody consumes their gradients, keep the range clean
            a.grad[s0:e0].zero_()
        ops.mark_zeroed(a.grad)                  # the whole gradient buffer holds zeros again: the next weight gradients are stored, not added (ops.gemm_accum)
        self.steps += 1
        a.mark_shadow_current()

    # ---- checkpoint format: torch.optim.SGD's (func/train.py:52-74 stores optimizer.state_dict(), :760-769 resumes it) ------
    def _flat_params(self):
        return [p for g in self.param_groups for p in g['params']]

    def state_dict(self):
        """Same layout as ``torch.optim.SGD.state_dict()``: ``state[idx]['momentum_buffer']`` per parameter (a copy of
        its slice of the flat momentum buffer) and ``param_groups`` with parameter indices, so a checkpoint written
        here resumes under the reference's torch optimizer and vice versa."""
        a = self.arena
        state, groups, idx = {}, [], 0
        for g in self.param_groups:
            ids = []
            for p in g['params']:
                if self.steps > 0 and g['momentum'] != 0:            # torch keeps no buffer for momentum-free groups
                    n = a.name_of[id(p)]
                    o = a.offsets[n]
                    state[idx] = {'momentum_buffer': self.momentum_buf[o:o + p.numel()].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
            d = {k: v for k, v in g.items() if k != 'params'}
            d.setdefault('dampening', 0)
            d.setdefault('maximize', False)
            d.setdefault('foreach', None)
            d.setdefault('differentiable', False)
            d.setdefault('fused', None)
            d['params'] = ids
            groups.append(d)
        return {'state': state, 'param_groups': groups, 'avt_steps': self.steps}      # extra key: ignored by torch.optim.SGD.load_state_dict

    def load_state_dict(self, sd):
        a = self.arena
        if 'momentum_buf' in sd:                                   # round-1 private format
            self.momentum_buf.copy_(sd['momentum_buf'])
            self.steps = sd['steps']
            for g, s in zip(self.param_groups, sd['param_groups']):
                g.update(s)
            return
        params = self._flat_params()
        saved_groups = sd['param_groups']
        if len(saved_groups) != len(self.param_groups) or any(len(sg['params']) != len(g['params'])
                                                              for sg, g in zip(saved_groups, self.param_groups)):
            raise ValueError('loaded state dict has a different number of parameter groups / parameters')
        order = [i for sg in saved_groups for i in sg['params']]
        loaded = 0
        with torch.no_grad():
            for p, i in zip(params, order):
                st = sd['state'].get(i, sd['state'].get(str(i)))
                if st is None or st.get('momentum_buffer') is None:
                    continue
                n = a.name_of[id(p)]
                o = a.offsets[n]
                self.momentum_buf[o:o + p.numel()].copy_(st['momentum_buffer'].reshape(-1).to(self.momentum_buf.device, torch.float32))
                loaded += 1
        # torch initialises the buffer with the first gradient and after that it is live; a torch-written file has no step count
        self.steps = int(sd.get('avt_steps', 1 if loaded else 0))
        for g, sg in zip(self.param_groups, saved_groups):
            g.update({k: v for k, v in sg.items() if k != 'params'})
        self._ranges = None
