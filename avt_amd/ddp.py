"""Data-parallel gradient exchange for the flat arena -- the MI355X counterpart of the reference's
``DistributedDataParallel(model, device_ids=[gpu])`` wrap (func/train.py:771-778).

One process per GPU; ``torch.distributed`` backend ``nccl`` (= RCCL, xGMI inside the node).  Because every gradient
lives in ONE flat fp32 buffer whose layout follows forward order, backward finishes the buffer from its END towards
its start.  The backward autograd nodes report finished segments through ``grad_ready_hook(first_param, last_param)``;
the reducer cuts the finished region into buckets and launches ``all_reduce(SUM)`` for each bucket on a side stream as
soon as the kernels producing it have been enqueued (event-ordered), so the 1.2 GB of AVT-h gradients -- produced
first -- travel while the long ViT backward still runs.  ``finish()`` makes the compute stream wait for the tail.
The 1/world averaging is folded into the fused optimizer (``grad_scale``), not into a separate pass.
Ring all-reduce on xGMI is per-link bound, so buckets are large (default 64 MiB: big enough to keep the link pipelines full,
small enough that only the last bucket -- the ViT's first block and the patch embedding -- is exposed after backward).
"""
import torch
import torch.distributed as dist

from .arena import ParamArena


class GradReducer:
    def __init__(self, model, bucket_bytes=64 << 20, process_group=None, overlap=True, always=False, mode='all_reduce'):
        self.model = model
        self.arena: ParamArena = model.arena
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.group = process_group
        # 'all_reduce': one RCCL all-reduce per bucket (RCCL picks ring / tree / direct for the xGMI mesh itself).
        # 'rs_ag': the same sum as an explicit reduce-scatter + all-gather pair per bucket (SURVEY 8e's full-mesh form: every
        # rank reduces 1/world of the bucket, then the shards are exchanged) -- selectable so the two can be compared on a
        # multi-GPU node; bucket edges are multiples of 64 * world elements so that every shard is 16-byte aligned.
        assert mode in ('all_reduce', 'rs_ag')
        self.mode = mode
        quantum = 64 * max(self.world, 1)
        self.bucket_elems = max(bucket_bytes // 4 // quantum, 1) * quantum
        self.overlap = overlap and self.arena.grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.arena.grad.is_cuda else None
        self._lo = self.arena.total         # everything in [_lo, total) has been produced
        self._sent = self.arena.total       # everything in [_sent, total) has been handed to the collective
        self._handles = []
        self.always = always                # run the collectives even for a world of one (exercises RCCL on a one-GPU box)
        self.launched = 0                   # collectives launched in the current step
        self._hooked = [m for m in model.modules() if hasattr(m, 'grad_ready_hook')]
        self._seen = {}
        for m in self._hooked:
            m.grad_ready_hook = (lambda first, last, _m=m: self._segment_ready(_m, first, last))

    @staticmethod
    def broadcast_parameters(model, src=0):
        """DDP constructor semantics: every rank starts from rank ``src``'s parameters (func/train.py:775-778)."""
        arena = model.arena
        dist.broadcast(arena.master, src=src)
        arena.refresh_shadow(force=True)

    def start_step(self):
        self._lo = self._sent = self.arena.total
        self._handles = []
        self._seen = {}
        self.launched = 0
        for m in self._hooked:
            m._fwd_calls = 0

    def _segment_ready(self, module, first_param, last_param):
        """A fused node finished writing the gradients of [first_param, last_param].  A module that ran forward k times
        this step (multi-crop clips: one node per crop, models/base_model.py:251-273) accumulates into the same range k
        times: the range only counts as finished after the k-th backward."""
        a = self.arena
        key = id(first_param)
        self._seen[key] = self._seen.get(key, 0) + 1
        if self._seen[key] < max(getattr(module, '_fwd_calls', 1), 1):
            return
        start = a.offsets[a.name_of[id(first_param)]]
        self._lo = min(self._lo, start)
        if (self.world > 1 or self.always) and self.overlap:
            while self._sent - self._lo >= self.bucket_elems:
                self._launch(self._sent - self.bucket_elems, self._sent)

    def _launch(self, s, e):
        a = self.arena
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                h = self._reduce(a.grad[s:e])
        else:
            h = self._reduce(a.grad[s:e])
        self._handles.append(h)
        self.launched += 1
        self._sent = s

    def _reduce(self, buf):
        n = buf.numel()
        if self.mode == 'rs_ag' and n % self.world == 0 and dist.get_backend(self.group) == 'nccl':
            shard = buf.view(self.world, n // self.world)[dist.get_rank(self.group)]       # in-place: the rank's own chunk
            dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group)
            return dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=True)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Reduce whatever is left (everything, if no hook fired) and order the compute stream after the collectives."""
        if self.world <= 1 and not self.always:
            return
        if self._sent > 0:
            self._launch(0, self._sent)
        for h in self._handles:
            h.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._handles = []
