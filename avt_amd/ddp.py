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
    def __init__(self, model, bucket_bytes=64 << 20, process_group=None, overlap=True, always=False, mode='all_reduce',
                 wire_dtype=torch.float32, tail_bytes=None, transport='torch', comm=None):
        self.model = model
        self.arena: ParamArena = model.arena
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.group = process_group
        # 'all_reduce': one RCCL all-reduce per bucket (RCCL picks ring / tree / direct for the xGMI mesh itself).
        # 'rs_ag': the same sum as an explicit reduce-scatter + all-gather pair per bucket (SURVEY 8e's full-mesh form: every
        # rank reduces 1/world of the bucket, then the shards are exchanged) -- selectable so the two can be compared on a
        # multi-GPU node; bucket edges are multiples of 64 * world elements so that every shard is 16-byte aligned.
        if mode not in ('all_reduce', 'rs_ag'):
            raise ValueError(f"GradReducer mode must be 'all_reduce' or 'rs_ag' (got {mode!r})")
        self.mode = mode
        # transport 'torch': torch.distributed collectives (backend 'nccl' = RCCL; gloo for the functional checks).  'abi': the same exchange through
        # the library's own RCCL entry points (avt_allreduce_bucket / avt_reduce_scatter_bucket / avt_allgather_bucket, include/avt_hip.h) -- what a
        # maintainer who binds only the .so runs; ``comm`` = an avt_amd.comm.RcclComm, or None to build one over the torch group's side channel.
        if transport not in ('torch', 'abi'):
            raise ValueError(f"GradReducer transport must be 'torch' or 'abi' (got {transport!r})")
        self.transport = transport
        self.comm = comm
        if transport == 'abi' and comm is None:
            from .comm import RcclComm
            self.comm = RcclComm.from_torch_group(self.arena.grad.device.index, process_group)
        if transport == 'abi' and self.comm.nranks != max(self.world, 1):
            raise ValueError(f'GradReducer: the RCCL communicator has {self.comm.nranks} ranks, the job {self.world}')
        # wire_dtype=torch.bfloat16: each bucket is cast to bf16, summed on the wire in bf16 and widened back into the fp32
        # buffer (half the xGMI bytes: 0.79 GB instead of 1.58 GB per step) -- an OPTION, because a bf16 sum over the
        # ranks keeps 8 bits of the gradient's mantissa; the default exchanges fp32.
        if wire_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('wire_dtype must be torch.float32 or torch.bfloat16')
        self.wire_dtype = wire_dtype
        quantum = 64 * max(self.world, 1)
        self.bucket_elems = max(bucket_bytes // 4 // quantum, 1) * quantum
        # The exchange that cannot hide behind backward is the LAST one (sent by finish(), after the first layers' backward).  Keep it
        # small: as soon as the not-yet-produced head of the buffer is at most ``tail_bytes`` (default: half a bucket -- at 64 MiB that
        # is the ViT's patch embedding + block 0, 31.5 MB), everything produced so far goes out at once instead of waiting for a
        # full bucket, so finish() only has that head left.
        # (never more than a bucket: with a 'single bucket' setting the default would otherwise turn every segment into its own collective)
        self.tail_elems = min((bucket_bytes // 2 if tail_bytes is None else tail_bytes) // 4, self.bucket_elems)
        self._tail_sent = False             # the early tail flush happens ONCE per step
        self.overlap = overlap
        self.paused = False                 # True: no exchange at all (bench: the same step without collectives in flight)
        self.comm_stream = torch.cuda.Stream() if self.arena.grad.is_cuda else None
        self._lo = self.arena.total         # everything in [_lo, total) has been produced
        self._sent = self.arena.total       # everything in [_sent, total) has been handed to the collective
        self._handles = []
        self.always = always                # run the collectives even for a world of one (exercises RCCL on a one-GPU box)
        self.launched = 0                   # buckets handed to the collective in the current step
        self.bytes_on_wire = 0              # payload bytes of those buckets (per rank, before the algorithm's own factor)
        self._timing = []                   # per step: (backward done on the compute stream, last collective done on the comm stream)
        # on_final(lo, stream): everything in [lo, total) of the gradient buffer now holds its FINAL value for this step -- produced and, in a data-parallel
        # job, exchanged.  `stream` = the stream that is current and on which that is true (the exchange's side stream), or None = the compute stream,
        # right behind the producing kernels.  func/train.py::Trainer applies the fused optimizer to that suffix at once (round 6): the head's 78 % of the
        # parameters are updated while the ViT's backward still runs, instead of in one weight-sized pass after it.
        self.on_final = None
        self._hooked = [m for m in model.modules() if hasattr(m, 'grad_ready_hook')]
        self._seen = {}
        self._fwd_calls = {}
        for m in self._hooked:
            m.grad_ready_hook = (lambda first, last, _m=m: self._segment_ready(_m, first, last))
            # a module that runs forward k times per step (multi-crop) accumulates k times into its gradient range: count the
            # forwards here, for EVERY hooked module, instead of trusting each module to do its own book-keeping
            m.register_forward_pre_hook(self._count_forward)

    def _count_forward(self, module, _inputs):
        if torch.is_grad_enabled():
            self._fwd_calls[id(module)] = self._fwd_calls.get(id(module), 0) + 1

    @staticmethod
    def broadcast_parameters(model, src=0):
        """DDP constructor semantics: every rank starts from rank ``src``'s parameters (func/train.py:775-778)."""
        arena = model.arena
        dist.broadcast(arena.master, src=src)
        arena.refresh_shadow(force=True)

    def broadcast_parameters_abi(self, src=0):
        """The same through the C ABI's communicator (transport 'abi')."""
        self.comm.broadcast(self.arena.master, root=src)
        self.arena.refresh_shadow(force=True)

    @staticmethod
    def broadcast_optimizer_state(optimizer, src=0):
        """Every rank continues from rank ``src``'s momentum buffers (after steps in which the replicas were allowed to drift apart)."""
        buf = getattr(optimizer, 'momentum_buf', None)             # avt_amd.optim.FusedSGD: one flat buffer
        if torch.is_tensor(buf):
            dist.broadcast(buf, src=src)
        for st in getattr(optimizer, 'state', {}).values():        # a torch optimizer: per-parameter state tensors
            for v in (st.values() if isinstance(st, dict) else ()):
                if torch.is_tensor(v) and v.numel() > 1:
                    dist.broadcast(v, src=src)

    def start_step(self):
        self._lo = self._sent = self.arena.total
        self._handles = []
        self._seen = {}
        self._fwd_calls = {}
        self.launched = 0
        self.bytes_on_wire = 0
        self._tail_sent = False

    def _segment_ready(self, module, first_param, last_param):
        """A fused node finished writing the gradients of [first_param, last_param].  A module that ran forward k times
        this step (multi-crop clips: one node per crop, models/base_model.py:251-273) accumulates into the same range k
        times: the range only counts as finished after the k-th backward."""
        a = self.arena
        key = id(first_param)
        self._seen[key] = self._seen.get(key, 0) + 1
        if self._seen[key] < max(self._fwd_calls.get(id(module), 1), 1):
            return
        start = a.offsets[a.name_of[id(first_param)]]
        self._lo = min(self._lo, start)
        if (self.world > 1 or self.always) and self.overlap and not self.paused:
            while self._sent - self._lo >= self.bucket_elems:
                self._launch(self._sent - self.bucket_elems, self._sent)
            if not self._tail_sent and 0 < self._lo <= self.tail_elems and self._sent > self._lo:
                # one early flush per step (round-4 advisor finding: without the flag every later segment launched its own small collective,
                # each with an event and a stream wait); what is produced after it goes out with finish()
                quantum = 64 * max(self.world, 1)
                lo = (self._lo + quantum - 1) // quantum * quantum          # keep bucket edges on shard boundaries (rs_ag)
                if lo < self._sent:
                    self._launch(lo, self._sent)
                    self._tail_sent = True
        elif self.on_final is not None and self.world <= 1 and not self.always and not self.paused:
            self.on_final(self._lo, None)       # a single process: final as produced

    def _launch(self, s, e, notify=True):
        a = self.arena
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                h = self._reduce(a.grad[s:e])
                if notify and self.on_final is not None:
                    h.wait()                    # (nccl: orders the side stream behind the collective; gloo: blocks the host -- a functional path)
                    self.on_final(s, self.comm_stream)
        else:
            h = self._reduce(a.grad[s:e])
            if notify and self.on_final is not None:
                h.wait()
                self.on_final(s, None)
        self._handles.append(h)
        self.launched += 1
        self._sent = s

    def _reduce(self, buf):
        """Sum ``buf`` (a slice of the flat fp32 gradient buffer) over the ranks, in place; returns the last async handle."""
        if self.wire_dtype != torch.float32:
            wire = buf.to(self.wire_dtype)                      # on the comm stream, ordered after the producers
            self._exchange(wire).wait()                         # (stream-ordered on nccl: wait() blocks the stream, not the host)
            buf.copy_(wire)
            return _Done()
        return self._exchange(buf)

    def _exchange(self, buf):
        n = buf.numel()
        self.bytes_on_wire += n * buf.element_size()
        if self.transport == 'abi':                             # stream-ordered RCCL calls through the C ABI, in place
            if self.mode == 'rs_ag' and n % self.world == 0 and (n // self.world * buf.element_size()) % 16 == 0:
                self.comm.reduce_scatter(buf)
                self.comm.all_gather(buf)
            else:
                self.comm.all_reduce(buf)
            return _Done()
        if self.mode == 'rs_ag' and n % self.world == 0:
            # (bucket edges are multiples of 64 * world, so finish()'s bucket [0, lo) and every full bucket divide; the bucket that can fail
            #  to is the one whose upper end is the arena's end -- the first full bucket or the early-tail bucket [lo, total): the arena's size
            #  is a multiple of 64, not of 64 * world, e.g. on 3, 5, 6 or 7 ranks -- and it goes out as a plain all-reduce below: the same sum)
            shard = buf.view(self.world, n // self.world)[dist.get_rank(self.group)]       # in place: the rank's own chunk
            dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group)
            src = shard if dist.get_backend(self.group) == 'nccl' else shard.clone()       # only RCCL gathers in place
            return dist.all_gather_into_tensor(buf, src, group=self.group, async_op=True)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Reduce whatever is left (everything, if no hook fired) and order the compute stream after the collectives."""
        if (self.world <= 1 and not self.always) or self.paused:
            return
        timed = self.comm_stream is not None
        if timed:
            bwd_done = torch.cuda.Event(enable_timing=True)
            bwd_done.record(torch.cuda.current_stream())
        tail = self._sent > 0
        if tail:
            self._launch(0, self._sent, notify=False)
        for h in self._handles:
            h.wait()
        if timed:
            comm_done = torch.cuda.Event(enable_timing=True)
            comm_done.record(self.comm_stream)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._timing.append((bwd_done, comm_done))
            del self._timing[:-64]
        self._handles = []
        if tail and self.on_final is not None:
            self.on_final(0, None)              # (the compute stream is ordered behind every collective here)

    def stats(self, last=None):
        """Per-rank exchange accounting for the bench line: buckets and payload bytes of the last step, and how long the
        optimizer had to wait for the collectives after backward had finished (``comm_exposed_ms``: mean / max over the last
        ``last`` steps; 0 when the exchange was fully hidden behind backward).  Synchronises the device."""
        out = {'mode': self.mode, 'transport': self.transport, 'wire_dtype': str(self.wire_dtype).replace('torch.', ''), 'buckets_per_step': self.launched,
               'bytes_per_step': self.bytes_on_wire, 'bucket_bytes': self.bucket_elems * 4}
        if self._timing:
            torch.cuda.synchronize()
            ms = [max(a.elapsed_time(b), 0.0) for a, b in (self._timing[-last:] if last else self._timing)]
            out['comm_exposed_ms'] = round(sum(ms) / len(ms), 3)
            out['comm_exposed_ms_max'] = round(max(ms), 3)
        return out


class _Done:
    """Handle of an exchange whose work has already been enqueued in stream order."""
    def wait(self):
        return True
