"""RCCL communicator behind the C ABI (include/avt_hip.h, "gradient exchange over RCCL"): the collectives of the data-parallel step without
torch.distributed underneath -- what a maintainer who binds only ``libavt_hip.so`` gets (reference: func/train.py:771-778, common/utils.py:145-148).
``GradReducer(transport='abi')`` runs the bucketed gradient exchange through it; the default transport stays torch.distributed (backend 'nccl' = the
same RCCL).  One process per GPU; the 128-byte id travels over whatever side channel the host has -- here an already initialised torch.distributed
group of any backend (``from_torch_group``) or the caller's own bytes."""
import ctypes

import torch

from . import lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


class RcclComm:
    def __init__(self, nranks, rank, device, uid: bytes):
        if len(uid) != 128:
            raise ValueError('the RCCL unique id is 128 bytes (RcclComm.unique_id() on rank 0)')
        self._h = ctypes.c_void_p()
        self.device = int(device)
        lib.call('avt_comm_init_rank', ctypes.byref(self._h), int(nranks), int(rank), self.device, ctypes.c_char_p(uid))
        self.nranks, self.rank = self.size()

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        lib.call('avt_comm_unique_id', buf)
        return buf.raw

    @classmethod
    def from_torch_group(cls, device, group=None):
        """Rank / world size and the id's side channel from an initialised torch.distributed group (any backend); world 1 without one."""
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls(1, 0, device, cls.unique_id())
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(world, rank, device, box[0])

    def size(self):
        n, r = ctypes.c_int(), ctypes.c_int()
        lib.call('avt_comm_size', self._h, ctypes.byref(n), ctypes.byref(r))
        return n.value, r.value

    def _args(self, t):
        if not (t.is_cuda and t.is_contiguous() and t.dtype in _DT and t.device.index == self.device):
            raise lib.AvtHipError(f'RcclComm: a contiguous fp32 / bf16 tensor on cuda:{self.device} is needed (got {t.dtype} on {t.device})')
        return ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(t.numel()), _DT[t.dtype]

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def all_reduce(self, t):
        """Sum over the ranks, in place, enqueued on the current stream."""
        p, n, d = self._args(t)
        lib.call('avt_allreduce_bucket', self._h, p, n, d, self._stream())

    def reduce_scatter(self, t):
        """In place: this rank's shard [rank n / W, (rank + 1) n / W) of ``t`` receives the sum over the ranks."""
        p, n, d = self._args(t)
        lib.call('avt_reduce_scatter_bucket', self._h, p, n, d, self._stream())

    def all_gather(self, t):
        """In place: every rank's shard of ``t`` is distributed to all ranks."""
        p, n, d = self._args(t)
        lib.call('avt_allgather_bucket', self._h, p, n, d, self._stream())

    def broadcast(self, t, root=0):
        p, n, d = self._args(t)
        lib.call('avt_broadcast_bucket', self._h, p, n, d, int(root), self._stream())

    def destroy(self):
        if self._h:
            torch.cuda.synchronize(self.device)
            lib.call('avt_comm_destroy', self._h)
            self._h = ctypes.c_void_p()
