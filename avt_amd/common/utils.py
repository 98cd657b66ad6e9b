"""Host helpers of the path: top-k accuracy from the fused cross-entropy kernel's target ranks (the numbers of the reference's
common/utils.py:17-44 ``accuracy``) and one-process-per-GPU distributed init (:106-150)."""
import os

import torch
import torch.distributed as dist


def accuracy_from_rank(rank, target, topk=(1,)):
    """The reference's top-k accuracy (percent over ALL rows) from the target-rank vector the fused cross-entropy kernel emits (rank = number of logits
    strictly above the target's, -1 for ignored rows): a row is a top-k hit iff 0 <= rank < k.  Rows with ignored targets
    count as misses and an all-ignored batch gives zeros, as in the reference (common/utils.py:17-44)."""
    with torch.no_grad():
        rank = rank.flatten()
        batch_size = target.numel()
        valid = rank >= 0
        return [((rank < k) & valid).sum(dtype=torch.float32) * (100.0 / batch_size) for k in topk]


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def init_distributed_mode(backend=None, allow_single=False):
    """One process per GPU, rank/world from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    ``backend='nccl'`` is RCCL on ROCm (xGMI inside a node); ``gloo`` is used by the CPU tests.  A single process stays
    un-initialised unless ``allow_single`` (used to exercise RCCL on a one-GPU box)."""
    if 'RANK' not in os.environ or (int(os.environ.get('WORLD_SIZE', '1')) <= 1 and not allow_single):
        return False, 0, 1, 0
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    if not dist.is_initialized():
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend=backend, init_method='env://', world_size=world, rank=rank, **kw)
    return True, rank, world, local
