"""A training step recorded once into a hipGraph and replayed (include/avt_hip.h "captured steps", ABI 9).

The reference's train_one_epoch body (func/train.py:203-239) -- op(data), weighted loss, backward, optimizer step, scheduler step -- is ~640 kernel launches at
its own 3 clips per GPU, and Python + the HIP runtime need as long to issue them (10.7 ms) as the device to run them (12.3 ms); at 1-2 clips the host is the
limit.  ``CapturedStep`` records the launches of one ``Trainer`` step through ``torch.cuda.CUDAGraph`` (stream capture: every entry point of the C ABI enqueues
on the caller's stream, none allocates or synchronises) and replays them; what changes from step to step does not sit in the launches' arguments:

  * the batch: copied into the captured step's own input tensors;
  * dropout seeds: indirect (avt_amd/seeds.py) -- the base seeds of the step are drawn on the host exactly as the eager modules draw them and written to
    their device slots before the replay;
  * learning rates: device-resident (``avt_sgd_step_dev``), one slot per parameter group, written from ``param_groups[i]['lr']`` before the replay; the
    scheduler is stepped on the host after it, as in the eager loop.

Replays and eager steps give the same bits (tests/test_model_gpu.py::test_captured_step_equals_the_eager_steps).  Single process, fused SGD, no gradient clipping
(the clip coefficient is a host decision on a device value); a data-parallel job keeps the eager loop (its exchange runs on a second stream with host-side
bucket logic).
"""
import torch

from .. import seeds
from ..optim import FusedSGD


def _clone_tree(x):
    if torch.is_tensor(x):
        return x.clone()
    if isinstance(x, dict):
        return {k: _clone_tree(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_clone_tree(v) for v in x)
    return x


def _copy_tree(dst, src):
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_tree(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s_ in zip(dst, src):
            _copy_tree(d, s_)


class CapturedStep:
    def __init__(self, trainer, data, warmup=2):
        opt = trainer.optimizer
        if not isinstance(opt, FusedSGD) or trainer.reducer is not None or trainer.max_norm is not None:
            raise ValueError('CapturedStep: single process, FusedSGD, no gradient clipping')
        self.trainer, self.opt = trainer, opt
        dev = trainer.model.arena.master.device
        self.data = _clone_tree(data)
        self.stream = torch.cuda.Stream(device=dev)
        # eager steps on the capturing stream first: allocations, per-stream workspaces and ticket blocks, lazily built weight copies, first_step of the SGD
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(max(int(warmup), 1)):
                trainer.step(self.data)
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        assert opt.steps > 0
        self.seeds = seeds.SeedCapture(dev)
        self.lr_dev = torch.zeros(len(opt.param_groups), dtype=torch.float32, device=dev)
        self._lr_last = [None] * len(opt.param_groups)
        self.graph = torch.cuda.CUDAGraph()
        opt.lr_dev = self.lr_dev
        opt.grad_scale = 1.0 / trainer.world
        steps_before = opt.steps
        try:
            with seeds.capturing(self.seeds), torch.cuda.graph(self.graph, stream=self.stream):
                _, self.outputs, self.losses, self.accuracies = trainer.op(self.data, train_mode=True)
                self.loss = trainer.total_loss(self.losses)
                self.loss.backward()
                opt.step()
        finally:
            opt.lr_dev = None
        opt.steps = steps_before              # (recording ran no kernel: no step was taken)
        self.replays = 0

    def step(self, data=None):
        """One training step.  ``data`` (same structure and shapes as at capture) is copied into the step's input tensors; None = train on what is there.
        Returns (loss, outputs, losses, accuracies): the captured step's own tensors, overwritten by the next replay."""
        opt, dev = self.opt, self.lr_dev.device
        cur = torch.cuda.current_stream(dev)
        if data is not None:
            _copy_tree(self.data, data)
        self.seeds.draw()
        for i, g in enumerate(opt.param_groups):          # (a fill kernel per changed rate: the value travels in the launch, not through host memory
            lr = float(g['lr'])                           #  that a later step could overwrite before the copy has run)
            if lr != self._lr_last[i]:
                self.lr_dev[i].fill_(lr)
                self._lr_last[i] = lr
        self.graph.replay()
        opt.steps += 1
        self.replays += 1
        if self.trainer.lr_scheduler is not None:
            self.trainer.lr_scheduler.step()
        return self.loss, self.outputs, self.losses, self.accuracies
