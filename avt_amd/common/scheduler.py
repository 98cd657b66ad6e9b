"""Per-iteration LR schedules with the reference's semantics (common/scheduler.py:57-75 CosineLR, :88-135 Warmup),
re-stated in closed/recursive form without subclassing torch's scheduler classes, so they drive any optimizer that
exposes ``param_groups`` (torch.optim.SGD or avt_amd.optim.FusedSGD).

Quirk reproduced on purpose (pinned by tests/golden/g4_lr_schedules.npz): after W warm-up iterations
the cosine phase continues *recursively* from base*(W-1)/W, so the nominal peak LR is never reached.
"""
import math


class CosineLR:
    def __init__(self, optimizer, num_epochs, iters_per_epoch=None, world_size=None, eta_min=0.0, **kwargs):
        self.optimizer = optimizer
        self.T_max = num_epochs * iters_per_epoch
        self.eta_min = eta_min * (world_size or 1)
        self.base_lrs = [g['lr'] for g in optimizer.param_groups]
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        e, T = self.last_epoch, self.T_max
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            if e >= T:
                g['lr'] = 0.0
            elif (e - 1 - T) % (2 * T) == 0:
                g['lr'] = g['lr'] + (base - self.eta_min) * (1 - math.cos(math.pi / T)) / 2
            else:
                g['lr'] = ((1 + math.cos(math.pi * e / T)) / (1 + math.cos(math.pi * (e - 1) / T))
                           * (g['lr'] - self.eta_min) + self.eta_min)

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'base_lrs': self.base_lrs, 'T_max': self.T_max, 'eta_min': self.eta_min}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class Warmup:
    """Linear warm-up from ``init_lr_ratio`` x base over ``num_epochs x iters_per_epoch`` iterations, then hands over
    to the wrapped scheduler."""
    def __init__(self, optimizer, scheduler, init_lr_ratio=0.0, num_epochs=5, last_epoch=-1, iters_per_epoch=None,
                 world_size=None):
        del world_size, last_epoch
        self.optimizer, self.base_scheduler = optimizer, scheduler
        self.warmup_iters = max(num_epochs * iters_per_epoch, 1)
        self.init_lr_ratio = init_lr_ratio if self.warmup_iters > 1 else 1.0
        self.base_lrs = [g.get('initial_lr', g['lr']) for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def _apply(self):
        f = self.init_lr_ratio + (1 - self.init_lr_ratio) * (float(self.last_epoch) / self.warmup_iters)
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g['lr'] = base * f

    def step(self):
        if self.last_epoch < self.warmup_iters - 1:
            self.last_epoch += 1
            self._apply()
        else:
            self.base_scheduler.step()

    def state_dict(self):
        return {'base_sched_dict': self.base_scheduler.state_dict(),
                'other_stuff': {'last_epoch': self.last_epoch, 'warmup_iters': self.warmup_iters,
                                'init_lr_ratio': self.init_lr_ratio, 'base_lrs': self.base_lrs}}

    def load_state_dict(self, sd):
        self.base_scheduler.load_state_dict(sd['base_sched_dict'])
        self.__dict__.update(sd['other_stuff'])
