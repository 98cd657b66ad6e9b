"""Dropout seeds of the HIP kernels, and their indirect form for captured steps (include/avt_hip.h, ABI 9).

Every dropout mask of the path is a pure function of (seed, element index) (csrc/common.hpp: drop_keep), so backward re-derives it; a module draws ONE
base seed per forward -- a host counter mixed with the process's torch seed -- and derives its layers' seeds as ``base + small constants``.  In a step
that is replayed from a hipGraph no Python runs, so the base seeds live in device memory: during capture ``fresh`` hands out ``DevSeed`` objects --
"bit 63 | offset << 48 | device address" once turned into the kernel argument, ``+`` adds to the offset -- and remembers the generator of each slot;
``SeedCapture.draw`` then writes, before every replay, the seeds the eager code would have drawn in that step.  Replays and eager steps agree bit for bit.
"""
import torch


class DevSeed:
    __slots__ = ('ptr', 'off')

    def __init__(self, ptr, off=0):
        self.ptr, self.off = int(ptr), int(off)
        assert 0 < self.ptr < (1 << 48) and self.ptr % 8 == 0, 'device address of a uint64'

    def __add__(self, k):
        return DevSeed(self.ptr, self.off + int(k))

    __radd__ = __add__

    def __int__(self):
        assert 0 <= self.off < (1 << 15), 'offset of an indirect seed'
        return (1 << 63) | (self.off << 48) | self.ptr

    __index__ = __int__

    def __repr__(self):
        return f'DevSeed(0x{self.ptr:x} + {self.off})'


class SeedCapture:
    """The base-seed slots of one captured step."""
    def __init__(self, device, max_slots=64):
        self.dev = torch.zeros(max_slots, dtype=torch.int64, device=device)
        self.gens = []

    def slot(self, gen):
        i = len(self.gens)
        if i >= self.dev.numel():
            raise RuntimeError('SeedCapture: more base seeds in one step than slots')
        self.gens.append(gen)
        return DevSeed(self.dev.data_ptr() + 8 * i)

    def draw(self):
        """Before a replay (on the replay's stream): this step's base seeds, drawn in the order the eager code draws them."""
        for i, g in enumerate(self.gens):       # (a fill kernel per slot: the value travels in the launch's arguments -- an asynchronous copy from a
            self.dev[i].fill_(g())              #  host buffer would read it when the copy RUNS, after later steps have overwritten it)


_capture = None


def fresh(gen):
    """A module's base seed for one forward: ``gen()`` (advances the module's counter) -- or, inside a capture, a DevSeed whose slot ``gen`` will fill."""
    return gen() if _capture is None else _capture.slot(gen)


class capturing:
    def __init__(self, cap):
        self.cap = cap

    def __enter__(self):
        global _capture
        assert _capture is None, 'nested seed captures'
        _capture = self.cap
        return self.cap

    def __exit__(self, *exc):
        global _capture
        _capture = None
