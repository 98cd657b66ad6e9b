#!/usr/bin/env python
"""Entry point mirroring the reference's train_net.py (:22-35): compose conf/ + an expts/*.txt override list, seed, and
dispatch to ``func.<cfg.train.fn>.main(cfg)`` -- here ``avt_amd.func.train.main`` on synthetic clips.

    python train_net.py -c expts/01_ek100_avt.txt [extra.override=value ...] [--steps N] [--batch B]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_net.py -c expts/01_ek100_avt.txt
"""
import argparse
import importlib
import os
import random

import torch

from avt_amd.config import compose, read_overrides

ROOT = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-c', '--cfg', default=os.path.join(ROOT, 'expts', '01_ek100_avt.txt'))
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--ckpt', default=None, help='resume from / store to this checkpoint.pth (reference format)')
    ap.add_argument('overrides', nargs='*')
    args = ap.parse_args()
    cfg = compose(os.path.join(ROOT, 'conf'), read_overrides(args.cfg) + list(args.overrides))
    random.seed(cfg.seed)
    torch.manual_seed(cfg.seed)
    mod = importlib.import_module(f'avt_amd.func.{cfg.train.fn}')
    mod.main(cfg, steps=args.steps, batch_size=args.batch, ckpt=args.ckpt)


if __name__ == '__main__':
    main()
