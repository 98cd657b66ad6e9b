"""Whole training step with the early optimizer step (fused SGD on finished suffixes of the gradient buffer, on a side stream under the rest of backward)
against the one pass after backward.   usage: python tools/lab/early_step_ab.py {early|late} [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avt_amd.func.train import Trainer
Trainer.EARLY_STEP = sys.argv[1] == 'early'
import bench
bench.main(sys.argv[2:])
