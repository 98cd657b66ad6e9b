"""Whole training step with the all-token ViT blocks' weight gradients on a second stream (models/vit.py: HipViT.wgrad_side_stream) against one stream.
usage: python tools/lab/wgrad_side_ab.py {side|auto|one} [bench.py arguments]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avt_amd.models import vit
vit.WGRAD_SIDE_STREAM = {'side': 'always', 'auto': True}.get(sys.argv[1], False)
import bench
bench.main(sys.argv[2:])
