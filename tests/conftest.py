import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fold_layernorm_at_every_size():
    """The product folds LayerNorm into the GEMMs only from HipViT.fold_min_rows token rows on (below, the unfolded path is the faster one); the parity tests run
    toy sizes and are there to pin the folded kernels too, so inside the test processes the fold applies at every size.  (Subprocesses -- bench.py, train_net.py --
    and ``tests/test_host_cpu.py::test_layernorm_fold_threshold`` see the product's default.)"""
    try:
        from avt_amd.models.vit import HipViT
    except Exception:          # (collection on a box without torch / the package: nothing to patch)
        yield
        return
    old = HipViT.fold_min_rows
    HipViT.fold_min_rows = 0
    try:
        yield
    finally:
        HipViT.fold_min_rows = old
