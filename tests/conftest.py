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


# The two routes a ViT step can take (HipViT.fold_min_rows): 'fold' = LayerNorm folded into the qkv / fc1 GEMMs at every size (what the product runs from
# 60000 token rows on, i.e. the bench's batches), 'product' = the product's own threshold, which at the parity tests' sizes (and at the reference's own 3 clips
# per GPU, expts/01_ek100_avt.txt:5) means LayerNorm kernels + the unfolded backward.  Whole-model oracle tests are parametrised over both (``route`` fixture).
ROUTES = ('fold', 'product')


@pytest.fixture(autouse=True)
def _fold_layernorm_at_every_size(request):
    """The product folds LayerNorm into the GEMMs only from HipViT.fold_min_rows token rows on (below, the unfolded path is the faster one).  Tests that take the
    ``route`` fixture run once per route; every other test runs toy sizes and is there to pin the folded kernels too, so it folds at every size.  (Subprocesses --
    bench.py, train_net.py -- and ``tests/test_host_cpu.py::test_layernorm_fold_threshold`` see the product's default.)"""
    try:
        from avt_amd.models.vit import HipViT
    except Exception:          # (collection on a box without torch / the package: nothing to patch)
        yield
        return
    old = HipViT.fold_min_rows
    if not hasattr(HipViT, '_product_fold_min_rows'):
        HipViT._product_fold_min_rows = old
    route = request.node.callspec.params.get('route', 'fold') if hasattr(request.node, 'callspec') else 'fold'
    HipViT.fold_min_rows = HipViT._product_fold_min_rows if route == 'product' else 0
    try:
        yield
    finally:
        HipViT.fold_min_rows = old


@pytest.fixture(params=ROUTES)
def route(request):
    """'fold' | 'product' -- see ROUTES; the autouse fixture above has already set HipViT.fold_min_rows accordingly."""
    return request.param
