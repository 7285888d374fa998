import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: the gpu-marked tests are skipped, not failed."""
    gpu_items = [it for it in items if it.get_closest_marker('gpu')]
    if not gpu_items:
        return
    try:
        from rescan_line_sted_amd import _lib
        have = _lib.device_count() >= 1
    except Exception:
        have = False
    if not have:
        skip = pytest.mark.skip(reason='no GPU (librlsted.so reports no device)')
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'))
        return cache[name]
    return load


def max_rel(a, b):
    """Normwise max relative error: max|a-b| / max|b|."""
    cplx = np.iscomplexobj(a) or np.iscomplexobj(b)
    a = np.asarray(a, dtype=np.complex128 if cplx else np.float64)
    b = np.asarray(b, dtype=np.complex128 if cplx else np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def fuzz_seeds(default):
    """Seeds of a randomised test: `default` of them in the suite, RLSTED_FUZZ_SEEDS of them in a soak run
    (RLSTED_FUZZ_SEEDS=5000 python -m pytest tests -m gpu -k random; RLSTED_FUZZ_FIRST=5000 for the next 5000)."""
    import os
    first = int(os.environ.get('RLSTED_FUZZ_FIRST', 0))          # (a later soak run continues where the last one stopped)
    return range(first, first + int(os.environ.get('RLSTED_FUZZ_SEEDS', default)))
