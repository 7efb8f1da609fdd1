import os
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault('MPLBACKEND', 'Agg')
os.environ.setdefault('OPENBLAS_NUM_THREADS', '4')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def device_ctx():
    """One device context for the whole GPU session; fails loudly (no fallback) if the library or GPU is missing."""
    from fokl_gpy_amd import _capi
    ctx = _capi.DeviceContext(int(os.environ.get('FOKL_DEVICE', '0')))
    yield ctx
    ctx.close()
