"""The C-ABI shared library loads on a GPU-less host and exports exactly what include/fokl_hip.h (the boundary) and
include/fokl_hip_internal.h (this package's own plumbing) declare."""
import ctypes
import os
import re

import pytest

from helpers import ROOT
from fokl_gpy_amd import _capi

HEADER = os.path.join(ROOT, 'include', 'fokl_hip.h')                       # the C ABI: what a maintainer binds
INTERNAL = os.path.join(ROOT, 'include', 'fokl_hip_internal.h')            # this package's own plumbing (_capi.py only)


def declared_symbols(path=HEADER):
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(fokl_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_documented_entry_points():
    syms = declared_symbols()
    for must in ('fokl_ctx_create', 'fokl_upload', 'fokl_build_terms', 'fokl_gram', 'fokl_bic_resid',
                 'fokl_gibbs_chain', 'fokl_predict', 'fokl_comm_allgather_f64', 'fokl_comm_allreduce_sum_f64'):
        assert must in syms


def test_the_boundary_is_separate_from_the_plumbing():
    """include/fokl_hip.h is the stable boundary (SURVEY 8(b)'s set + the K1-K3 launch forms, predict, derivatives, clean,
    GP_Integrate, comm); the search / pool / stream / device-engine plumbing lives in fokl_hip_internal.h and nowhere else."""
    public, internal = declared_symbols(HEADER), declared_symbols(INTERNAL)
    assert not set(public) & set(internal)
    prefixes = ('fokl_search_', 'fokl_pool_', 'fokl_stream_', 'fokl_dchain_', 'fokl_dspectral_', 'fokl_outcome_',
                'fokl_spectrum_')
    assert not [s for s in public if s.startswith(prefixes)]
    assert len(public) <= 50
    # both headers are plain C (the boundary needs nothing of the plumbing)
    import shutil
    import subprocess
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc:
        for header in (HEADER, INTERNAL):
            res = subprocess.run([cc, '-fsyntax-only', '-x', 'c', '-std=c99', '-Wall', header], capture_output=True, text=True)
            assert res.returncode == 0, res.stderr


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [s for s in declared_symbols(HEADER) + declared_symbols(INTERNAL) if not hasattr(lib, s)]
    assert missing == []


def test_binding_table_matches_header():
    assert sorted(_capi.SIGNATURES.keys()) == sorted(declared_symbols(HEADER) + declared_symbols(INTERNAL))


def test_no_device_is_reported_not_hidden():
    """Without a GPU the context constructor must raise (no CPU fallback); with one it must succeed."""
    n = _capi.device_count()
    if n == 0:
        try:
            _capi.DeviceContext(0)
        except _capi.FoklNativeError as exc:
            assert exc.code == -1
        else:
            raise AssertionError("DeviceContext(0) succeeded without a device")
    else:
        ctx = _capi.DeviceContext(0)
        ctx.close()


def test_version():
    assert _capi.load().fokl_version() >= 100


def test_development_build_of_the_hip_translation_unit_still_parses():
    """`make DEV=1` (the retired Gram kernels kept for A/B runs, -DFOKL_DEV_KERNELS) shares fokl_hip.hip with the product
    build: a host-side syntax pass over it catches an edit that only compiles without the flag."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    src = os.path.join(os.path.dirname(os.path.abspath(_capi.__file__)), 'csrc', 'fokl_hip.hip')
    res = subprocess.run([hipcc, '-std=c++17', '-ffp-contract=off', '--offload-arch=gfx950', '-DFOKL_DEV_KERNELS',
                          '--cuda-host-only', '-fsyntax-only', '-Wno-unused-command-line-argument', src],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
