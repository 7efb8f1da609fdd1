"""The C-ABI shared library loads on a GPU-less host and exports exactly what include/fokl_hip.h declares."""
import ctypes
import os
import re

import pytest

from helpers import ROOT
from fokl_gpy_amd import _capi

HEADER = os.path.join(ROOT, 'include', 'fokl_hip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(fokl_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_documented_entry_points():
    syms = declared_symbols()
    for must in ('fokl_ctx_create', 'fokl_upload', 'fokl_build_terms', 'fokl_gram', 'fokl_bic_resid',
                 'fokl_gibbs_chain', 'fokl_predict', 'fokl_comm_allgather_f64', 'fokl_comm_allreduce_sum_f64'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert missing == []


def test_binding_table_matches_header():
    assert sorted(_capi.SIGNATURES.keys()) == declared_symbols()


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
