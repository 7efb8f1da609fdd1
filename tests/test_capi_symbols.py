"""The C-ABI shared library loads on a GPU-less host and exports exactly what include/fokl_hip.h declares."""
import ctypes
import os
import re

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
