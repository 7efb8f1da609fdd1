"""
ctypes binding of libfokl_hip.so (C ABI: include/fokl_hip.h).

This is the only place the Python host code touches native code.  There is deliberately NO CPU fallback:
if the library is missing, or no gfx950 device is present when a device context is requested, the error
is raised to the caller.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FOKL_HIP_LIBRARY', os.path.join(_HERE, 'libfokl_hip.so'))   # override: A/B builds

UNIQUE_ID_BYTES = 128
K_BASIS, K_GRAM, K_RESID, K_PREDICT, K_RESID_MF, K_GRAM_MFMA, K_GRAM_REDUCE = 0, 1, 2, 3, 4, 5, 6
RESID_TERMS_MAX_FACTORS = 32
RESID_TERMS_MAX_ORDER = 8
RESID_TERMS_LAYOUTS = ((8, 1), (16, 1), (8, 2), (4, 4), (2, 8), (8, 4), (16, 2), (4, 8))    # inputs x orders per input (csrc/fokl_hip.hip)
SLOT_ONES, SLOT_Y, SLOT_FIRST_FREE = 0, 1, 2

c_int, c_i32, c_i64, c_dbl, c_vp = ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/fokl_hip.h one to one (tests/test_capi_symbols.py checks it)
SIGNATURES = {
    'fokl_version': (c_int, []),
    'fokl_device_count': (c_int, [c_vp]),
    'fokl_ctx_create': (c_int, [c_int, c_vp]),
    'fokl_ctx_destroy': (None, [c_vp]),
    'fokl_last_error': (ctypes.c_char_p, [c_vp]),
    'fokl_sync': (c_int, [c_vp]),
    'fokl_upload': (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_int, c_int]),
    'fokl_stage_inputs': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    'fokl_upload_staged': (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    'fokl_download_inputs': (c_int, [c_vp, c_vp]),
    'fokl_column_min_max': (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_int]),
    'fokl_normalize_columns': (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_int]),
    'fokl_reserve_slots': (c_int, [c_vp, c_int]),
    'fokl_slot_capacity': (c_int, [c_vp]),
    'fokl_rows': (c_i64, [c_vp]),
    'fokl_build_terms': (c_int, [c_vp, c_vp, c_int, c_vp]),
    'fokl_build_terms_deriv': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_dbl]),
    'fokl_gram': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int]),
    'fokl_gram_launch': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int]),
    'fokl_gram_fetch': (c_int, [c_vp, c_vp, c_i64]),
    'fokl_gram_ready': (c_int, [c_vp]),
    'fokl_gram_plan': (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    'fokl_bic_resid': (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_int]),
    'fokl_bic_resid_launch': (c_int, [c_vp, c_vp, c_int, c_vp]),
    'fokl_bic_resid_fetch': (c_int, [c_vp, c_vp, c_int]),
    'fokl_bic_resid_terms_launch': (c_int, [c_vp, c_vp, c_int, c_vp]),
    'fokl_predict': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    'fokl_read_slot': (c_int, [c_vp, c_int, c_i64, c_i64, c_vp]),
    'fokl_write_slot': (c_int, [c_vp, c_int, c_i64, c_i64, c_vp]),
    'fokl_timing_enable': (c_int, [c_vp, c_int]),
    'fokl_timing_reset': (c_int, [c_vp]),
    'fokl_timing_get': (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_probe': (c_int, [c_vp, c_int, c_vp]),
    'fokl_gibbs_chain': (c_int, [c_vp, c_vp, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_int,
                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_noise_tape': (c_int, [c_int, c_int, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                c_vp]),
    'fokl_gibbs_chain_from_tape': (c_int, [c_vp, c_vp, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_int, c_vp, c_vp, c_vp,
                                           c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_pool_create': (c_int, [c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    'fokl_pool_stream': (c_vp, [c_vp]),
    'fokl_pool_use_dsyevd': (c_int, [c_vp, c_vp, c_int]),
    'fokl_pool_use_dgemm': (c_int, [c_vp, c_vp]),
    'fokl_device_dgemm': (None, [c_vp] * 13),               # (called by the pool's threads through its address, not from here)
    'fokl_device_dgemm_configure': (c_int, [c_int, c_vp, c_int, c_vp]),
    'fokl_device_dgemm_stats': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_pool_spectral_affinity': (c_int, [c_vp, c_vp, c_int]),
    'fokl_pool_release_hold': (c_int, [c_vp, ctypes.c_uint64]),
    'fokl_pool_destroy': (None, [c_vp]),
    'fokl_pool_submit_noise': (c_int, [c_vp, c_int, c_int, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                                       c_vp, c_int, c_int, c_vp, c_vp]),
    'fokl_tape_ready': (c_int, [c_vp, c_int, c_vp, c_int]),
    'fokl_pool_resolve': (c_int, [c_vp, c_int]),
    'fokl_pool_submit_chain': (c_int, [c_vp, c_vp, c_vp, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_int, c_vp, c_vp,
                                       c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_pool_submit_spectral': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                          c_vp]),
    'fokl_pool_submit_spectral_update': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp,
                                                 c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_pool_poll': (c_int, [c_vp]),
    'fokl_pool_wait': (c_int, [c_vp]),
    'fokl_pool_busy_seconds': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_pool_noise_waits': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_thread_cpu_seconds': (c_int, [c_vp, c_int]),
    'fokl_pool_stream_stats': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_stream_create': (c_int, [c_vp, c_i32, c_i32, c_dbl, c_int, c_vp, c_int, c_vp]),
    'fokl_stream_prestates_published': (c_i64, [c_vp]),
    'fokl_stream_given_gauss': (c_dbl, [c_vp]),
    'fokl_stream_destroy': (None, [c_vp]),
    'fokl_stream_walk': (c_int, [c_vp, c_int, c_int, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp]),
    'fokl_stream_tell': (c_int, [c_vp, c_vp]),
    'fokl_stream_seek': (c_int, [c_vp, c_vp]),
    'fokl_stream_hold': (c_int, [c_vp, c_vp]),
    'fokl_stream_release': (c_int, [c_vp, ctypes.c_uint64]),
    'fokl_stream_advance_floor': (c_int, [c_vp]),
    'fokl_stream_state': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_stream_expand': (c_int, [c_vp, c_int, c_dbl, c_dbl, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_stream_stats': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_stream_set_helpers': (c_int, [c_vp, c_int, c_vp]),
    'fokl_stream_place_bulk': (c_int, [c_vp, c_vp, c_int]),
    'fokl_stream_fast_ln_error': (c_dbl, [c_i64]),
    'fokl_search_create': (c_int, [c_vp, c_vp, c_vp, c_vp]),
    'fokl_search_bind_spectral': (c_int, [c_vp, c_vp, c_int, c_dbl, c_int]),
    'fokl_search_hold_spectral': (c_int, [c_vp, c_int]),
    'fokl_search_set_update': (c_int, [c_vp, c_int, c_int, c_int]),
    'fokl_search_set_decide': (c_int, [c_vp, c_int, c_dbl]),
    'fokl_search_set_deterministic': (c_int, [c_vp, c_int]),
    'fokl_search_destroy': (None, [c_vp]),
    'fokl_search_error': (ctypes.c_char_p, [c_vp]),
    'fokl_search_mispredicted': (c_int, [c_vp]),
    'fokl_search_set_substage': (c_int, [c_vp, c_vp, c_int]),
    'fokl_search_speculate': (c_int, [c_vp, c_vp, c_vp, c_int]),
    'fokl_search_drop_speculation': (c_int, [c_vp]),
    'fokl_search_spectral': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp]),
    'fokl_search_spectral_from': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    'fokl_spectrum_done': (c_int, [c_vp]),
    'fokl_spectrum_wait': (c_int, [c_vp, c_vp, c_vp, c_vp]),
    'fokl_spectrum_release': (None, [c_vp, c_vp]),
    'fokl_spectrum_retain': (c_int, [c_vp, c_vp]),
    'fokl_search_model_begin': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    'fokl_search_model_commit': (c_int, [c_vp, c_vp, c_vp, c_dbl, c_int, c_vp]),
    'fokl_run_create': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_run_set_update': (c_int, [c_vp, c_int, c_int, c_int]),
    'fokl_run_search': (c_int, [c_vp, c_vp]),
    'fokl_run_result': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_run_arrays': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_run_error': (ctypes.c_char_p, [c_vp]),
    'fokl_run_destroy': (None, [c_vp]),
    'fokl_search_score': (c_int, [c_vp, c_vp, c_dbl, c_dbl, c_int, c_int, c_vp]),
    'fokl_outcome_info': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_outcome_spectrum': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_outcome_chain_ready': (c_int, [c_vp]),
    'fokl_outcome_draws': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_outcome_intercept_scale': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_outcome_new_term_stats': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    'fokl_outcome_release': (None, [c_vp, c_vp]),
    'fokl_outcome_drop': (None, [c_vp, c_vp]),
    'fokl_search_verify': (c_int, [c_vp, c_int]),
    'fokl_search_register_forecast': (c_int, [c_vp, c_vp, c_int, c_vp, c_dbl]),
    'fokl_search_clear_forecasts': (None, [c_vp]),
    'fokl_search_likely_first_tests': (c_int, [c_vp, c_vp, c_int, c_dbl, c_vp, c_vp, c_vp]),
    'fokl_search_stats': (c_int, [c_vp, c_vp, c_int]),
    'fokl_search_trace': (c_i64, [c_vp, c_vp, c_i64]),
    'fokl_search_kill_tests': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_finish_tape_blocks': (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'fokl_gibbs_chain_from_finished_tape': (c_int, [c_vp, c_vp, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_int, c_vp,
                                                    c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    'fokl_rng_normals': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'fokl_rng_gammas': (c_int, [c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_i64, c_vp]),
    'fokl_gp_integrate': (c_int, [c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                                  c_dbl, c_vp, c_vp]),
    'fokl_dchain_create': (c_int, [c_int, c_int, c_vp]),
    'fokl_dchain_destroy': (None, [c_vp]),
    'fokl_dchain_submit': (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_vp, c_vp, c_vp,
                                   c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    'fokl_dchain_prestate_ring': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_dchain_bind_stream': (c_int, [c_vp, c_vp]),
    'fokl_dchain_submit_rows': (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_vp,
                                        c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    'fokl_dchain_stream_stats': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_dchain_try_release': (c_int, [c_vp, c_i64]),
    'fokl_dchain_poll': (c_int, [c_vp, c_i64]),
    'fokl_dchain_flush': (c_int, [c_vp]),
    'fokl_dchain_wait': (c_int, [c_vp, c_i64, c_vp]),
    'fokl_dchain_fetch_w': (c_int, [c_vp, c_i64, c_vp]),
    'fokl_dchain_release': (c_int, [c_vp, c_i64]),
    'fokl_dchain_stats': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_dspectral_create': (c_int, [c_int, c_vp]),
    'fokl_dspectral_destroy': (None, [c_vp]),
    'fokl_dspectral_max_columns': (c_int, []),
    'fokl_dspectral_set_signs': (c_int, [c_vp, c_int]),
    'fokl_dspectral_submit': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    'fokl_dspectral_flush': (c_int, [c_vp]),
    'fokl_dspectral_poll': (c_int, [c_vp, c_i64]),
    'fokl_dspectral_wait': (c_int, [c_vp, c_i64]),
    'fokl_dspectral_release': (c_int, [c_vp, c_i64]),
    'fokl_dspectral_stats': (c_int, [c_vp, c_vp, c_vp]),
    'fokl_host_alloc': (c_int, [ctypes.c_size_t, c_vp]),
    'fokl_host_free': (c_int, [c_vp]),
    'fokl_comm_unique_id': (c_int, [c_vp]),
    'fokl_comm_init': (c_int, [c_vp, c_vp, c_int, c_int]),
    'fokl_comm_destroy': (c_int, [c_vp]),
    'fokl_comm_init_detached': (c_int, [c_int, c_vp, c_int, c_int, ctypes.POINTER(c_vp)]),
    'fokl_comm_adopt': (c_int, [c_vp, c_vp, c_int, c_int]),
    'fokl_comm_release_detached': (c_int, [c_vp]),
    'fokl_comm_allgather_f64': (c_int, [c_vp, c_vp, c_int, c_vp]),
    'fokl_comm_allreduce_sum_f64': (c_int, [c_vp, c_vp, c_int]),
}


class FoklNativeError(RuntimeError):
    """A libfokl_hip call returned a non-zero status."""

    def __init__(self, code, message):
        super().__init__(f"libfokl_hip error {code}: {message}")
        self.code = code


_lib = None


def load():
    """Load libfokl_hip.so (built in-tree by ``__graft_entry__.build()`` / ``make -C fokl_gpy_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FoklNativeError(-1, f"{LIB_PATH} not found -- build it with `python -c 'import __graft_entry__ as g; "
                                  f"g.build()'` or `make -C fokl_gpy_amd/csrc` (there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    host_only = os.environ.get('FOKL_HOST_ONLY_LIBRARY', '0') == '1'      # sanitizer builds of the host side (csrc/Makefile)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None) if host_only else getattr(lib, name)
        if fn is None:
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(arr):
    """Address of a numpy buffer as a plain integer (accepted by every c_void_p parameter; far cheaper than
    ``arr.ctypes.data_as``, which showed up at 3 us a call on the search's critical thread)."""
    return arr.__array_interface__['data'][0] if arr is not None else None


def _check(rc, ctx=None):
    if rc != 0:
        msg = load().fokl_last_error(ctx)
        raise FoklNativeError(rc, msg.decode() if msg else "unknown error")


def device_count():
    n = c_int(0)
    rc = load().fokl_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


# ---------------------------------------------------------------------------------------------------------
# numpy legacy RNG state <-> the C sampler
# ---------------------------------------------------------------------------------------------------------

def gram_plan(row_slots, col_slots, kind=0):
    """fokl_gram_plan: the tile lists fokl_gram's MFMA path would use for this block (host arithmetic, no device);
    kind 0 = gram_tiles_kernel (16x16x4 MFMA; the default), kind 1 = gram_tiles4s_kernel (4x4x4 MFMA; FOKL_GRAM_MFMA4=2).
    -> dict(nci, i_tiles, j_tiles, nt, ct, rows_per_chunk, ks, depth, waves, icols, perm, staged [G, 16],
            tiles [G, 4, 10, 4] = (local row-side tile, local column-side tile, out i-tile, out j-tile; -1: padding),
            half [G, 4, 10] = entries of a ragged last row tile that gram_tiles_dma_kernel forms with 4x4x4 MFMAs)"""
    rs = np.ascontiguousarray(row_slots, dtype=np.int32)
    cs = np.ascontiguousarray(col_slots, dtype=np.int32)
    info = np.zeros(10, dtype=np.int32)
    lib = load()
    _check(lib.fokl_gram_plan(_ptr(rs), rs.shape[0], _ptr(cs), cs.shape[0], int(kind), _ptr(info), None, None, None,
                              None, 0))
    groups = int(info[3])
    icols = np.empty(int(info[0]), dtype=np.int32)
    perm = np.empty(cs.shape[0], dtype=np.int32)
    staged = np.empty((groups, 16), dtype=np.int32)
    tiles = np.empty((groups, 4, 10, 4), dtype=np.int32)
    _check(lib.fokl_gram_plan(_ptr(rs), rs.shape[0], _ptr(cs), cs.shape[0], int(kind), _ptr(info), _ptr(icols),
                              _ptr(perm), _ptr(staged), _ptr(tiles), groups))
    half = (tiles[..., 0] >= 256) & (tiles[..., 2] >= 0)       # entries the LDS-DMA kernel computes as half tiles
    tiles[..., 0] &= 255
    return dict(nci=int(info[0]), i_tiles=int(info[1]), j_tiles=int(info[2]), nt=int(info[4]), ct=int(info[5]),
                rows_per_chunk=32 << int(info[6]), ks=int(info[7]), depth=int(info[8]), waves=int(info[9]), icols=icols,
                perm=perm, staged=staged, tiles=tiles, half=half)


def device_dgemm_stats():
    """(calls of fokl_device_dgemm so far in this process, of which ran on the device, device attempts that fell back)"""
    v = [ctypes.c_int64(0) for _ in range(3)]
    if load().fokl_device_dgemm_stats(*[ctypes.byref(x) for x in v]) != 0:
        return 0, 0, 0
    return tuple(x.value for x in v)


class TraceRecords:
    """The evaluations of a search in order, one dict(cols, built, ev, kill, b0) each -- formed when somebody looks (a fit has
    hundreds of them and the search's last half millisecond is no place to build them)."""

    def __init__(self, rec):
        self._rec = rec

    def __len__(self):
        return self._rec.shape[0]

    @staticmethod
    def _record(r):
        return dict(cols=int(r[0]), built=int(r[1]), ev=float(r[2]), kill=bool(r[3]), b0=float(r[4]))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._record(r) for r in self._rec[i]]
        return self._record(self._rec[i])

    def __iter__(self):
        return (self._record(r) for r in self._rec)

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(list(self))


class LegacyStream:
    """Mutable copy of numpy's global legacy RNG state in the layout the C sampler updates in place."""

    def __init__(self, state=None):
        st = np.random.get_state() if state is None else state
        if st[0] != 'MT19937':
            raise ValueError("numpy's global RNG is not MT19937")
        self.key = np.array(st[1], dtype=np.uint32, copy=True)
        self.pos = ctypes.c_int32(int(st[2]))
        self.has_gauss = ctypes.c_int32(int(st[3]))
        self.cache = ctypes.c_double(float(st[4]))

    def as_numpy_state(self):
        return ('MT19937', self.key.copy(), int(self.pos.value), int(self.has_gauss.value), float(self.cache.value))

    def publish(self):
        """Write the advanced state back to numpy's global generator."""
        np.random.set_state(self.as_numpy_state())

    def args(self):
        return (_ptr(self.key), ctypes.byref(self.pos), ctypes.byref(self.has_gauss), ctypes.byref(self.cache))

    def normals(self, n):
        out = np.empty(int(n), dtype=np.float64)
        _check(load().fokl_rng_normals(*self.args(), c_i64(int(n)), _ptr(out)))
        return out

    def gammas(self, shape, scale, n):
        out = np.empty(int(n), dtype=np.float64)
        _check(load().fokl_rng_gammas(*self.args(), c_dbl(shape), c_dbl(scale), c_i64(int(n)), _ptr(out)))
        return out


THREAD_KINDS = ('walker', 'chain', 'finish', 'spectral', 'bulk', 'device_chain_dispatcher')


def thread_cpu_seconds():
    """CPU-seconds used so far by the library's own threads, by kind (fokl_thread_cpu_seconds)."""
    v = np.zeros(8)
    n = load().fokl_thread_cpu_seconds(_ptr(v), v.shape[0])
    if n < 0:
        _check(n)
    return dict(zip(THREAD_KINDS, v[:n].tolist()))


def _host_threads():
    try:
        return max(1, min(8, len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        return 1


def column_min_max(x):
    """(minima, maxima) of the columns of a C-contiguous float64 [n, m] array on several host threads (fokl_column_min_max:
    np.min / np.max per column, exact)."""
    n, m = x.shape
    lows, highs = np.empty(m), np.empty(m)
    _check(load().fokl_column_min_max(_ptr(x), n, m, _ptr(lows), _ptr(highs), _host_threads()))
    return lows, highs


def normalize_columns(x, lows, spans):
    """x <- (x - lows) / spans in place, element by element as numpy does it (fokl_normalize_columns)."""
    lows = np.ascontiguousarray(lows, dtype=np.float64)
    spans = np.ascontiguousarray(spans, dtype=np.float64)
    _check(load().fokl_normalize_columns(_ptr(x), x.shape[0], x.shape[1], _ptr(lows), _ptr(spans), _host_threads()))
    return x


def gibbs_chain(lamb, qty, astar, atau_star, b, btau, dtd, sigsqd0, tausqd0, draws, stream, want_sig_tau=False):
    """G3 in the eigenbasis (include/fokl_hip.h: fokl_gibbs_chain).  Returns w [draws, p1] (betas = w @ Q.T)."""
    lamb = np.ascontiguousarray(lamb, dtype=np.float64)
    qty = np.ascontiguousarray(qty, dtype=np.float64)
    p1 = lamb.shape[0]
    w = np.empty((int(draws), p1), dtype=np.float64)
    sigs = np.empty(int(draws)) if want_sig_tau else None
    taus = np.empty(int(draws)) if want_sig_tau else None
    _check(load().fokl_gibbs_chain(_ptr(lamb), _ptr(qty), p1, float(astar), float(atau_star), float(b), float(btau),
                                   float(dtd), float(sigsqd0), float(tausqd0), int(draws), *stream.args(),
                                   _ptr(w), _ptr(sigs), _ptr(taus)))
    if want_sig_tau:
        return w, sigs, taus
    return w


class NoiseTape:
    """The data-independent random numbers of one candidate's chain (include/fokl_hip.h: fokl_noise_tape): per
    iteration p1 normals -- accepted polar pairs left unfinished, see the header -- and two standard gammas.
    ``progress[0]`` counts the iterations recorded so far (-1 = the producer failed); ``block_done`` are the flags of
    fokl_finish_tape_blocks (blocks of BLOCK iterations).
    One allocation, carved up by address: a tape is made for every model evaluation on the search's critical thread."""
    __slots__ = ('p1', 'draws', 'finishing_requested', 'materialised_by_pool', '_buf', '_addr', '_off', '_ints')
    BLOCK = 16                                                # = FOKL_TAPE_BLOCK

    @classmethod
    def _layout(cls, p1, d):
        half = p1 // 2 + 1
        nblocks = -(-d // cls.BLOCK)
        pad = lambda count: -(-count // 8) * 8                # regions start on 64-byte lines (in doubles)
        # int32 area: progress (a line of its own: polled by other threads) | block_done [nblocks] | lead [d]
        ints = (16, 16 + -(-nblocks // 16) * 16)
        off = [0]
        for count in (d * p1 + 16, d * half + 8, d, d):     # slack: the recorder stores whole vectors (see the header)
            off.append(off[-1] + pad(count))
        # behind the int32 area: the tape as the pool's walker leaves it (fokl_tape_row [d]: 4 x uint64 each)
        rows = off[4] + pad((ints[1] + d + 1) // 2) + 8
        off.append(rows)
        return tuple(off), ints, rows + 4 * d + 8

    @classmethod
    def doubles_needed(cls, p1, draws):
        return cls._layout(int(p1), int(draws))[2]

    def __init__(self, p1, draws, raw=None):
        """raw: a float64 buffer of at least doubles_needed(p1, draws) elements to build the tape in (recycled memory
        is already mapped and probably cached), else a fresh allocation."""
        self.p1, self.draws = p1, d = int(p1), int(draws)
        off, ints, total = self._layout(p1, d)
        self._off, self._ints = off, ints
        if raw is None:
            raw = np.empty(total, dtype=np.float64)
        elif raw.shape[0] < total:
            raise ValueError("buffer too small for this tape")
        shift = ((-raw.__array_interface__['data'][0]) % 64) // 8
        self._buf = raw[shift:]
        self._addr = self._buf.__array_interface__['data'][0]
        self._buf[off[4]:off[4] + ints[1] // 2] = 0.0         # progress and block flags start at zero
        self.finishing_requested = False                      # HostPool.submit_noise(..., finish=True) was given this tape
        self.materialised_by_pool = False                     # block_done flags: set by the pool's finish threads

    def _int_area(self):
        return self._buf[self._off[4]:].view(np.int32)

    normals = property(lambda self: self._buf[:self.draws * self.p1].reshape(self.draws, self.p1))
    pair_r2 = property(lambda self: self._buf[self._off[1]:self._off[1] + self.draws * (self.p1 // 2 + 1)]
                       .reshape(self.draws, self.p1 // 2 + 1))
    gam_sig = property(lambda self: self._buf[self._off[2]:self._off[2] + self.draws])
    gam_tau = property(lambda self: self._buf[self._off[3]:self._off[3] + self.draws])
    progress = property(lambda self: self._int_area()[:1])
    block_done = property(lambda self: self._int_area()[self._ints[0]:self._ints[0] + -(-self.draws // self.BLOCK)])
    lead = property(lambda self: self._int_area()[self._ints[1]:self._ints[1] + self.draws])

    def pointers(self):
        """normals, pair_r2, lead, gam_sig, gam_tau -- the argument order of the C entry points."""
        a, off = self._addr, self._off
        return (a, a + 8 * off[1], a + 8 * off[4] + 4 * self._ints[1], a + 8 * off[2], a + 8 * off[3])

    def progress_pointer(self):
        return self._addr + 8 * self._off[4]

    def rows_pointer(self):
        return self._addr + 8 * self._off[5]

    rows = property(lambda self: self._buf[self._off[5]:self._off[5] + 4 * self.draws].view(np.uint64)
                    .reshape(self.draws, 4))

    def block_done_pointer(self):
        return self._addr + 8 * self._off[4] + 4 * self._ints[0]


def record_noise_tape(tape, astar, atau_star, stream):
    """Fill ``tape`` from the stream on the calling thread."""
    try:
        _check(load().fokl_noise_tape(tape.p1, tape.draws, float(astar), float(atau_star), *stream.args(),
                                      *tape.pointers(), tape.progress_pointer()))
    except BaseException:
        tape.progress[0] = -1
        raise
    return tape


def noise_tape(p1, draws, astar, atau_star, stream):
    return record_noise_tape(NoiseTape(p1, draws), astar, atau_star, stream)


class _StreamCursor(ctypes.Structure):
    _fields_ = [('position', ctypes.c_uint64), ('gauss_source', ctypes.c_uint64), ('has_gauss', ctypes.c_int32)]


class StreamEngine:
    """include/fokl_hip.h: fokl_stream_* -- the random stream as bulk threads + one serial walk, outside a HostPool
    (tests, tools; a fit's engine lives inside its pool).  ``walk`` fills a NoiseTape's rows, ``expand`` turns them into
    the numbers fokl_noise_tape records; the part of the stream a tape covers is held from its walk until ``release``."""

    def __init__(self, stream, bulk_threads=2, prestates=None):
        """prestates: (address, entries) of DeviceChainEngine.prestate_ring() when a device will expand this stream's
        tapes from rows."""
        self._lib = load()
        self._h = None
        self.stream = stream
        h = c_vp(0)
        _check(self._lib.fokl_stream_create(_ptr(stream.key), stream.pos, stream.has_gauss, stream.cache,
                                            int(bulk_threads), prestates[0] if prestates else None,
                                            prestates[1] if prestates else 0, ctypes.byref(h)))
        self._h = h

    def close(self, write_back=True):
        """Stops the bulk threads; the walker's position goes back to ``stream`` as numpy's state tuple."""
        if self._h:
            if write_back:
                _check(self._lib.fokl_stream_state(self._h, *self.stream.args()))
            self._lib.fokl_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close(write_back=False)

    def walk(self, tape, astar, atau_star):
        """One model evaluation's draws: -> the hold that keeps the tape's part of the stream readable."""
        hold = ctypes.c_uint64(0)
        _check(self._lib.fokl_stream_hold(self._h, ctypes.byref(hold)))
        a = tape._addr
        rc = self._lib.fokl_stream_walk(self._h, tape.p1, tape.draws, float(astar), float(atau_star), tape.rows_pointer(),
                                        a + 8 * tape._off[2], a + 8 * tape._off[3], tape.progress_pointer())
        if rc:
            self._lib.fokl_stream_release(self._h, hold)
        _check(rc)
        _check(self._lib.fokl_stream_advance_floor(self._h))
        return hold.value

    def expand(self, tape, astar, atau_star, first=0, last=None):
        ptr = tape.pointers()
        _check(self._lib.fokl_stream_expand(self._h, tape.p1, float(astar), float(atau_star), tape.rows_pointer(),
                                            int(first), tape.draws if last is None else int(last), ptr[0], ptr[1], ptr[2],
                                            ptr[3], ptr[4]))
        return tape

    def release(self, hold):
        _check(self._lib.fokl_stream_release(self._h, ctypes.c_uint64(hold)))

    def tell(self):
        cur = _StreamCursor()
        _check(self._lib.fokl_stream_tell(self._h, ctypes.byref(cur)))
        return cur

    def seek(self, cursor):
        _check(self._lib.fokl_stream_seek(self._h, ctypes.byref(cursor)))

    def numpy_state(self):
        st = LegacyStream(self.stream.as_numpy_state())
        _check(self._lib.fokl_stream_state(self._h, *st.args()))
        return st.as_numpy_state()

    def stats(self):
        b, w, seg, ga, ge = c_dbl(0), c_dbl(0), c_i64(0), c_i64(0), c_i64(0)
        _check(self._lib.fokl_stream_stats(self._h, *[ctypes.byref(x) for x in (b, w, seg, ga, ge)]))
        return dict(bulk_s=b.value, walker_wait_s=w.value, segments=seg.value, gamma_attempts=ga.value,
                    gamma_attempts_exact=ge.value)


def tape_ready(tape):
    """fokl_tape_ready: 1 once the tape is recorded and -- a tape the pool's finish threads work on -- every block is there,
    0 before, -1 if it never will be."""
    return load().fokl_tape_ready(tape.progress_pointer(), tape.draws,
                                  tape.block_done_pointer() if tape.materialised_by_pool else None, tape.BLOCK)


def gibbs_chain_from_tape(lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape, want_sig_tau=False, follow=False):
    """Replay the chain arithmetic on a tape -- with ``follow=True`` on one that is still being recorded.
    Returns (w, bstar_negative[, sigs, taus])."""
    lamb = np.ascontiguousarray(lamb, dtype=np.float64)
    qty = np.ascontiguousarray(qty, dtype=np.float64)
    p1 = lamb.shape[0]
    if p1 != tape.p1:
        raise ValueError("tape was recorded for a different model size")
    w = np.empty((tape.draws, p1), dtype=np.float64)
    sigs = np.empty(tape.draws) if want_sig_tau else None
    taus = np.empty(tape.draws) if want_sig_tau else None
    flag = ctypes.c_int32(0)
    _check(load().fokl_gibbs_chain_from_tape(_ptr(lamb), _ptr(qty), p1, float(b), float(btau), float(dtd),
                                             float(sigsqd0), float(tausqd0), tape.draws, *tape.pointers(), _ptr(w),
                                             _ptr(sigs), _ptr(taus), ctypes.byref(flag),
                                             tape.progress_pointer() if follow else None))
    if want_sig_tau:
        return w, bool(flag.value), sigs, taus
    return w, bool(flag.value)


def finish_tape_blocks(tape, part=0, parts=1, follow=False):
    """Complete the normals of the tape's blocks part, part + parts, ... in place (fokl_finish_tape_blocks)."""
    ptr = tape.pointers()
    _check(load().fokl_finish_tape_blocks(tape.p1, tape.draws, ptr[0], ptr[1], ptr[2],
                                          tape.progress_pointer() if follow else None, int(part), int(parts),
                                          tape.BLOCK, tape.block_done_pointer()))


def gibbs_chain_from_finished_tape(lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape, follow=False):
    """The recursion on a tape completed by finish_tape_blocks.  Returns (w, bstar_negative)."""
    lamb = np.ascontiguousarray(lamb, dtype=np.float64)
    qty = np.ascontiguousarray(qty, dtype=np.float64)
    p1 = lamb.shape[0]
    if p1 != tape.p1:
        raise ValueError("tape was recorded for a different model size")
    w = np.empty((tape.draws, p1), dtype=np.float64)
    flag = ctypes.c_int32(0)
    ptr = tape.pointers()
    _check(load().fokl_gibbs_chain_from_finished_tape(_ptr(lamb), _ptr(qty), p1, float(b), float(btau), float(dtd),
                                                      float(sigsqd0), float(tausqd0), tape.draws, ptr[0], ptr[3],
                                                      ptr[4], tape.block_done_pointer() if follow else None,
                                                      tape.BLOCK, _ptr(w), None, None, ctypes.byref(flag)))
    return w, bool(flag.value)


# ---------------------------------------------------------------------------------------------------------
# host threads of one fit
# ---------------------------------------------------------------------------------------------------------

def _scipy_dsyevd_address():
    """Address of scipy's dsyevd (Fortran ABI, 32-bit integers), or None."""
    try:
        from scipy.linalg import cython_lapack
        capsule = cython_lapack.__pyx_capi__['dsyevd']
        api = ctypes.pythonapi
        api.PyCapsule_GetName.restype = ctypes.c_char_p
        api.PyCapsule_GetName.argtypes = [ctypes.py_object]
        api.PyCapsule_GetPointer.restype = ctypes.c_void_p
        api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
        name = api.PyCapsule_GetName(capsule)
        if name is None or name.count(b'int *') != 6 or b'long' in name:
            return None
        return api.PyCapsule_GetPointer(capsule, name)
    except (ImportError, KeyError):
        return None


def _scipy_dgemm_address():
    """Address of scipy's dgemm (Fortran ABI, 32-bit integers), or None."""
    try:
        from scipy.linalg import cython_blas
        capsule = cython_blas.__pyx_capi__['dgemm']
        api = ctypes.pythonapi
        api.PyCapsule_GetName.restype = ctypes.c_char_p
        api.PyCapsule_GetName.argtypes = [ctypes.py_object]
        api.PyCapsule_GetPointer.restype = ctypes.c_void_p
        api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
        name = api.PyCapsule_GetName(capsule)
        if name is None or name.count(b'int *') != 6 or name.count(b'char *') != 2 or b'long' in name:
            return None
        return api.PyCapsule_GetPointer(capsule, name)
    except (ImportError, KeyError):
        return None


def _scipy_dsyevr_address():
    """Address of the dsyevr that scipy.linalg.eigh itself calls (Fortran ABI, 32-bit integers)."""
    from scipy.linalg import cython_lapack
    capsule = cython_lapack.__pyx_capi__['dsyevr']
    api = ctypes.pythonapi
    api.PyCapsule_GetName.restype = ctypes.c_char_p
    api.PyCapsule_GetName.argtypes = [ctypes.py_object]
    api.PyCapsule_GetPointer.restype = ctypes.c_void_p
    api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
    name = api.PyCapsule_GetName(capsule)
    if name is None or name.count(b'int *') != 11 or b'long' in name:
        raise FoklNativeError(-2, f"unexpected signature of scipy's dsyevr: {name!r}")
    return api.PyCapsule_GetPointer(capsule, name)


class PoolJob:
    """A job on a HostPool.  Keeps the buffers the job reads / writes alive; ``wait()`` may be called repeatedly."""
    __slots__ = ('_h', 'keep', 'result', 'recycle', 'held', 'unresolved', 'ignore_failure', 'co_reader')

    def __init__(self, handle, keep, result=None, tentative=False):
        self._h, self.keep, self.result = handle, keep, result
        self.recycle = None                 # raw buffers the owner of the pool may reuse once the job has run
        self.held = None                    # raw buffer a LATER job will still read: not reusable when this one is done
        self.unresolved = bool(tentative)   # a tentative noise job that has not been given its verdict yet
        self.ignore_failure = False         # a chain started ahead of the decision that it is needed: its tape may go
        self.co_reader = None               # noise job: such a chain, given up, that may still be reading the tape

    def done(self):
        """True once the job has run (its native record is then released, the buffers may go)."""
        if self._h is not None and load().fokl_pool_poll(self._h):
            self.wait()
        return self._h is None

    def resolve(self, commit):
        """Verdict on a tentative noise job (fokl_pool_resolve)."""
        if self.unresolved:
            self.unresolved = False
            _check(load().fokl_pool_resolve(self._h, int(bool(commit))))

    def wait(self):
        if self._h is not None:
            h, self._h = self._h, None
            rc = load().fokl_pool_wait(h)
            if not self.ignore_failure:
                _check(rc)
        return self.result


class SpectralResult:
    """Outputs of fokl_pool_submit_spectral: lamb, Qt (row j = eigenvector j), qty = Q'Xty, betahat, and the residual
    moments (sum r, sum r^2) of y - X betahat formed from the Gram."""
    __slots__ = ('lamb', 'Qt', 'qty', 'betahat', 'moments', '_buf', '_addr')

    @staticmethod
    def doubles(p1):
        return p1 * (p1 + 3) + 2

    def __init__(self, p1, buf=None):
        """buf: an existing float64 buffer of doubles(p1) values holding a result computed elsewhere (another rank)."""
        buf = self._buf = np.empty(p1 * (p1 + 3) + 2, dtype=np.float64) if buf is None else buf
        self._addr = buf.__array_interface__['data'][0]
        self.lamb, self.qty, self.betahat = buf[:p1], buf[p1:2 * p1], buf[2 * p1:3 * p1]
        self.Qt = buf[3 * p1:3 * p1 + p1 * p1].reshape(p1, p1)
        self.moments = buf[3 * p1 + p1 * p1:]

    def pointers(self, p1):
        """lamb_out, qt_out, qty_out, betahat_out, moments_out"""
        a = self._addr
        return (a, a + 24 * p1, a + 8 * p1, a + 16 * p1, a + 8 * p1 * (p1 + 3))


class HostPool:
    """include/fokl_hip.h: fokl_pool_* -- the noise thread (owns ``stream`` until close()), chain threads and
    spectral threads of one fit."""

    def __init__(self, stream, chain_threads=2, finish_threads=2, spectral_threads=3, noise_cpu=-1, bulk_threads=2,
                 prestates=None, device_dgemm=None):
        """prestates: (address, entries) of a pre-state ring (DeviceChainEngine.prestate_ring()) for a device that
        regenerates the stream itself.  device_dgemm: (device, columns) -- the eigen-update's product on that device's matrix
        cores for models of at least that many columns (fokl_device_dgemm), scipy's dgemm below."""
        self._lib = load()
        self.stream = stream
        self._h = None
        h = c_vp(0)
        fn = _scipy_dsyevr_address() if spectral_threads > 0 else None
        self.finish_threads = int(finish_threads)
        _check(self._lib.fokl_pool_create(int(chain_threads), self.finish_threads, int(spectral_threads),
                                          max(1, int(bulk_threads)), int(noise_cpu), c_vp(fn),
                                          *stream.args(), prestates[0] if prestates else None,
                                          prestates[1] if prestates else 0, ctypes.byref(h)))
        self._h = h
        # models of FOKL_EIGH_DC_FROM columns or more (default 80; 0: never): LAPACK's divide-and-conquer driver, which
        # shares dsyevr's tridiagonal reduction and returns its eigenpairs to ~3e-12 in 2/3 of the time (half at 585 columns)
        self.dsyevd_from = 0
        dc_from = int(os.environ.get('FOKL_EIGH_DC_FROM', '80'))
        if spectral_threads > 0 and dc_from > 0 and os.environ.get('FOKL_EIGH_SIGNS', 'canonical') != 'lapack':
            fd = _scipy_dsyevd_address()
            if fd:
                _check(self._lib.fokl_pool_use_dsyevd(self._h, c_vp(fd), dc_from))
                self.dsyevd_from = dc_from
        # the product of the eigen-update (submit_spectral_update; the search core's FOKL_EIGH_UPDATE)
        self.has_dgemm = False
        if spectral_threads > 0:
            fg = _scipy_dgemm_address()
            self.device_dgemm_from = 0
            if fg and device_dgemm is not None:
                entry = c_vp(0)
                if self._lib.fokl_device_dgemm_configure(int(device_dgemm[0]), c_vp(fg), int(device_dgemm[1]),
                                                         ctypes.byref(entry)) == 0 and entry.value:
                    fg, self.device_dgemm_from = entry.value, int(device_dgemm[1])
            if fg:
                _check(self._lib.fokl_pool_use_dgemm(self._h, c_vp(fg)))
                self.has_dgemm = True

    def spectral_affinity(self, cpus):
        cpus = np.ascontiguousarray(sorted(cpus), dtype=np.int32)
        _check(self._lib.fokl_pool_spectral_affinity(self._h, _ptr(cpus), cpus.shape[0]))

    def stream_handle(self):
        """The pool's fokl_stream (for DeviceChainEngine.bind)."""
        return self._lib.fokl_pool_stream(self._h)

    def set_walk_helpers(self, count, cpus=None):
        """fokl_stream_set_helpers on the pool's stream: helper threads for the serial walk (positions + accept tests of the
        blocks the walking thread chases), each on the logical CPU given for it."""
        arr = None
        if cpus is not None:
            arr = np.ascontiguousarray(list(cpus)[:count] + [-1] * max(0, count - len(cpus)), dtype=np.int32)
        _check(self._lib.fokl_stream_set_helpers(c_vp(self.stream_handle()), int(count), _ptr(arr) if arr is not None else None))

    def place_bulk_threads(self, cpus):
        """fokl_stream_place_bulk on the pool's stream: bulk thread i on logical CPU cpus[i % len(cpus)]."""
        arr = np.ascontiguousarray(list(cpus), dtype=np.int32)
        _check(self._lib.fokl_stream_place_bulk(c_vp(self.stream_handle()), _ptr(arr), arr.shape[0]))

    def close(self):
        """Runs everything still queued (each submitted tape advances the stream), then stops the threads."""
        if self._h:
            self._lib.fokl_pool_destroy(self._h)
            self._h = None

    __del__ = close

    def submit_noise(self, tape, astar, atau_star, tentative=False, finish=False):
        """tentative: record ahead of the decision; the job must then get ``resolve(commit)`` (see the header).
        finish: the finish threads complete the tape while it is recorded, before a chain is asked for."""
        h = c_vp(0)
        finish = bool(finish) and self.finish_threads > 0
        _check(self._lib.fokl_pool_submit_noise(self._h, tape.p1, tape.draws, float(astar), float(atau_star),
                                                tape.rows_pointer(), *tape.pointers(), tape.progress_pointer(),
                                                int(bool(tentative)), tape.block_done_pointer(), tape.BLOCK, int(finish),
                                                None, ctypes.byref(h)))
        tape.finishing_requested = finish
        tape.materialised_by_pool = True
        return PoolJob(h, (tape,), tape, tentative)

    def submit_chain(self, lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape, w_raw=None):
        """Returns a job whose result is (w [draws, p1], bstar_negative int32[1]); w is carved out of the float64 buffer
        w_raw if that is given."""
        p1 = lamb.shape[0]
        if p1 != tape.p1:
            raise ValueError("tape was recorded for a different model size")
        if w_raw is None:
            w = np.empty((tape.draws, p1), dtype=np.float64)
        else:
            w = w_raw[:tape.draws * p1].reshape(tape.draws, p1)
        flag = np.zeros(1, dtype=np.int32)
        h = c_vp(0)
        _check(self._lib.fokl_pool_submit_chain(self._h, _ptr(lamb), _ptr(qty), p1, float(b), float(btau), float(dtd),
                                                float(sigsqd0), float(tausqd0), tape.draws, *tape.pointers(),
                                                tape.progress_pointer(), tape.block_done_pointer(),
                                                tape.BLOCK, int(tape.finishing_requested), _ptr(w), _ptr(flag),
                                                None, None, ctypes.byref(h)))
        return PoolJob(h, (lamb, qty, tape, w, flag), (w, flag))

    def submit_spectral(self, gram, idx, ycol):
        """gram: C-contiguous float64 [L, L]; idx: int32 [p1].  Returns a job whose result is a SpectralResult."""
        p1 = idx.shape[0]
        res = SpectralResult(p1)
        h = c_vp(0)
        _check(self._lib.fokl_pool_submit_spectral(self._h, _ptr(gram), gram.shape[1], _ptr(idx), p1, int(ycol),
                                                   *res.pointers(p1), ctypes.byref(h)))
        return PoolJob(h, (gram, idx, res), res)

    def submit_spectral_update(self, gram, idx, ycol, parent, parent_pos):
        """G2 of gram[idx][:, idx] from the finished SpectralResult ``parent`` of the model that has one more column, at
        position ``parent_pos`` of its column list.  -> (job, updated): updated[0] is 1 when the eigenpairs were derived
        from the parent's, 0 when the model was decomposed afresh after all (once the job has run)."""
        p1 = idx.shape[0]
        res = SpectralResult(p1)
        updated = np.full(1, -1, dtype=np.int32)
        h = c_vp(0)
        _check(self._lib.fokl_pool_submit_spectral_update(self._h, _ptr(gram), gram.shape[1], _ptr(idx), p1, int(ycol),
                                                          _ptr(parent.lamb), _ptr(parent.Qt), int(parent_pos), None,
                                                          *res.pointers(p1), _ptr(updated), ctypes.byref(h)))
        return PoolJob(h, (gram, idx, res, parent, updated), res), updated

    def busy_seconds(self):
        v = [c_dbl(0) for _ in range(4)]
        _check(self._lib.fokl_pool_busy_seconds(self._h, *[ctypes.byref(x) for x in v]))
        w = [c_dbl(0), c_dbl(0)]
        _check(self._lib.fokl_pool_noise_waits(self._h, ctypes.byref(w[0]), ctypes.byref(w[1])))
        b, ww, seg, ga, ge = c_dbl(0), c_dbl(0), c_i64(0), c_i64(0), c_i64(0)
        _check(self._lib.fokl_pool_stream_stats(self._h, *[ctypes.byref(x) for x in (b, ww, seg, ga, ge)]))
        return dict(noise=v[0].value, chain=v[1].value, finish=v[2].value, spectral=v[3].value,
                    noise_queue_wait=w[0].value, noise_verdict_wait=w[1].value, bulk=b.value, walker_wait=ww.value,
                    stream_segments=seg.value, gamma_attempts=ga.value, gamma_attempts_exact=ge.value)


# ---------------------------------------------------------------------------------------------------------
# the native half of a search (include/fokl_hip.h: fokl_search_*)
# ---------------------------------------------------------------------------------------------------------

class _SearchParams(ctypes.Structure):
    _fields_ = [('n', c_i64), ('a', c_dbl), ('b', c_dbl), ('atau', c_dbl), ('btau', c_dbl), ('threshav', c_dbl),
                ('threshstda', c_dbl), ('threshstdb', c_dbl), ('guess_margin', c_dbl), ('draws', c_i32), ('half0', c_i32),
                ('aic', c_i32), ('lookahead', c_i32), ('foresight', c_i32), ('speculation_max', c_i32),
                ('tentative_tapes', c_i32), ('test_rewinds', c_i32), ('device_chain_columns', c_i32),
                ('finish_threads', c_i32), ('flip_guess', c_i32), ('device_rows', c_i32)]


class _OutcomeView(ctypes.Structure):
    _fields_ = [('spectrum', c_vp), ('idx', c_vp), ('ev', c_dbl), ('siglik', c_dbl), ('intercept_scale', c_dbl),
                ('p1', c_i32), ('on_device', c_i32)]


_FORESEE_CB = ctypes.CFUNCTYPE(None, c_vp, ctypes.POINTER(c_i32), c_int)
_IDLE_CB = ctypes.CFUNCTYPE(c_int, c_vp)
_RESIDUAL_CB = ctypes.CFUNCTYPE(c_int, c_vp, ctypes.POINTER(c_i32), c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl),
                                ctypes.POINTER(c_dbl))


class _KillTestsArgs(ctypes.Structure):
    _fields_ = [('gram', c_vp), ('columns', c_vp), ('mean_abs', c_vp), ('rel_std', c_vp), ('slots', c_vp), ('best', c_vp),
                ('ahead_keys', c_vp), ('ahead_offsets', c_vp), ('ahead_spectra', c_vp), ('user', c_vp),
                ('foresee', _FORESEE_CB), ('idle_work', _IDLE_CB), ('residual', _RESIDUAL_CB), ('active', c_i32),
                ('proposals', c_i32), ('n_prev', c_i32), ('vm_next', c_i32), ('ahead_count', c_i32)]


class _KillTestsResult(ctypes.Structure):
    _fields_ = [('killed', c_vp), ('best', c_vp), ('evmin', c_dbl), ('killed_count', c_i32), ('best_is_new', c_i32)]


SEARCH_STATS = ('gibbs_calls', 'kill_tests', 'terms_logical', 't_eigh', 't_chain', 'chains_materialised', 'bic_from_gram',
                'tapes_rewound', 'tapes_wasted', 'chains_ahead', 'chains_ahead_unused', 'chains_skipped',
                'spectral_submitted', 'device_chains', 'chains_fetched', 'guessed', 'guess_waits', 'guesses_verified',
                'dchain_kernel_s', 'dchain_timed', 't_resid', 't_kill_loop', 'tapes_materialised', 'rows_chains', 'path_repredicted', 'spectral_device',
                'spectral_updated', 'direct_tests', 'direct_max_rel', 'chains_cancelled', 't_settle',
                'guess_max_dev', 'direct_in_band', 'stats_by_chain_thread')


class NativeSearch:
    """include/fokl_hip.h: fokl_search_* -- the tapes on order, G2 jobs, chains and the kill-test loop of one fit, next to
    its HostPool.  Handles (spectra, tapes, outcomes) are plain integers; engine.NativeOutcome wraps the last kind."""

    def __init__(self, pool, dchain, **params):
        self._lib = load()
        self._h = None
        self._pool, self._dchain = pool, dchain             # keep them alive: the search uses them until close()
        prm = _SearchParams(**params)
        h = c_vp(0)
        _check(self._lib.fokl_search_create(pool._h, dchain._h if dchain is not None else None, ctypes.byref(prm),
                                            ctypes.byref(h)))
        self._h = h
        self.draws = int(params['draws'])

    def close(self):
        if self._h:
            self._lib.fokl_search_destroy(self._h)
            self._h = None

    __del__ = close

    def _checked(self, rc):
        if rc:
            if self._lib.fokl_search_mispredicted(self._h):
                from .host_pipeline import Misprediction
                raise Misprediction(self._lib.fokl_search_error(self._h).decode())
            _check(rc)

    def set_substage(self, term_ids):
        ids = np.ascontiguousarray(term_ids, dtype=np.int64)
        self._checked(self._lib.fokl_search_set_substage(self._h, _ptr(ids), ids.shape[0]))

    def speculate(self, sizes):
        """sizes: [(columns, is_model)] the stream will probably serve next, in order."""
        n = len(sizes)
        cols = np.array([s for s, _ in sizes], dtype=np.int32)
        model = np.array([int(m) for _, m in sizes], dtype=np.int32)
        self._checked(self._lib.fokl_search_speculate(self._h, _ptr(cols), _ptr(model), n))

    def drop_speculation(self):
        self._checked(self._lib.fokl_search_drop_speculation(self._h))

    def bind_spectral(self, engine, max_columns=None, slack=-1.0, lookahead=-1):
        """G2 of models of up to max_columns columns on the device (a DeviceSpectralEngine; None: the pool's threads) when
        requested `slack` kernel durations ahead of need (0: always; < 0: the library's default)."""
        self._spectral_engine = engine                   # kept alive as long as the search
        limit = (engine.max_columns if max_columns is None else int(max_columns)) if engine is not None else 0
        self._checked(self._lib.fokl_search_bind_spectral(self._h, engine._h if engine is not None else None, limit,
                                                          float(slack), int(lookahead)))

    def hold_spectral(self, hold):
        self._checked(self._lib.fokl_search_hold_spectral(self._h, 1 if hold else 0))

    def set_update(self, from_columns, depth, lookahead=0):
        """Kill tests' G2 from the tested-against model's eigenpairs for parents of from_columns columns or more (0: never), at
        most `depth` steps from a fresh decomposition; `lookahead` (0: the search's own): G2 look-ahead while that is on, in
        sub-stages whose model has fewer than 192 columns."""
        self._checked(self._lib.fokl_search_set_update(self._h, int(from_columns), int(depth), int(lookahead)))

    def set_decide(self, mode, tolerance=0.0):
        """Kill tests' BIC decisions: 0 from G2 of every trial model, 1 from the downdated least-squares model of the
        sub-stage, confirmed by the accepted models' eigenpairs to `tolerance` (relative; 0: 1e-9)."""
        self._checked(self._lib.fokl_search_set_decide(self._h, int(mode), float(tolerance)))

    def set_deterministic(self, on=True):
        """Ranks repeating one search side by side: no decision may depend on when a chain's statistics arrive."""
        self._checked(self._lib.fokl_search_set_deterministic(self._h, int(bool(on))))

    def spectral(self, gram, idx, parent=None, parent_pos=-1):
        """parent / parent_pos: a spectrum handle of this search for the model that has one more column, and which of its
        columns this model lacks (G2 may then follow from the parent's eigenpairs, set_update)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        h = c_vp(0)
        if parent is None:
            self._checked(self._lib.fokl_search_spectral(self._h, _ptr(gram), gram.shape[1], _ptr(idx), idx.shape[0],
                                                         ctypes.byref(h)))
        else:
            self._checked(self._lib.fokl_search_spectral_from(self._h, _ptr(gram), gram.shape[1], _ptr(idx), idx.shape[0],
                                                              c_vp(parent), int(parent_pos), ctypes.byref(h)))
        return h.value

    def spectrum_done(self, spectrum):
        return bool(self._lib.fokl_spectrum_done(c_vp(spectrum)))

    def spectrum_view(self, spectrum):
        """-> SpectralResult over the search's own buffer (valid while the spectrum is referenced)."""
        buf, p1 = c_vp(0), c_int(0)
        self._checked(self._lib.fokl_spectrum_wait(self._h, c_vp(spectrum), ctypes.byref(buf), ctypes.byref(p1)))
        n = p1.value
        arr = np.ctypeslib.as_array((ctypes.c_double * SpectralResult.doubles(n)).from_address(buf.value))
        return SpectralResult(n, arr)

    def spectrum_retain(self, spectrum):
        """-> the same handle, holding one more reference (spectrum_release when done with it)."""
        self._checked(self._lib.fokl_spectrum_retain(self._h, c_vp(spectrum)))
        return spectrum

    def spectrum_release(self, spectrum):
        if self._h:
            self._lib.fokl_spectrum_release(self._h, c_vp(spectrum))

    def model_begin(self, gram, idx, given, then):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        cols = np.array([s for s, _ in then], dtype=np.int32)
        model = np.array([int(m) for _, m in then], dtype=np.int32)
        sp, tape = c_vp(0), c_vp(0)
        self._checked(self._lib.fokl_search_model_begin(self._h, _ptr(gram), gram.shape[1], _ptr(idx), idx.shape[0],
                                                        c_vp(given) if given else None, _ptr(cols), _ptr(model),
                                                        len(then), ctypes.byref(sp), ctypes.byref(tape)))
        return sp.value, tape.value

    def model_commit(self, spectrum, tape, dtd, new_terms=0):
        """new_terms: the model's last `new_terms` active columns are a sub-stage's new terms -- their statistics
        (outcome_new_term_stats) are formed by the chain thread as soon as the chain has run."""
        out = c_vp(0)
        self._checked(self._lib.fokl_search_model_commit(self._h, c_vp(spectrum), c_vp(tape), float(dtd), int(new_terms),
                                                         ctypes.byref(out)))
        return out.value

    def score(self, outcome, s1, s2, n_prev, kill):
        ev = c_dbl(0)
        self._checked(self._lib.fokl_search_score(self._h, c_vp(outcome), float(s1), float(s2), int(n_prev), int(kill),
                                                  ctypes.byref(ev)))
        return ev.value

    def outcome_info(self, outcome):
        view = _OutcomeView()
        self._checked(self._lib.fokl_outcome_info(self._h, c_vp(outcome), ctypes.byref(view)))
        return view

    def outcome_spectrum(self, outcome):
        h = c_vp(0)
        self._checked(self._lib.fokl_outcome_spectrum(self._h, c_vp(outcome), ctypes.byref(h)))
        return h.value

    def outcome_chain_ready(self, outcome):
        return bool(self._lib.fokl_outcome_chain_ready(c_vp(outcome)))

    def outcome_draws(self, outcome, p1):
        w = c_vp(0)
        self._checked(self._lib.fokl_outcome_draws(self._h, c_vp(outcome), ctypes.byref(w)))
        return np.ctypeslib.as_array((ctypes.c_double * (self.draws * p1)).from_address(w.value)).reshape(self.draws, p1)

    def outcome_intercept_scale(self, outcome):
        v = c_dbl(0)
        self._checked(self._lib.fokl_outcome_intercept_scale(self._h, c_vp(outcome), ctypes.byref(v)))
        return v.value

    def outcome_new_term_stats(self, outcome, cols, half0, half1):
        """(|mean beta| over rows half1 .., std over rows half1 .. / |mean over rows half0 ..|) of the model's active columns
        `cols` (FR:1656-1658); waits for the chain."""
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        mean_abs, rel_std = np.empty(cols.shape[0]), np.empty(cols.shape[0])
        self._checked(self._lib.fokl_outcome_new_term_stats(self._h, c_vp(outcome), _ptr(cols), cols.shape[0], int(half0),
                                                            int(half1), _ptr(mean_abs), _ptr(rel_std)))
        return mean_abs, rel_std

    def outcome_release(self, outcome):
        if self._h:
            self._lib.fokl_outcome_release(self._h, c_vp(outcome))

    def outcome_drop(self, outcome):
        if self._h:
            self._lib.fokl_outcome_drop(self._h, c_vp(outcome))

    def verify(self, block=False):
        self._checked(self._lib.fokl_search_verify(self._h, int(bool(block))))

    def register_forecast(self, key, spectrum, dtd):
        key = np.ascontiguousarray(key, dtype=np.int32)
        self._checked(self._lib.fokl_search_register_forecast(self._h, _ptr(key), key.shape[0], c_vp(spectrum), float(dtd)))

    def clear_forecasts(self):
        self._lib.fokl_search_clear_forecasts(self._h)

    def likely_first_tests(self, spectrum, n_new, siglik=None):
        """-> [(active column, probably accepted)] in testing order (fokl_search_likely_first_tests)."""
        out = np.empty(max(1, int(n_new)), dtype=np.int32)
        acc = np.zeros(max(1, int(n_new)), dtype=np.int32)
        count = c_int(0)
        self._checked(self._lib.fokl_search_likely_first_tests(self._h, c_vp(spectrum), int(n_new),
                                                               float('nan') if siglik is None else float(siglik),
                                                               _ptr(out), _ptr(acc), ctypes.byref(count)))
        return [(int(c), bool(a)) for c, a in zip(out[:count.value], acc[:count.value])]

    def stats(self):
        v = np.zeros(len(SEARCH_STATS) + 8)
        n = self._lib.fokl_search_stats(self._h, _ptr(v), v.shape[0])
        if n != len(SEARCH_STATS):
            raise FoklNativeError(-3, "fokl_search_stats: the library's counters do not match SEARCH_STATS")
        return dict(zip(SEARCH_STATS, v[:n].tolist()))

    def trace(self):
        n = self._lib.fokl_search_trace(self._h, None, 0)
        rec = np.zeros((max(1, n), 5))
        self._lib.fokl_search_trace(self._h, _ptr(rec), n)
        return TraceRecords(rec[:n])

    def kill_tests(self, gram, columns, mean_abs, rel_std, slots, best, n_prev, vm_next, ahead, foresee=None,
                   idle_work=None, residual=None):
        """fokl_search_kill_tests.  ahead: {frozenset of active columns: spectrum handle}; the callbacks are Python
        callables -- foresee(list of killed columns), idle_work(), residual(idx, betahat) -> (s1, s2) -- whose exceptions
        are re-raised here after the native loop has returned.  -> (killed columns, evmin, best handle, best_is_new)."""
        A = int(gram.shape[0]) - 1
        vm = int(columns.shape[0])
        columns = np.ascontiguousarray(columns, dtype=np.int32)
        mean_abs = np.ascontiguousarray(mean_abs, dtype=np.float64)
        rel_std = np.ascontiguousarray(rel_std, dtype=np.float64)
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        keys, offsets, handles = [], [0], []
        for key, h in ahead.items():
            keys.extend(sorted(key))
            offsets.append(len(keys))
            handles.append(h)
        keys = np.array(keys if keys else [0], dtype=np.int32)
        offsets = np.array(offsets, dtype=np.int32)
        handle_arr = (c_vp * max(1, len(handles)))(*handles)
        raised = []

        def guard(fn, *args):
            try:
                return fn(*args)
            except BaseException as exc:                     # noqa: B902 -- carried over the native frame
                raised.append(exc)
                return None

        def cb_foresee(_user, killed, count):
            if not raised:
                guard(foresee, [int(killed[i]) for i in range(count)])

        def cb_idle(_user):
            if not raised:
                guard(idle_work)
            return -3 if raised else 0

        def cb_residual(_user, idx, p1, betahat, s1, s2):
            if not raised:
                got = guard(residual, np.array([idx[i] for i in range(p1)], dtype=np.int32),
                            np.array([betahat[i] for i in range(p1)]))
                if got is not None:
                    s1[0], s2[0] = float(got[0]), float(got[1])
            return -3 if raised else 0

        killed = np.zeros(max(1, vm), dtype=np.int32)
        args = _KillTestsArgs(
            gram=_ptr(gram), columns=_ptr(columns), mean_abs=_ptr(mean_abs), rel_std=_ptr(rel_std),
            slots=_ptr(slots), best=best, ahead_keys=_ptr(keys), ahead_offsets=_ptr(offsets),
            ahead_spectra=ctypes.cast(handle_arr, c_vp).value, user=None,
            foresee=_FORESEE_CB(cb_foresee) if foresee is not None else _FORESEE_CB(),
            idle_work=_IDLE_CB(cb_idle) if idle_work is not None else _IDLE_CB(),
            residual=_RESIDUAL_CB(cb_residual) if residual is not None else _RESIDUAL_CB(),
            active=A, proposals=vm, n_prev=int(n_prev), vm_next=-1 if vm_next is None else int(vm_next),
            ahead_count=len(handles))
        res = _KillTestsResult(killed=_ptr(killed))
        rc = self._lib.fokl_search_kill_tests(self._h, ctypes.byref(args), ctypes.byref(res))
        if raised:
            raise raised[0]
        self._checked(rc)
        return [int(c) for c in killed[:res.killed_count]], float(res.evmin), res.best, bool(res.best_is_new)


# ---------------------------------------------------------------------------------------------------------
# G3 on the device
# ---------------------------------------------------------------------------------------------------------

class _PinnedBlock:
    """Owner of one fokl_host_alloc allocation.  When the last array view of it goes the block returns to a free list
    of its size (page-locking and unlocking cost milliseconds; tapes come and go by the hundred per fit)."""
    __slots__ = ('ptr', 'nbytes')
    FREE = {}                                                     # nbytes -> [addresses]
    KEEP_BYTES = int(float(os.environ.get('FOKL_PINNED_POOL_MB', '1024')) * (1 << 20))
    kept = 0

    def __init__(self, nbytes):
        spare = _PinnedBlock.FREE.get(nbytes)
        if spare:
            self.ptr = spare.pop()
            _PinnedBlock.kept -= nbytes
        else:
            ptr = c_vp(0)
            _check(load().fokl_host_alloc(ctypes.c_size_t(nbytes), ctypes.byref(ptr)))
            self.ptr = ptr.value
        self.nbytes = nbytes

    def __del__(self):
        try:
            if self.ptr:
                if _PinnedBlock.kept + self.nbytes <= _PinnedBlock.KEEP_BYTES:
                    _PinnedBlock.FREE.setdefault(self.nbytes, []).append(self.ptr)
                    _PinnedBlock.kept += self.nbytes
                else:
                    load().fokl_host_free(c_vp(self.ptr))
        except Exception:                                          # interpreter shutdown: the driver reclaims it anyway
            pass
        self.ptr = None


def pinned_empty(doubles):
    """A float64 array of ``doubles`` elements in page-locked host memory (tapes that the device chains read in place)."""
    block = _PinnedBlock(int(doubles) * 8)
    raw = (ctypes.c_double * int(doubles)).from_address(block.ptr)
    raw._fokl_owner = block                                       # the ctypes array is the ndarray's base: keeps the block
    return np.ctypeslib.as_array(raw)


class DeviceChainJob:
    """One chain on a DeviceChainEngine (include/fokl_hip.h: fokl_dchain_*).  ``wait()`` -> (mean of w over the rows
    from ``stat_first`` on, bstar-negative flag); ``fetch_w()`` -> the draws in the eigenbasis [draws, p1];
    ``release()`` frees the device slot (idempotent).  Keeps the host buffers the job reads alive.
    Completion is read from the job's statistics area in page-locked host memory (the kernel stores the ticket there
    last): ``done()`` is a memory load, not a call into the runtime."""
    __slots__ = ('_engine', '_ticket', 'p1', 'draws', 'keep', 'recycle', '_stats', 'ignore_failure', '_area', '_flag',
                 '_mark', '_misses')
    unresolved = False                  # PoolJob's interface: only tentative noise jobs wait for a verdict

    def __init__(self, engine, ticket, p1, draws, keep, stats_address=None):
        self._engine, self._ticket, self.p1, self.draws, self.keep = engine, ticket, p1, draws, keep
        self.recycle = None
        self._stats = None
        self.ignore_failure = False
        self._area = self._flag = None
        self._mark = float(ticket) if ticket is not None else 0.0
        self._misses = 0
        if stats_address:
            self._area = np.ctypeslib.as_array((ctypes.c_double * (6 + p1)).from_address(stats_address))
            self._flag = ctypes.c_double.from_address(stats_address + 8 * (4 + p1))

    def resolve(self, commit):
        pass

    def _ran(self):
        return self._flag is not None and self._flag.value == self._mark

    def done(self):
        """True once the device is through with the job's host buffers (the chain has run, or the job has failed)."""
        if self._ticket is None or self._stats is not None or self._ran():
            return True
        if self._flag is not None:
            # a failed job never sets the flag: the engine is asked now and then (it knows a failed job: poll reports it as
            # finished, wait() / release() report the error), so that such a job does not hold its buffers for good
            self._misses += 1
            if self._misses % 64:
                return False
        return bool(self._engine._lib.fokl_dchain_poll(self._engine._h, self._ticket))

    def wait(self):
        if self._stats is None:
            if self._ticket is None:
                raise RuntimeError("device chain: released before anybody read its statistics")
            if self._ran():
                self._stats = np.array(self._area[:4 + self.p1])
            else:
                buf = np.empty(4 + self.p1, dtype=np.float64)
                _check(self._engine._lib.fokl_dchain_wait(self._engine._h, self._ticket, _ptr(buf)))
                self._stats = buf
        return self._stats[4:], np.array([int(self._stats[0])], dtype=np.int32)

    @property
    def kernel_seconds(self):
        """How long the chain's wavefront ran, by the kernel's own clock (0.0 if not known)."""
        if self._area is not None and self._ran():
            return float(self._area[5 + self.p1])
        return 0.0

    @property
    def last_state(self):
        """(sigma^2, tau^2) after the last iteration."""
        self.wait()
        return float(self._stats[1]), float(self._stats[2])

    def fetch_w(self, out=None):
        w = np.empty((self.draws, self.p1), dtype=np.float64) if out is None else out
        _check(self._engine._lib.fokl_dchain_fetch_w(self._engine._h, self._ticket, _ptr(w)))
        return w

    def try_release(self):
        """release() if that takes no waiting.  -> True when the slot is free."""
        if self._ticket is None or not self._engine._h:
            return True
        if self._flag is not None and not self._ran() and self._stats is None:
            return False
        if self._engine._lib.fokl_dchain_try_release(self._engine._h, self._ticket):
            self._ticket, self.keep, self._area, self._flag = None, None, None, None
            return True
        return False

    def release(self):
        if self._ticket is not None and self._engine._h:
            self._engine._lib.fokl_dchain_release(self._engine._h, self._ticket)
        self._ticket = None
        self.keep = self._area = self._flag = None


class DeviceChainEngine:
    """include/fokl_hip.h: fokl_dchain_* -- finishing of the polar normals and the Gibbs recursion on the GPU."""
    max_columns = 768                   # one wavefront per chain, up to 12 eigen-directions per lane

    def __init__(self, device=0, slots=96):
        self._lib = load()
        self._h = None
        h = c_vp(0)
        _check(self._lib.fokl_dchain_create(int(device), int(slots), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if self._h:
            self._lib.fokl_dchain_destroy(self._h)
            self._h = None

    __del__ = close

    def submit(self, lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape, stat_first=0, follow=True):
        """Queue the chain of ``tape`` (a NoiseTape, possibly still on record: ``follow``) for the model (lamb, qty)."""
        lamb = np.ascontiguousarray(lamb, dtype=np.float64)
        qty = np.ascontiguousarray(qty, dtype=np.float64)
        p1 = lamb.shape[0]
        if p1 != tape.p1:
            raise ValueError("tape was recorded for a different model size")
        ptr = tape.pointers()
        finished = bool(tape.finishing_requested)
        ticket, area = c_i64(0), c_vp(0)
        _check(self._lib.fokl_dchain_submit(self._h, p1, tape.draws, _ptr(lamb), _ptr(qty), float(b), float(btau),
                                            float(dtd), float(sigsqd0), float(tausqd0), ptr[0], ptr[2], ptr[3], ptr[4],
                                            tape.progress_pointer() if follow else None,
                                            tape.block_done_pointer() if finished or tape.materialised_by_pool else None,
                                            tape.BLOCK, int(finished),
                                            int(stat_first), ctypes.byref(ticket), ctypes.byref(area)))
        return DeviceChainJob(self, ticket.value, p1, tape.draws, (tape,), area.value)

    def prestate_ring(self):
        """(address, entries) of the page-locked ring a stream's bulk threads leave its pre-states in: hand it to the
        HostPool / StreamEngine whose tapes this engine will expand from rows, then ``bind`` that stream."""
        ring, entries = c_vp(0), c_int(0)
        _check(self._lib.fokl_dchain_prestate_ring(self._h, ctypes.byref(ring), ctypes.byref(entries)))
        return ring.value, entries.value

    def bind(self, stream_handle):
        _check(self._lib.fokl_dchain_bind_stream(self._h, stream_handle))

    def submit_rows(self, lamb, qty, b, btau, dtd, sigsqd0, tausqd0, astar, atau_star, tape, span, stat_first=0):
        """The chain of a tape that exists as rows (``tape.rows``, its gamma arrays and progress word in page-locked
        memory); span: uint64 [2] as fokl_pool_submit_noise fills it (position held from, position behind the tape)."""
        lamb = np.ascontiguousarray(lamb, dtype=np.float64)
        qty = np.ascontiguousarray(qty, dtype=np.float64)
        p1 = lamb.shape[0]
        ptr = tape.pointers()
        ticket, area = c_i64(0), c_vp(0)
        _check(self._lib.fokl_dchain_submit_rows(self._h, p1, tape.draws, _ptr(lamb), _ptr(qty), float(b), float(btau),
                                                 float(dtd), float(sigsqd0), float(tausqd0), float(astar), float(atau_star),
                                                 tape.rows_pointer(), ptr[3], ptr[4], tape.progress_pointer(), _ptr(span),
                                                 int(stat_first), ctypes.byref(ticket), ctypes.byref(area)))
        return DeviceChainJob(self, ticket.value, p1, tape.draws, (tape, span), area.value)

    def stream_stats(self):
        seg, rows = c_i64(0), c_i64(0)
        _check(self._lib.fokl_dchain_stream_stats(self._h, ctypes.byref(seg), ctypes.byref(rows)))
        return dict(segments_made=seg.value, rows_jobs=rows.value)

    def flush(self):
        """Issue what is queued now (the caller knows that nothing more is coming for a while)."""
        _check(self._lib.fokl_dchain_flush(self._h))

    def stats(self):
        busy, issued, launches, staged = c_dbl(0), c_i64(0), c_i64(0), c_i64(0)
        _check(self._lib.fokl_dchain_stats(self._h, ctypes.byref(busy), ctypes.byref(issued), ctypes.byref(launches),
                                           ctypes.byref(staged)))
        return dict(dispatch_s=busy.value, issued=issued.value, launches=launches.value, staged=staged.value)


class DeviceSpectralJob:
    """One eigen-decomposition on the device; the arrays are views of the engine's page-locked result area (valid until
    ``release``)."""

    def __init__(self, engine, ticket, result, p1):
        self.engine, self.ticket, self.p1 = engine, int(ticket), int(p1)
        n = self.p1
        area = (ctypes.c_double * (n * n + 3 * n + 7)).from_address(result)
        self._area = np.frombuffer(area, dtype=np.float64)
        self._released = False

    def done(self):
        rc = self.engine._lib.fokl_dspectral_poll(self.engine._h, self.ticket)
        if rc < 0:
            _check(-rc)
        return rc == 1

    def wait(self):
        """-> (lamb, Qt, qty, betahat, moments) like HostPool's spectral job."""
        _check(self.engine._lib.fokl_dspectral_wait(self.engine._h, self.ticket))
        n, a = self.p1, self._area
        return a[:n], a[3 * n:3 * n + n * n].reshape(n, n), a[n:2 * n], a[2 * n:3 * n], a[3 * n + n * n:3 * n + n * n + 2]

    def info(self):
        """-> dict(sweeps, rotations, seconds on the device) of a job that has run."""
        a = self._area[3 * self.p1 + self.p1 * self.p1 + 2:]
        return dict(sweeps=int(a[0]), rotations=int(a[1]), seconds=float(a[2]), not_converged=bool(a[3]))

    def release(self):
        if not self._released and self.engine._h:
            self._released = True
            _check(self.engine._lib.fokl_dspectral_release(self.engine._h, self.ticket))


class DeviceSpectralEngine:
    """include/fokl_hip.h: fokl_dspectral_* -- G2 (eigh of XtX sub-blocks, Q'Xty, betahat, residual moments) on the GPU."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = None
        h = c_vp(0)
        _check(self._lib.fokl_dspectral_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.max_columns = int(self._lib.fokl_dspectral_max_columns())

    def close(self):
        if self._h:
            self._lib.fokl_dspectral_destroy(self._h)
            self._h = None

    __del__ = close

    def set_signs(self, canonical=True):
        _check(self._lib.fokl_dspectral_set_signs(self._h, 1 if canonical else 0))

    def submit(self, gram, idx, launch=True):
        """Queue the decomposition of gram[idx][:, idx] (gram: [(A + 1), (A + 1)], ones column first, y last)."""
        gram = np.ascontiguousarray(gram, dtype=np.float64)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        ticket, result = c_i64(0), c_vp(0)
        _check(self._lib.fokl_dspectral_submit(self._h, _ptr(gram), int(gram.shape[0]), _ptr(idx), int(idx.size),
                                               int(gram.shape[0]) - 1, 1 if launch else 0, ctypes.byref(ticket),
                                               ctypes.byref(result)))
        return DeviceSpectralJob(self, ticket.value, result.value, idx.size)

    def flush(self):
        _check(self._lib.fokl_dspectral_flush(self._h))

    def stats(self):
        submitted, launches = c_i64(0), c_i64(0)
        _check(self._lib.fokl_dspectral_stats(self._h, ctypes.byref(submitted), ctypes.byref(launches)))
        return dict(submitted=submitted.value, launches=launches.value)


def gibbs_chain_device(engine, lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape, stat_first=0):
    """One chain on the device, start to finish: -> (w [draws, p1], mean of w from row stat_first, bstar-negative)."""
    job = engine.submit(lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape, stat_first, follow=False)
    try:
        mean_w, flag = job.wait()
        return job.fetch_w(), np.array(mean_w), bool(flag[0])
    finally:
        job.release()


# ---------------------------------------------------------------------------------------------------------
# device context
# ---------------------------------------------------------------------------------------------------------

_OPS_SIGS = (('reserve_slots', [c_vp, c_int]),
             ('build_terms', [c_vp, c_vp, c_int, c_vp]),
             ('gram', [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int]),
             ('gram_launch', [c_vp, c_vp, c_int, c_vp, c_int, c_int]),
             ('gram_fetch', [c_vp, c_vp, c_i64]),
             ('gram_ready', [c_vp]),
             ('bic_resid', [c_vp, c_vp, c_int, c_vp, c_vp, c_int]),
             ('bic_resid_launch', [c_vp, c_vp, c_int, c_vp]),
             ('bic_resid_fetch', [c_vp, c_vp, c_int]),
             ('bic_resid_terms_launch', [c_vp, c_vp, c_int, c_vp]))
_OPS_TYPES = {name: ctypes.CFUNCTYPE(c_int, *args) for name, args in _OPS_SIGS}


class _BackendOps(ctypes.Structure):
    _fields_ = [('ctx', c_vp)] + [(name, c_vp) for name, _ in _OPS_SIGS] + [('kernel_id', c_i32)]


class _RunParams(ctypes.Structure):
    _fields_ = [(name, c_i32) for name in (
        'm', 'n_phis', 'way3', 'tolerance', 'gimmie', 'draws', 'half0', 'lookahead', 'lookahead_native', 'foresight',
        'speculate_across', 'forecast_early', 'forecast_polls', 'matrix_free', 'update_from', 'update_depth',
        'update_lookahead', 'head_start', 'slot_capacity')]


RUN_STATS = ('terms_physical', 'substages', 'forecasts_used', 'forecasts_early', 'resid_matrix_free', 't_resid',
             'phase_prepare', 'phase_model', 'phase_statistics', 'phase_tests', 'phase_wrap_up')


def backend_ops(backend, kernel_id):
    """include/fokl_hip_internal.h fokl_backend_ops for a search backend -> (struct, what must stay alive with it).
    A backend on a DeviceContext hands over the library's own entry points (nothing of the loop passes through Python); any
    other object with HipBackend's methods (the checker backend of the CPU tests) is reached through callbacks."""
    lib = load()
    ops = _BackendOps()
    ops.kernel_id = int(kernel_id)
    ctx = getattr(backend, 'ctx', None)
    if isinstance(ctx, DeviceContext):
        ops.ctx = ctx._h
        for name, _ in _OPS_SIGS:
            setattr(ops, name, ctypes.cast(getattr(lib, 'fokl_' + name), c_vp).value)
        return ops, (ctx,)
    pending, raised = {}, []

    def arr(ptr, count, ctype, dtype):
        return np.array(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype * count)).contents), dtype=dtype)

    def guard(fn):
        def call(*args):
            try:
                fn(*args)
                return 0
            except BaseException as exc:                     # noqa: B902 -- carried over the native frame
                raised.append(exc)
                return -3
        return call

    def out(ptr, values):
        values = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        ctypes.memmove(ptr, values.ctypes.data, values.nbytes)

    def reserve_slots(_c, n):
        backend.reserve_slots(int(n))

    def build_terms(_c, terms, T, slots):
        sl = arr(slots, T, ctypes.c_int32, np.int32)
        m = pending['m']
        backend.build_terms(arr(terms, T * m, ctypes.c_int32, np.int32).reshape(T, m), [int(v) for v in sl])

    def gram(_c, rows, nr, cols, nc, dst, _path, allreduce):
        out(dst, backend.gram([int(v) for v in arr(rows, nr, ctypes.c_int32, np.int32)],
                              [int(v) for v in arr(cols, nc, ctypes.c_int32, np.int32)], bool(allreduce)))

    def gram_launch(_c, rows, nr, cols, nc, allreduce):
        pending['gram'] = backend.gram([int(v) for v in arr(rows, nr, ctypes.c_int32, np.int32)],
                                       [int(v) for v in arr(cols, nc, ctypes.c_int32, np.int32)], bool(allreduce))

    def gram_fetch(_c, dst, count):
        block = pending.pop('gram')
        assert block.size == count
        out(dst, block)

    def bic_resid(_c, slots, nc, betahat, dst, allreduce):
        out(dst, backend.bic_resid([int(v) for v in arr(slots, nc, ctypes.c_int32, np.int32)],
                                   arr(betahat, nc, ctypes.c_double, np.float64), bool(allreduce)))

    def bic_resid_launch(_c, slots, nc, betahat):
        pending['resid'] = backend.bic_resid([int(v) for v in arr(slots, nc, ctypes.c_int32, np.int32)],
                                             arr(betahat, nc, ctypes.c_double, np.float64), False)

    def bic_resid_fetch(_c, dst, _allreduce):
        out(dst, pending.pop('resid'))

    table = dict(reserve_slots=reserve_slots, build_terms=build_terms, gram=gram, gram_launch=gram_launch,
                 gram_fetch=gram_fetch, bic_resid=bic_resid, bic_resid_launch=bic_resid_launch,
                 bic_resid_fetch=bic_resid_fetch)
    keep = [pending, raised]
    for name, fn in table.items():
        cb = _OPS_TYPES[name](guard(fn))
        keep.append(cb)
        setattr(ops, name, ctypes.cast(cb, c_vp).value)
    return ops, tuple(keep)


class NativeRun:
    """include/fokl_hip_internal.h: fokl_run_* -- the sub-stage loop of a fit (csrc/fokl_run.cpp) on a NativeSearch.
    Created BEFORE the pool (its head start puts the first sub-stage's columns and Gram block under way)."""

    def __init__(self, backend, kernel_id, **params):
        self._lib = load()
        self._h = None
        self._ops, self._keep = backend_ops(backend, kernel_id)
        if isinstance(self._keep[0], dict):
            self._keep[0]['m'] = int(params['m'])
        self._prm = _RunParams(**{k: int(v) for k, v in params.items()})
        self.m = int(params['m'])
        h = c_vp(0)
        _check(self._lib.fokl_run_create(ctypes.byref(self._ops), ctypes.byref(self._prm), ctypes.byref(h)))
        self._h = h
        self._reraise()

    def _reraise(self):
        raised = self._keep[1] if isinstance(self._keep[0], dict) else None
        if raised:
            exc = raised[0]
            del raised[:]
            raise exc

    def set_update(self, from_columns, depth, lookahead):
        _check(self._lib.fokl_run_set_update(self._h, int(from_columns), int(depth), int(lookahead)))

    def search(self, native_search):
        """The loop.  Raises what a backend callback raised, or FoklNativeError with the search's / the run's message."""
        rc = self._lib.fokl_run_search(self._h, native_search._h)
        self._reraise()
        if rc != 0:
            native_search._checked(rc)
        return self

    def result(self):
        """-> (mtx [rows, m] float64, evs, per sub-stage (mean_abs, rel_std), handle of the returned model, handle of the
        last sub-stage's survivor, stats)"""
        rows, n_evs, subs = c_i32(0), c_i32(0), c_i32(0)
        best, last = c_vp(0), c_vp(0)
        _check(self._lib.fokl_run_result(self._h, ctypes.byref(rows), ctypes.byref(n_evs), ctypes.byref(subs),
                                         ctypes.byref(best), ctypes.byref(last)))
        mtx = np.zeros((max(rows.value, 0), self.m), dtype=np.int32)
        evs = np.zeros(n_evs.value)
        sizes = np.zeros(max(subs.value, 1), dtype=np.int32)
        stats = np.zeros(16)
        total = self._lib.fokl_run_arrays(self._h, _ptr(mtx) if mtx.size else None, _ptr(evs) if evs.size else None,
                                          _ptr(sizes), None, None, _ptr(stats))
        mean_abs, rel_std = np.zeros(max(total, 1)), np.zeros(max(total, 1))
        self._lib.fokl_run_arrays(self._h, None, None, None, _ptr(mean_abs), _ptr(rel_std), None)
        per, at = [], 0
        for size in sizes[:subs.value]:
            per.append(dict(mean_abs=mean_abs[at:at + size].copy(), rel_std=rel_std[at:at + size].copy()))
            at += int(size)
        return (mtx.astype(np.float64), evs, per, best.value, last.value, dict(zip(RUN_STATS, stats[:len(RUN_STATS)])))

    def close(self):
        if self._h is not None and self._h:
            self._lib.fokl_run_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceContext:
    """One HIP stream on one MI355X plus the resident dataset and column slots."""

    def __init__(self, device=0):
        self._lib = load()
        h = c_vp(0)
        _check(self._lib.fokl_ctx_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.fokl_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _check(rc, self._h)

    def sync(self):
        self._ck(self._lib.fokl_sync(self._h))

    # A model whose normalised inputs exist on this context only (upload_staged) is told to fetch them before the dataset
    # is replaced (owner._materialise_inputs()).
    _lazy_owner = None

    def _settle_lazy_inputs(self, keep=None):
        owner, self._lazy_owner = self._lazy_owner, None
        model = owner() if owner is not None else None
        if model is not None and model is not keep:
            model._materialise_inputs()

    def upload(self, x, y, kernel_id, phis_packed, n_basis, width):
        self._settle_lazy_inputs()
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(np.reshape(y, -1), dtype=np.float64)
        if x.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError("inputs must be [n, m] and data [n]")
        phis_packed = np.ascontiguousarray(phis_packed, dtype=np.float64)
        self._ck(self._lib.fokl_upload(self._h, _ptr(x), _ptr(y), x.shape[0], x.shape[1], int(kernel_id),
                                       _ptr(phis_packed), int(n_basis), int(width)))
        self.n, self.m = x.shape

    def stage_inputs(self, x, owner=None):
        """Raw inputs [n, m] (float64, C-contiguous) to the device; -> (column minima, column maxima)."""
        if x.dtype != np.float64 or x.ndim != 2 or not x.flags.c_contiguous:
            raise ValueError("stage_inputs: float64 C-contiguous [n, m] expected")
        self._settle_lazy_inputs(keep=owner)
        lows, highs = np.empty(x.shape[1]), np.empty(x.shape[1])
        self._ck(self._lib.fokl_stage_inputs(self._h, _ptr(x), x.shape[0], x.shape[1], _ptr(lows), _ptr(highs)))
        self._staged_shape = x.shape
        return lows, highs

    def upload_staged(self, y, kernel_id, phis_packed, n_basis, width, lows, spans, owner=None):
        """fokl_upload from the staged raw inputs, normalised on the device as (x - lows) / spans.  owner: the model whose
        ``inputs`` these are (asked to fetch them before the next dataset replaces them here)."""
        n, m = self._staged_shape
        y = np.ascontiguousarray(np.reshape(y, -1), dtype=np.float64)
        if y.shape[0] != n:
            raise ValueError("inputs must be [n, m] and data [n]")
        phis_packed = np.ascontiguousarray(phis_packed, dtype=np.float64)
        lows, spans = np.ascontiguousarray(lows, dtype=np.float64), np.ascontiguousarray(spans, dtype=np.float64)
        self._ck(self._lib.fokl_upload_staged(self._h, _ptr(y), n, m, int(kernel_id), _ptr(phis_packed), int(n_basis),
                                              int(width), _ptr(lows), _ptr(spans)))
        self.n, self.m = n, m
        if owner is not None:
            import weakref
            self._lazy_owner = weakref.ref(owner)

    def download_inputs(self):
        out = np.empty((getattr(self, 'n', 0) or 1, getattr(self, 'm', 0) or 1), dtype=np.float64)   # (no dataset: the call says so)
        self._ck(self._lib.fokl_download_inputs(self._h, _ptr(out)))
        return out

    def reserve_slots(self, n_slots):
        self._ck(self._lib.fokl_reserve_slots(self._h, int(n_slots)))

    @property
    def slot_capacity(self):
        return self._lib.fokl_slot_capacity(self._h)

    def build_terms(self, terms, slots):
        terms = np.ascontiguousarray(np.atleast_2d(terms), dtype=np.int32)
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        if terms.shape[0] != slots.shape[0]:
            raise ValueError("one slot per term")
        self._ck(self._lib.fokl_build_terms(self._h, _ptr(terms), terms.shape[0], _ptr(slots)))

    def build_terms_deriv(self, terms, slots, wrt_input, order, divisor):
        terms = np.ascontiguousarray(np.atleast_2d(terms), dtype=np.int32)
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        if terms.shape[0] != slots.shape[0]:
            raise ValueError("one slot per term")
        self._ck(self._lib.fokl_build_terms_deriv(self._h, _ptr(terms), terms.shape[0], _ptr(slots), int(wrt_input),
                                                  int(order), float(divisor)))

    def gram(self, row_slots, col_slots, path=0, allreduce=False):
        rs = np.ascontiguousarray(row_slots, dtype=np.int32)
        cs = np.ascontiguousarray(col_slots, dtype=np.int32)
        out = np.empty((rs.shape[0], cs.shape[0]), dtype=np.float64)
        self._ck(self._lib.fokl_gram(self._h, _ptr(rs), rs.shape[0], _ptr(cs), cs.shape[0], _ptr(out), int(path),
                                     int(bool(allreduce))))
        return out

    def gram_launch(self, row_slots, col_slots, allreduce=False):
        """fokl_gram_launch: -> shape of the block gram_fetch will return."""
        rs = np.ascontiguousarray(row_slots, dtype=np.int32)
        cs = np.ascontiguousarray(col_slots, dtype=np.int32)
        self._ck(self._lib.fokl_gram_launch(self._h, _ptr(rs), rs.shape[0], _ptr(cs), cs.shape[0],
                                            int(bool(allreduce))))
        return rs.shape[0], cs.shape[0]

    def gram_ready(self):
        """True once gram_fetch would not wait."""
        rc = self._lib.fokl_gram_ready(self._h)
        if rc < 0:
            self._ck(rc)
        return rc == 1

    def gram_fetch(self, shape):
        out = np.empty(shape, dtype=np.float64)
        self._ck(self._lib.fokl_gram_fetch(self._h, _ptr(out), out.size))
        return out

    def bic_resid(self, slots, betahat, allreduce=False):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        bh = np.ascontiguousarray(np.reshape(betahat, -1), dtype=np.float64)
        if s.shape[0] != bh.shape[0]:
            raise ValueError("one coefficient per column")
        out = np.empty(2, dtype=np.float64)
        self._ck(self._lib.fokl_bic_resid(self._h, _ptr(s), s.shape[0], _ptr(bh), _ptr(out), int(bool(allreduce))))
        return out[0], out[1]

    def bic_resid_launch(self, slots, betahat):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        bh = np.ascontiguousarray(np.reshape(betahat, -1), dtype=np.float64)
        if s.shape[0] != bh.shape[0]:
            raise ValueError("one coefficient per column")
        self._ck(self._lib.fokl_bic_resid_launch(self._h, _ptr(s), s.shape[0], _ptr(bh)))

    def bic_resid_terms_launch(self, terms, betahat):
        """Matrix-free residual pass of the model [intercept] + terms (rows of the interaction matrix)."""
        terms = np.ascontiguousarray(terms, dtype=np.int32)
        beta = np.ascontiguousarray(np.reshape(betahat, -1), dtype=np.float64)
        n_terms = terms.shape[0] if terms.size else 0
        if beta.shape[0] != n_terms + 1:
            raise ValueError("betahat needs one coefficient for the intercept and one per term")
        self._ck(self._lib.fokl_bic_resid_terms_launch(self._h, _ptr(terms), n_terms, _ptr(beta)))

    def bic_resid_fetch(self, allreduce=False):
        out = np.empty(2, dtype=np.float64)
        self._ck(self._lib.fokl_bic_resid_fetch(self._h, _ptr(out), int(bool(allreduce))))
        return out[0], out[1]

    def predict(self, slots, betas, cut=None):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        betas = np.ascontiguousarray(betas, dtype=np.float64)
        draws, nc = betas.shape
        if nc != s.shape[0]:
            raise ValueError("betas columns must match the slot list")
        mean = np.empty(self.n, dtype=np.float64)
        bounds = np.empty((self.n, 2), dtype=np.float64) if cut is not None else None
        self._ck(self._lib.fokl_predict(self._h, _ptr(s), nc, _ptr(betas), draws, int(cut or 0), _ptr(mean),
                                        _ptr(bounds)))
        return (mean, bounds) if cut is not None else mean

    def read_slot(self, slot, row0=0, nrows=None):
        nrows = self.n - row0 if nrows is None else nrows
        out = np.empty(int(nrows), dtype=np.float64)
        self._ck(self._lib.fokl_read_slot(self._h, int(slot), int(row0), int(nrows), _ptr(out)))
        return out

    def write_slot(self, slot, values, row0=0):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._ck(self._lib.fokl_write_slot(self._h, int(slot), int(row0), v.shape[0], _ptr(v)))

    def probe(self, what):
        """Sustained rate of a trivial kernel on this device: what = 0 HBM read, 1 HBM write, 2 one read : seven
        writes (bytes/s), 3 fp64 MFMA on register operands (flop/s) -- include/fokl_hip.h: fokl_probe."""
        rate = c_dbl(0)
        self._ck(self._lib.fokl_probe(self._h, int(what), ctypes.byref(rate)))
        return rate.value

    def timing_enable(self, on=True):
        self._ck(self._lib.fokl_timing_enable(self._h, int(bool(on))))

    def timing_reset(self):
        self._ck(self._lib.fokl_timing_reset(self._h))

    def timing_get(self, kernel_id):
        ms, launches, nbytes, flops, ideal = c_dbl(0), c_i64(0), c_dbl(0), c_dbl(0), c_dbl(0)
        self._ck(self._lib.fokl_timing_get(self._h, int(kernel_id), ctypes.byref(ms), ctypes.byref(launches),
                                           ctypes.byref(nbytes), ctypes.byref(flops), ctypes.byref(ideal)))
        return dict(ms=ms.value, launches=launches.value, bytes=nbytes.value, flops=flops.value,
                    ideal_ms=ideal.value)

    def timing_get_gram(self):
        """All Gram launches: the HBM-bound ones (K_GRAM) and the fp64-MFMA-bound ones (K_GRAM_MFMA) together."""
        a, b = self.timing_get(K_GRAM), self.timing_get(K_GRAM_MFMA)
        return {key: a[key] + b[key] for key in a}

    # -- RCCL ------------------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        _check(load().fokl_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        buf = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        self._ck(self._lib.fokl_comm_init(self._h, buf, int(rank), int(world)))

    def comm_destroy(self):
        self._ck(self._lib.fokl_comm_destroy(self._h))

    @staticmethod
    def comm_init_detached(device, unique_id, rank, world):
        """ncclCommInitRank without a context (fokl_comm_init_detached) -> opaque communicator for comm_adopt."""
        buf = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        comm = c_vp(0)
        _check(load().fokl_comm_init_detached(int(device), buf, int(rank), int(world), ctypes.byref(comm)))
        return comm

    def comm_adopt(self, comm, rank, world):
        self._ck(self._lib.fokl_comm_adopt(self._h, comm, int(rank), int(world)))

    @staticmethod
    def comm_release_detached(comm):
        load().fokl_comm_release_detached(comm)

    def allgather(self, values, world):
        v = np.ascontiguousarray(values, dtype=np.float64)
        out = np.empty((int(world), v.shape[0]), dtype=np.float64)
        self._ck(self._lib.fokl_comm_allgather_f64(self._h, _ptr(v), v.shape[0], _ptr(out)))
        return out

    def allreduce_sum(self, values):
        v = np.array(values, dtype=np.float64, copy=True)
        self._ck(self._lib.fokl_comm_allreduce_sum_f64(self._h, _ptr(v), v.size))
        return v
