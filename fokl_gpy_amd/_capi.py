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
K_BASIS, K_GRAM, K_RESID, K_PREDICT = 0, 1, 2, 3
SLOT_ONES, SLOT_Y, SLOT_FIRST_FREE = 0, 1, 2

c_int, c_i64, c_dbl, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/fokl_hip.h one to one (tests/test_capi_symbols.py checks it)
SIGNATURES = {
    'fokl_version': (c_int, []),
    'fokl_device_count': (c_int, [c_vp]),
    'fokl_ctx_create': (c_int, [c_int, c_vp]),
    'fokl_ctx_destroy': (None, [c_vp]),
    'fokl_last_error': (ctypes.c_char_p, [c_vp]),
    'fokl_sync': (c_int, [c_vp]),
    'fokl_upload': (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_int, c_int]),
    'fokl_reserve_slots': (c_int, [c_vp, c_int]),
    'fokl_slot_capacity': (c_int, [c_vp]),
    'fokl_rows': (c_i64, [c_vp]),
    'fokl_build_terms': (c_int, [c_vp, c_vp, c_int, c_vp]),
    'fokl_build_terms_deriv': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_dbl]),
    'fokl_gram': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int]),
    'fokl_bic_resid': (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_int]),
    'fokl_bic_resid_launch': (c_int, [c_vp, c_vp, c_int, c_vp]),
    'fokl_bic_resid_fetch': (c_int, [c_vp, c_vp, c_int]),
    'fokl_predict': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    'fokl_read_slot': (c_int, [c_vp, c_int, c_i64, c_i64, c_vp]),
    'fokl_write_slot': (c_int, [c_vp, c_int, c_i64, c_i64, c_vp]),
    'fokl_timing_enable': (c_int, [c_vp, c_int]),
    'fokl_timing_reset': (c_int, [c_vp]),
    'fokl_timing_get': (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    'fokl_gibbs_chain': (c_int, [c_vp, c_vp, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_int,
                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_noise_tape': (c_int, [c_int, c_int, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_gibbs_chain_from_tape': (c_int, [c_vp, c_vp, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl, c_int, c_vp, c_vp, c_vp,
                                           c_vp, c_vp, c_vp, c_vp, c_vp]),
    'fokl_rng_normals': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'fokl_rng_gammas': (c_int, [c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_i64, c_vp]),
    'fokl_comm_unique_id': (c_int, [c_vp]),
    'fokl_comm_init': (c_int, [c_vp, c_vp, c_int, c_int]),
    'fokl_comm_destroy': (c_int, [c_vp]),
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
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(arr):
    return arr.ctypes.data_as(c_vp) if arr is not None else c_vp(0)


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
    """The data-independent random numbers of one candidate's chain (include/fokl_hip.h: fokl_noise_tape).
    ``progress[0]`` counts the iterations recorded so far (-1 = the producer failed)."""
    __slots__ = ('p1', 'draws', 'normals', 'gam_sig', 'gam_tau', 'progress')

    def __init__(self, p1, draws):
        self.p1, self.draws = int(p1), int(draws)
        self.normals = np.empty((self.draws, self.p1), dtype=np.float64)
        self.gam_sig = np.empty(self.draws, dtype=np.float64)
        self.gam_tau = np.empty(self.draws, dtype=np.float64)
        self.progress = np.zeros(1, dtype=np.int32)


def record_noise_tape(tape, astar, atau_star, stream):
    """Fill ``tape`` from the stream (runs on the worker thread of engine.NoisePipeline; the GIL is released)."""
    try:
        _check(load().fokl_noise_tape(tape.p1, tape.draws, float(astar), float(atau_star), *stream.args(),
                                      _ptr(tape.normals), _ptr(tape.gam_sig), _ptr(tape.gam_tau),
                                      _ptr(tape.progress)))
    except BaseException:
        tape.progress[0] = -1
        raise
    return tape


def noise_tape(p1, draws, astar, atau_star, stream):
    return record_noise_tape(NoiseTape(p1, draws), astar, atau_star, stream)


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
                                             float(sigsqd0), float(tausqd0), tape.draws, _ptr(tape.normals),
                                             _ptr(tape.gam_sig), _ptr(tape.gam_tau), _ptr(w), _ptr(sigs), _ptr(taus),
                                             ctypes.byref(flag), _ptr(tape.progress) if follow else c_vp(0)))
    if want_sig_tau:
        return w, bool(flag.value), sigs, taus
    return w, bool(flag.value)


# ---------------------------------------------------------------------------------------------------------
# device context
# ---------------------------------------------------------------------------------------------------------

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

    def upload(self, x, y, kernel_id, phis_packed, n_basis, width):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(np.reshape(y, -1), dtype=np.float64)
        if x.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError("inputs must be [n, m] and data [n]")
        phis_packed = np.ascontiguousarray(phis_packed, dtype=np.float64)
        self._ck(self._lib.fokl_upload(self._h, _ptr(x), _ptr(y), x.shape[0], x.shape[1], int(kernel_id),
                                       _ptr(phis_packed), int(n_basis), int(width)))
        self.n, self.m = x.shape

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

    def timing_enable(self, on=True):
        self._ck(self._lib.fokl_timing_enable(self._h, int(bool(on))))

    def timing_reset(self):
        self._ck(self._lib.fokl_timing_reset(self._h))

    def timing_get(self, kernel_id):
        ms, launches, nbytes, flops = c_dbl(0), c_i64(0), c_dbl(0), c_dbl(0)
        self._ck(self._lib.fokl_timing_get(self._h, int(kernel_id), ctypes.byref(ms), ctypes.byref(launches),
                                           ctypes.byref(nbytes), ctypes.byref(flops)))
        return dict(ms=ms.value, launches=launches.value, bytes=nbytes.value, flops=flops.value)

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

    def allgather(self, values, world):
        v = np.ascontiguousarray(values, dtype=np.float64)
        out = np.empty((int(world), v.shape[0]), dtype=np.float64)
        self._ck(self._lib.fokl_comm_allgather_f64(self._h, _ptr(v), v.shape[0], _ptr(out)))
        return out

    def allreduce_sum(self, values):
        v = np.array(values, dtype=np.float64, copy=True)
        self._ck(self._lib.fokl_comm_allreduce_sum_f64(self._h, _ptr(v), v.size))
        return v
