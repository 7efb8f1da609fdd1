"""Shared test helpers: golden-fixture loading and a CPU stand-in for the device backend.

``OracleBackend`` implements the backend protocol of ``fokl_gpy_amd.engine`` with the oracle's column builder and
numpy so that the HOST logic (search driver, slot bookkeeping, Gram cache, sampler hand-off, class surface) can be
exercised by the ``-m "not gpu"`` suite.  It lives under tests/ on purpose: the product never routes through it.
"""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

from oracle import fokl_oracle as O  # noqa: E402
from fokl_gpy_amd import getKernels, _capi  # noqa: E402

FIT_CASES = ['bern_m1', 'bern_m3', 'bern_m3_gimmie_tol1', 'bern_m4_way3', 'bern_m6', 'bern_m8_capped',
             'testdata10_default', 'testdata10_changed', 'splines_m4', 'sigmoid_splines']


def host_fingerprint():
    """Behavioural fingerprint of the numerical stack underneath the reference: BLAS ``dot``, LAPACK ``eigh`` (values AND
    eigenvector signs), libm ``pow`` / ``log`` -- the third-party arithmetic of SURVEY 8(c).  Fixtures made by importing
    the reference store the fingerprint of the host that made them (``host_fingerprint`` in the .npz,
    tests/golden/make_golden.py); a test may ask for the reference's numbers bit for bit only where the two agree --
    the reference's own tests are same-host regressions too (/root/reference/test/test_FoKL.py:42-56).  Sizes cover the
    fixtures' (N up to 2 000 rows, up to 75 columns) so that a BLAS whose blocking depends on its thread count shows up."""
    import hashlib
    import scipy.linalg
    h = hashlib.sha256()
    rng = np.random.default_rng(20240229)
    for n, p in ((10, 6), (300, 24), (441, 50), (1500, 45), (2000, 75)):
        A = rng.random((n, p))
        A[:, 0] = 1.0
        y = rng.random((n, 1))
        G = np.dot(A.T, A)
        lam, Q = scipy.linalg.eigh(G)
        for part in (G, np.dot(A.T, y), lam, Q, np.dot(A, Q[:, -1])):
            h.update(np.ascontiguousarray(part).tobytes())
    x = rng.random(4096)
    for k in (2, 3, 7, 20):
        h.update((x ** k).tobytes())
        h.update(np.array([float(v) ** k for v in x[:256]]).tobytes())
    h.update(np.log(x).tobytes())
    return h.hexdigest()[:16]


_FINGERPRINT = []


def same_host_as(fixture):
    """True when this host reproduces the numerical stack of the host that generated ``fixture`` (an open .npz)."""
    if 'host_fingerprint' not in getattr(fixture, 'files', ()):
        return False
    if not _FINGERPRINT:
        _FINGERPRINT.append(host_fingerprint())
    return str(fixture['host_fingerprint']) == _FINGERPRINT[0]


def load_case(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    hy = dict(zip([str(k) for k in g['hyper_keys']], [float(v) for v in g['hyper_vals']]))
    for k in ('burnin', 'draws', 'tolerance'):
        if k in hy:
            hy[k] = int(hy[k])
    for k in ('gimmie', 'way3', 'aic'):
        if k in hy:
            hy[k] = bool(hy[k])
    for k in ('a', 'atau'):
        if k in hy and float(hy[k]).is_integer():
            hy[k] = int(hy[k])
    kernel_name = str(g['kernel'])
    if kernel_name == 'Cubic Splines':
        phis = getKernels.table_to_phis(np.load(os.path.join(GOLDEN, 'spline_phis.npz'))['table'])
        kid = O.KERNEL_SPLINES
    else:
        phis = getKernels.bernoulli()
        cap = int(g['phis_cap'])
        if cap > 0:
            phis = phis[:cap]
        kid = O.KERNEL_BERNOULLI
    return g, hy, kernel_name, kid, phis


def unpack_phis(packed, kernel_id, n_basis, width):
    if kernel_id == O.KERNEL_SPLINES:
        return getKernels.table_to_phis(np.asarray(packed).reshape(n_basis, 4, width))
    tab = np.asarray(packed).reshape(n_basis, width)
    return tuple(list(tab[i, :i + 2]) for i in range(n_basis))


class OracleBackend:
    """CPU stand-in for ``HipBackend`` built from the oracle (tests only)."""

    def __init__(self):
        self.cols = {}
        self.capacity = 0
        self.calls = dict(build=0, gram=0, resid=0)

    def upload(self, inputs, data, kernel_id, packed, n_basis, width):
        self.inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        self.kernel = kernel_id
        self.phis = unpack_phis(packed, kernel_id, n_basis, width)
        n = self.inputs.shape[0]
        self.cols = {0: np.ones(n), 1: np.asarray(data, dtype=np.float64).reshape(-1).copy()}
        if kernel_id == O.KERNEL_SPLINES:
            self.phind, self.xsm = O.inputs_to_phind(self.inputs, len(self.phis[0][0]))
        else:
            self.phind, self.xsm = None, self.inputs
        self.capacity = 16
        self.n = n

    def reserve_slots(self, count):
        self.capacity = max(self.capacity, count)

    def build_terms(self, terms, slots):
        self.calls['build'] += 1
        X = O.build_columns_c(self.xsm, self.phind, self.phis, self.kernel, np.atleast_2d(terms))
        for j, s in enumerate(slots):
            assert 2 <= s < self.capacity, "slot outside the reserved range"
            self.cols[int(s)] = X[:, j].copy()

    def build_terms_deriv(self, terms, slots, wrt_input, order, divisor):
        """Derivative columns with the oracle's scalar evaluate_basis at the twice-normalised coordinate (FR:778-782)."""
        terms = np.atleast_2d(terms)
        if self.kernel == O.KERNEL_SPLINES:
            X, phind = O.twice_normalised(self.inputs, len(self.phis[0][0]))
        else:
            X, phind = self.inputs, None
        for j, s in enumerate(slots):
            col = np.ones(self.n)
            for n in range(self.n):
                phi = 1
                for md in range(terms.shape[1]):
                    num = int(terms[j, md])
                    if not num:
                        continue
                    if self.kernel == O.KERNEL_SPLINES:
                        c = [self.phis[num - 1][k][int(phind[n, md])] for k in range(4)]
                    else:
                        c = self.phis[num - 1]
                    if md == wrt_input:
                        phi *= O.evaluate_basis(c, X[n, md], self.kernel, d=order) / divisor
                    else:
                        phi *= O.evaluate_basis(c, X[n, md], self.kernel)
                col[n] = phi
            self.cols[int(s)] = col

    def read_slot(self, slot):
        return self.cols[int(slot)].copy()

    def gram(self, row_slots, col_slots, allreduce=False):
        self.calls['gram'] += 1
        A = np.stack([self.cols[int(s)] for s in row_slots], axis=1)
        B = np.stack([self.cols[int(s)] for s in col_slots], axis=1)
        return A.T @ B

    def bic_resid(self, slots, betahat, allreduce=False):
        self.calls['resid'] += 1
        X = np.stack([self.cols[int(s)] for s in slots], axis=1)
        r = self.cols[1] - X @ np.reshape(betahat, -1)
        return float(np.sum(r)), float(np.sum(r * r))

    def predict(self, slots, betas, cut=None):
        X = np.stack([self.cols[int(s)] for s in slots], axis=1)
        mod = X @ np.asarray(betas).T
        mean = np.mean(mod, axis=1)
        if cut is None:
            return mean
        srt = np.sort(mod, axis=1)
        draws = mod.shape[1]
        return mean, np.stack([srt[:, cut], srt[:, draws - cut]], axis=1)

UNITS_PATH = os.path.join(GOLDEN, 'units.npz')


class StandInChainJob(_capi.DeviceChainJob):
    """DeviceChainJob whose chain runs on a host thread (tests of the search's device-chain logic without a GPU)."""
    __slots__ = ('_future', '_released', '_engine_ref')

    def __init__(self, engine, future, p1, draws, tape):
        super().__init__(None, -1, p1, draws, (tape,))
        self._future, self._released, self._engine_ref = future, False, engine

    def done(self):
        return self._released or self._future.done()

    def _result(self):
        if self._released:
            raise RuntimeError("device chain: released before anybody read its statistics")
        return self._future.result()

    def wait(self):
        w, flag, first = self._result()
        return w[first:].mean(axis=0) if w.shape[0] > first else np.zeros(w.shape[1]), np.array([int(flag)], dtype=np.int32)

    def fetch_w(self, out=None):
        return np.array(self._result()[0])

    def try_release(self):
        if self._released:
            return True
        if not self._future.done():
            return False
        self.release()
        return True

    def release(self):
        if not self._released:
            self._future.result()                              # the real engine waits until the tape has been read
            self._released = True
            self._engine_ref.alive -= 1
            self._future = None
            self.keep = None


class StandInChainEngine:
    """CPU stand-in for _capi.DeviceChainEngine (tests only; installed through host_pipeline._chain_engine_factory): the same
    interface, the chain computed by the host sampler on a worker thread after `delay` seconds -- late enough for the
    search to take its kill-test decisions from guesses, as it does with a real device."""
    wants_pinned_tapes = False

    def __init__(self, slots=64, delay=0.004):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(2)
        self.slots, self.delay, self.alive, self.issued, self.refused = slots, delay, 0, 0, 0

    def submit(self, lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape, stat_first=0, follow=True):
        if self.alive >= self.slots:
            self.refused += 1
            raise _capi.FoklNativeError(-3, 'stand-in engine: every slot holds a chain that was not released')
        lamb, qty = np.array(lamb, dtype=np.float64), np.array(qty, dtype=np.float64)

        def work():
            import time
            time.sleep(self.delay)
            if tape.finishing_requested:
                w, flag = _capi.gibbs_chain_from_finished_tape(lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape, follow=True)
            else:
                # a raw tape of the pool is walked first and expanded block by block afterwards: wait for the blocks, not
                # for the walk (what the real engine's dispatcher does)
                while True:
                    state = _capi.tape_ready(tape)
                    if state < 0:
                        raise _capi.FoklNativeError(-3, 'stand-in engine: the tape was sent back')
                    if state:
                        break
                    time.sleep(0.0005)
                w, flag = _capi.gibbs_chain_from_tape(lamb, qty, b, btau, dtd, sigsqd0, tausqd0, tape)
            return w, flag, int(stat_first)

        self.alive += 1
        self.issued += 1
        return StandInChainJob(self, self._pool.submit(work), lamb.shape[0], tape.draws, tape)

    def stats(self):
        return dict(dispatch_s=0.0, issued=self.issued, launches=self.issued)

    def close(self):
        self._pool.shutdown(wait=True)
