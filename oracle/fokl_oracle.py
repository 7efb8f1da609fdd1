"""
ORACLE -- test infrastructure, NOT the product.

CPU restatement (numpy + a small C helper, oracle_c.c) of the forward-variable-selection hot path of
FoKL-GPy's ``FoKLRoutines.FoKL.fit``.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product (``fokl_gpy_amd``) never does.

Parity pin: ``tests/golden/*.npz`` were produced by importing the real reference in the build container
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks every function here against
them (selected interaction matrix exactly, BIC trace and posterior draws to ~1e-12).

Every function cites the reference lines it restates; ``FR`` = /root/reference/src/FoKL/FoKLRoutines.py.

Third-party arithmetic the reference leans on and this oracle uses *as is* (it is the same dependency,
not a re-implementation): numpy's legacy global RandomState (``np.random.normal`` / ``np.random.gamma``,
MT19937 + polar Gaussian with cache + Marsaglia-Tsang gamma; call sites FR:1527, 1541, 1547),
``scipy.linalg.eigh`` (FR:1499) and BLAS ``dot`` (FR:1492-1494).
"""
import ctypes
import math
import os
import subprocess

import numpy as np
from scipy.linalg import eigh as _scipy_eigh

KERNEL_SPLINES = 0
KERNEL_BERNOULLI = 1
_HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------------------------------------
# eigen-decomposition with a deterministic sign convention
# ---------------------------------------------------------------------------------------------------------

def canonical_signs(Q):
    """Flip every eigenvector so that its largest-magnitude component is positive."""
    Q = np.array(Q, dtype=np.float64, copy=True)
    piv = np.argmax(np.abs(Q), axis=0)
    sgn = np.sign(Q[piv, np.arange(Q.shape[1])])
    sgn[sgn == 0] = 1.0
    return Q * sgn


def eigh_reference(A):
    """``scipy.linalg.eigh`` exactly as the reference calls it (FR:17, FR:1499)."""
    return _scipy_eigh(A)


def eigh_canonical(A):
    """Reference eigh followed by the sign canonicalisation shared by patched reference, oracle and product."""
    lam, Q = _scipy_eigh(A)
    return lam, canonical_signs(Q)


# ---------------------------------------------------------------------------------------------------------
# F1, F2: spline indexing and scalar basis evaluation
# ---------------------------------------------------------------------------------------------------------

def inputs_to_phind(inputs, l_phis):
    """F1 (FR:570-589): piece index and local coordinate of normalised inputs for the spline kernel."""
    phind = np.array(np.ceil(inputs * l_phis), dtype=np.uint16)
    if phind.ndim == 1:
        phind = phind[:, np.newaxis]
    phind = phind + (phind == 0)
    phind = phind - 1
    xsm = np.array(l_phis * inputs - phind, dtype=inputs.dtype)
    if np.max(phind) > 499 or np.min(phind) < 0:
        raise ValueError('Inputs are not normalized correctly')
    return phind, xsm


def evaluate_basis(c, x, kernel, d=0):
    """F2 (FR:834-847).  ``x`` must be a numpy float64 scalar so that ``**`` is libm pow as in the reference."""
    if kernel == KERNEL_SPLINES:
        if d == 0:
            return c[0] + c[1] * x + c[2] * (x ** 2) + c[3] * (x ** 3)
        if d == 1:
            return c[1] + 2 * c[2] * x + 3 * c[3] * (x ** 2)
        return 2 * c[2] + 6 * c[3] * x
    if d == 0:
        return c[0] + sum(c[k] * (x ** k) for k in range(1, len(c)))
    if d == 1:
        return c[1] + sum(k * c[k] * (x ** (k - 1)) for k in range(2, len(c)))
    return sum((k - 1) * k * c[k] * (x ** (k - 2)) for k in range(2, len(c)))


# ---------------------------------------------------------------------------------------------------------
# F4: term enumeration
# ---------------------------------------------------------------------------------------------------------

def distinct_arrangements(indvec):
    """
    F4 (FR:1350-1354, FR:1616): ``np.unique(perms(indvec), axis=0)`` == all distinct arrangements of the
    multiset ``indvec`` in ascending lexicographic order, as float64 rows.  Generated directly with the
    classic next-permutation step instead of enumerating M! tuples.
    """
    cur = sorted(float(v) for v in indvec)
    rows = [list(cur)]
    n = len(cur)
    while True:
        i = n - 2
        while i >= 0 and cur[i] >= cur[i + 1]:
            i -= 1
        if i < 0:
            break
        j = n - 1
        while cur[j] <= cur[i]:
            j -= 1
        cur[i], cur[j] = cur[j], cur[i]
        cur[i + 1:] = reversed(cur[i + 1:])
        rows.append(list(cur))
    return np.array(rows, dtype=np.float64)


def deal_indvec(ind, m, sett):
    """FR:1605-1613: spread ``ind`` units round-robin over the first ``sett`` slots of a length-m vector."""
    indvec = np.zeros(m)
    left = ind
    while left:
        for j in range(sett):
            indvec[j] += 1
            left -= 1
            if left == 0:
                break
    return indvec


# ---------------------------------------------------------------------------------------------------------
# F3: basis-matrix columns (faithful scalar path in Python; same arithmetic in C for anything bigger)
# ---------------------------------------------------------------------------------------------------------

def pack_phis(phis, kernel):
    if kernel == KERNEL_SPLINES:
        nb, npiece = len(phis), len(phis[0][0])
        out = np.empty((nb, 4, npiece))
        for i in range(nb):
            for k in range(4):
                out[i, k] = phis[i][k]
        return np.ascontiguousarray(out), nb, npiece
    nb = len(phis)
    width = max(len(p) for p in phis)
    out = np.zeros((nb, width))
    for i in range(nb):
        out[i, :len(phis[i])] = phis[i]
    return np.ascontiguousarray(out), nb, width


def build_columns_scalar(xsm, phind, phis, kernel, terms):
    """F3 (FR:1461-1485) with the reference's own per-element loop structure.  Slow: small cases only."""
    n, m = xsm.shape
    terms = np.atleast_2d(terms)
    out = np.zeros((n, terms.shape[0]))
    for i in range(n):
        for j in range(terms.shape[0]):
            phi = 1
            for k in range(m):
                num = terms[j][k]
                if num != 0:
                    nid = int(num - 1)
                    if kernel == KERNEL_SPLINES:
                        coeffs = [phis[nid][order][phind[i, k]] for order in range(4)]
                    else:
                        coeffs = phis[nid]
                    phi = phi * evaluate_basis(coeffs, xsm[i, k], kernel)
            out[i][j] = phi
    return out


_LIB = None


def build_c_helper(force=False):
    """Compile oracle_c.c -> oracle/liboracle_c.so (gcc, no FMA contraction) and return the path."""
    so = os.path.join(_HERE, 'liboracle_c.so')
    src = os.path.join(_HERE, 'oracle_c.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fPIC', '-shared', '-o', so, src, '-lm'])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c_helper())
        _LIB.oracle_build_columns.restype = None
        _LIB.oracle_gram.restype = None
        _LIB.oracle_inputs_to_phind.restype = None
    return _LIB


def build_columns_c(xsm, phind, phis, kernel, terms, threads=1):
    """F3 through oracle_c.c -- identical arithmetic (libm pow, no FMA) to ``build_columns_scalar``; returns [N, T].
    threads > 1 splits the ROWS over that many native calls (every element is computed by the same scalar code, so
    the result does not depend on the split); used by the golden generators for the full-size configurations."""
    lib = _lib()
    xsm = np.ascontiguousarray(xsm, dtype=np.float64)
    n, m = xsm.shape
    terms = np.ascontiguousarray(np.atleast_2d(terms), dtype=np.int32)
    t = terms.shape[0]
    packed, nb, width = pack_phis(phis, kernel)
    out = np.empty((t, n), dtype=np.float64)
    ph = np.ascontiguousarray(phind, dtype=np.uint16) if kernel == KERNEL_SPLINES else None

    def rows(lo, hi):
        ph_ptr = ctypes.c_void_p(ph.ctypes.data + 2 * lo * m) if ph is not None else ctypes.c_void_p(0)
        lib.oracle_build_columns(ctypes.c_void_p(xsm.ctypes.data + 8 * lo * m), ph_ptr, ctypes.c_int64(hi - lo),
                                 ctypes.c_int(m), ctypes.c_int(kernel), packed.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_int(nb), ctypes.c_int(width), terms.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_int(t), ctypes.c_void_p(out.ctypes.data + 8 * lo), ctypes.c_int64(n))

    threads = max(1, min(int(threads), n // 4096 or 1))
    if threads == 1:
        rows(0, n)
    else:
        from concurrent.futures import ThreadPoolExecutor
        cuts = [n * k // threads for k in range(threads + 1)]
        with ThreadPoolExecutor(threads) as ex:                      # ctypes releases the GIL during the call
            list(ex.map(lambda k: rows(cuts[k], cuts[k + 1]), range(threads)))
    return np.ascontiguousarray(out.T)


def build_columns_numpy(xsm, phind, phis, kernel, terms):
    """F3 vectorised over the rows: the same expressions as ``evaluate_basis`` (FR:836, FR:843) applied to whole input
    columns with numpy ufuncs (``x ** k`` -> np.power), factors multiplied in ascending input order starting from 1
    (FR:1466-1483).  This is the "fair" CPU baseline of SURVEY 8(d)(ii): what a NumPy user would write instead of the
    reference's per-element Python loops; values agree with the scalar path to the last bit or two of np.power."""
    xsm = np.asarray(xsm, dtype=np.float64)
    n, m = xsm.shape
    terms = np.atleast_2d(terms)
    out = np.empty((n, terms.shape[0]))
    cache = {}
    for j in range(terms.shape[0]):
        phi = None
        for k in range(m):
            num = int(terms[j][k])
            if num == 0:
                continue
            fac = cache.get((k, num))
            if fac is None:
                x = xsm[:, k]
                if kernel == KERNEL_SPLINES:
                    p = phind[:, k]
                    c = [np.asarray(phis[num - 1][order])[p] for order in range(4)]
                    fac = c[0] + c[1] * x + c[2] * (x ** 2) + c[3] * (x ** 3)
                else:
                    c = phis[num - 1]
                    fac = c[0] + sum(c[q] * (x ** q) for q in range(1, len(c)))
                cache[(k, num)] = fac
            phi = fac if phi is None else phi * fac
        out[:, j] = phi
    return out


# ---------------------------------------------------------------------------------------------------------
# G1-G4: one gibbs() call
# ---------------------------------------------------------------------------------------------------------

class GibbsResult:
    __slots__ = ('betas', 'sigs', 'taus', 'betahat', 'X', 'ev', 'XtX', 'Xty', 'lamb', 'Q')


def gibbs(inputs, data, phis, kernel, Xin, discmtx, a, b, atau, btau, draws, phind, xsm, sigsqd, tausqd, dtd,
          eigh=eigh_reference, build=build_columns_c, timing=None):
    """
    One model evaluation (FR:1396-1558): build the missing columns of X (F3), XtX / Xty (G1, FR:1492-1494),
    spectral least squares (G2, FR:1499-1505), ``draws`` Gibbs iterations (G3, FR:1519-1548) and the
    BIC (G4, FR:1551-1554).  Consumes numpy's global legacy RNG exactly like the reference.
    ``timing`` (optional dict, bench.py's CPU baseline): seconds spent in the parts whose cost grows with the number
    of rows (``rows_s``: X-build, XtX / Xty, residual pass) and in the parts that do not (``chain_s``: eigh + sampler).
    """
    import time as _time
    _t0 = _time.perf_counter()
    n_obs = inputs.shape[0]
    discmtx = np.atleast_2d(discmtx)
    mmtx = discmtx.shape[0]
    if np.size(Xin) == 0:
        Xin = np.ones((n_obs, 1))
    nxin = Xin.shape[1]
    if mmtx + 1 > nxin:
        new_cols = build(xsm, phind, phis, kernel, discmtx[nxin - 1:mmtx])
        X = np.concatenate([Xin, new_cols], axis=1)
    else:
        X = Xin

    XtX = np.transpose(X).dot(X)
    Xty = np.transpose(X).dot(data)

    _t1 = _time.perf_counter()
    lamb, Q = eigh(XtX)
    betahat = Q.dot(np.diag(1 / lamb)).dot(np.transpose(Q)).dot(Xty)

    n = len(data)
    astar = a + 1 + n / 2 + (mmtx + 1) / 2
    atau_star = atau + mmtx / 2

    betas = np.zeros((draws, mmtx + 1))
    sigs = np.zeros((draws, 1))
    taus = np.zeros((draws, 1))
    eye = np.identity(mmtx + 1)
    for k in range(draws):
        shifted = np.diag(lamb) + (1 / tausqd) * eye
        shifted_inv = np.diag(1 / np.diag(shifted))
        mun = Q.dot(shifted_inv).dot(np.transpose(Q)).dot(Xty)
        S = Q.dot(np.diag(np.diag(shifted_inv) ** (1 / 2)))
        vec = np.random.normal(loc=0, scale=1, size=(mmtx + 1, 1))
        betas[k][:] = np.transpose(mun + sigsqd ** (1 / 2) * S.dot(vec))
        bk = betas[k][:]
        bstar = b + 0.5 * (bk.dot(XtX.dot(np.transpose([bk]))) - 2 * bk.dot(Xty) + dtd
                           + bk.dot(np.transpose([bk])) / tausqd)
        if bstar < 0:
            sigsqd = math.nan
        else:
            sigsqd = 1 / np.random.gamma(astar, 1 / bstar)
        sigs[k] = sigsqd
        btau_star = (1 / (2 * sigsqd)) * (bk.dot(np.reshape(bk, (len(bk), 1)))) + btau
        tausqd = 1 / np.random.gamma(atau_star, 1 / btau_star)
        taus[k] = tausqd

    _t2 = _time.perf_counter()
    siglik = np.var(data - np.matmul(X, betahat))
    lik = -(n / 2) * np.log(siglik) - (n - 1) / 2
    ev = (mmtx + 1) * np.log(n) - 2 * np.max(lik)
    if timing is not None:
        timing['rows_s'] = timing.get('rows_s', 0.0) + (_t1 - _t0) + (_time.perf_counter() - _t2)
        timing['chain_s'] = timing.get('chain_s', 0.0) + (_t2 - _t1)

    res = GibbsResult()
    res.betas, res.sigs, res.taus, res.betahat = betas, sigs, taus, betahat
    res.X, res.ev, res.XtX, res.Xty, res.lamb, res.Q = X[:, :mmtx + 1], ev, XtX, Xty, lamb, Q
    return res


# ---------------------------------------------------------------------------------------------------------
# G5, G6, D0: the forward-selection driver
# ---------------------------------------------------------------------------------------------------------

DEFAULT_HYPERS = dict(a=4, b=None, atau=4, btau=None, tolerance=3, burnin=1000, draws=1000, gimmie=False,
                      way3=False, threshav=0.05, threshstda=0.5, threshstdb=2, aic=False)


def default_b_btau(data, a, atau, b=None, btau=None):
    """D0 (FR:1322-1348): data-driven defaults of the two inverse-gamma scale parameters."""
    sigmasq = np.var(data)
    data_mean = np.mean(data)
    if b is None:
        b = sigmasq * (a + 1)
    if btau is None:
        btau = (np.abs(data_mean) / sigmasq) * (atau + 1)
    return b, btau


def fit(inputs, data, phis, kernel, eigh=eigh_reference, build=build_columns_c, trace=None, **hypers):
    """
    Forward variable selection (FR:1350-1760) on already normalised ``inputs`` [N, M] and ``data`` [N, 1].

    Returns (betas[-draws:], mtx, evs).  ``trace`` (optional list) receives one dict per gibbs() call with
    the number of columns, how many were (re)built, the BIC and whether it was a kill test -- the
    "logical candidate-term" count of SURVEY 8(d) is ``sum(t['built'] for t in trace)`` -- and what the kill tests read
    from that call's chain (instrumentation only): ``b0`` = mean(betas[half0:, 0]), whose magnitude is the scale of
    FR:1671 once the call's model is the accepted one, and, for a sub-stage's own call, ``mean_abs`` / ``rel_std`` of its new
    terms in the order of the interaction matrix (betavs / betavs2, FR:1656-1658).
    """
    hp = dict(DEFAULT_HYPERS)
    for key, val in hypers.items():
        if key not in hp:
            raise ValueError(f"Unexpected keyword argument: '{key}'")
        hp[key] = val
    a, atau = hp['a'], hp['atau']
    b, btau = default_b_btau(data, a, atau, hp['b'], hp['btau'])
    tolerance, way3, aic = hp['tolerance'], hp['way3'], hp['aic']
    threshav, threshstda, threshstdb = hp['threshav'], hp['threshstda'], hp['threshstdb']
    draws = hp['burnin'] + hp['draws']

    if kernel == KERNEL_SPLINES:
        phind, xsm = inputs_to_phind(inputs, len(phis[0][0]))
    else:
        phind, xsm = None, inputs

    sigsqd0 = b / (1 + a)
    tausqd0 = btau / (1 + atau)
    dtd = np.transpose(data).dot(data)

    def run(Xin, discmtx, kill):
        nxin = 1 if np.size(Xin) == 0 else Xin.shape[1]
        res = gibbs(inputs, data, phis, kernel, Xin, discmtx, a, b, atau, btau, draws, phind, xsm,
                    sigsqd0, tausqd0, dtd, eigh=eigh, build=build)
        if trace is not None:
            trace.append(dict(cols=discmtx.shape[0] + 1, built=discmtx.shape[0] + 1 - nxin, ev=float(res.ev),
                              kill=kill, xtx=res.XtX if len(trace) < 12 else None,
                              b0=float(np.mean(res.betas[int(np.ceil(draws / 2)):draws, 0]))))
        return res

    n, m = inputs.shape
    damtx = np.zeros((0, m))
    evs = np.array([])
    X = []
    ind = 1
    greater = 0
    finished = False
    sett = 1 if m == 1 else (3 if way3 else 2)
    half1 = int(np.ceil((draws / 2) + 1))
    half0 = int(np.ceil(draws / 2))
    betas = mtx = beters = None

    while True:
        indvec = deal_indvec(ind, m, sett)
        while True:
            vecs = distinct_arrangements(indvec)
            vm = vecs.shape[0]
            damtx = np.append(damtx, vecs, axis=0)
            dam = damtx.shape[0]

            res = run(X, damtx, False)
            beters, xers, ev = res.betas, res.X, res.ev
            if aic:
                ev = ev + (2 - np.log(n)) * (dam + 1)

            new = slice(dam - vm + 1, dam + 1)
            mean_abs = np.abs(np.mean(beters[half1:draws, new], axis=0))
            rel_std = np.divide(np.std(np.array(beters[half1:draws, new]), axis=0),
                                np.abs(np.mean(beters[half0:draws, dam - vm + 1:dam + 2], axis=0)))
            ids = np.array(range(dam - vm + 2, dam + 2))
            if trace is not None:
                trace[-1].update(mean_abs=np.array(mean_abs, dtype=np.float64), rel_std=np.array(rel_std, dtype=np.float64))
            table = np.transpose(np.array([mean_abs, rel_std, ids]))
            if table.shape[1] > 0:
                table = table[np.argsort(table[:, 0])]

            killset = []
            evmin = ev
            for i in range(vm):
                if table[i, 1] > threshstdb or table[i, 1] > threshstda and table[i, 0] < threshav * \
                        np.mean(np.abs(np.mean(beters[half0:draws, 0]))):
                    killtest = np.append(killset, (table[i, 2] - 1))
                    if killtest.size > 1:
                        killtest[::-1].sort()
                    damtx_test = damtx
                    for k in range(np.size(killtest)):
                        damtx_test = np.delete(damtx_test, int(np.array(killtest[k]) - 1), 0)
                    damtest = damtx_test.shape[0]
                    res_t = run(X, damtx_test, True)
                    evtest = res_t.ev
                    if aic:
                        evtest = evtest + (2 - np.log(n)) * (damtest + 1)
                    if evtest < evmin:
                        killset = killtest
                        evmin = evtest
                        xers = res_t.X
                        beters = res_t.betas
            for k in range(np.size(killset)):
                damtx = np.delete(damtx, int(np.array(killset[k]) - 1), 0)

            ev = evmin
            X = xers

            if np.size(evs) > 0:
                if ev < np.min(evs):
                    betas, mtx, greater = beters, damtx, 1
                    evs = np.append(evs, ev)
                elif greater < tolerance:
                    greater += 1
                    evs = np.append(evs, ev)
                else:
                    finished = True
                    evs = np.append(evs, ev)
                    break
            else:
                greater += 1
                betas, mtx = beters, damtx
                evs = np.append(evs, ev)

            if m == 1:
                break
            elif way3:
                if indvec[1] > indvec[2]:
                    indvec[0] += 1
                    indvec[1] -= 1
                elif indvec[2]:
                    indvec[1] += 1
                    indvec[2] -= 1
                    if indvec[1] > indvec[0]:
                        indvec[0] += 1
                        indvec[1] -= 1
                else:
                    break
            elif indvec[1]:
                indvec[0] += 1
                indvec[1] -= 1
            else:
                break

        if finished:
            break
        ind += 1
        if ind > len(phis):
            break

    if hp['gimmie']:
        betas, mtx = beters, damtx
    return betas[-hp['draws']::, :], mtx, evs


# ---------------------------------------------------------------------------------------------------------
# N3: fitupdate, first call (no prior model): gibbs_Xin_update "case 1" + its driver loop
# ---------------------------------------------------------------------------------------------------------

def gibbs_update_case1(sigsqd0, inputs, data, phis, kernel, Xin, discmtx, a, b, atau, btau, phind, xsm, draws,
                       eigh=eigh_reference, build=build_columns_c):
    """gibbs_Xin_update without a prior model (FR:2010-2152): X-build as in gibbs(), then a sampler that starts from
    sigsqd0 / tausqd = 1 / sigsqd0, evaluates the log-likelihood of every draw and scores the model with the best one:
    ev = (mmtx + 1) log n - 2 max(lik).  Returns (betas, X, ev)."""
    n_obs = inputs.shape[0]
    discmtx = np.atleast_2d(discmtx)
    mmtx = discmtx.shape[0]
    if np.size(Xin) == 0:
        Xin = np.ones((n_obs, 1))
    nxin = Xin.shape[1]
    if mmtx + 1 > nxin:
        X = np.concatenate([Xin, build(xsm, phind, phis, kernel, discmtx[nxin - 1:mmtx])], axis=1)
    else:
        X = Xin

    tausqd = 1 / sigsqd0
    XtX = np.transpose(X).dot(X)
    Xty = np.transpose(X).dot(data)
    lamb, Q = eigh(XtX)
    betahat = Q.dot(np.diag(1 / lamb)).dot(np.transpose(Q)).dot(Xty)
    squerr = np.linalg.norm(data - X.dot(betahat)) ** 2
    astar = a + 1 + len(data) / 2 + (mmtx + 1) / 2
    atau_star = atau + mmtx / 2
    dtd = np.transpose(data).dot(data)

    betas = np.zeros((draws, mmtx + 1))
    sigsqd = sigsqd0
    lik = np.zeros((draws, 1))
    n = len(data)
    eye = np.identity(mmtx + 1)
    for k in range(draws):
        shifted_inv = np.diag(1 / np.diag(np.diag(lamb) + (1 / tausqd) * eye))
        mun = Q.dot(shifted_inv).dot(np.transpose(Q)).dot(Xty)
        S = Q.dot(np.diag(np.diag(shifted_inv) ** (1 / 2)))
        vec = np.random.normal(loc=0, scale=1, size=(mmtx + 1, 1))
        betas[k][:] = np.transpose(mun + sigsqd ** (1 / 2) * S.dot(vec))

        comp1 = -(n / 2) * np.log(sigsqd)
        comp2 = np.transpose(betahat) - betas[k][:]
        comp3 = betahat - np.reshape(betas[k][:], (len(betas[k][:]), 1))
        lik[k] = comp1 - (squerr + comp2.dot(XtX).dot(comp3)) / (2 * sigsqd)

        vecc = mun - np.reshape(betas[k][:], (len(betas[k][:]), 1))
        bstar = b + (0.5 * np.transpose(vecc)).dot((XtX + (1 / tausqd) * eye).dot(vecc)) + 0.5 * dtd \
            - 0.5 * np.transpose(mun).dot(Xty)
        if bstar < 0:
            sigsqd = math.nan
        else:
            sigsqd = 1 / np.random.gamma(astar, 1 / bstar)
        btau_star = (1 / (2 * sigsqd)) * (betas[k][:].dot(np.reshape(betas[k][:], (len(betas[k][:]), 1)))) + btau
        tausqd = 1 / np.random.gamma(atau_star, 1 / btau_star)

    ev = (mmtx + 1) * np.log(n) - 2 * np.max(lik)
    return betas, X[:, 0:mmtx + 1], ev


def fitupdate_first(inputs, data, phis, kernel, eigh=eigh_reference, build=build_columns_c, trace=None, sigsqd0=0.5,
                    **hypers):
    """
    ``FoKL.fit`` with ``update=True`` on a model that has not been built yet (FR:1365-1367 -> fitupdate, FR:1850-2583):
    sub-stages (ind - i, i), i = floor(ind / 2) .. 0, of 2-way terms only, every sub-stage adds all its terms (no kill
    tests), one gibbs_Xin_update per sub-stage, `ev == min(evs)` / `greater <= tolerance` stop rule (FR:2551-2566).
    ``relats_in`` never excludes anything in the variants the reference can run (probed: arrays with one row, lists of
    non-zero ints).  Returns (betas of the best model -- all burnin + draws rows --, mtx, evs, built).
    """
    hp = dict(DEFAULT_HYPERS)
    for key, val in hypers.items():
        if key not in hp:
            raise ValueError(f"Unexpected keyword argument: '{key}'")
        hp[key] = val
    a, atau = hp['a'], hp['atau']
    b, btau = default_b_btau(data, a, atau, hp['b'], hp['btau'])
    tolerance, aic = hp['tolerance'], hp['aic']
    draws = hp['burnin'] + hp['draws']
    if kernel == KERNEL_SPLINES:
        phind, xsm = inputs_to_phind(inputs, len(phis[0][0]))
    else:
        phind, xsm = None, inputs
    n, m = inputs.shape
    if m == 1:
        raise ValueError("not enough values to unpack (expected 2, got 0)")      # FR:2528 on the scalar damtx
    damtx = np.zeros((0, m))
    evs = []
    ind = 1
    greater = 0
    finished = False
    built = False
    X = []
    betas_best = mtx = betas = None
    while True:
        i_list = [0] if ind == 1 else np.arange(0, math.floor(ind / 2) + 0.1, 1)[::-1]
        for i in i_list:
            vecs = np.zeros(m)
            vecs[0] = ind - i
            vecs[1] = i
            vecs = distinct_arrangements(vecs)
            damtx = np.concatenate((damtx, vecs), axis=0)
            nxin = 1 if np.size(X) == 0 else X.shape[1]
            betas, X, ev = gibbs_update_case1(sigsqd0, inputs, data, phis, kernel, X, damtx, a, b, atau, btau, phind,
                                              xsm, draws, eigh=eigh, build=build)
            if aic:
                ev = ev + (2 - np.log(n)) * damtx.shape[0]
            if trace is not None:
                trace.append(dict(cols=damtx.shape[0] + 1, built=damtx.shape[0] + 1 - nxin, ev=float(ev), kill=False))
            evs = np.concatenate((evs, [ev])) if np.size(evs) else np.array([ev])
            if ev == np.min(evs):
                betas_best, mtx, greater = betas, damtx, 1
            elif greater <= tolerance:
                greater = greater + 1
            else:
                finished = True
                built = True
                break
        if finished:
            break
        ind = ind + 1
        if ind > len(phis):
            break
    if hp['gimmie']:
        betas_best, mtx = betas, damtx
    return betas_best, mtx, evs, built


# ---------------------------------------------------------------------------------------------------------
# N3: fitupdate on a built model: priors from the previous posterior (gibbs_Xin_update cases 2 and 3)
# ---------------------------------------------------------------------------------------------------------
# The reference mixes np.matrix and ndarray operands; the statements below keep its operand types (np.asmatrix where it
# has them) so that every product dispatches to the same BLAS call and every `*` means what it means there.

def gibbs_update_case2(sigsqd0, data, X, mu_old, Sigma_old, a, b, atau, btau, draws, eigh=eigh_reference):
    """Same number of terms as the prior model (FR:2153-2264): all coefficients keep the Gaussian prior
    N(mu_old, sigsqd tausqd Sigma_old); eigh and two inverses per Gibbs iteration."""
    mmtx = X.shape[1] - 1
    num_old_terms = np.shape(mu_old)[1]
    X_old = X
    tausqd = 1 / sigsqd0
    XotXo = np.asmatrix(np.transpose(X_old).dot(X_old))
    Xoty = np.transpose(X_old).dot(data)
    Sigma_old_inverse = np.linalg.inv(Sigma_old)
    astar = a + len(data) / 2 + (mmtx + 1) / 2
    atau_star = atau + (mmtx + 1) / 2
    yty = np.transpose(data).dot(data)
    ytXo = np.transpose(data).dot(X_old)
    betas_old = np.asmatrix(np.zeros((draws, num_old_terms)))
    sigsqd = sigsqd0
    lik = np.zeros((draws, 1))
    n = len(data)
    mu_old = mu_old.transpose()
    for k in range(draws):
        Sigma_old_post = np.linalg.inv(XotXo + (1 / tausqd) * Sigma_old_inverse)
        Lamb_old, Q_old = eigh(XotXo + (1 / tausqd) * Sigma_old_inverse)
        Lamb_tausqd_inv_old = 1 / Lamb_old
        mu_old_post = Sigma_old_post.dot(Xoty + (1 / tausqd * Sigma_old_inverse).dot(mu_old))
        S_old = Q_old.dot(np.diag(Lamb_tausqd_inv_old) ** (1 / 2))
        vec_old = np.random.normal(loc=0, scale=1, size=(num_old_terms, 1))
        betas_old[k][:] = np.transpose(mu_old_post + sigsqd ** (1 / 2) * S_old.dot(vec_old))
        bk = betas_old[k][:]
        comp1 = 0.5 * (yty - ytXo.dot(bk.transpose()))
        comp2 = 0.5 * (-bk.dot(Xoty) + bk.dot(XotXo).dot(bk.transpose()))
        comp3 = 0.5 * (1 / tausqd) * (bk.dot(Sigma_old_inverse).dot(bk.transpose())
                                      - bk.dot(Sigma_old_inverse).dot(mu_old))
        comp4 = 0.5 * (1 / tausqd) * (-np.transpose(mu_old).dot(Sigma_old_inverse).dot(bk.transpose())
                                      + np.transpose(mu_old).dot(Sigma_old_inverse).dot(mu_old))
        bstar = comp1 + comp2 + comp3 + comp4 + b
        if bstar < 0:
            sigsqd = math.nan
        else:
            sigsqd = 1 / np.random.gamma(astar, 1 / bstar)
        comp1 = 0.5 * (1 / sigsqd) * (bk.dot(Sigma_old_inverse).dot(bk.transpose())
                                      - bk.dot(Sigma_old_inverse).dot(mu_old))
        comp2 = 0.5 * (1 / sigsqd) * (-np.transpose(mu_old).dot(Sigma_old_inverse).dot(bk.transpose())
                                      + np.transpose(mu_old).dot(Sigma_old_inverse).dot(mu_old))
        btau_star = comp1 + comp2 + btau
        tausqd = 1 / np.random.gamma(atau_star, 1 / btau_star)
        comp1 = -(n / 2) * np.log(sigsqd)
        comp2 = yty - ytXo.dot(bk.transpose())
        comp3 = -bk.dot(Xoty) + bk.dot(XotXo).dot(bk.transpose())
        lik[k] = comp1 - 0.5 / sigsqd * (comp2 + comp3)
    ev = (mmtx + 1) * np.log(n) - 2 * max(lik)
    return betas_old, X[:, 0:mmtx + 1], ev


def gibbs_update_case3(sigsqd0, data, X, mu_old, Sigma_old, a, b, atau, btau, draws, eigh=eigh_reference):
    """More terms than the prior model (FR:2266-2425): the first `num_old_terms` coefficients keep the prior
    N(mu_old, sigsqd Sigma_old) (no tausqd there), the new ones the usual N(0, sigsqd tausqd); blocked Gibbs."""
    mmtx = X.shape[1] - 1
    num_old_terms = np.shape(mu_old)[1]
    length_old = num_old_terms
    length_new = mmtx - num_old_terms + 1
    X_old = X[:, 0:length_old]
    X_new = X[:, length_old:length_old + length_new]
    tausqd = 1 / sigsqd0
    XotXo = np.asmatrix(np.transpose(X_old).dot(X_old))
    Xoty = np.transpose(X_old).dot(data)
    Sigma_old_inverse = np.linalg.inv(Sigma_old)
    Sigma_old_post = np.linalg.inv(XotXo + Sigma_old_inverse)
    Lamb_old, Q_old = eigh(XotXo + Sigma_old_inverse)
    XntXn = np.transpose(X_new).dot(X_new)
    Xnty = np.transpose(X_new).dot(data)
    Lamb_new, Q_new = eigh(XntXn)
    XotXn = np.transpose(X_old).dot(X_new)
    XntXo = np.transpose(X_new).dot(X_old)
    Lamb_tausqd_inv_old = np.diag(np.linalg.inv(np.diag(Lamb_old)))
    astar = a + len(data) / 2 + (mmtx + 1) / 2
    atau_star = atau + length_new / 2
    yty = np.transpose(data).dot(data)
    ytXo = np.transpose(data).dot(X_old)
    ytXn = np.transpose(data).dot(X_new)
    betas_old = np.asmatrix(np.zeros((draws, num_old_terms)))
    betas_new = np.asmatrix(np.zeros((draws, mmtx - num_old_terms + 1)))
    sigsqd = sigsqd0
    lik = np.zeros((draws, 1))
    n = len(data)
    mu_old = mu_old.transpose()
    for k in range(draws):
        mu_old_post = Sigma_old_post.dot(Xoty - XotXn.dot(betas_new[k - 1].transpose()) + Sigma_old_inverse.dot(mu_old))
        S_old = Q_old.dot(np.diag(Lamb_tausqd_inv_old) ** (1 / 2))
        vec_old = np.random.normal(loc=0, scale=1, size=(length_old, 1))
        betas_old[k][:] = np.transpose(mu_old_post + sigsqd ** (1 / 2) * S_old.dot(vec_old))
        bo = betas_old[k][:]
        Lamb_tausqd_inv_new = np.diag(np.linalg.inv(np.diag(Lamb_new) + (1 / tausqd) * np.identity(length_new)))
        mu_new_post = np.linalg.inv(XntXn + (1 / tausqd) * np.identity(length_new)).dot(
            Xnty - XntXo.dot(betas_old[k].transpose()))
        S_new = Q_new.dot(np.diag(Lamb_tausqd_inv_new) ** (1 / 2))
        vec_new = np.random.normal(loc=0, scale=1, size=(length_new, 1))
        betas_new[k][:] = np.transpose(mu_new_post + sigsqd ** (1 / 2) * S_new.dot(vec_new))
        bn = betas_new[k][:]
        comp1 = 0.5 * (yty - ytXo.dot(bo.transpose()) - ytXn.dot(bn.transpose()))
        comp2 = 0.5 * (-bo.dot(Xoty) + bo.dot(XotXo).dot(bo.transpose()) + bo.dot(XotXn).dot(bn.transpose()))
        comp3 = 0.5 * (-bn.dot(Xnty) + bn.dot(XntXo).dot(bo.transpose()) + bn.dot(XntXn).dot(bn.transpose()))
        comp4 = 0.5 / tausqd * (bn.dot(bn.transpose()))
        comp5 = 0.5 * (bo.dot(Sigma_old_inverse).dot(bo.transpose()) - bo.dot(Sigma_old_inverse).dot(mu_old))
        comp6 = 0.5 * (-np.transpose(mu_old).dot(Sigma_old_inverse).dot(bo.transpose())
                       + np.transpose(mu_old).dot(Sigma_old_inverse).dot(mu_old))
        bstar = comp1 + comp2 + comp3 + comp4 + comp5 + comp6 + b
        if bstar < 0:
            sigsqd = math.nan
        else:
            sigsqd = 1 / np.random.gamma(astar, 1 / bstar)
        btau_star = (1 / (2 * sigsqd)) * (bn.dot(bn.transpose())) + btau
        tausqd = 1 / np.random.gamma(atau_star, 1 / btau_star)
        comp1 = -(n / 2) * np.log(sigsqd)
        comp2 = yty - ytXo.dot(bo.transpose()) - ytXn.dot(bn.transpose())
        comp3 = -bo.dot(Xoty) + bo.dot(XotXo).dot(bo.transpose()) + bo.dot(XotXn).dot(bn.transpose())
        comp4 = -bn.dot(Xnty) + bn.dot(XntXo).dot(bo.transpose()) + bn.dot(XntXn).dot(bn.transpose())
        lik[k] = comp1 - 0.5 / sigsqd * (comp2 + comp3 + comp4)
    ev = (mmtx + 1) * np.log(n) - 2 * max(lik)
    return np.concatenate((betas_old, betas_new), axis=1), X[:, 0:mmtx + 1], ev


def fitupdate_next(inputs, data, phis, kernel, betas_prev, burn=500, eigh=eigh_reference, build=build_columns_c,
                   sigsqd0=0.5, **hypers):
    """``fit(update=True)`` on a model that HAS been built (FR:1939-1943, 2473-2583): prior mean / covariance from the
    previous draws ``betas_prev[burn:-1]``; sub-stages are skipped until the interaction matrix is as long as the prior
    model (FR:2530), then case 2 ("same") once and case 3 ("new") for every longer model.  Returns (betas, mtx, evs,
    built) with evs as the reference leaves it: a [k, 1] array."""
    hp = dict(DEFAULT_HYPERS)
    for key, val in hypers.items():
        if key not in hp:
            raise ValueError(f"Unexpected keyword argument: '{key}'")
        hp[key] = val
    a, atau = hp['a'], hp['atau']
    b, btau = default_b_btau(data, a, atau, hp['b'], hp['btau'])
    tolerance, aic = hp['tolerance'], hp['aic']
    draws = hp['burnin'] + hp['draws']
    mu_old = np.asmatrix(np.mean(betas_prev[burn:-1], axis=0))
    sigma_old = np.cov(betas_prev[burn:-1].transpose())
    num_old_terms = np.shape(mu_old)[1]
    if kernel == KERNEL_SPLINES:
        phind, xsm = inputs_to_phind(inputs, len(phis[0][0]))
    else:
        phind, xsm = None, inputs
    n, m = inputs.shape
    damtx = np.zeros((0, m))
    evs = []
    ind = 1
    greater = 0
    finished = False
    built = True
    X = None
    betas_best = mtx = betas = None
    while True:
        i_list = [0] if ind == 1 else np.arange(0, math.floor(ind / 2) + 0.1, 1)[::-1]
        for i in i_list:
            vecs = np.zeros(m)
            vecs[0] = ind - i
            vecs[1] = i
            damtx = np.concatenate((damtx, distinct_arrangements(vecs)), axis=0)
            if num_old_terms - 1 <= damtx.shape[0]:
                have = 1 if X is None else X.shape[1]
                Xin = np.ones((n, 1)) if X is None else X
                if damtx.shape[0] + 1 > have:
                    Xin = np.concatenate([Xin, build(xsm, phind, phis, kernel, damtx[have - 1:])], axis=1)
                if num_old_terms == damtx.shape[0] + 1:
                    betas, X, ev = gibbs_update_case2(sigsqd0, data, Xin, mu_old, sigma_old, a, b, atau, btau, draws,
                                                      eigh=eigh)
                else:
                    betas, X, ev = gibbs_update_case3(sigsqd0, data, Xin, mu_old, sigma_old, a, b, atau, btau, draws,
                                                      eigh=eigh)
                if aic:
                    ev = ev + (2 - np.log(n)) * damtx.shape[0]
                evs = np.concatenate((evs, [ev])) if np.size(evs) else [ev]
                if ev == np.min(evs):
                    betas_best, mtx, greater = betas, damtx, 1
                elif greater <= tolerance:
                    greater = greater + 1
                else:
                    finished = True
                    break
        if finished:
            break
        ind = ind + 1
        if ind > len(phis):
            break
    if hp['gimmie']:
        betas_best, mtx = betas, damtx
    return betas_best, mtx, np.array(evs), built


# ---------------------------------------------------------------------------------------------------------
# evaluate / coverage3 numerics (kept class surface; FR:929-978, FR:1193)
# ---------------------------------------------------------------------------------------------------------

def evaluate(inputs, betas, mtx, phis, kernel, draws, setnos, return_bounds=False, build=build_columns_c):
    """Posterior-mean prediction (FR:941-978) for normalised ``inputs`` with an explicit draw selection ``setnos``."""
    inputs = np.asarray(inputs)
    n = inputs.shape[0]
    if kernel == KERNEL_SPLINES:
        phind, xsm = inputs_to_phind(inputs, len(phis[0][0]))
    else:
        phind, xsm = None, inputs
    X = np.concatenate([np.ones((n, 1)), build(xsm, phind, phis, kernel, np.atleast_2d(mtx))], axis=1)
    modells = np.zeros((n, draws))
    for i in range(draws):
        modells[:, i] = np.transpose(np.matmul(X, np.transpose(np.array(betas[setnos[i], :]))))
    mean = np.mean(modells, 1)
    if not return_bounds:
        return mean
    bounds = np.zeros((n, 2))
    cut = int(np.floor(draws * 0.025) + 1)
    for i in range(n):
        drawset = np.sort(modells[i, :])
        bounds[i, 0] = drawset[cut]
        bounds[i, 1] = drawset[draws - cut]
    return mean, bounds


def coverage_rmse(mean, data):
    """FR:1193 computes ``sqrt(mean(mean - data) ** 2)`` with an (n,) - (n,1) broadcast; its value is
    ``|mean(mean) - mean(data)|`` up to rounding.  Evaluated here without the n x n temporary."""
    return np.sqrt((np.mean(mean) - np.mean(data)) ** 2)


def twice_normalised(inputs, l_phis):
    """The other output of _inputs_to_phind (FR:584-586): X = (inputs - (phind - 1) r) / r with the 1-based phind."""
    phind = np.array(np.ceil(inputs * l_phis), dtype=np.uint16)
    if phind.ndim == 1:
        phind = phind[:, np.newaxis]
    phind = phind + (phind == 0)
    r = 1 / l_phis
    xmin = np.array((phind - 1) * r, dtype=inputs.dtype)
    return (inputs - xmin) / r, phind - 1


def bss_derivatives(inputs, betas, mtx, phis, kernel, minmax, d1, d2, draws):
    """
    N2 (FR:735-791): first / second partial derivatives of the fitted model, per draw.  ``d1`` / ``d2`` are boolean
    masks over the inputs.  Returns dy [N, M, 2, draws] (the array the reference holds at FR:791, before its
    averaging / reshaping tail).  Same loop order and arithmetic as the reference; small cases only.
    """
    inputs = np.asarray(inputs, dtype=np.float64)
    mtx = np.atleast_2d(np.asarray(mtx))
    N, M = inputs.shape
    B = mtx.shape[0]
    span_m = [minmax[m][1] - minmax[m][0] for m in range(M)]
    if kernel == KERNEL_SPLINES:
        X, phind = twice_normalised(inputs, len(phis[0][0]))
        L_phis = len(phis[0][0])
    else:
        X, phind, L_phis = inputs, None, 1
    derv = [np.asarray(d1, dtype=bool), np.asarray(d2, dtype=bool)]
    dy = np.zeros([draws, N, M, 2])
    basis_nm = np.zeros([N, M, B])
    for n in range(N):
        for m in range(M):
            for di in (0, 1):
                if not derv[di][m]:
                    continue
                span_L = span_m[m] / L_phis
                span_L = [1, span_L, span_L ** 2]
                for b in range(B):
                    phi = 1
                    for md in range(M):
                        num = int(mtx[b, md])
                        derp = di + 1 if md == m else 0
                        if num:
                            if kernel == KERNEL_SPLINES:
                                c = [phis[num - 1][k][int(phind[n, md])] for k in range(4)]
                            else:
                                c = phis[num - 1]
                            if derp == 0:
                                if basis_nm[n, md, b] == 0:
                                    basis_nm[n, md, b] = evaluate_basis(c, X[n, md], kernel)
                                phi *= basis_nm[n, md, b]
                            else:
                                phi *= evaluate_basis(c, X[n, md], kernel, d=derp) / span_L[derp]
                        elif derp:
                            phi = 0
                            break
                    dy[:, n, m, di] = dy[:, n, m, di] + betas[-draws:, b + 1] * phi
    return np.transpose(dy, (1, 2, 3, 0))
