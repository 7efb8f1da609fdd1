"""
N3 of SURVEY 8(f): sequential updating -- ``fit(update=True)`` / ``fitupdate`` (FoKLRoutines.py 1850-2583), first and later
calls (round 3: split out of engine.py).  Same device calls as the search (K1, K2, K3), the reference's own host samplers.
"""
import math
import os
import time

import numpy as np

from . import _capi
from .engine import (SLOT_ONES, SLOT_Y, SLOT_FIRST_FREE, ForwardSelection, SlotPool, distinct_arrangements,
                     eigh_canonical)

# ---------------------------------------------------------------------------------------------------------
# N3: sequential updating, first call (fitupdate without a prior model)
# ---------------------------------------------------------------------------------------------------------

def update_substage_patterns(m, n_phis):
    """(ind, indvec) of the sub-stages of fitupdate in the reference's order (FR:2480-2496, 2574-2578): for every total
    order ind the 2-way patterns (ind - i, i), i = floor(ind / 2) .. 0."""
    ind = 1
    while ind <= n_phis:
        for i in ([0] if ind == 1 else range(ind // 2, -1, -1)):
            vec = [0] * m
            vec[0], vec[1] = ind - i, i
            yield ind, vec
        ind += 1


def fit_update_first(backend, n, m, n_phis, a, b, atau, btau, tolerance, draws_total, gimmie, aic, sigsqd0, stream,
                     console=False):
    """
    ``fit(update=True)`` on a model that has not been built (FR:1850-2583 with `mu_old` empty: gibbs_Xin_update
    "case 1", FR:2060-2152, under the driver loop FR:2473-2583) on the device backend.

    Every sub-stage appends all distinct arrangements of its pattern (no kill tests), so the design only grows:
    K1 builds the new columns once, K2 extends the Gram by the new block, K3 gives the squared error of the
    least-squares fit (`squerr`, FR:2083), and the sampler -- same recursion and same use of the random stream as
    FR:1519-1548, started from sigsqd0 and tausqd = 1 / sigsqd0 -- runs in the eigenbasis (fokl_gibbs_chain).  The
    model is scored with the best log-likelihood among its draws (FR:2113-2118, 2146): in the eigenbasis
    (betahat - beta_k)' XtX (betahat - beta_k) = sum_i lamb_i (qty_i / lamb_i - w_ki)^2.
    Returns (betas [draws_total, P + 1] of the best model, mtx, evs, built, stats).
    """
    if m == 1:
        raise ValueError("not enough values to unpack (expected 2, got 0)")     # as the reference, FR:2528
    pool = SlotPool(backend)
    gram = np.array(backend.gram([SLOT_ONES, SLOT_Y], [SLOT_ONES, SLOT_Y]), dtype=np.float64)
    model_slots = []
    damtx = np.zeros((0, m))
    evs = np.array([])
    greater = 0
    built = False
    best = mtx = last = None
    stats = dict(gibbs_calls=0, kill_tests=0, terms_logical=0, terms_physical=0, substages=0)
    trace = []
    for ind, indvec in update_substage_patterns(m, n_phis):
        vecs = distinct_arrangements(indvec)
        new_slots = pool.take(vecs.shape[0])
        backend.build_terms(vecs.astype(np.int32), new_slots)
        n_prev = 1 + len(model_slots)
        block = backend.gram(new_slots, [SLOT_ONES] + model_slots + new_slots + [SLOT_Y])
        gram = ForwardSelection._extend_gram(gram, list(range(n_prev)), block, list(range(n_prev)), n_prev)
        model_slots = model_slots + new_slots
        damtx = np.append(damtx, vecs, axis=0)
        p1 = 1 + len(model_slots)
        XtX, Xty, dtd = gram[:p1, :p1], gram[:p1, p1], gram[p1, p1]
        lamb, Q = eigh_canonical(XtX)
        qty = Q.T @ Xty
        betahat = Q @ (qty / lamb)                                   # FR:2079-2081
        _, squerr = backend.bic_resid([SLOT_ONES] + model_slots, betahat)   # ||y - X betahat||^2, FR:2083
        astar = a + 1 + n / 2 + p1 / 2                               # FR:2085
        atau_star = atau + (p1 - 1) / 2                              # FR:2086
        w, sigs, _ = _capi.gibbs_chain(lamb, qty, astar, atau_star, b, btau, dtd, sigsqd0, 1 / sigsqd0, draws_total,
                                       stream, want_sig_tau=True)
        sig_used = np.concatenate([[sigsqd0], sigs[:-1]])            # iteration k looks at the sigsqd it starts with
        quad = np.sum(lamb * (qty / lamb - w) ** 2, axis=1)
        lik = -(n / 2) * np.log(sig_used) - (squerr + quad) / (2 * sig_used)     # FR:2113-2118
        ev = p1 * math.log(n) - 2 * np.max(lik)                      # FR:2146
        if aic:
            ev = ev + (2 - math.log(n)) * damtx.shape[0]             # FR:2541-2547 (dam, not dam + 1)
        stats['gibbs_calls'] += 1
        stats['terms_logical'] += vecs.shape[0]
        stats['terms_physical'] += vecs.shape[0]
        stats['substages'] += 1
        trace.append(dict(cols=p1, built=vecs.shape[0], ev=float(ev), kill=False))
        if console:
            print(ind, ev)
        last = (w, Q, damtx)
        evs = np.append(evs, ev)
        if ev == np.min(evs):                                        # FR:2556-2566
            best, mtx, greater = (w, Q), damtx, 1
        elif greater <= tolerance:
            greater += 1
        else:
            built = True
            break
    if gimmie:
        best, mtx = last[:2], last[2]
    betas = best[0] @ best[1].T
    return betas, np.array(mtx, dtype=np.float64), evs, built, stats, trace


# ---------------------------------------------------------------------------------------------------------
# N3: sequential updating of a built model (priors from the previous posterior)
# ---------------------------------------------------------------------------------------------------------
# The N-dependent half -- the basis columns of the new batch and their Gram matrix -- runs on the device as everywhere
# else (K1 + incremental K2); what is left is algebra on (P + 1) x (P + 1) matrices with a dense prior precision, which
# does not diagonalise once and for all (case 2 re-factorises in every Gibbs iteration, FR:2206-2212), so the samplers
# below stay on the host and draw from numpy's global generator directly, call for call as the reference does.

def _update_sampler_same(G, p, mu, Sinv, a, b, atau, btau, sigsqd0, n, draws):
    """All P = p coefficients keep the prior N(mu, sigsqd tausqd Sigma_old) (gibbs_Xin_update case 2, FR:2153-2264).
    G: Gram of [X | y].  Returns (betas [draws, p], ev)."""
    XtX, Xty, yty = G[:p, :p], G[:p, p:p + 1], G[p, p]
    ytX = Xty.T
    astar = a + n / 2 + p / 2                                        # FR:2178 (no "+ 1" in the update cases)
    atau_star = atau + p / 2                                         # FR:2179
    betas = np.zeros((draws, p))
    lik = np.zeros(draws)
    sigsqd, tausqd = sigsqd0, 1 / sigsqd0
    mu_prec = Sinv @ mu                                              # Sigma_old^-1 mu_old
    mu_quad = float(mu.T @ mu_prec)
    for k in range(draws):
        prec = XtX + (1 / tausqd) * Sinv                             # FR:2202
        cov = np.linalg.inv(prec)
        lam, Q = eigh_canonical(prec)                                # FR:2206
        mean = cov @ (Xty + (1 / tausqd * Sinv) @ mu)                # FR:2212-2213
        S = Q @ (np.diag(1 / lam) ** (1 / 2))
        vec = np.random.normal(loc=0, scale=1, size=(p, 1))
        bk = (mean + sigsqd ** (1 / 2) * (S @ vec)).T                # [1, p]
        betas[k] = bk
        fit_quad = float(-(bk @ Xty) + bk @ XtX @ bk.T)              # -b'X'y + b'X'X b
        data_part = float(yty - ytX @ bk.T)                          # y'y - y'X b
        prior_quad = float(bk @ Sinv @ bk.T - bk @ mu_prec) + float(-(mu.T @ Sinv @ bk.T) + mu_quad)
        bstar = 0.5 * data_part + 0.5 * fit_quad + 0.5 * (1 / tausqd) * prior_quad + b        # FR:2222-2231
        sigsqd = math.nan if bstar < 0 else 1 / np.random.gamma(astar, 1 / bstar)
        btau_star = 0.5 * (1 / sigsqd) * prior_quad + btau           # FR:2242-2248
        tausqd = 1 / np.random.gamma(atau_star, 1 / btau_star)
        lik[k] = -(n / 2) * np.log(sigsqd) - 0.5 / sigsqd * (data_part + fit_quad)            # FR:2254-2258
    return betas, p * math.log(n) - 2 * np.max(lik)


def _update_sampler_grown(G, p_old, p, mu, Sinv, a, b, atau, btau, sigsqd0, n, draws):
    """The first p_old coefficients keep the prior N(mu, sigsqd Sigma_old), the p - p_old new ones N(0, sigsqd tausqd)
    (gibbs_Xin_update case 3, FR:2266-2425): blocked Gibbs, old block first.  Returns (betas [draws, p], ev)."""
    q = p - p_old
    XoXo, XoXn, XnXn = G[:p_old, :p_old], G[:p_old, p_old:p], G[p_old:p, p_old:p]
    Xoy, Xny, yty = G[:p_old, p:p + 1], G[p_old:p, p:p + 1], G[p, p]
    XnXo = XoXn.T
    cov_old = np.linalg.inv(XoXo + Sinv)                             # FR:2289-2290: no tausqd on the old block
    lam_old, Q_old = eigh_canonical(XoXo + Sinv)
    lam_new, Q_new = eigh_canonical(XnXn)
    S_old = Q_old @ (np.diag(np.diag(np.linalg.inv(np.diag(lam_old)))) ** (1 / 2))
    astar = a + n / 2 + p / 2                                        # FR:2334
    atau_star = atau + q / 2                                         # FR:2335
    betas_old, betas_new = np.zeros((draws, p_old)), np.zeros((draws, q))
    lik = np.zeros(draws)
    sigsqd, tausqd = sigsqd0, 1 / sigsqd0
    mu_prec = Sinv @ mu
    mu_quad = float(mu.T @ mu_prec)
    eye = np.identity(q)
    for k in range(draws):
        prev_new = betas_new[k - 1:k].T if k else betas_new[-1:].T   # the reference reads row k - 1 (row -1 at k = 0)
        mean_old = cov_old @ (Xoy - XoXn @ prev_new + mu_prec)       # FR:2357-2358
        vec_old = np.random.normal(loc=0, scale=1, size=(p_old, 1))
        bo = (mean_old + sigsqd ** (1 / 2) * (S_old @ vec_old)).T
        betas_old[k] = bo
        shrink = np.diag(np.linalg.inv(np.diag(lam_new) + (1 / tausqd) * eye))               # FR:2366-2367
        mean_new = np.linalg.inv(XnXn + (1 / tausqd) * eye) @ (Xny - XnXo @ bo.T)            # FR:2370-2372
        S_new = Q_new @ (np.diag(shrink) ** (1 / 2))
        vec_new = np.random.normal(loc=0, scale=1, size=(q, 1))
        bn = (mean_new + sigsqd ** (1 / 2) * (S_new @ vec_new)).T
        betas_new[k] = bn
        data_part = float(yty - Xoy.T @ bo.T - Xny.T @ bn.T)
        old_part = float(-(bo @ Xoy) + bo @ XoXo @ bo.T + bo @ XoXn @ bn.T)
        new_part = float(-(bn @ Xny) + bn @ XnXo @ bo.T + bn @ XnXn @ bn.T)
        prior_quad = float(bo @ Sinv @ bo.T - bo @ mu_prec) + float(-(mu.T @ Sinv @ bo.T) + mu_quad)
        bn2 = float(bn @ bn.T)
        bstar = 0.5 * data_part + 0.5 * old_part + 0.5 * new_part + 0.5 / tausqd * bn2 + 0.5 * prior_quad + b
        sigsqd = math.nan if bstar < 0 else 1 / np.random.gamma(astar, 1 / bstar)            # FR:2379-2397
        tausqd = 1 / np.random.gamma(atau_star, 1 / ((1 / (2 * sigsqd)) * bn2 + btau))       # FR:2402-2404
        lik[k] = -(n / 2) * np.log(sigsqd) - 0.5 / sigsqd * (data_part + old_part + new_part)    # FR:2408-2415
    return np.concatenate([betas_old, betas_new], axis=1), p * math.log(n) - 2 * np.max(lik)


def fit_update_next(backend, n, m, n_phis, betas_prev, burn, a, b, atau, btau, tolerance, draws_total, gimmie, aic,
                    sigsqd0, console=False):
    """
    ``fit(update=True)`` on a model that has been built (FR:1939-1943 + the driver FR:2473-2583): prior mean and
    covariance from the previous posterior draws ``betas_prev[burn:-1]``; the interaction matrix grows sub-stage by
    sub-stage as in the first call, nothing is sampled until it is as long as the prior model (FR:2530), then the model
    of the same size ("same", case 2) and every longer one ("new", case 3) is.  Consumes numpy's GLOBAL generator.
    Returns (betas, mtx, evs [k, 1] as the reference leaves them, stats, trace).
    """
    if m == 1:
        raise ValueError("not enough values to unpack (expected 2, got 0)")
    prev = np.asarray(betas_prev)[burn:-1]
    mu = np.mean(prev, axis=0)[:, None]                              # [P, 1]
    Sinv = np.linalg.inv(np.cov(prev.transpose()))
    p_old = mu.shape[0]
    pool = SlotPool(backend)
    gram = np.array(backend.gram([SLOT_ONES, SLOT_Y], [SLOT_ONES, SLOT_Y]), dtype=np.float64)
    model_slots = []
    damtx = np.zeros((0, m))
    pending = np.zeros((0, m))                                       # terms of skipped sub-stages, built when first needed
    evs = []
    greater = 0
    best = mtx = last = None
    stats = dict(gibbs_calls=0, kill_tests=0, terms_logical=0, terms_physical=0, substages=0)
    trace = []
    for ind, indvec in update_substage_patterns(m, n_phis):
        vecs = distinct_arrangements(indvec)
        damtx = np.append(damtx, vecs, axis=0)
        pending = np.append(pending, vecs, axis=0)
        if p_old - 1 > damtx.shape[0]:
            continue                                                 # FR:2530: not as long as the prior model yet
        new_slots = pool.take(pending.shape[0])
        backend.build_terms(pending.astype(np.int32), new_slots)
        n_prev = 1 + len(model_slots)
        block = backend.gram(new_slots, [SLOT_ONES] + model_slots + new_slots + [SLOT_Y])
        gram = ForwardSelection._extend_gram(gram, list(range(n_prev)), block, list(range(n_prev)), n_prev)
        model_slots = model_slots + new_slots
        built_now, pending = pending.shape[0], np.zeros((0, m))
        p = 1 + len(model_slots)
        if p == p_old:
            if console:
                print('same')
            betas, ev = _update_sampler_same(gram, p, mu, Sinv, a, b, atau, btau, sigsqd0, n, draws_total)
        else:
            if console:
                print('new')
            betas, ev = _update_sampler_grown(gram, p_old, p, mu, Sinv, a, b, atau, btau, sigsqd0, n, draws_total)
        if aic:
            ev = ev + (2 - math.log(n)) * damtx.shape[0]
        stats['gibbs_calls'] += 1
        stats['terms_logical'] += built_now
        stats['terms_physical'] += built_now
        stats['substages'] += 1
        trace.append(dict(cols=p, built=built_now, ev=float(ev), kill=False))
        if console:
            print(ind, ev)
        last = (betas, damtx)
        evs.append(ev)
        if ev == min(evs):
            best, mtx, greater = betas, damtx, 1
        elif greater <= tolerance:
            greater += 1
        else:
            break
    if gimmie and last is not None:
        best, mtx = last
    if best is None:
        raise ValueError("the prior model has more terms than the basis set can enumerate (FR:2530 never holds)")
    return best, np.array(mtx, dtype=np.float64), np.array(evs, dtype=np.float64)[:, None], stats, trace
