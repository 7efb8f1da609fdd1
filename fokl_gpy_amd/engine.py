"""
Forward-variable-selection driver: the host orchestration of the hot path of ``FoKL.fit``.

Restates the control flow of the reference's driver loop (FoKLRoutines.py "FR" 1602-1760: term enumeration,
one ``gibbs`` evaluation per sub-stage, sequential kill tests, BIC stop rule) on top of a device backend:

    K1  backend.build_terms   -- new basis columns, written once per sub-stage into device "slots"
    K2  backend.gram          -- Gram block of the new columns against [1 | model | new | y]
    K3  backend.bic_resid     -- residual moments of a candidate model (any column subset)
    G3  _capi.gibbs_chain     -- the Gibbs chain in the eigenbasis, numpy-legacy random stream (host C++)

Where the reference rebuilds every surviving new column and the full Gram matrix for each kill test
(FR:1676-1683), every kill-test design is a column subset of the sub-stage's own design, so its XtX / Xty
are sub-blocks of the sub-stage Gram held on the host; the BIC still comes from an explicit
residual pass over the device columns, never from the cancellation-prone Gram identity.
The *logical* candidate-term count (what the reference would have built) is tallied per call.
"""
import collections
import itertools
import threading
import math
import os
import time

import numpy as np
from scipy.linalg import eigh as _eigh

from . import _capi

SLOT_ONES, SLOT_Y, SLOT_FIRST_FREE = _capi.SLOT_ONES, _capi.SLOT_Y, _capi.SLOT_FIRST_FREE
from .host_pipeline import (  # noqa: F401  (re-exported: tests and tools reach them through this module)
    HostPipeline, ShardedSpectralJob, GibbsOutcome, EagerOutcome, NativeOutcome, NativeSpectrum, Misprediction, ModelSize,
    chain_engine_for, spectral_engine_for,
    close_chain_engines, drop_spare_buffers, _mark, _flush_marks, _cpu_budget, _thread_plan, _thread_spares,
    _place_host_threads, _SPARES)


# ---------------------------------------------------------------------------------------------------------
# term enumeration (F4)
# ---------------------------------------------------------------------------------------------------------

def distinct_arrangements(indvec):
    """All distinct orderings of the multiset ``indvec`` in ascending lexicographic order, float64 rows.

    Equals ``np.unique(perms(indvec), axis=0)`` of the reference (FR:1350-1354, FR:1616) without enumerating
    M! tuples -- the reference's itertools.permutations cannot run beyond M ~ 10 inputs.
    """
    cur = sorted(float(v) for v in indvec)
    n = len(cur)
    rows = [tuple(cur)]
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
        cur[i + 1:] = cur[:i:-1]
        rows.append(tuple(cur))
    return np.array(rows, dtype=np.float64).reshape(len(rows), n)


def deal_indvec(ind, m, sett):
    """Spread ``ind`` units round-robin over the first ``sett`` entries of a length-m vector (FR:1605-1613)."""
    if min(int(ind), int(sett)) > m:
        raise IndexError(f"index {m} is out of bounds for axis 0 with size {m}")   # as the reference's indvec[j]
    indvec = np.zeros(m)
    q, r = divmod(int(ind), int(sett))
    indvec[:sett] = q
    indvec[:r] += 1
    return indvec


def advance_indvec(indvec, m, way3):
    """Next sub-stage pattern of the same total order (FR:1722-1740).  Returns False when the stage is exhausted."""
    if m == 1:
        return False
    if way3:
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
            return False
        return True
    if indvec[1]:
        indvec[0] += 1
        indvec[1] -= 1
        return True
    return False


# ---------------------------------------------------------------------------------------------------------
# eigen-decomposition with the shared sign convention
# ---------------------------------------------------------------------------------------------------------

def eigh_canonical(A):
    """``scipy.linalg.eigh`` (as the reference, FR:1499) + "largest-magnitude component positive" per eigenvector.

    LAPACK's eigenvector signs flip under 1-ulp changes of XtX, and each flip changes the realised draw
    Q diag(d^1/2) vec (FR:1525-1528); the convention makes draws comparable across summation orders.
    """
    lam, Q = _eigh(A)
    if os.environ.get('FOKL_EIGH_SIGNS', 'canonical') == 'lapack':
        return lam, Q                   # LAPACK's own signs: the untouched reference's draws (see README, parity)
    piv = np.argmax(np.abs(Q), axis=0)
    sgn = np.sign(Q[piv, np.arange(Q.shape[1])])
    sgn[sgn == 0] = 1.0
    return lam, Q * sgn


# ---------------------------------------------------------------------------------------------------------
# device backend protocol
# ---------------------------------------------------------------------------------------------------------

class HipBackend:
    """The shipped backend: a ``_capi.DeviceContext`` on one MI355X.  No CPU path exists in the product."""

    def __init__(self, ctx):
        self.ctx = ctx

    def upload(self, inputs, data, kernel_id, packed, n_basis, width):
        self.ctx.upload(inputs, data, kernel_id, packed, n_basis, width)
        self.kernel_id = kernel_id

    # FoKL.clean's normalisation on the device (fokl_stage_inputs / fokl_upload_staged): the model's `inputs` attribute
    # then exists on the device only until somebody asks for it (FoKLRoutines._DeviceInputs) -- or until the next dataset
    # arrives on this context, which makes the previous owner fetch its copy first (_capi.DeviceContext).
    def stage_inputs(self, x, owner):
        return self.ctx.stage_inputs(x, owner)

    def upload_staged(self, data, kernel_id, packed, n_basis, width, lows, spans, owner):
        self.ctx.upload_staged(data, kernel_id, packed, n_basis, width, lows, spans, owner)
        self.kernel_id = kernel_id

    def download_inputs(self):
        return self.ctx.download_inputs()

    def reserve_slots(self, count):
        self.ctx.reserve_slots(count)

    def build_terms(self, terms, slots):
        self.ctx.build_terms(terms, slots)

    def build_terms_deriv(self, terms, slots, wrt_input, order, divisor):
        self.ctx.build_terms_deriv(terms, slots, wrt_input, order, divisor)

    def read_slot(self, slot):
        return self.ctx.read_slot(slot)

    def gram(self, row_slots, col_slots, allreduce=False):
        return self.ctx.gram(row_slots, col_slots, 0, allreduce)

    def gram_launch(self, row_slots, col_slots, allreduce=False):
        return self.ctx.gram_launch(row_slots, col_slots, allreduce)

    def gram_fetch(self, shape):
        return self.ctx.gram_fetch(shape)

    def gram_ready(self):
        return self.ctx.gram_ready()

    def bic_resid(self, slots, betahat, allreduce=False):
        return self.ctx.bic_resid(slots, betahat, allreduce)

    def bic_resid_launch(self, slots, betahat):
        self.ctx.bic_resid_launch(slots, betahat)

    def bic_resid_fetch(self, allreduce=False):
        return self.ctx.bic_resid_fetch(allreduce)

    def bic_resid_terms_launch(self, terms, betahat):
        self.ctx.bic_resid_terms_launch(terms, betahat)

    def resid_terms_supported(self, terms):
        """Can a model made of (a subset of) these terms take the matrix-free residual pass?  (The limits of
        fokl_bic_resid_terms_launch, include/fokl_hip.h: one or two inputs per term, the distinct (input, order) factors
        within one of the kernel's slot layouts.)  Measured on MI355X (tools/k3_probe.py, profiles/k3_ab_r05.txt)."""
        nz = terms != 0
        if not nz.any() or int(nz.sum(axis=1).max()) > 2:
            return False
        if getattr(self, 'kernel_id', 1) != 0 and int(terms.max()) > _capi.RESID_TERMS_MAX_ORDER:
            return False
        inputs, deepest = self._factor_shape(terms)
        self._shape_of_last = (terms, inputs, deepest)          # (resid_terms_pay_from is asked about the same terms next)
        return any(inputs <= gm and deepest <= km for gm, km in _capi.RESID_TERMS_LAYOUTS)

    @staticmethod
    def _factor_shape(terms):
        """(inputs used, most distinct orders on one input) of a set of terms."""
        per_input = [len(set(col) - {0}) for col in terms.T.tolist()]     # (a few hundred small ints: no numpy call pays)
        return sum(1 for c in per_input if c), max(per_input)

    def resid_terms_pay_from(self, terms):
        """Model columns from which the matrix-free pass is the faster one (its time depends on the factor slots of its
        layout, the stored pass's on the columns it reads: profiles/k3_quadratic_r05.txt -- Bernoulli factors: always;
        spline factors, four table gathers each: 38 us at 8 slots / 72 us at 16 against 1.4 us per stored column at
        N = 1e6)."""
        if getattr(self, 'kernel_id', 1) != 0:
            return 0
        last = getattr(self, '_shape_of_last', None)
        inputs, deepest = last[1:] if last is not None and last[0] is terms else self._factor_shape(terms)
        slots = min(gm * km for gm, km in _capi.RESID_TERMS_LAYOUTS if inputs <= gm and deepest <= km)
        return int(3.5 * slots)

    def predict(self, slots, betas, cut=None):
        return self.ctx.predict(slots, betas, cut)


class SlotPool:
    """Free list of device column slots (slot 0 = ones, slot 1 = y are never handed out)."""

    def __init__(self, backend, initial=64):
        self.backend = backend
        self.capacity = 0
        self.free = []
        self.grow(initial)

    def grow(self, capacity):
        if capacity <= self.capacity:
            return
        self.backend.reserve_slots(capacity)
        lo = max(self.capacity, SLOT_FIRST_FREE)
        self.free.extend(range(capacity - 1, lo - 1, -1))
        self.capacity = capacity

    def take(self, count):
        if len(self.free) < count:
            self.grow(self.capacity + max(count - len(self.free), self.capacity // 2, 32))
        out = [self.free.pop() for _ in range(count)]
        return out

    def give(self, slots):
        self.free.extend(sorted(slots, reverse=True))


# ---------------------------------------------------------------------------------------------------------
# the driver
# ---------------------------------------------------------------------------------------------------------


# Deepest look-ahead of the native kill-test loop that every BASELINE configuration has been run with (0 .. 24: parity green;
# 24 already costs configs[3] time).  48 on configs[3] orders work so far ahead that a kill test's tape is submitted to the
# device after the stream has overwritten the pre-states of its segments (512 segments are kept) and the fit stops with
# "the pre-state of a segment is not (or no longer) in the ring" -- a loud error, but not one a documented knob should reach.
LOOKAHEAD_MAX = 24


def _bounded_lookahead(name, value):
    depth = max(0, int(value))
    if depth > LOOKAHEAD_MAX:
        import warnings
        warnings.warn(f"{name}={depth} is deeper than the native search has been verified with; using {LOOKAHEAD_MAX}",
                      RuntimeWarning)
        depth = LOOKAHEAD_MAX
    return depth


class ForwardSelection:
    """
    One ``fit`` worth of forward selection on an uploaded dataset.

    Parameters mirror the locals of FoKL.fit (FR:1302-1315, 1371-1374): hyper-parameters, ``n`` rows, ``m``
    inputs, ``n_phis = len(phis)`` (the search stops when the stage index exceeds it, FR:1747).
    ``stream`` is the numpy legacy RNG state the chain continues.  With ``row_sharded=True`` the backend holds one
    shard of the rows (``n`` local, ``n_global`` in total) and sums its Gram blocks / residual moments over ranks
    inside ``gram`` / ``bic_resid``; the sampler then runs replicated on every rank from identical inputs.
    ``comm`` is not used by the search itself (collectives live in the backend) and is kept for callers that want
    the communicator next to the search object.
    """

    def __init__(self, backend, n, m, n_phis, a, b, atau, btau, tolerance, draws_total, draws_keep, gimmie, way3,
                 threshav, threshstda, threshstdb, aic, stream, console=False, comm=None, row_sharded=False,
                 n_global=None, candidate_sharded=False):
        self.backend = backend
        self.n_local = int(n)
        self.n = int(n_global) if n_global is not None else int(n)
        self.n_local = int(n)               # rows resident on this rank's device
        self.m = int(m)
        self.n_phis = int(n_phis)
        self.a, self.b, self.atau, self.btau = a, b, atau, btau
        self.tolerance = tolerance
        self.draws = int(draws_total)
        self.draws_keep = int(draws_keep)
        self.gimmie, self.way3, self.aic = bool(gimmie), bool(way3), bool(aic)
        self.threshav, self.threshstda, self.threshstdb = threshav, threshstda, threshstdb
        self.stream = stream
        self.console = console
        self.comm = comm
        self.allreduce = bool(row_sharded)
        # candidate sharding (north_star): every rank holds all rows and drives the same search from the
        # same random stream; the RNG-free half of every model evaluation -- G2 and the BIC of the speculative kill-test
        # candidates -- is computed by one rank each and all-gathered (HostPipeline._exchange)
        self.candidate_sharded = bool(candidate_sharded) and comm is not None and (
            comm.world > 1 or os.environ.get('FOKL_CANDIDATE_SHARD_FORCE', '0') == '1')
        self._replicated_native = False     # a search replicated over ranks on the native driver (decided in run())
        self.substage_stats = []            # per sub-stage: |mean beta| and std / |mean| of its new terms (FR:1656-1658)
        self._update_args = None            # (from columns, depth, look-ahead) of the derived eigenpairs, when they are on
        self._predicted_kills = None        # columns the current sub-stage's tests will probably remove (_guess_first_tests)
        # FOKL_FORECAST_EARLY=0: G2 of the coming sub-stage's model is requested when the kill set is final, not before
        self._forecast_early = os.environ.get('FOKL_FORECAST_EARLY', '1') != '0'
        # two ctypes calls per look, ~1 us: the model's chain takes ~1.2 ms; FOKL_SPIN scales it like the native spins
        self._forecast_polls = int(2000 * float(os.environ.get('FOKL_SPIN', '1') or 1))
        # Both at once is the hybrid split: every rank holds N / G rows (K1, K2, K3 on its rows, the small Gram blocks and
        # residual moments all-reduced on the device -- the DEVICE work is divided by G) and the replicated search deals its
        # eigen-decompositions over the ranks as well (the HOST work that bounds configs[3] is divided by G too).  The Gram
        # is identical on every rank after the all-reduce, so the owner of a spectral job computes what any rank would.
        self.sigsqd0 = b / (1 + a)          # FR:1371
        self.tausqd0 = btau / (1 + atau)    # FR:1372
        self.pool = SlotPool(backend)
        self.host = None                    # HostPipeline while run() is active (b > 0 only)
        # csrc/fokl_search.cpp: the tapes on order, the G2 jobs, the chains and the kill-test loop as native code next to the
        # pool (_capi.NativeSearch) while run() is active -- single-process searches; FOKL_SEARCH=python keeps this
        # file's own statement of the same logic (the reference for tests/test_native_search.py, and what searches
        # replicated over ranks run)
        self.native = None
        self._term_ids = {}                 # term -> its number (native BIC cache: identical models score identically)
        self._async_resid = hasattr(backend, 'bic_resid_launch')
        # K3 without the stored columns (fokl_bic_resid_terms_launch): the residual pass re-forms the basis columns from
        # the inputs, 8 N (M_used + 1) bytes instead of 8 N (P + 2).  Decided per sub-stage (every model evaluated in a
        # sub-stage is a column subset of its full model): FOKL_K3=columns keeps the stored-column pass for A/B runs.
        self._matrix_free = (hasattr(backend, 'bic_resid_terms_launch') and
                             os.environ.get('FOKL_K3', 'matrixfree') != 'columns')
        self._terms_arr = None              # [A, m] int32 terms of the active columns (row 0 = intercept) or None
        self._arrangement_memo = {}         # pattern -> its distinct arrangements (this search's)
        self._terms_pay_from = 0            # ... and the model size from which the matrix-free pass is the faster one
        # spectral jobs submitted ahead of the kill tests: three along the guessed path (the Python loop: every miss costs
        # the jobs); the native loop predicts its path (csrc/fokl_search.cpp PathModel) and goes twelve deep
        self.lookahead = int(os.environ.get('FOKL_LOOKAHEAD', '3'))
        self._lookahead_native = _bounded_lookahead('FOKL_LOOKAHEAD', os.environ.get('FOKL_LOOKAHEAD', '12'))
        # next test's tape requested before the decision that the test is run (rewound when it is not; data-driven, so
        # replicated drivers of a row-sharded fit stay in step) -- FOKL_TENTATIVE_TAPES=0 disables, =test forces rewinds
        # the next sub-stage's columns and Gram block are built, and G2 of its predicted model started, before this
        # sub-stage's kill tests are over (once at most `foresight` likely tests remain); FOKL_FORESIGHT=0 disables
        # (building ahead takes the Gram block against every column that can still be in the model, killed ones
        # included: about a third more K2 work, free while the GPU idles behind the host -- N = 1e6: 0.110 -> 0.105 s per
        # fit -- and a loss once the device bounds the fit -- N = 1e7: 0.141 vs 0.134 s, N = 5e7: 0.339 vs 0.303 s
        # without it: off beyond 5e6 rows per rank unless FOKL_FORESIGHT says otherwise)
        self.foresight = int(os.environ.get('FOKL_FORESIGHT', '8' if self.n_local <= 5_000_000 else '0'))
        mode = os.environ.get('FOKL_TENTATIVE_TAPES', '1')
        self.tentative_tapes = mode != '0'
        self._test_rewinds = mode == 'test'
        # how many tapes may be on order ahead of the decisions that they are needed (_speculate, _drop_speculation)
        # (the native search's tapes on order are rows the device expands, 96 KB each, and its kill tests are decided within
        # microseconds: its book is deep enough for a whole sub-stage's likely tests, so that the walker does not idle between
        # a model's tape and the moment the model's statistics have ordered the proposals)
        self.speculation_max = max(1, min(16, int(os.environ.get('FOKL_SPECULATION', '12'))))
        self._speculation_native = max(1, min(64, int(os.environ.get('FOKL_SPECULATION', '48'))))
        # FOKL_SPECULATE_ACROSS=0: tapes are not ordered beyond the sub-stage whose tests are being guessed
        self._speculate_across = os.environ.get('FOKL_SPECULATE_ACROSS', '1') != '0'
        self._speculation = self.speculation_max
        self._spec = collections.deque()    # (model size, tentative noise job): on order, in stream order, no verdict yet
        self._prechain = None               # (noise job, spec, chain job, buffer of w): chain started ahead (_chain_ahead)
        # BIC of kill-test candidates (pipelined search): 'gram' (= 'auto', the default) = residual moments from the
        # sub-stage's Gram, computed by the spectral thread along with G2 (no device work per candidate, no
        # wait of the driver; agrees with the device pass to < 1e-10 relative on the BIC; candidates that (nearly)
        # interpolate the data get the device pass after all, see _score); 'device' = the K3 residual pass, as for
        # every sub-stage model; 'check' = device, recording the largest disagreement with 'gram' seen.  (Up to round 2
        # 'auto' meant the device pass while it hid behind the candidate's noise tape: with tapes recorded ahead of the
        # driver the pass is latency on the driver's path instead, 0.115 of 0.175 ms per test at 60 columns.)
        self.kill_bic = os.environ.get('FOKL_KILL_BIC', 'auto')
        if self.kill_bic not in ('auto', 'gram', 'device', 'check'):
            raise ValueError("FOKL_KILL_BIC must be auto, gram, device or check")
        if self.candidate_sharded:
            # the candidate's BIC travels with its spectral factors (moments from the Gram, computed by the owner): a
            # K3 pass per candidate would be replicated on every GPU; and enough G2 jobs must be in flight ahead of the
            # tests to keep the spectral threads of all ranks busy
            self.kill_bic = 'gram'
            self.lookahead = max(self.lookahead, 3 * comm.world)
        # G3 of kill-test candidates on the device (chain_engine_for; decided in run()).  A device chain answers a
        # millisecond or two after its tape, so nothing in the search waits for one: the only thing the tests want from
        # the chain of the model accepted so far is the scale of its intercept draws (second clause of FR:1670), and that
        # is its least-squares intercept to a fraction of a per cent.  A decision is taken from that guess when the
        # proposal's |mean beta| is further than `guess_margin` (relative) from the threshold, noted with the model, and
        # CONFIRMED against the chain's own statistics when they arrive (_verify); a decision that does not hold raises
        # Misprediction and the search is repeated without guessing -- results never depend on a guess.
        self.allow_device_chains = True
        self.chain_engine = None
        self.spectral_engine = None         # G2 on the device (spectral_engine_for; decided in run())
        # ... for models of up to this many columns.  Beyond, the chain stays on the host threads: a 585-column tape is 14 MB
        # of page-locked memory and dozens of them are alive at a time (configs[3]: 1.74 s per fit with device chains,
        # 1.49 s with host chains -- and the eigen-decompositions, not the chains, are what that fit waits for)
        # (round 4: with tapes expanded on the device from rows nothing of a 585-column tape crosses the bus or is
        # materialised on the host -- every kill test's chain goes to the device, up to the engine's 768 columns:
        # configs[3] 1.76 -> 1.11 s per fit; FOKL_DCHAIN_ROWS=0 falls back to 256)
        self.device_chain_columns = int(os.environ.get(
            'FOKL_DCHAIN_MAX_COLUMNS', '768' if os.environ.get('FOKL_DCHAIN_ROWS', '1') != '0' else '256'))
        # (under direct decisions the native loop widens this floor, per model, to 32 standard errors of the chain's mean
        # intercept as its least-squares fit predicts them -- csrc/fokl_search.cpp guess_margin_for; measured on configs[2]:
        # the chains' means lie within 2.6e-5 of the least-squares intercepts)
        direct = os.environ.get('FOKL_KILL_DECIDE', 'direct') == 'direct' and os.environ.get('FOKL_SEARCH', 'native') != 'python'
        self.guess_margin = float(os.environ.get('FOKL_GUESS_MARGIN', '0.002' if direct else '0.02'))
        self._flip_guess = int(os.environ.get('FOKL_GUESS_TEST_FLIP', '0'))   # tests: the n-th guess is taken wrong
        self._unverified = collections.deque()   # (device-chained outcome, half0) whose checks are open, oldest first
        self._zombies = collections.deque()      # device jobs nobody will look at, released once they have run
        self._device_jobs = []                   # every device chain of this search (all released when it ends)
        self.trace = []                     # one record per gibbs evaluation
        self._outcomes = []                 # pipelined evaluations whose draws sit in pooled buffers (see _retire)
        self._retiring = []                 # ... and those of them nobody will look at any more
        self._ev_cache = {}                 # model (set of terms) -> its BIC, see _same_model_same_ev
        self._active_terms = [()]           # term of every active column of the current sub-stage (() = intercept)
        self.stats = dict(gibbs_calls=0, kill_tests=0, terms_logical=0, terms_physical=0, substages=0,
                          t_eigh=0.0, t_resid=0.0, t_chain=0.0, chains_materialised=0, bic_from_gram=0,
                          bic_gram_max_rel=0.0, tapes_rewound=0, tapes_wasted=0, chains_ahead=0, chains_ahead_unused=0, forecasts_used=0, resid_matrix_free=0, chains_skipped=0,
                          spectral_submitted=0, device_chains=0, chains_fetched=0, guessed=0, guess_waits=0,
                          guesses_verified=0, dchain_kernel_s=0.0, dchain_timed=0)

    # -- one model evaluation (G1-G4) -------------------------------------------------------------------
    def _same_model_same_ev(self, idx, ev):
        """The reference scores identical models identically (a sub-stage whose new terms are all killed ends on the
        model of the one before, FR:1691-1721, and `ev < min(evs)` then is an exact tie).  Here a model can be scored
        along different routes (device residual pass or Gram identity, Gram blocks from different launches), equal to
        rounding only: the first score of a model is the score of every later evaluation of it."""
        key = frozenset(self._active_terms[c] for c in idx)
        return self._ev_cache.setdefault(key, ev)

    def _ev_from_moments(self, s1, s2, p1):
        n = self.n
        siglik = s2 / n - (s1 / n) ** 2                              # np.var(y - X betahat), FR:1551
        self._last_siglik = siglik
        lik = -(n / 2) * math.log(siglik) - (n - 1) / 2 if siglik > 0 else math.nan
        ev = p1 * math.log(n) - 2 * lik                              # FR:1553-1554
        if self.aic:
            ev = ev + (2 - math.log(n)) * p1                         # FR:1653-1654 / FR:1684-1685
        return ev

    def _record(self, p1, n_prev_cols, ev, kill):
        built = p1 - n_prev_cols
        self.stats['gibbs_calls'] += 1
        self.stats['kill_tests'] += int(kill)
        self.stats['terms_logical'] += built
        self.trace.append(dict(cols=p1, built=built, ev=float(ev), kill=bool(kill)))

    # The pipelined evaluation comes in three phases so that the kill-test loop can interleave them:
    #   _begin  : G2 (waits for the spectral job) and the launch of the K3 residual pass -- deterministic, free of
    #             random numbers, so it may run for a model whose test is not decided yet;
    #   _commit : the noise tape request and the chain job -- consumes the random stream, strictly in reference order;
    #   _score  : fetches the residual moments -> BIC.
    def _begin(self, gram, slots, idx, spectral_job=None, on_device=True):
        """on_device: measure the residual moments with K3 (always for a sub-stage's model); otherwise they come from
        the Gram (kill-test candidates: column subsets of that model, see self.kill_bic)."""
        t0 = time.perf_counter()
        if spectral_job is None:
            spectral_job = self.host.spectral(gram, idx)
        spec = spectral_job.wait()
        t1 = time.perf_counter()
        self.stats['t_eigh'] += t1 - t0
        cand_slots = None
        if on_device:
            cand_slots = [slots[i] for i in idx]
            if self._async_resid:
                self._launch_resid(cand_slots, idx, spec.betahat)
        ycol = gram.shape[0] - 1
        return spec, idx, cand_slots, gram[ycol, ycol], slots

    def _launch_resid(self, cand_slots, idx, betahat):
        if self._terms_arr is not None and len(idx) >= self._terms_pay_from:
            self.backend.bic_resid_terms_launch(self._terms_arr[idx[1:]], betahat)
            self.stats['resid_matrix_free'] += 1
        else:
            self.backend.bic_resid_launch(cand_slots, betahat)

    def _spectral(self, gram, idx, parent=None, parent_pos=-1):
        """Queue G2 for the model made of columns idx of gram (pool job, or the native search's).  parent / parent_pos
        (native search): the spectrum of the model with one more column and which of its columns this one lacks."""
        if self.native is not None:
            return NativeSpectrum(self.native, self.native.spectral(gram, idx, getattr(parent, 'h', None), parent_pos),
                                  gram)
        return self.host.spectral(gram, idx)

    def _set_active_terms(self, damtx):
        """Terms of the active columns of the sub-stage that begins (column 0 = intercept)."""
        self._active_terms = [()] + list(map(tuple, np.asarray(damtx, dtype=np.int64).tolist()))
        if self.native is not None:
            self.native.set_substage([self._term_ids.setdefault(t, len(self._term_ids)) for t in self._active_terms])
        self._terms_arr = None
        if self._matrix_free and damtx.shape[0]:
            arr = np.ascontiguousarray(damtx, dtype=np.int32)
            if self.backend.resid_terms_supported(arr):
                self._terms_arr = np.vstack([np.zeros((1, arr.shape[1]), dtype=np.int32), arr])
                self._terms_pay_from = getattr(self.backend, 'resid_terms_pay_from', lambda a: 0)(arr)

    def _request_noise(self, p1, tentative=False, model=True):
        """model: the tape is meant for a sub-stage's model (host chain: finished by host threads while it is recorded);
        False: for a kill-test candidate, whose chain runs on the device when there is an engine (the tape stays raw)."""
        astar = self.a + 1 + self.n / 2 + p1 / 2                     # FR:1508 (mmtx + 1 == p1)
        atau_star = self.atau + (p1 - 1) / 2                         # FR:1510
        on_device = not model and self.chain_engine is not None and p1 <= self.device_chain_columns
        return self.host.request(p1, astar, atau_star, tentative, finish=not on_device)

    # Tapes on order.  A tape's content depends on nothing but the model size and the position of the stream, so the
    # driver keeps the noise thread supplied with the sizes the search will PROBABLY ask for next -- several deep, across
    # kill tests and sub-stage boundaries -- as tentative requests; the evaluation that really comes takes the oldest one
    # if the size fits (commit) and otherwise sends them all back (the stream is rewound to where the oldest began).  The
    # stream's consumption therefore is exactly the reference's whatever was guessed.
    def _drop_speculation(self, keep=0):
        """Send back the tapes on order beyond the first `keep`.  What a wrong guess costs is the recorder's time on
        tapes it had begun: each of those takes two off the depth of the speculation (every tape that is used grows it
        back by one); orders the recorder had not reached yet cost nothing."""
        if self.native is not None:
            self.native.drop_speculation()
            return
        while len(self._spec) > keep:
            _, job = self._spec.pop()                                 # youngest first
            if self._prechain is not None and self._prechain[0] is job:
                self._drop_prechain()
            begun = job.result.progress[0] > 0
            self.host.discard(job)
            self.stats['tapes_rewound'] += 1
            if begun:
                self.stats['tapes_wasted'] += 1
                self._speculation = max(1, self._speculation - 2)

    def _tape_for(self, p1, model=True):
        """The tape of the model evaluation that happens now (p1 columns; model: see _request_noise -- a tape on order
        is taken whatever role it was ordered for)."""
        if self._spec:
            size, job = self._spec[0]
            if size == p1:
                self._spec.popleft()
                job.resolve(True)
                self._speculation = min(self.speculation_max, self._speculation + 1)
                _mark('tape_committed', str(p1))
                return job
            self._drop_speculation()
        _mark('tape_requested', str(p1))
        return self._request_noise(p1, model=model)

    def _speculate(self, sizes):
        """sizes: the models the stream will probably serve next, in order (as far as the caller can see).  Orders that
        agree with them stay, the others are sent back, missing ones are placed until self._speculation tapes are on
        order."""
        if not self.tentative_tapes or self.host is None:
            return
        sizes = list(sizes)
        if self.native is not None:
            self.native.speculate([(int(size), isinstance(size, ModelSize)) for size in sizes])
            return
        k = 0
        while k < len(self._spec) and k < len(sizes) and self._spec[k][0] == sizes[k]:
            k += 1
        if k < len(self._spec):
            self._drop_speculation(k)
        for size in sizes[k:max(k, self._speculation)]:
            if self._test_rewinds:                                    # tests: a recorded tape that is then discarded
                bogus = self._request_noise(size + 1, tentative=True)
                while bogus.result.progress[0] < self.draws:
                    time.sleep(0)
                self.host.discard(bogus)
                self.stats['tapes_rewound'] += 1
            self._spec.append((size, self._request_noise(size, tentative=True, model=isinstance(size, ModelSize))))
        _mark('speculating', ' '.join(str(size) for size, _ in self._spec))

    def _chain_ahead(self, spectral_job, p1, dtd):
        """Start the chain of the evaluation expected next -- G2 `spectral_job`, p1 columns -- if its G2 has run and its
        tape is the oldest on order: by the time the driver gets there (it may have to wait for exactly these draws to
        decide the test after: second clause of FR:1670) they are under way.  Costs a chain thread some work when the
        guess is wrong; results are untouched (same tape, same arithmetic, whoever submits the job)."""
        if not self._spec or self._spec[0][0] != p1 or not getattr(spectral_job, 'done', lambda: False)():
            return
        if self.chain_engine is not None and not isinstance(self._spec[0][0], ModelSize):
            return                                                    # a kill test's chain: submitted to the device at commit
        noise_job = self._spec[0][1]
        spec = spectral_job.wait()
        if self._prechain is not None:
            if self._prechain[0] is noise_job and self._prechain[1] is spec:
                return
            self._drop_prechain()
        chain_job, w_raw = self.host.chain_ahead(spec, self.b, self.btau, dtd, self.sigsqd0, self.tausqd0, noise_job)
        self._prechain = (noise_job, spec, chain_job, w_raw)
        self.stats['chains_ahead'] += 1

    def _drop_prechain(self):
        if self._prechain is not None:
            noise_job, _, chain_job, w_raw = self._prechain
            self._prechain = None
            self.host.disown(chain_job, w_raw, noise_job)
            self.stats['chains_ahead_unused'] += 1

    def _commit(self, pending, noise_job=None, test=False, stat_first=0):
        """-> (noise job, chain job, raw buffer of w): the trailing arguments of GibbsOutcome.
        test: a kill-test candidate -- its chain goes to the device engine if there is one (the mean of w over the rows
        from stat_first on comes back with it); sub-stage models keep the host chain that follows the recorder."""
        spec, idx, _, dtd, _ = pending
        if noise_job is None:
            noise_job = self._tape_for(idx.shape[0], model=not test)
        if (test and self.chain_engine is not None and idx.shape[0] <= self.device_chain_columns
                and not (self._prechain is not None and self._prechain[0] is noise_job)):
            job = self.host.chain_device(spec, self.b, self.btau, dtd, self.sigsqd0, self.tausqd0, noise_job, stat_first)
            if job is not None:
                self.stats['device_chains'] += 1
                self._device_jobs.append(job)
                return noise_job, job, None
        if self._prechain is not None:
            if self._prechain[0] is noise_job and self._prechain[1] is spec:
                _, _, chain_job, w_raw = self._prechain
                self._prechain = None
                self.host.adopt(chain_job, noise_job)
                return noise_job, chain_job, w_raw
            self._drop_prechain()
        chain_job, w_raw = self.host.chain(spec, self.b, self.btau, dtd, self.sigsqd0, self.tausqd0, noise_job)
        return noise_job, chain_job, w_raw

    def _score(self, pending):
        spec, idx, cand_slots, dtd, slots = pending
        p1 = idx.shape[0]
        launched = cand_slots is not None
        if not launched:
            # y'y - 2 b'Xty + b'XtX b cancels y'y / SSR digits: fine for a noisy fit (1e3 at the benchmark), useless
            # for a model that (nearly) interpolates the data -- those candidates get the residual pass after all
            if spec.moments[1] > 1e-6 * dtd:
                self.stats['bic_from_gram'] += 1
                return self._same_model_same_ev(idx, self._ev_from_moments(spec.moments[0], spec.moments[1], p1))
            cand_slots = [slots[i] for i in idx]
        t0 = time.perf_counter()
        if self._async_resid and launched:
            s1, s2 = self.backend.bic_resid_fetch(self.allreduce)
        else:
            s1, s2 = self.backend.bic_resid(cand_slots, spec.betahat, self.allreduce)
        self.stats['t_resid'] += time.perf_counter() - t0
        ev = self._ev_from_moments(s1, s2, p1)
        if self.kill_bic == 'check':
            other = self._ev_from_moments(spec.moments[0], spec.moments[1], p1)
            self.stats['bic_gram_max_rel'] = max(self.stats['bic_gram_max_rel'], abs(other - ev) / abs(ev))
        return self._same_model_same_ev(idx, ev)

    def _evaluate(self, gram, slots, idx, n_prev_cols, kill, spectral_job=None, overlap=None, then=(), after_begin=None,
                  new_terms=0):
        """
        gram  : Gram of the sub-stage's active columns, last row/column = y   [(A + 1) x (A + 1)]
        slots : device slot of each active column
        idx   : active-column indices of this candidate model (idx[0] == 0, the intercept)
        spectral_job : G2 of exactly this model submitted earlier (pipelined search), if any
        overlap : called once G2 is under way -- work for the driver thread in its shadow (pipelined search)
        then  : sizes of the models that will probably be evaluated after this one (tapes ordered ahead, _speculate)
        after_begin : (native search) called with the model's spectrum handle once G2 has run and K3 is launched
        new_terms : (native search) how many of the model's last columns are the sub-stage's new terms (their statistics)
        """
        idx = np.asarray(idx, dtype=np.int32)
        p1 = idx.shape[0]
        if self.native is not None:
            # the same steps with the tape, G2 and the chain on the native side (fokl_search_model_begin / _commit /
            # _score); this thread keeps the device's residual pass
            ns = self.native
            t0 = time.perf_counter()
            spectrum, tape = ns.model_begin(gram, idx, spectral_job.h if spectral_job is not None else None,
                                            [(int(size), isinstance(size, ModelSize)) for size in then])
            try:
                betahat = ns.spectrum_view(spectrum).betahat        # (the wait for G2 is counted by the native side)
                cand_slots = [slots[i] for i in idx]
                if self._async_resid:
                    self._launch_resid(cand_slots, idx, betahat)
                if after_begin is not None:
                    after_begin(spectrum)       # G2 is there, the device is busy with K3: what can be ordered ahead now
                ycol = gram.shape[0] - 1
                handle = ns.model_commit(spectrum, tape, gram[ycol, ycol], 0 if kill else new_terms)
            except BaseException:
                ns.spectrum_release(spectrum)
                raise
            t0 = time.perf_counter()
            if self._async_resid:
                s1, s2 = self.backend.bic_resid_fetch(self.allreduce)
            else:
                s1, s2 = self.backend.bic_resid(cand_slots, np.array(betahat), self.allreduce)
            self.stats['t_resid'] += time.perf_counter() - t0
            ns.score(handle, s1, s2, n_prev_cols, kill)
            outcome = NativeOutcome(self, ns, handle)
            self._outcomes.append(outcome)
            return outcome
        if self.host is not None:
            # not speculative: the tape is requested first, so that it is recorded while G2 runs
            noise_job = self._tape_for(p1)
            self._speculate(then)
            if spectral_job is None:
                spectral_job = self.host.spectral(gram, idx)
            _mark('eval_requested', str(p1))
            if overlap is not None:
                overlap()
                _mark('eval_overlap')
            pending = self._begin(gram, slots, idx, spectral_job)
            _mark('eval_begun')
            jobs = self._commit(pending, noise_job)
            ev = self._score(pending)
            _mark('eval_scored')
            self._record(p1, n_prev_cols, ev, kill)
            outcome = GibbsOutcome(self, pending[0], ev, idx, *jobs)
            outcome.siglik = self._last_siglik
            self._outcomes.append(outcome)
            return outcome

        n = self.n
        astar = self.a + 1 + n / 2 + p1 / 2                          # FR:1508
        atau_star = self.atau + (p1 - 1) / 2                         # FR:1510
        t0 = time.perf_counter()
        cand_slots = [slots[i] for i in idx]
        ycol = gram.shape[0] - 1
        dtd = gram[ycol, ycol]
        XtX = gram[np.ix_(idx, idx)]
        Xty = gram[idx, ycol]
        lamb, Q = eigh_canonical(XtX)
        qty = Q.T @ Xty
        betahat = Q @ (qty / lamb)                                  # FR:1502-1504
        self.stats['t_eigh'] += time.perf_counter() - t0
        if self._async_resid:
            self._launch_resid(cand_slots, idx, betahat)
        try:
            w = _capi.gibbs_chain(lamb, qty, astar, atau_star, self.b, self.btau, dtd, self.sigsqd0,
                                  self.tausqd0, self.draws, self.stream)
        finally:
            if self._async_resid:
                s1, s2 = self.backend.bic_resid_fetch(self.allreduce)
        if not self._async_resid:
            s1, s2 = self.backend.bic_resid(cand_slots, betahat, self.allreduce)
        ev = self._same_model_same_ev(idx, self._ev_from_moments(s1, s2, p1))
        self._record(p1, n_prev_cols, ev, kill)
        return EagerOutcome(w, Q, betahat, ev, idx)

    def _retire(self, *keep):
        """End of a sub-stage: the draws of every model evaluated so far except `keep` (the best model of the search and
        the model the next sub-stage starts from) can no longer be looked at -- their buffers go back to the pool, a
        little later (_release_retired): the noise thread is waiting for the coming sub-stage's orders right now."""
        self._retiring += [o for o in self._outcomes if not any(o is k for k in keep)]
        self._outcomes = [k for k in keep if isinstance(k, (GibbsOutcome, NativeOutcome))]

    def _release_retired(self):
        for outcome in self._retiring:
            outcome.release()
        self._retiring = []

    @staticmethod
    def _columns_without(count, removed):
        return np.array([c for c in range(count) if c not in removed], dtype=np.int32)

    def _likely_first_tests(self, spec, n_new, siglik=None):
        """The kill tests a sub-stage will probably run, guessed from the least-squares fit of its model (spec: G2 of that
        model, its last n_new columns are new) before the model's chain is there: the proposals will be ordered by
        |mean beta| ~ |betahat| and can pass FR:1670 only if std beta / |mean beta| ~ (siglik (XtX)^-1_jj)^1/2 /
        |betahat_j| exceeds the smaller threshold (siglik from the Gram: G2 brings the residual moments along).
        Returns the active-column indices in testing order."""
        A = spec.betahat.shape[0]
        new = np.arange(A - n_new, A)
        if siglik is None:
            siglik = spec.moments[1] / self.n - (spec.moments[0] / self.n) ** 2
        guess_mean = np.abs(spec.betahat[new])
        guess_std = np.sqrt(max(siglik, 0.0) * np.sum(spec.Qt[:, new] ** 2 / spec.lamb[:, None], axis=0))
        floor = min(self.threshstda, self.threshstdb)
        return [int(new[j]) for j in np.argsort(guess_mean) if guess_std[j] > floor * guess_mean[j]]

    def _guess_first_tests(self, gram, spec, n_new, siglik=None, before_model=False, spectrum=None, vm_next=None):
        """G2 jobs for the first kill tests of the sub-stage whose model's G2 is `spec` (_likely_first_tests), and the
        tapes of all the likely ones (before_model: the model's own tape has not been taken yet and leads them).
        Returns ({trial set -> job}, sizes of the likely tests); a wrong guess costs a spectral thread a few
        milliseconds."""
        A = gram.shape[0] - 1
        if self.native is not None and spectrum is not None:
            # the native side also says which of them will probably be ACCEPTED (the BIC without a column from a rank-one
            # formula): G2 goes out along the path the tests will take.  Four deep: the ORDER of the tests is still a
            # guess here (|betahat| for |mean beta|) -- twelve deep cost 40 wasted eigen-decompositions per fit and
            # took the CPUs the model's chain runs on (53.5 -> 56-58 ms per fit)
            guessed = self.native.likely_first_tests(spectrum.h, n_new, siglik)
            jobs, cur, sizes = {}, frozenset(), []
            against = spectrum                          # the model the next test is held against, where its G2 is known
            self.native.hold_spectral(True)             # device G2: the four jobs become one grid
            try:
                for c, accepted in guessed:
                    trial = cur | {c}
                    if len(jobs) <= min(self._lookahead_native, 3):
                        pos = -1
                        if against is not None:
                            pos = int(np.searchsorted(self._columns_without(A, cur), c))
                        jobs[trial] = self._spectral(gram, self._columns_without(A, trial), against, pos)
                    sizes.append(A - len(cur) - 1)
                    if accepted:
                        cur = trial
                        against = jobs.get(trial)
            finally:
                self.native.hold_spectral(False)
            self._predicted_kills = sorted(cur)          # (what the tests will have removed if they go as the downdate says)
            beyond = []
            if vm_next is not None and self._speculate_across:
                # ... and across the boundary: the coming sub-stage's model if these tests end as predicted, and its tests as
                # if every new term were tested and accepted (their G2 will say better; a tape of the wrong size is rewound)
                coming = A - len(cur) + vm_next
                beyond = [ModelSize(coming)] + [coming - t for t in range(1, vm_next + 1)]
            self._speculate(([ModelSize(A)] if before_model else []) + sizes + beyond)
            return jobs, sizes
        likely = self._likely_first_tests(spec, n_new, siglik)
        jobs, cur = {}, frozenset()
        # (their order is guessed from the least-squares fit before the chain's statistics are there: three or four
        # deep -- what lies further along is ordered by the kill-test loop once the proposals are known)
        for c in likely[:1 + min(self.lookahead, 3)]:
            cur = cur | {c}
            jobs[cur] = self._spectral(gram, self._columns_without(A, cur))
        # and their tapes: test t of the sub-stage has A - 1 - t columns if the tests before it were accepted
        sizes = [A - 1 - t for t in range(len(likely))]
        self._speculate(([ModelSize(A)] if before_model else []) + sizes)
        return jobs, sizes

    def _intercept_scale(self, outcome, half0):
        """np.mean(np.abs(np.mean(betas[half0:draws, 0]))) of FR:1671 for the model accepted so far (needs its chain)."""
        if outcome.intercept_scale is None:
            if getattr(outcome, 'on_device', False):
                outcome.intercept_scale = abs(outcome.mean_intercept_draw(half0))
            else:
                outcome.intercept_scale = np.mean(np.abs(np.mean(
                    outcome.beta_columns(np.array([0]), half0)[:, 0])))
        return outcome.intercept_scale

    def _second_clause_now(self, outcome, value, half0):
        """`value < threshav * |mean intercept draw of outcome|` (second clause of FR:1670) if it can be had without
        waiting for a chain: from the chain's statistics if they are there, else -- device chains only -- from the
        least-squares intercept when `value` is not within guess_margin of the threshold; the guess is noted with the
        outcome and confirmed by _verify.  None: the caller has to wait for the chain."""
        if outcome.intercept_scale is not None or (getattr(outcome, 'chain_ready', None) and outcome.chain_ready()):
            return bool(value < self.threshav * self._intercept_scale(outcome, half0))
        if not getattr(outcome, 'on_device', False):
            return None
        threshold = self.threshav * abs(float(outcome.betahat[0]))
        if threshold <= 0.0 or not math.isfinite(threshold) or abs(value - threshold) <= self.guess_margin * threshold:
            # too close to call from the guess, and the device's answer is milliseconds away: the chain once more, in
            # line on this thread (same tape, same arithmetic up to the last bit of log())
            if outcome._dtd is None:
                return None
            self.stats['guess_waits'] += 1
            t0 = time.perf_counter()
            outcome.intercept_scale = outcome.intercept_scale_on_host(half0)
            self.stats['t_chain'] += time.perf_counter() - t0
            return bool(value < self.threshav * outcome.intercept_scale)
        decision = bool(value < threshold)
        self.stats['guessed'] += 1
        if self._flip_guess and self.stats['guessed'] == self._flip_guess:
            decision = not decision                                   # tests: a guess that verification must catch
        if not outcome.checks:
            self._unverified.append((outcome, half0))
        outcome.checks.append((float(value), decision))
        return decision

    def _release_device_job(self, job):
        """A device chain nobody will look at again: its slot goes back once it has run (never a wait here)."""
        if not job.try_release():
            self._zombies.append(job)

    def _verify(self, block=False):
        """Confirm the decisions that were taken from guessed intercept scales against the chains' own statistics --
        those that have arrived, or (block) all of them.  Raises Misprediction if one does not hold."""
        if self.native is not None:
            self.native.verify(block)
            return
        # chains complete in the order they were submitted, near enough: only the oldest is polled (one call per turn)
        while self._zombies and (self._zombies[0].try_release() or block):
            self._zombies.popleft().release()
        while self._unverified and (block or self._unverified[0][0].chain_ready()):
            outcome, half0 = self._unverified.popleft()
            scale = self._intercept_scale(outcome, half0)
            for value, decision in outcome.checks:
                if bool(value < self.threshav * scale) != decision:
                    raise Misprediction(f"kill test decided from a guessed intercept scale "
                                        f"({abs(float(outcome.betahat[0]))!r}) that its chain does not confirm ({scale!r})")
                self.stats['guesses_verified'] += 1
            outcome.checks = []
            if outcome._release_wanted:
                outcome.release()

    def _kill_tests_pipelined(self, gram, slots, n_prev, cand_col, mean_abs, rel_std, best, half0, foresee=None,
                              ahead=None, vm_next=None, idle_work=None, peek=None, chain_coming=None):
        """FR:1666-1690 with the host pipeline: same tests, same order, same random-stream consumption.

        Whether proposal i is tested may hinge on the chain of the model accepted so far (second clause of FR:1670);
        what the test computes does not.  So G2 of the models the next few tests will need is submitted ahead, and
        the residual pass of the upcoming one is in flight while this thread waits for that chain.
        foresee(killed set) is told, towards the end of the loop, which columns the loop will probably have removed
        when it is done (the caller starts G2 of the next sub-stage's model with it).  `ahead` may bring G2 jobs the
        caller has already submitted (trial set -> job).  vm_next: number of columns the coming sub-stage adds (None:
        there is none) -- the tapes ordered ahead go on across the boundary with that sub-stage's model and first tests.
        idle_work: called once, when the first test's tape and G2 are under way (or, without a test, at the end): work of
        the driver that nothing in this loop waits for (the caller builds the coming sub-stage's columns with it).
        peek(killed set): how many kill tests the coming sub-stage will probably run if this one ends with that kill set
        (None: not known yet).  chain_coming(killed set): start the chain of the coming sub-stage's model if this one ends
        with that kill set and G2 of that model is there.
        """
        if self.native is not None:
            # csrc/fokl_search.cpp fokl_search_kill_tests: this loop as native code.  peek / chain_coming are its own (the
            # forecasts are registered with it by `foresee`)
            resid = lambda idx, betahat: self.backend.bic_resid([slots[i] for i in idx], betahat, self.allreduce)
            killed, evmin, handle, is_new = self.native.kill_tests(
                gram, cand_col, mean_abs, rel_std, slots, best.h, n_prev, vm_next,
                {key: job.h for key, job in (ahead or {}).items()}, foresee, idle_work, resid)
            if is_new:
                best = NativeOutcome(self, self.native, handle)
                self._outcomes.append(best)
            return killed, evmin, best
        A = len(slots)
        dtd = gram[A, A]
        vm = cand_col.shape[0]
        cols = [int(c) for c in cand_col]
        clause1 = [bool(rel_std[j] > self.threshstdb) for j in range(vm)]
        clause2a = [bool(rel_std[j] > self.threshstda) for j in range(vm)]
        proposal = [j for j in range(vm) if clause1[j] or clause2a[j]]     # the others cannot pass FR:1670
        # Guess at "mean_abs < threshav * |mean intercept draw|" for proposals further down the list: the posterior mean
        # of the intercept is close to its least-squares value, and it barely moves from one accepted model to the next.
        scale_guess = abs(float(best.betahat[0]))
        on_device = self.kill_bic in ('device', 'check')
        killed = frozenset()
        evmin = best.ev
        ahead = {} if ahead is None else ahead                        # trial set -> spectral job submitted ahead
        last_accepted = True                                          # predictor: proposals go the way the last went
        likely = lambda j: clause1[j] or mean_abs[j] < self.threshav * scale_guess

        def forecast(pos):
            # the kill set at the end of the loop if every remaining test that looks likely runs and is accepted
            rest = [j for j in proposal[pos:] if likely(j)]
            if foresee is not None and len(rest) <= self.foresight:
                foresee(killed | {cols[j] for j in rest})

        def order_tapes(pos):
            # The tapes of what the stream serves next if the search goes on as predicted -- the likely tests from
            # proposal[pos] on, each one column smaller than the one before while tests are being accepted, then the
            # coming sub-stage's model and its first test -- ordered as soon as this test's BIC is known (the sizes need
            # the kill set) instead of after the chains that decide whether those tests run: the noise thread goes from
            # one tape to the next without a pause.
            sizes, pred = [], set(killed)
            for j in proposal[pos:]:
                if len(sizes) >= self.speculation_max:
                    break
                if likely(j):
                    sizes.append(A - len(pred) - 1)
                    if last_accepted:
                        pred.add(cols[j])
            else:
                if vm_next is not None:
                    # across the boundary: the coming model, its first test (every first test is one column smaller
                    # whichever proposal it removes) -- or, if G2 of that model is there already, all the tests its
                    # least-squares fit makes likely
                    tests = peek(pred) if peek is not None else None
                    if tests is None:
                        tests = min(vm_next, 1)
                    coming = A - len(pred) + vm_next
                    sizes += [ModelSize(coming)] + [coming - t for t in range(1, tests + 1)]
            self._speculate(sizes)
            # ... and the chain of the very next evaluation, if its G2 is there
            nxt = next((j for j in proposal[pos:] if likely(j)), None)
            if nxt is not None:
                job = ahead.get(killed | {cols[nxt]})
                if job is not None:
                    self._chain_ahead(job, A - len(killed) - 1, dtd)
            elif chain_coming is not None and vm_next is not None:
                chain_coming(killed)

        forecast(0)
        order_tapes(0)
        for pos, i in enumerate(proposal):
            decided = clause1[i]
            _mark('test', f"{pos} decided={int(decided)} known={int(best.intercept_scale is not None)}")
            self._verify()
            if not decided:
                # the second clause without a wait: from the chain of `best` if it has run, from its least-squares
                # intercept (confirmed later) if that is a device chain and the proposal is not a borderline case
                quick = self._second_clause_now(best, mean_abs[i], half0)
                if quick is False:
                    continue
                if quick:
                    decided = True
                    if best.intercept_scale is not None:
                        scale_guess = best.intercept_scale
            if not decided and (best.intercept_scale is not None or not likely(i)):
                # second clause without G2 of a model that will probably not be needed: from the known scale, or --
                # the test looks unlikely -- after waiting for the chain of `best`
                scale_guess = self._intercept_scale(best, half0)
                if not mean_abs[i] < self.threshav * scale_guess:
                    continue
                decided = True
            # G2 of the models on the predicted path, self.lookahead tests deep (a wrong guess costs latency only)
            cur = killed
            upcoming = itertools.islice((j for j in proposal[pos + 1:] if likely(j)), self.lookahead)
            for j in itertools.chain((i,), upcoming):
                key = cur | {cols[j]}
                if key not in ahead:
                    ahead[key] = self.host.spectral(gram, self._columns_without(A, key))
                if last_accepted:
                    cur = key
            trial = killed | {cols[i]}
            idx = self._columns_without(A, trial)
            p1 = idx.shape[0]
            # the test runs for sure: its tape is committed (or, not on order after a wrong guess, requested) now, so
            # that the stream moves on while this thread waits for G2
            noise_job = self._tape_for(p1, model=False) if decided else None
            _mark('g2_submitted')
            if idle_work is not None:
                idle_work()
                idle_work = None
                _mark('idle_work')
            pending = self._begin(gram, slots, idx, ahead.pop(trial), on_device=on_device)
            _mark('begun')
            if not decided:
                scale_guess = self._intercept_scale(best, half0)      # waits for the chain of `best`
                _mark('chain_of_best')
                if not mean_abs[i] < self.threshav * scale_guess:
                    if self._async_resid and pending[2] is not None:
                        self.backend.bic_resid_fetch(self.allreduce)  # drains the speculative residual pass
                    continue
                noise_job = self._tape_for(p1, model=False)
            if pending[2] is None:
                # the BIC comes from the Gram and is known now, before anything is spent on the candidate's draws:
                # a rejected candidate only has to advance the random stream (its tape is recorded, never finished
                # nor chained -- nobody reads the draws of a model that loses, FR:1686-1690)
                ev = self._score(pending)
                if ev < evmin:
                    jobs = self._commit(pending, noise_job, test=True, stat_first=half0)
                else:
                    if self._prechain is not None and self._prechain[0] is noise_job:
                        self._drop_prechain()
                    self.host.abandon(noise_job)
                    self.stats['chains_skipped'] += 1
                    jobs = None
            else:
                jobs = self._commit(pending, noise_job, test=True, stat_first=half0)
                ev = self._score(pending)
            _mark('scored')
            self._record(p1, n_prev, ev, True)
            last_accepted = bool(ev < evmin)
            if last_accepted:
                killed, evmin = trial, ev
                best.release()                                        # the model it replaces: its draws are history
                best = GibbsOutcome(self, pending[0], ev, idx, *jobs)
                best._dtd = dtd
                self._outcomes.append(best)
            elif jobs is not None:
                if jobs[2] is None:
                    self._release_device_job(jobs[1])                 # nobody will read a rejected candidate's draws
                else:
                    jobs[1].recycle.append(jobs[2])
            forecast(pos + 1)
            order_tapes(pos + 1)
        order_tapes(len(proposal))                                    # the kill set is final
        if idle_work is not None:
            idle_work()
            forecast(len(proposal))
        return sorted(killed), evmin, best

    # -- the search ---------------------------------------------------------------------------------------
    def run(self):
        """The whole search.  With b > 0 (always, unless the user forces a non-positive scale) the random stream, the
        chains and the eigen-decompositions run on native threads (HostPipeline); otherwise every chain runs in line."""
        pipelined = self.b > 0 and os.environ.get('FOKL_NOISE_PIPELINE', '1') != '0'
        t_begin_run = time.perf_counter()
        # Head start: the seed Gram and K1 + K2 of the first sub-stage (its pattern is known before anything else is) go to
        # the device now, and run while the host threads of the pipeline are being created (0.6 ms) -- the first sub-stage
        # then finds its Gram block waiting, like every later one whose columns were built ahead.  Single-process searches
        # (a split over ranks gathers rows: a collective, kept where the other collectives are).  FOKL_HEAD_START=0: off.
        self._head = None
        # The sub-stage loop itself as native code (csrc/fokl_run.cpp, round 6) where the search is the common case: one
        # process, the native search, look-ahead on, default build-ahead / statistics.  Its head start replaces the one
        # below.  FOKL_SUBSTAGE_LOOP=python: this file's _run (what every other search runs, and the statement the native
        # loop is tested against).
        self._nrun = None
        if (pipelined and not (self.allreduce or self.candidate_sharded) and self._native_planned()
                and self.lookahead > 0 and self.tentative_tapes and not self._test_rewinds and not self.console
                and not (self.way3 and self.m == 2)        # (FR:1724 reads indvec[2]: the reference raises -- this file's loop too)
                and os.environ.get('FOKL_SUBSTAGE_LOOP', 'native') != 'python'
                and os.environ.get('FOKL_BUILD_AHEAD', 'model') == 'model'
                and os.environ.get('FOKL_STATS', 'native') == 'native'
                and not os.environ.get('FOKL_POOL_TRACE')):
            self._nrun = _capi.NativeRun(
                self.backend, getattr(self.backend, 'kernel_id', getattr(self.backend, 'kernel', 1)),
                m=self.m, n_phis=self.n_phis, way3=int(self.way3), tolerance=self.tolerance, gimmie=int(bool(self.gimmie)),
                draws=self.draws, half0=int(math.ceil(self.draws / 2)), lookahead=self.lookahead,
                lookahead_native=self._lookahead_native, foresight=self.foresight,
                speculate_across=int(self._speculate_across), forecast_early=int(self._forecast_early),
                forecast_polls=0 if os.environ.get('FOKL_SYNC') == 'blocking' else self._forecast_polls,
                matrix_free=int(self._matrix_free), update_from=0, update_depth=0, update_lookahead=0,
                head_start=int(os.environ.get('FOKL_HEAD_START', '1') != '0'), slot_capacity=self.pool.capacity)
        self.stats['t_head_start'] = time.perf_counter() - t_begin_run
        if (self._nrun is None and pipelined and hasattr(self.backend, 'gram_launch')
                and not (self.allreduce or self.candidate_sharded)
                and os.environ.get('FOKL_HEAD_START', '1') != '0'):
            base = self.backend.gram([SLOT_ONES, SLOT_Y], [SLOT_ONES, SLOT_Y], False)
            self._head = (base, self._build_ahead(next(self._patterns())[1], [SLOT_ONES]))
        if pipelined:
            try:
                # device chains: one search per process at a time drives an engine's guesses.  Searches replicated over ranks
                # (rows or candidates sharded) have them under the native driver, which then keeps the arrival time of a
                # chain out of every decision (fokl_search_set_deterministic); the Python loop keeps host chains there -- a
                # rank whose chain had arrived would decide from it, one that guesses would stop on a misprediction alone
                # and leave the others in a collective
                replicated = self.allreduce or self.candidate_sharded
                self._replicated_native = replicated and self._native_planned()
                if self.allow_device_chains and (not replicated or self._replicated_native):
                    self.chain_engine = chain_engine_for(getattr(getattr(self.backend, 'ctx', None), 'device', None))
                    if self.chain_engine is not None:
                        self.device_chain_columns = min(self.device_chain_columns,
                                                        getattr(self.chain_engine, 'max_columns', 768))
                    self._dchain_stats0 = self.chain_engine.stats() if self.chain_engine is not None else {}
                # (a sub-stage of 3-way terms over m inputs adds up to m (m - 1) (m - 2) columns: models of hundreds)
                wide = self.m * (self.m - 1) * (self.m - 2 if self.way3 else 1) >= 1000
                # (the Python loop's candidate sharding deals the G2 jobs over the ranks, HostPipeline._exchange; the native
                # driver's splits the Gram launches instead, _gram_of_new_columns, and keeps every G2 job at home)
                self.host = HostPipeline(self.stream, self.draws,
                                         self.comm if self.candidate_sharded and not self._replicated_native else None,
                                         chain_engine=self.chain_engine, wide_models=wide,
                                         device=getattr(getattr(self.backend, 'ctx', None), 'device', None))
            except (ImportError, KeyError, AttributeError, _capi.FoklNativeError) as exc:
                # e.g. a scipy without the cython_lapack capsule the spectral threads call through: same results in
                # line, only slower
                import warnings
                warnings.warn(f"host thread pipeline unavailable ({exc}); running the search in line", RuntimeWarning)
                self.host = None
        self.stats['t_pool_create'] = time.perf_counter() - t_begin_run
        if self.host is not None and self._native_wanted():
            try:
                half0 = int(math.ceil(self.draws / 2))
                self.native = _capi.NativeSearch(
                    self.host.pool, self.chain_engine, n=self.n, a=self.a, b=self.b, atau=self.atau, btau=self.btau,
                    threshav=self.threshav, threshstda=self.threshstda, threshstdb=self.threshstdb,
                    guess_margin=self.guess_margin, draws=self.draws, half0=half0, aic=int(self.aic),
                    lookahead=self._lookahead_native, foresight=self.foresight, speculation_max=self._speculation_native,
                    tentative_tapes=int(self.tentative_tapes), test_rewinds=int(self._test_rewinds),
                    device_chain_columns=self.device_chain_columns, finish_threads=self.host.pool.finish_threads,
                    flip_guess=self._flip_guess, device_rows=int(self.host.device_rows))
                # G2 of the kill tests' models on the device (Jacobi in LDS) where the search runs on one process and the
                # model fits the kernel; wider models, replicated searches and FOKL_EIGH=host keep LAPACK on the pool's threads
                self.spectral_engine = None
                if not self.allreduce and not self.candidate_sharded:       # (G2 on the device: single-process searches)
                    self.spectral_engine = spectral_engine_for(getattr(getattr(self.backend, 'ctx', None), 'device', None))
                if self.spectral_engine is not None:
                    # hybrid: only jobs the device finishes before their kill test comes up (FOKL_DSPECTRAL_SLACK kernel
                    # durations ahead, default 1.5; FOKL_DSPECTRAL_LOOKAHEAD tests ahead at most, default 32); device: all
                    everything = os.environ.get('FOKL_EIGH') == 'device'
                    self.native.bind_spectral(self.spectral_engine,
                                              slack=0.0 if everything else float(os.environ.get('FOKL_DSPECTRAL_SLACK', '-1')),
                                              lookahead=int(os.environ.get('FOKL_DSPECTRAL_LOOKAHEAD', '-1')))
                # Kill tests' G2 from the eigenpairs of the model each is tested against (secular equation + one product, a fifth
                # of a decomposition's time): FOKL_EIGH_UPDATE = columns of the smallest such parent (default 8; 0: every model
                # is decomposed afresh), at most FOKL_EIGH_UPDATE_DEPTH (default 6) such steps from a decomposition.  Not where
                # several ranks repeat one search: which models are derived depends on what was requested ahead, and the ranks
                # must agree to the last bit.
                update_from = int(os.environ.get('FOKL_EIGH_UPDATE', '8'))
                # (replicated searches: under direct decisions G2 is requested for accepted models only, at the moment of the
                # decision -- which models are derived is then a function of the decisions, the same on every rank; while the
                # loop orders G2 ahead along a path it predicts, FOKL_KILL_DECIDE=g2, it depends on timing: not there)
                direct_wanted = os.environ.get('FOKL_KILL_DECIDE', 'direct') == 'direct' and \
                    getattr(self, 'allow_direct_decisions', True)
                if update_from > 0 and getattr(self.host.pool, 'has_dgemm', False) \
                        and (direct_wanted or not (self.allreduce or self.candidate_sharded)) \
                        and os.environ.get('FOKL_EIGH_SIGNS', 'canonical') != 'lapack':
                    # FOKL_LOOKAHEAD_DERIVED (default 0: the ordinary look-ahead): a deeper G2 window while derivation is on, in
                    # sub-stages of fewer than 192 columns.  A chain of derivations advances slower than the loop tests, and 24
                    # deep keeps more of its pieces running side by side: configs[2] 40.0-40.1 ms per fit against 40.3-43.0,
                    # waiting for G2 8.0 -> 6.9 ms -- but configs[3] 0.69-1.07 s against 0.62-0.63 (its narrow sub-stages;
                    # not understood), so it stays a knob
                    derived_ahead = 0 if 'FOKL_LOOKAHEAD' in os.environ else _bounded_lookahead(
                        'FOKL_LOOKAHEAD_DERIVED', os.environ.get('FOKL_LOOKAHEAD_DERIVED', '0'))
                    # (depth: with the kill tests decided at once nobody waits for a link of the chain of derived models any more --
                    # their G2 only feeds chains whose statistics are confirmed later -- so the chain may be long: 24 steps cost 27 ms
                    # of spectral CPU per configs[2] fit against 41 at 6, same fit time, same bits of the returned draws; while
                    # the loop waits for G2, FOKL_KILL_DECIDE=g2, a step every 0.1-0.2 ms is what it waits for: 6)
                    depth_default = '24' if os.environ.get('FOKL_KILL_DECIDE', 'direct') == 'direct' and \
                        getattr(self, 'allow_direct_decisions', True) else '6'
                    self._update_args = (update_from, int(os.environ.get('FOKL_EIGH_UPDATE_DEPTH', depth_default)),
                                         derived_ahead)
                    self.native.set_update(*self._update_args)
                    self.stats['eigh_update_from'] = update_from
                # Kill tests' BICs from the sub-stage's least-squares model downdated column by column (microseconds on the
                # search thread; G2 then only feeds the accepted models' chains and confirms the BIC) instead of from G2 of every
                # trial model, which the loop had to wait for: FOKL_KILL_DECIDE = direct (default) | g2.  A fit repeated after a
                # misprediction runs with g2.
                decide = os.environ.get('FOKL_KILL_DECIDE', 'direct')
                if decide not in ('direct', 'g2'):
                    raise ValueError("FOKL_KILL_DECIDE must be direct or g2")
                if not getattr(self, 'allow_direct_decisions', True):
                    decide = 'g2'
                self.native.set_decide(1 if decide == 'direct' else 0, float(os.environ.get('FOKL_KILL_DECIDE_TOL', '0')))
                self.stats['kill_decide'] = decide
                if self.allreduce or self.candidate_sharded:
                    self.native.set_deterministic(True)
            except BaseException:
                # (a bad FOKL_EIGH / FOKL_KILL_DECIDE value, a device that went away: the pool's threads -- the walker still
                # attached to the stream --, the L3 pinning and the chain engine's binding do not outlive this search)
                if self.native is not None:
                    self.native.close()
                    self.native = None
                self.host.close()
                self.host = None
                if self._head is not None:          # the block launched ahead is fetched: the context outlives this search
                    self._ahead_block(self._head[1])
                    self.pool.give(self._head[1]['slots'])
                    self._head = None
                raise
        self.stats['search_driver'] = 'native' if self.native is not None else 'python'
        if self.native is None and self.host is not None and os.environ.get('FOKL_SEARCH', 'native') != 'python':
            import warnings
            warnings.warn("the native search driver is not available for this search (FOKL_KILL_BIC / FOKL_SEARCH_DIST / a "
                          "stand-in chain engine): it runs on the Python statement of the loop, several times slower -- "
                          "fit_stats['search_driver'] says which", RuntimeWarning)
        _mark('pool_up')
        dgemm0 = _capi.device_dgemm_stats() if self.host is not None else (0, 0, 0)
        self.stats['t_pool_up'] = time.perf_counter() - t_begin_run
        # which arithmetic produced the draws (the stream is numpy's either way): libmvec's vector log or libm's scalar
        # one for the normals finished on the host, host threads or the device for the kill tests' chains
        self.stats['finish_log'] = os.environ.get('FOKL_FINISH_LOG', 'fast')
        self.stats['chain_mode'] = 'device' if self.chain_engine is not None else 'host'
        self.stats['eigh_mode'] = (os.environ.get('FOKL_EIGH', 'host') if getattr(self, 'spectral_engine', None) is not None
                                   and self.native is not None else 'host')
        t_up = time.perf_counter()
        try:
            return self._run()
        finally:
            _mark('run_end')
            t_down = time.perf_counter()
            self.stats['t_search_body'] = t_down - t_up
            if self.native is not None:
                # what the native side counted (its evaluations, waits and guesses) joins this side's counters; the
                # trace of evaluations is its (model evaluations are recorded there too: one sequence)
                for key, value in self.native.stats().items():
                    self.stats[key] = self.stats.get(key, 0) + value
                self.trace = self.native.trace()
                self._outcomes, self._retiring = [], []
                self.native.close()         # sends back what is on order, waits for what is in flight
                self.native = None
                self.stats['t_teardown_search'] = time.perf_counter() - t_down
            if self.host is not None:
                # every device chain of this search gives its slot back (the engine outlives the fit); idempotent, and
                # a chain that has not run yet is waited for -- its tape is committed, so it will
                for job in self._device_jobs:
                    job.release()
                self._device_jobs = []
                self._zombies.clear()
                self._unverified.clear()
                if self.chain_engine is not None:
                    now = self.chain_engine.stats()               # the engine outlives the fit: this fit's share
                    self.stats.update({'dchain_' + k: v - self._dchain_stats0.get(k, 0) for k, v in now.items()})
                t_close = time.perf_counter()
                busy = self.host.close()     # all requested tapes are recorded -> the stream ends where it must
                self.stats['t_teardown_pool'] = time.perf_counter() - t_close
                dgemm1 = _capi.device_dgemm_stats()
                self.stats.update(eigen_update_products=dgemm1[0] - dgemm0[0], eigen_update_products_on_device=dgemm1[1] - dgemm0[1],
                                  eigen_update_products_fell_back=dgemm1[2] - dgemm0[2])
                self.stats.update(pool_bulk_s=busy.get('bulk', 0.0), walker_wait_s=busy.get('walker_wait', 0.0),
                                  stream_segments=busy.get('stream_segments', 0),
                                  gamma_attempts_exact=busy.get('gamma_attempts_exact', 0),
                                  pool_noise_s=busy['noise'], pool_chain_s=busy['chain'],
                                  pool_finish_s=busy['finish'], pool_spectral_s=busy['spectral'],
                                  noise_queue_wait_s=busy['noise_queue_wait'],
                                  noise_verdict_wait_s=busy['noise_verdict_wait'],
                                  spectral_remote=self.host.remote_results, exchanges=self.host.exchanges,
                                  spectral_submitted=self.stats.get('spectral_submitted', 0) + self.host.spectral_submitted)
                self.host = None
                self.stats['t_teardown'] = time.perf_counter() - t_down
                _mark('pool_down')
                _flush_marks()

    def _native_wanted(self):
        """The native search core drives single-process searches whose kill-test BICs come from the Gram; replicated
        searches (rows or candidates over ranks), FOKL_KILL_BIC=device|check and stand-in chain engines (tests) keep the
        Python statement of the loop.  FOKL_SEARCH=python forces that one."""
        if not self._native_planned():
            return False
        if self.chain_engine is not None and not isinstance(self.chain_engine, _capi.DeviceChainEngine):
            return False
        return True

    def _native_planned(self):
        """What can be said before the host pipeline is up.  Searches replicated over ranks run the native driver too
        (round 5; FOKL_SEARCH_DIST=python keeps the Python loop for them, with its G2 jobs dealt over the ranks)."""
        from . import host_pipeline
        if os.environ.get('FOKL_SEARCH', 'native') == 'python':
            return False
        if (self.allreduce or self.candidate_sharded) and os.environ.get('FOKL_SEARCH_DIST', 'native') == 'python':
            return False
        if self.kill_bic not in ('auto', 'gram'):
            return False
        return host_pipeline._chain_engine_factory is None

    def _patterns(self):
        """(stage ind, indvec) of every sub-stage in the reference's order (FR:1602-1613, FR:1722-1747)."""
        m = self.m
        sett = 1 if m == 1 else (3 if self.way3 else 2)   # FR:1595-1600
        ind = 1
        while True:
            indvec = deal_indvec(ind, m, sett)
            while True:
                yield ind, list(indvec)
                if not advance_indvec(indvec, m, self.way3):
                    break
            ind += 1
            if ind > self.n_phis:                  # FR:1747
                return

    # -- candidates sharded over ranks, native driver (round 5) ---------------------------------------------
    # Every rank holds all rows and repeats the same search.  With the kill tests decided from the sub-stage's Gram nothing
    # of a kill test depends on N any more; what does is the forward step itself: the T candidate terms' columns (K1) and
    # their Gram rows against [model | candidates | y] (K2, 2 N T (P + T + 1) flops -- the launch the matrix pipe bounds).
    # Each rank takes the Gram rows of its share of the candidates -- every rank builds all T columns, 8 N T bytes of
    # stores, since a candidate's row needs every other candidate's column -- and ONE all-gather per forward step brings the
    # per-candidate rows (from which each candidate's BIC follows on the host) to every rank: north_star's split.
    def _candidate_split(self):
        return self._replicated_native and self.candidate_sharded and not self.allreduce

    def _share_of(self, slots):
        """This rank's share of a forward step's candidate columns, padded to the common length by repeating its last
        column (padding rows are dropped after the gather).  -> (slots of the share, common length)"""
        world, rank = self.comm.world, self.comm.rank
        share = -(-len(slots) // world)
        mine = list(slots[rank * share:(rank + 1) * share])
        mine += [slots[-1]] * (share - len(mine))
        return mine, share

    def _gather_rows(self, local, total, share):
        """The candidates' Gram rows of all ranks in candidate order: one all-gather of share x columns doubles."""
        t0 = time.perf_counter()
        parts = self.comm.allgather(np.ascontiguousarray(local).reshape(-1)).reshape(self.comm.world, share, -1)
        self.stats['candidate_gathers'] = self.stats.get('candidate_gathers', 0) + 1
        self.stats['t_candidate_gather'] = self.stats.get('t_candidate_gather', 0.0) + time.perf_counter() - t0
        return np.concatenate([parts[r, :max(0, min(share, total - r * share))] for r in range(self.comm.world)], axis=0)

    def _gram_of_new_columns(self, new_slots, col_slots):
        """Gram rows of a forward step's candidate columns (blocking form)."""
        if not self._candidate_split():
            return self.backend.gram(new_slots, col_slots, self.allreduce)
        mine, share = self._share_of(new_slots)
        return self._gather_rows(self.backend.gram(mine, col_slots, False), len(new_slots), share)

    def _arrangements(self, indvec):
        """distinct_arrangements(indvec), enumerated once per search (a sub-stage's pattern is looked at when it is the coming
        one -- how many terms will it bring -- and again when its columns are built)."""
        key = tuple(float(v) for v in indvec)
        hit = self._arrangement_memo.get(key)
        if hit is None:
            hit = self._arrangement_memo[key] = distinct_arrangements(indvec)
            hit.setflags(write=False)
        return hit

    def _build_ahead(self, indvec, active_slots):
        """K1 + K2 of a coming sub-stage while the current one is still being decided: its columns, and their Gram
        block against every column that can still be in the model then (all of the current sub-stage's) and y."""
        vecs = self._arrangements(indvec)
        slots = self.pool.take(vecs.shape[0])
        self.backend.build_terms(vecs.astype(np.int32), slots)
        self.stats['terms_physical'] += vecs.shape[0]
        ahead = dict(indvec=indvec, vecs=vecs, slots=slots, over=len(active_slots))
        if hasattr(self.backend, 'gram_launch'):
            # the driver does not wait: the block is fetched when somebody looks at it (_ahead_block)
            if self._candidate_split():
                mine, share = self._share_of(slots)
                ahead['share'] = share
                ahead['pending'] = self.backend.gram_launch(mine, active_slots + slots + [SLOT_Y], False)
            else:
                ahead['pending'] = self.backend.gram_launch(slots, active_slots + slots + [SLOT_Y], self.allreduce)
        else:
            ahead['block'] = self._gram_of_new_columns(slots, active_slots + slots + [SLOT_Y])
        return ahead

    def _ahead_block(self, ahead):
        if 'block' not in ahead:
            block = self.backend.gram_fetch(ahead.pop('pending'))
            if 'share' in ahead:                    # this rank's share of the candidates: the others' rows are gathered
                block = self._gather_rows(block, len(ahead['slots']), ahead.pop('share'))
            ahead['block'] = block
        return ahead['block']

    @staticmethod
    def _extend_gram(gram, keep, block, block_kept, over):
        """Gram of [columns `keep` of gram's model] + [new columns] + y.  gram: (A + 1)^2 with y last; block: the new
        columns' rows of the Gram, whose columns block_kept belong to the kept columns, over .. over + vm - 1 to the
        new columns themselves and over + vm to y (_build_ahead: block_kept = keep, over = A; a K2 call on exactly
        the kept columns: block_kept = 0 .. len(keep) - 1, over = len(keep))."""
        A = gram.shape[0] - 1
        n_prev, vm = len(keep), block.shape[0]
        A2 = n_prev + vm
        out = np.empty((A2 + 1, A2 + 1))
        out[:n_prev, :n_prev] = gram[np.ix_(keep, keep)]
        out[:n_prev, A2] = out[A2, :n_prev] = gram[keep, A]
        out[A2, A2] = gram[A, A]
        cross = block[:, block_kept]
        out[n_prev:A2, :n_prev] = cross
        out[:n_prev, n_prev:A2] = cross.T
        out[n_prev:A2, n_prev:A2] = block[:, over:over + vm]
        out[n_prev:A2, A2] = out[A2, n_prev:A2] = block[:, over + vm]
        return out

    def _run_native(self):
        """csrc/fokl_run.cpp: the sub-stage loop as native code on this search's NativeSearch; this side keeps what follows
        the loop (_finish)."""
        nrun, self._nrun = self._nrun, None
        if self._update_args is not None:
            nrun.set_update(*self._update_args)
        try:
            nrun.search(self.native)
            mtx, evs, per_substage, best_h, last_h, st = nrun.result()
        finally:
            nrun.close()
        self.substage_stats = per_substage
        for key in ('terms_physical', 'substages', 'forecasts_used', 'resid_matrix_free'):
            self.stats[key] += int(st[key])
        self.stats['forecasts_early'] = self.stats.get('forecasts_early', 0) + int(st['forecasts_early'])
        self.stats['t_resid'] += st['t_resid']
        self.stats['substage_loop'] = 'native'
        phase = self.stats.setdefault('phases', dict(prepare=0.0, model=0.0, statistics=0.0, tests=0.0, wrap_up=0.0))
        for key in phase:
            phase[key] += st['phase_' + key]
        betas = NativeOutcome(self, self.native, best_h)
        last = betas if last_h == best_h else NativeOutcome(self, self.native, last_h)
        self._outcomes = [betas] if last is betas else [betas, last]
        if self.chain_engine is not None and hasattr(self.chain_engine, 'flush'):
            self.chain_engine.flush()              # the last kill tests' chains go out now, not when their batch has aged
        return self._finish(betas, mtx, evs, last)

    def _run(self):
        if getattr(self, '_nrun', None) is not None:
            if self.native is not None:
                return self._run_native()
            self._nrun.close()                      # the native search did not come up after all: this file's loop from scratch
            self._nrun = None
        m, n = self.m, self.n
        draws = self.draws
        half1 = int(math.ceil(draws / 2 + 1))      # FR:1656
        half0 = int(math.ceil(draws / 2))          # FR:1658, FR:1671

        # Gram of [ones, y] seeds the cache: n, sum y, y'y
        model_slots = []                            # device slots of accepted terms (aligned with damtx rows)
        damtx = np.zeros((0, m))
        head = getattr(self, '_head', None)
        self._head = None
        if head is not None:
            base = head[0]
        else:
            base = self.backend.gram([SLOT_ONES, SLOT_Y], [SLOT_ONES, SLOT_Y], self.allreduce)
        gram = np.array(base, dtype=np.float64)     # Gram over [ones] + model columns + y
        keep = [0]

        evs = np.array([])
        betas = mtx = None
        last = None
        last_damtx = damtx
        greater = 0
        patterns = self._patterns()
        pattern = next(patterns)
        ahead = head[1] if head is not None else None   # the coming sub-stage, built early (pipelined search only)
        forecasts = {}                              # survivors' slots -> (G2 job of the coming model, its Gram)
        look_ahead = self.host is not None and self.foresight > 0

        phase = self.stats.setdefault('phases', dict(prepare=0.0, model=0.0, statistics=0.0, tests=0.0, wrap_up=0.0))
        tick = time.perf_counter()

        def lap(name):
            # where the driver thread's time goes, sub-stage by sub-stage (waits included)
            nonlocal tick
            now = time.perf_counter()
            phase[name] += now - tick
            tick = now

        while pattern is not None:
            ind, indvec = pattern
            pattern = next(patterns, None)
            n_prev = 1 + len(model_slots)
            spectral_job = None
            if ahead is not None:
                vecs, new_slots = ahead['vecs'], ahead['slots']
                hit = forecasts.pop(tuple(model_slots), None)
                if hit is not None:
                    spectral_job, gram = hit
                    self.stats['forecasts_used'] += 1
                else:
                    gram = self._extend_gram(gram, keep, self._ahead_block(ahead), keep, ahead['over'])
                self._ahead_block(ahead)                    # a launched block is always fetched
                if self.native is not None:
                    self.native.clear_forecasts()           # before their Grams go: waits for G2 jobs nobody else holds
                ahead, forecasts = None, {}
            else:
                # K1 + K2: build the new columns once, extend the Gram
                vecs = self._arrangements(indvec)
                new_slots = self.pool.take(vecs.shape[0])
                self.backend.build_terms(vecs.astype(np.int32), new_slots)
                self.stats['terms_physical'] += vecs.shape[0]
                block = self._gram_of_new_columns(new_slots, [SLOT_ONES] + model_slots + new_slots + [SLOT_Y])
                gram = self._extend_gram(gram, keep, block, list(range(n_prev)), n_prev)
            _mark('substage', f"{ind} hit={int(spectral_job is not None)}")
            vm = vecs.shape[0]
            damtx = np.append(damtx, vecs, axis=0)
            dam = damtx.shape[0]
            self._set_active_terms(damtx)
            active_slots = [SLOT_ONES] + model_slots + new_slots
            A = len(active_slots)
            pipelined = self.host is not None
            # G2 of this model may be there already (started while the sub-stage before was being decided): the first
            # kill tests are guessed from it -- their G2 jobs and tapes -- before anything else happens
            early, then = None, [A - 1] if vm > 0 and A > 1 else []
            vm_next = None
            if pipelined and pattern is not None:
                vm_next = self._arrangements(pattern[1]).shape[0]
            if pipelined and self.lookahead > 0 and spectral_job is not None and getattr(
                    spectral_job, 'done', lambda: False)():
                early, then = self._guess_first_tests(gram, spectral_job.wait(), vm, before_model=True,
                                                      spectrum=spectral_job if self.native is not None else None,
                                                      vm_next=vm_next)

            def build_next(coming=pattern, active=active_slots):
                nonlocal ahead
                self._release_retired()
                if look_ahead and coming is not None:
                    ahead = self._build_ahead(coming[1], active)

            # the tape of the first kill test, whichever proposal that will be (every first test has A - 1 columns), is
            # ordered together with the model's own (which may be on its way already, see _kill_tests_pipelined): the
            # noise thread goes straight on instead of idling until the model's chain has finished and its statistics
            # have ordered the proposals
            lap('prepare')
            def first_tests_now(spectrum, gram=gram, vm=vm):
                # G2 of the model has just arrived (the native search's model_begin waited for it): the likely first tests'
                # G2 jobs and tapes go out now, under the device's residual pass, not after it
                nonlocal early
                if early is None and self.lookahead > 0 and vm > 0:
                    own = NativeSpectrum(self.native, self.native.spectrum_retain(spectrum), gram)
                    early, _ = self._guess_first_tests(gram, None, vm, spectrum=own, vm_next=vm_next)

            full = self._evaluate(gram, active_slots, np.arange(A), n_prev, kill=False, spectral_job=spectral_job,
                                  overlap=None if pipelined else build_next, then=then,
                                  after_begin=first_tests_now if pipelined and self.native is not None else None,
                                  new_terms=vm)
            best = full
            ev = full.ev
            _mark('full_evaluated', str(A))
            lap('model')

            def foresee(pred_killed, gram=gram, active=active_slots, A=A):
                # G2 of the coming sub-stage's model if the kill tests end as predicted (at most two guesses)
                if ahead is None or len(forecasts) >= 2:
                    return
                gone = set(pred_killed)
                keep_pred = [c for c in range(A) if c not in gone]
                key = tuple(active[c] for c in keep_pred[1:])
                if key not in forecasts:
                    g = self._extend_gram(gram, keep_pred, self._ahead_block(ahead), keep_pred, ahead['over'])
                    forecasts[key] = (self._spectral(g, np.arange(g.shape[0] - 1, dtype=np.int32)), g)
                    if self.native is not None:
                        self.native.register_forecast(key, forecasts[key][0].h, g[-1, -1])

            guesses = {}

            def coming_tests(pred_killed, active=active_slots, A=A):
                # kill tests the coming sub-stage will probably run if this one ends with that kill set -- known once
                # G2 of its model (foresee) is there
                key = tuple(active[c] for c in range(1, A) if c not in pred_killed)
                if key not in guesses:
                    hit = forecasts.get(key)
                    if hit is None or not getattr(hit[0], 'done', lambda: False)():
                        return None
                    guesses[key] = len(self._likely_first_tests(hit[0].wait(), vm_next))
                return guesses[key]

            def chain_coming(pred_killed, active=active_slots, A=A):
                # the chain of the coming sub-stage's model, should this one end with that kill set
                hit = forecasts.get(tuple(active[c] for c in range(1, A) if c not in pred_killed))
                if hit is not None:
                    self._chain_ahead(hit[0], A - len(pred_killed) + vm_next, hit[1][-1, -1])

            if early is None:
                early = {}
                if pipelined and self.lookahead > 0:
                    # guess the first tests now, while the chain of the sub-stage model is still running
                    own = None
                    if self.native is not None:
                        own = NativeSpectrum(self.native, self.native.outcome_spectrum(full.h), gram)
                    early, _ = self._guess_first_tests(gram, full, vm, siglik=full.siglik, spectrum=own, vm_next=vm_next)
            # K1 + K2 of the coming sub-stage now, while this thread would only wait for the model's chain (its BIC pass
            # has left the device): inside the kill tests -- where it used to hide behind the first test's decomposition --
            # a derived G2 answers in a fifth of the time this takes.  FOKL_BUILD_AHEAD=tests: there, as before
            build_in_tests = build_next
            if pipelined and self.native is not None and os.environ.get('FOKL_BUILD_AHEAD', 'model') != 'tests':
                build_next()
                build_in_tests = None

            # While this thread would only wait for the model's chain: G2 of the COMING sub-stage's model for the kill set the
            # least-squares downdate predicts (a decomposition of up to 1.2 ms that otherwise starts when the tests are over
            # and is what the next sub-stage then waits for).  Only in the time that wait takes: as soon as the coming
            # columns' Gram block has arrived -- if the chain is there first, the tests go ahead and say it exactly.
            if (self._forecast_early and pipelined and self.native is not None and ahead is not None
                    and 'pending' in ahead and 'share' not in ahead and self._predicted_kills is not None
                    and hasattr(self.backend, 'gram_ready') and not forecasts):
                # (a bounded look, not a wait: a Gram block that is late leaves this to the tests -- and processes that share
                # a CPU quota, FOKL_SYNC=blocking, do not spin here at all)
                polls = 0 if os.environ.get('FOKL_SYNC') == 'blocking' else self._forecast_polls
                while polls > 0 and not full.chain_ready():
                    polls -= 1
                    if self.backend.gram_ready():
                        foresee(self._predicted_kills)
                        self.stats['forecasts_early'] = self.stats.get('forecasts_early', 0) + 1
                        break
            self._predicted_kills = None

            # statistics of the new terms (FR:1656-1664)
            if self.native is not None and os.environ.get('FOKL_STATS', 'native') != 'numpy':
                # (one native pass over the draws instead of a matmul and five numpy reductions: 0.25 -> 0.05 ms per sub-stage
                # of the driver's time, right where the walker waits for the proposals' order)
                mean_abs, rel_std = self.native.outcome_new_term_stats(full.h, np.arange(dam - vm + 1, dam + 1), half0, half1)
            else:
                tail = full.beta_columns(np.arange(dam - vm + 1, dam + 1), half0)     # draws half0 .. of the new terms
                mean_abs = np.abs(np.mean(tail[half1 - half0:], axis=0))
                rel_std = np.divide(np.std(tail[half1 - half0:], axis=0), np.abs(np.mean(tail, axis=0)))
            _mark('full_statistics')
            lap('statistics')
            # (what orders and gates this sub-stage's kill tests, in the order of the interaction matrix: kept for whoever
            # wants to hold the chain's numbers -- not only the decisions they lead to -- against another implementation's)
            self.substage_stats.append(dict(mean_abs=np.array(mean_abs), rel_std=np.array(rel_std)))
            order = np.argsort(mean_abs)
            cand_col = np.arange(dam - vm + 1, dam + 1)[order]     # active-column index of each proposal
            mean_abs, rel_std = mean_abs[order], rel_std[order]

            # sequential kill tests (FR:1666-1690): proposals in ascending |mean beta|
            killed = []                                           # active-column indices removed so far
            evmin = ev
            if self.host is None:
                for i in range(vm):
                    # FR:1670-1671.  The second clause needs the intercept draws of the model accepted so far
                    # (Python's short-circuit `or` / `and`, exactly as in the reference's expression).
                    if rel_std[i] > self.threshstdb or (
                            rel_std[i] > self.threshstda and
                            mean_abs[i] < self.threshav * self._intercept_scale(best, half0)):
                        trial = set(killed)
                        trial.add(int(cand_col[i]))
                        res = self._evaluate(gram, active_slots, self._columns_without(A, trial), n_prev, kill=True)
                        if res.ev < evmin:
                            killed = sorted(trial)
                            evmin = res.ev
                            best = res
            else:
                if self.native is not None and self._update_args is not None and self._update_args[1] > 6:
                    # The sub-stage after which the stop rule may end the search (FR:1708-1718: `greater` has reached the
                    # tolerance): the chains of its accepted models are what the search's last act -- confirming the guessed
                    # decisions -- waits for, and those chains wait for G2.  Short pieces of derived models there (six steps
                    # behind a decomposition: many pieces side by side), long ones everywhere else (nobody waits: less CPU).
                    last_chance = evs.size > 0 and greater >= self.tolerance
                    self.native.set_update(self._update_args[0], 6 if last_chance else self._update_args[1],
                                           self._update_args[2])
                killed, evmin, best = self._kill_tests_pipelined(gram, active_slots, n_prev, cand_col, mean_abs,
                                                                 rel_std, best, half0, foresee, early, vm_next,
                                                                 build_in_tests, coming_tests, chain_coming)
            ev = evmin
            _mark('tests_over', str(len(killed)))
            lap('tests')

            # commit the surviving columns (FR:1691-1695)
            keep = [c for c in range(A) if c not in set(killed)]
            if killed:
                damtx = np.delete(damtx, [c - 1 for c in killed], axis=0)
                self.pool.give([active_slots[c] for c in killed])
            model_slots = [active_slots[c] for c in keep[1:]]
            self.stats['substages'] += 1
            last, last_damtx = best, damtx

            if self.console:
                print([ind, float(ev)])

            # best-model bookkeeping and stop rule (FR:1701-1721)
            if evs.size > 0:
                if ev < np.min(evs):
                    betas, mtx, greater = best, damtx, 1
                    evs = np.append(evs, ev)
                elif greater < self.tolerance:
                    greater += 1
                    evs = np.append(evs, ev)
                else:
                    evs = np.append(evs, ev)
                    break
            else:
                greater += 1
                betas, mtx = best, damtx
                evs = np.append(evs, ev)
            self._retire(betas, best)
            lap('wrap_up')

        lap('wrap_up')
        if self.chain_engine is not None and hasattr(self.chain_engine, 'flush'):
            self.chain_engine.flush()              # the last kill tests' chains go out now, not when their batch has aged
        if self.host is not None:                  # the search stopped: tapes on order for a sub-stage that does not come
            self._drop_speculation()
        if self.native is not None:
            self.native.clear_forecasts()
        if ahead is not None:                      # ... and the columns built ahead are not needed
            self._ahead_block(ahead)
            self.pool.give(ahead['slots'])

        if self.gimmie:                            # FR:1751-1753
            betas, mtx = last, last_damtx
        return self._finish(betas, mtx, evs, last)

    def _finish(self, betas, mtx, evs, last):
        """What follows the sub-stage loop: the returned model's draws, the confirmation of every guessed decision."""
        # The returned model's draws are formed while the last confirmations are still on their way (eigenpairs of the last
        # sub-stage's accepted models, then their chains on the device: 3-4 ms in which this thread only waits); should a
        # confirmation fail, Misprediction discards them with everything else.
        t0 = time.perf_counter()
        self._verify(block=False)                  # (starts the chains whose eigenpairs have arrived)
        t1 = time.perf_counter()
        out_betas = betas.betas[-self.draws_keep::, :]
        t2 = time.perf_counter()
        self.stats['t_final_draws'] = t2 - t1
        self._verify(block=True)                   # every decision taken from a guess is confirmed before anything returns
        self.stats['t_final_verify'] = (t1 - t0) + (time.perf_counter() - t2)
        for keep_alive in (betas, last):           # the returned draws are on the host now: the device slots go back
            if getattr(keep_alive, 'on_device', False):
                keep_alive.release()
        lamb = np.sort(np.asarray(getattr(betas, 'lamb', ()), dtype=np.float64))
        if lamb.shape[0] > 1:
            # conditioning of the returned model's eigenproblem: betas = w Q', and an eigenvector moves by about
            # eps ||XtX|| / gap under a rounding-level change of XtX -- what the draws' distance to another
            # implementation's is made of (tools/draw_margin.py)
            self.stats.update(final_cond=float(lamb[-1] / lamb[0]),
                              final_eps_norm_over_gap=float(2.220446049250313e-16 * lamb[-1] / np.min(np.diff(lamb))))
        return out_betas, np.array(mtx, dtype=np.float64), evs
