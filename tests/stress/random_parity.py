"""Random small problems through the device path (threaded search) against the oracle (in-line search, CPU restatement of
the reference): selected model, BIC trace, draws and the final numpy stream must agree (development aid / stress run;
problems whose model outgrows the data -- terms >= rows / 3 -- are reported but not compared beyond the model)."""
import os, sys, warnings, hashlib
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fokl_gpy_amd import FoKLRoutines, getKernels
from helpers import OracleBackend
warnings.simplefilter('ignore')
spl = getKernels.table_to_phis(np.load(os.path.join(ROOT, 'tests', 'golden', 'spline_phis.npz'))['table'])

MAX_ROWS = 3000


def problem(seed, max_rows=None):
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(300, max_rows or MAX_ROWS)), int(rng.integers(1, 6))
    x = rng.random((n, m))
    y = np.sin(3 * x[:, 0]) + (x[:, 1 % m] * x[:, 2 % m] if m > 1 else 0) + 0.1 * rng.standard_normal(n)
    kw = dict(burnin=int(rng.integers(20, 100)), draws=int(rng.integers(20, 100)), tolerance=int(rng.integers(1, 4)),
              way3=bool(rng.integers(0, 2)) and m >= 3, aic=bool(rng.integers(0, 2)), UserWarnings=False, ConsoleOutput=False)
    if rng.integers(0, 2):
        kw.update(kernel=1)
    else:
        kw.update(kernel=0, phis=spl)
    return x, y, kw


TALLY = dict(device_chains=0, guessed=0, guesses_verified=0, guess_waits=0, searches_repeated=0, kill_tests=0, direct_tests=0,
             chains_cancelled=0, spectral_submitted=0, spectral_updated=0)


def fit(x, y, kw, seed, oracle):
    previous = os.environ.get('FOKL_NOISE_PIPELINE')
    os.environ['FOKL_NOISE_PIPELINE'] = '0' if oracle else '1'
    try:
        model = FoKLRoutines.FoKL(**kw)
        if oracle:
            model._backend_override = OracleBackend()
        np.random.seed(seed + 11)
        b, mtx, evs = model.fit(x, y, clean=True)
        st = np.random.get_state()
    finally:
        if previous is None:
            del os.environ['FOKL_NOISE_PIPELINE']
        else:
            os.environ['FOKL_NOISE_PIPELINE'] = previous
    if not oracle:
        for key in TALLY:
            TALLY[key] += model.fit_stats.get(key, 0)
    return b, mtx, evs, hashlib.sha256(st[1].tobytes()).hexdigest() + str(st[2:])


def compare(seed, max_rows=None):
    """-> (ok, outgrown, terms of the oracle's model, terms of the device path's model, rows x inputs, kernel)"""
    x, y, kw = problem(seed, max_rows)
    ref, got = fit(x, y, kw, seed, True), fit(x, y, kw, seed, False)
    same_model = ref[1].shape == got[1].shape and np.array_equal(ref[1], got[1])
    grown = ref[1].shape[0] >= x.shape[0] / 3
    ok = same_model and (grown or (ref[3] == got[3] and np.allclose(got[2], ref[2], rtol=1e-9) and
                                   np.allclose(got[0], ref[0], rtol=1e-6, atol=1e-8 * np.abs(ref[0]).max())))
    return ok, grown, ref[1].shape[0], got[1].shape[0], x.shape, kw['kernel']


if __name__ == '__main__':
    if len(sys.argv) > 3:
        MAX_ROWS = int(sys.argv[3])
    bad = 0
    import signal

    class TooSlow(Exception):
        pass

    def alarm(*_):
        raise TooSlow()

    signal.signal(signal.SIGALRM, alarm)
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        # (a problem whose model outgrows its data keeps the oracle's in-line loop busy for a quarter of an hour -- seed 1672:
        # hundreds of columns per gibbs() call; it would be reported as outgrown and not compared anyway)
        signal.alarm(int(os.environ.get('RANDOM_PARITY_DEADLINE', '120')))
        try:
            ok, grown, t_ref, t_got, shape, kernel = compare(seed)
        except TooSlow:
            print(seed, 'SKIPPED: over the deadline (a model that outgrows its data)', flush=True)
            continue
        finally:
            signal.alarm(0)
        if not ok:
            bad += 1
        print(seed, 'rows', shape, 'kernel', kernel, 'terms', t_ref, t_got, 'OK' if ok else 'MISMATCH',
              '(outgrown)' if grown else '', flush=True)
    print('mismatches', bad, 'device path:', TALLY)
