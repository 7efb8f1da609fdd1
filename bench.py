#!/usr/bin/env python3
"""
Benchmark of the forward-selection hot path on MI355X -- BASELINE.json's metric:

    candidate-terms/sec (basis build + Gibbs + BIC), N = 1e6, M = 8

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4}] [--mode fits|rows|candidates]
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One *step* = one complete forward-selection fit (FoKL.fit's search: every sub-stage's basis build, Gram,
Gibbs chains, BIC and kill tests) over a synthetic BASELINE configuration (default configs[2]: N = 1e6 rows,
M = 8 inputs, Bernoulli-polynomial kernel, 2-way interactions, reference-default hyper-parameters: burnin 1000,
draws 1000, tolerance 3) with the normalised inputs already resident in HBM when the timed region starts.
--config 4: a step is --fits-per-step independent fits (N = 1e5, M = 8 each) on resident datasets, dealt over --procs
worker processes that share the GPU (throughput mode).

`value` = candidate terms / second, where the numerator is the reference-equivalent (logical) count: the sum
over all gibbs evaluations of the columns the reference builds for that evaluation (FoKLRoutines.py:1461),
tallied per call by the search driver.  `value_physical` counts what the GPU really built.

After the timed region the last fit of rank 0 is compared with the oracle-generated golden of the same workload
(tests/golden/cfg*.npz: interaction matrix and gibbs-call sequence exact, BIC trace 1e-9, draws 1e-9 of the column
scale, numpy stream equal): `parity_checked` / `parity` in the JSON line, exit code 3 on a mismatch.

N > 1 (--mode):
  fits        every rank fits its own dataset(s) (independent fits are the unit that shards without a data-path
              collective, BASELINE configs[4]); one RCCL all-gather of the per-rank counters; "weak" scaling.
  candidates  ONE fit; the RNG-free half of every model evaluation (eigen-decomposition + BIC of the speculative
              kill-test candidates) is dealt over the ranks and gathered with RCCL all-gathers (north_star's split;
              default for --config 3); "strong" scaling.
  rows        ONE dataset, rows sharded, Gram blocks / residual moments all-reduced on the device; "strong" scaling.
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIDE_FITS = int(os.environ.get('FOKL_BENCH_SIDE_FITS', '12'))   # fits per process in the throughput side measurement
THROUGHPUT_SPECULATION = '24'    # FOKL_SPECULATION of the processes that share a GPU (throughput mode; a fit alone: 48)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6300 GB/s achievable)
FP64_MFMA_PEAK_TFLOPS = 78.6   # dense fp64 matrix peak


# BASELINE.json's configs as concrete synthetic inputs: U[0,1) inputs from default_rng(seed), a smooth
# low-order response + 0.05 N(0,1); the fit's chain starts from np.random.seed(seed_fit).  `unit` selects one of the
# independent datasets of configs[4] (dataset seed 100 + unit, chain seed 1000 + unit).
CONFIGS = {
    1: dict(rows=100_000, inputs=4, kernel='Cubic Splines', seed=11, seed_fit=1000, fit={},
            label='configs[1]: synthetic N={rows}, M=4, Cubic Splines, burnin 1000 + draws 1000'),
    2: dict(rows=1_000_000, inputs=8, kernel='Bernoulli Polynomials', seed=12, seed_fit=1000, fit={},
            label='configs[2]: synthetic N={rows}, M=8, Bernoulli Polynomials, 2-way interactions, burnin 1000 + '
                  'draws 1000'),
    3: dict(rows=1_000_000, inputs=16, kernel='Bernoulli Polynomials', seed=13, seed_fit=1000, fit=dict(way3=True),
            phis_cap=3,
            label='configs[3]: synthetic N={rows}, M=16, Bernoulli Polynomials, 3-way interactions, stages capped '
                  'at 3 (phis[:3]: the uncapped search needs eigen-decompositions of 3 360-column models per kill '
                  'test), burnin 1000 + draws 1000'),
    4: dict(rows=100_000, inputs=8, kernel='Bernoulli Polynomials', seed=100, seed_fit=1000, fit={},
            label='configs[4]: independent fits of synthetic N={rows}, M=8 datasets (Bernoulli Polynomials, 2-way, '
                  'burnin 1000 + draws 1000), dataset seeds 100 + i, chain seeds 1000 + i'),
}


def make_workload(seed, n, m):
    """U[0,1) inputs; the response uses as many of the structures below as there are inputs.
    M = 4: sin(4 x0) + x1 x2 + 0.3 x3^2;  M = 8: ... + 0.5 x4 x5;  M = 16 (3-way config): sin(4 x0) + x1 x2 x3 +
    0.3 x4^2 + 0.5 x5 x6.  Always + 0.05 N(0,1)."""
    rng = np.random.default_rng(seed)
    x = rng.random((n, m))
    if m >= 16:
        y = np.sin(4.0 * x[:, 0]) + x[:, 1] * x[:, 2] * x[:, 3] + 0.3 * x[:, 4] ** 2 + 0.5 * x[:, 5] * x[:, 6]
    else:
        y = np.sin(4.0 * x[:, 0]) + x[:, 1] * x[:, 2] + 0.3 * x[:, 3] ** 2
        if m >= 6:
            y = y + 0.5 * x[:, 4] * x[:, 5]
    y = y + 0.05 * rng.standard_normal(n)
    return x, y


def config_workload(config, unit=0, rows=None):
    """-> (x, y, spec): the dataset of BASELINE configs[config] (unit-th dataset for configs[4]) and its description:
    spec = dict(rows, inputs, kernel, phis_cap, fit keywords, seed, seed_fit, label)."""
    spec = dict(CONFIGS[config])
    spec['rows'] = int(rows) if rows else spec['rows']
    spec['seed'] = spec['seed'] + unit
    spec['seed_fit'] = spec['seed_fit'] + unit
    spec['label'] = spec['label'].format(rows=spec['rows'])
    spec.setdefault('phis_cap', None)
    x, y = make_workload(spec['seed'], spec['rows'], spec['inputs'])
    return x, y, spec


GOLDENS = {  # (config, unit, rows, fit overrides) -> fixture made by tests/golden/make_config_golden.py
    (2, 0, 1_000_000): ('cfg2_n1e6_m8', {}),
    (4, 0, 100_000): ('cfg4_unit0_n1e5_m8', {}),
    (4, 5, 100_000): ('cfg4_unit5_n1e5_m8', {}),
    (4, 17, 100_000): ('cfg4_unit17_n1e5_m8', {}),
    (4, 29, 100_000): ('cfg4_unit29_n1e5_m8', {}),
    (4, 42, 100_000): ('cfg4_unit42_n1e5_m8', {}),
    (4, 63, 100_000): ('cfg4_unit63_n1e5_m8', {}),
    (2, 7, 1_000_000): ('cfg2_unit7_n1e6_m8', {}),
    (1, 0, 100_000): ('cfg1_n1e5_m4_splines', {}),
    (3, 0, 100_000): ('cfg3_n1e5_m16_way3', dict(burnin=30, draws=30)),
    (3, 0, 1_000_000): ('cfg3_n1e6_m16_way3', dict(burnin=30, draws=30)),
}
# configs[3] at its benchmarked N with chains long enough for the kill tests' Monte-Carlo statistics to settle (round 4:
# 250 + 250 draws, five hours of oracle): preferred over the 30 + 30 twin when the fixture is there
_LONG = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden', 'cfg3_n1e6_m16_way3_d250.npz')
if os.path.exists(_LONG):
    GOLDENS[(3, 0, 1_000_000)] = ('cfg3_n1e6_m16_way3_d250', dict(burnin=250, draws=250))


def kernel_and_phis(spec):
    """-> (kernel name, phis or None, oracle kernel id) of a workload spec."""
    from fokl_gpy_amd import getKernels
    if spec['kernel'] == 'Cubic Splines':
        tab = np.load(os.path.join(ROOT, 'tests', 'golden', 'spline_phis.npz'))['table']
        return spec['kernel'], getKernels.table_to_phis(tab), 0
    phis = getKernels.bernoulli()
    if spec['phis_cap']:
        phis = phis[:spec['phis_cap']]
    return spec['kernel'], phis, 1


def compare_with_golden(name, model, betas, mtx, evs, state):
    """The tolerances of tests/test_config_goldens.py.  -> dict(ok, ...)"""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), allow_pickle=False)
    out = dict(golden=name + '.npz')
    trace = model.fit_trace
    out['mtx_equal'] = bool(mtx.shape == g['mtx'].shape and np.array_equal(mtx, g['mtx']))
    out['gibbs_calls_equal'] = bool([t['cols'] for t in trace] == g['call_cols'].tolist() and
                                    [t['built'] for t in trace] == g['call_built'].tolist())
    out['terms_logical'] = int(np.sum(g['call_built']))
    if out['mtx_equal'] and out['gibbs_calls_equal']:
        out['max_rel_bic'] = float(np.max(np.abs(np.array([t['ev'] for t in trace]) - g['call_ev']) / np.abs(g['call_ev'])))
        scale = np.max(np.abs(g['betas']), axis=0)
        out['max_draw_err_over_scale'] = float(np.max(np.abs(betas - g['betas']) / scale))
        out['numpy_stream_equal'] = bool(np.array_equal(state[1], g['rng_key']) and state[2] == int(g['rng_pos']) and
                                         state[3] == int(g['rng_has_gauss']) and state[4] == float(g['rng_cached']))
        # The gate is the stated one: BIC 1e-9 relative, draws 1e-9 of the column scale -- the tolerances of
        # tests/test_config_goldens.py, fixed.  `draw_bound` is information only: betas = w Q', an eigenvector of XtX turns
        # by about ||dXtX|| / gap when XtX changes by dXtX, and the GPU's Gram differs from the golden's BLAS Gram in its
        # last bits; 1/64 of eps ||XtX|| / (smallest gap) of the RETURNED model (a statistic of the fit under test, hence
        # never part of the gate) is what those bits can move a draw by, relative to its column's scale.
        sens = model.fit_stats.get('final_eps_norm_over_gap')
        out['draw_bound'] = float(sens) / 64.0 if sens else None
        out['draw_gate'] = 1e-9
        out['ok'] = bool(out['max_rel_bic'] < 1e-9 and out['max_draw_err_over_scale'] < out['draw_gate'] and
                         out['numpy_stream_equal'] and len(evs) == len(g['evs']))
        # how far inside the stated tolerances (BIC 1e-9 relative, draws 1e-9 of the column scale) the fit
        # sits: limit / measured.  What the draws' distance is made of: betas = w Q', and an eigenvector of XtX moves by
        # about eps ||XtX|| / gap when XtX changes in its last bits (the GPU's Gram against the oracle's BLAS Gram) --
        # `eigvec_sensitivity` is that figure for the returned model.  The finishing mode of the normals (libmvec's
        # vector log or libm's scalar one) does not show: the same fit gives the same figure to 10 digits in both
        # (profiles/draw_margin_r03.json).
        out['margin'] = dict(draws=1e-9 / max(out['max_draw_err_over_scale'], 1e-300),
                             bic=1e-9 / max(out['max_rel_bic'], 1e-300),
                             eigvec_sensitivity=model.fit_stats.get('final_eps_norm_over_gap'),
                             condition_number=model.fit_stats.get('final_cond'),
                             finish_log=os.environ.get('FOKL_FINISH_LOG', 'fast'))
    else:
        out['ok'] = False
    return out


def mfma_flops_issued_over_algorithmic(trace_lines):
    """What the MFMA-bound Gram launches of a fit issue on the matrix cores (16 x 16 tiles of the launch plan; half tiles on a
    ragged last row tile; the symmetric part's lower tiles never computed) over the algorithmic 2 N nr nc they are booked with
    (SURVEY 8(d)).  trace_lines: FOKL_GRAM_TRACE's, one per launch: `nr nc distinct class`."""
    from fokl_gpy_amd import _capi
    run = alg = 0.0
    for line in trace_lines:
        nr, nc, _, klass = line.split()
        nr, nc = int(nr), int(nc)
        if klass != 'gram_mfma':
            continue
        rs = np.arange(2, 2 + nr, dtype=np.int32)                                 # the fit's shape: new columns against
        cs = np.concatenate([[0], np.arange(1000, 1000 + nc - nr - 2), rs, [1]])   # [ones | model | new | y]
        plan = _capi.gram_plan(rs, cs.astype(np.int32))
        real = plan['tiles'][..., 2] >= 0
        run += 512.0 * (np.count_nonzero(real & ~plan['half']) + 0.5 * np.count_nonzero(real & plan['half']))
        alg += 2.0 * nr * nc
    return run / alg if alg else None


def cgroup_cpu():
    """What the container's CPU controller says (cgroup v2): quota in CPUs (None: unlimited / unknown) and the counters of
    cpu.stat -- usage_usec, nr_periods, nr_throttled, throttled_usec.  Fits side by side on one GPU are bounded by this quota
    long before the device is busy: the throughput line reports how much of it they used and how often they were throttled."""
    out = dict(quota_cpus=None)
    try:
        q, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            out['quota_cpus'] = float(q) / float(period)
    except (OSError, ValueError):
        pass
    try:
        for line in open('/sys/fs/cgroup/cpu.stat'):
            k, v = line.split()
            if k in ('usage_usec', 'nr_periods', 'nr_throttled', 'throttled_usec'):
                out[k] = int(v)
    except (OSError, ValueError):
        pass
    return out


def cpu_baselines(x, y, spec, seconds_target=15.0, which=('scalar', 'vectorised')):
    """
    The reference's algorithm on the host cores of this box (oracle, kind "port"), on a bounded sample of the same
    workload: its first gibbs evaluations (FoKLRoutines.py:1396-1558) with the configuration's own number of Gibbs
    iterations -- the first two sub-stage models (8 main effects, then + 28 two-way terms at M = 8) and the first kill
    test of the second sub-stage (FR:1673-1683: the model without one of the new terms, all 27 surviving new columns
    rebuilt), i.e. three evaluations building 63 candidate terms.

      scalar      the reference's own per-element Python loop structure for the basis matrix (FR:1446-1485), one core,
                  on the first n_sample rows.  Only the work that grows with the rows (X-build, XtX / Xty, residual
                  pass: `rows_s`) is scaled to the full N; the eigen-decomposition and the Gibbs iterations (`chain_s`)
                  do not depend on N and are added as measured:  value = terms / (rows_s * N / n_sample + chain_s);
      vectorised  the same evaluations at FULL N with the basis matrix built by whole-column numpy expressions
                  (oracle.build_columns_numpy) and numpy's multi-threaded BLAS for XtX -- the fair NumPy baseline of
                  SURVEY 8(d)(ii).
    Both are baselines, not targets.
    """
    from oracle import fokl_oracle as O
    kernel, phis, kid = kernel_and_phis(spec)
    hp = dict(O.DEFAULT_HYPERS)
    hp.update(spec['fit'])
    a, atau = hp['a'], hp['atau']
    draws = hp['burnin'] + hp['draws']
    n_full, m = x.shape
    sett = 1 if m == 1 else (3 if hp['way3'] else 2)
    t1 = O.distinct_arrangements(O.deal_indvec(1, m, sett))
    t2 = O.distinct_arrangements(O.deal_indvec(2, m, sett)) if len(phis) >= 2 and m > 1 else t1[:0]
    if t1.shape[0] + t2.shape[0] > 64:                  # M = 16: the 137-column chain alone (O(P^3) numpy products per
        t2 = t1[:0]                                     # Gibbs iteration, FR:1521-1528) would take a minute
    n_calls = 3 if t2.shape[0] > 1 else 1
    terms = t1.shape[0] + (2 * t2.shape[0] - 1 if n_calls == 3 else 0)
    lo, hi = x.min(axis=0), x.max(axis=0)
    xn_full = (x - lo) / (hi - lo)                      # clean()'s min-max normalisation (FR:420-470)
    state = np.random.get_state()
    out = {}

    def evaluations(xs, ys, build):
        b, btau = O.default_b_btau(ys, a, atau)
        dtd = ys.T.dot(ys)
        if kid == O.KERNEL_SPLINES:
            phind, xsm = O.inputs_to_phind(xs, len(phis[0][0]))
        else:
            phind, xsm = None, xs
        np.random.seed(12345)
        timing = {}
        common = (a, b, atau, btau, draws, phind, xsm, b / (1 + a), btau / (1 + atau), dtd)
        t0 = time.perf_counter()
        r1 = O.gibbs(xs, ys, phis, kid, [], t1, *common, build=build, timing=timing)
        if n_calls == 3:
            O.gibbs(xs, ys, phis, kid, r1.X, np.vstack([t1, t2]), *common, build=build, timing=timing)
            O.gibbs(xs, ys, phis, kid, r1.X, np.vstack([t1, t2[:-1]]), *common, build=build, timing=timing)
        return time.perf_counter() - t0, timing['rows_s'], timing['chain_s']

    if 'scalar' in which:
        # ~8 us per (row, term) for the reference's loop; the chains add ~2 s
        n_sample = int(min(n_full, max(2000, seconds_target / (terms * 8e-6))))
        xs = np.ascontiguousarray(xn_full[:n_sample])
        ys = np.ascontiguousarray(y[:n_sample])[:, None]
        dt, rows_s, chain_s = evaluations(xs, ys, O.build_columns_scalar)
        full_s = rows_s * n_full / n_sample + chain_s
        out['cpu_baseline'] = dict(
            value=terms / full_s, unit='candidate-terms/s', cores=1, kind='port',
            sample=f'oracle scalar path (reference loop structure, FR:1446-1485), first {n_calls} gibbs evaluation(s) ({terms} '
                   f'terms, {draws} Gibbs iterations each) on the first {n_sample} of {n_full} rows in {dt:.1f} s: '
                   f'{rows_s:.1f} s of row-dependent work (X-build, XtX, residual pass) scaled by {n_full}/{n_sample}, '
                   f'+ {chain_s:.1f} s of eigh / Gibbs iterations as measured (independent of N)',
            seconds=dt, rows_seconds=rows_s, chain_seconds=chain_s, seconds_at_full_n=full_s, terms=terms,
            evaluations=n_calls)
    if 'vectorised' in which:
        try:
            from threadpoolctl import threadpool_info
            blas_threads = max([lib.get('num_threads', 1) for lib in threadpool_info() if lib.get('user_api') == 'blas']
                               or [1])
        except Exception:
            blas_threads = 1
        dt, rows_s, chain_s = evaluations(xn_full, y[:, None], O.build_columns_numpy)
        out['cpu_baseline_vectorised'] = dict(
            value=terms / dt, unit='candidate-terms/s', cores=int(blas_threads), kind='port',
            sample=f'oracle with whole-column numpy expressions for the basis matrix and numpy BLAS ({blas_threads} '
                   f'threads) for XtX: first {n_calls} gibbs evaluation(s) ({terms} terms, {draws} Gibbs iterations each) at '
                   f'the full N = {n_full} in {dt:.1f} s; no extrapolation',
            seconds=dt, rows_seconds=rows_s, chain_seconds=chain_s, terms=terms, evaluations=n_calls)
    np.random.set_state(state)
    return out


def l3_domain_of(cpu):
    """Logical CPUs sharing the last-level cache with `cpu` that this process may use (None if unknown)."""
    try:
        allowed = set(os.sched_getaffinity(0))
        with open(f'/sys/devices/system/cpu/cpu{cpu}/cache/index3/shared_cpu_list') as fh:
            want = set()
            for part in fh.read().strip().split(','):
                lo, _, hi = part.partition('-')
                want.update(range(int(lo), int(hi or lo) + 1))
        want = sorted(want & allowed)
        return want if len(want) >= 2 else None
    except (OSError, ValueError, AttributeError):
        return None


def l3_domains_by_load(cpus, sample_s=0.25, near=None, avoid=()):
    """[(busy fraction, domain)] of the L3 domains that hold the logical CPUs `cpus`, in CPU order: /proc/stat sampled
    twice, `sample_s` apart.  Domains that are not wholly inside `near` (the GPU's NUMA node) or touch `avoid` are left
    out.  The hosts of the pool are shared: other tenants' load on the cores a fit's threads sit on shows up one to one
    in the (host-bound) fit time."""
    def snap():
        out = {}
        with open('/proc/stat') as fh:
            for line in fh:
                if line.startswith('cpu') and line[3].isdigit():
                    p = line.split()
                    out[int(p[0][3:])] = (sum(map(int, p[1:9])), int(p[4]) + int(p[5]))
        return out
    avoid = set(avoid or ())
    a = snap()
    time.sleep(sample_s)
    b = snap()
    domains, seen = [], set()
    for cpu in sorted(cpus):
        if cpu in seen:
            continue
        dom = l3_domain_of(cpu)
        if not dom:
            continue
        seen.update(dom)
        if avoid & set(dom) or (near is not None and not set(dom) <= near):
            continue                                        # e.g. the other socket: far from the GPU's page-locked memory
        busy = [1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in dom if c in a and c in b]
        domains.append((sum(busy) / max(1, len(busy)), dom))
    return domains


def quietest_l3_domain(local, ranks, sample_s=0.25):
    """The least busy L3 domain among those this rank may claim (domains d with d % ranks == local, so that the ranks of a
    node never pick the same one)."""
    try:
        domains = l3_domains_by_load(set(os.sched_getaffinity(0)), sample_s, near=gpu_numa_cpus(local))
        mine = [d for i, d in enumerate(domains) if i % max(1, ranks) == local % max(1, ranks)]
        if not mine:
            return None
        return min(mine, key=lambda d: d[0])[1]
    except (OSError, ValueError, KeyError, IndexError):
        return None


def gpu_numa_cpus(local=0):
    """Logical CPUs of the NUMA node the rank's GPU hangs off (sysfs: the AMD render nodes this process can read, in
    order), or None.  Tapes live in page-locked memory next to the GPU: a recorder on the other socket writes them
    three times slower (measured: 0.23 instead of 0.08 s per fit), so threads are only ever pinned inside this set."""
    try:
        nodes = []
        for name in sorted(os.listdir('/sys/class/drm'), key=lambda n: (len(n), n)):
            if not name.startswith('renderD'):
                continue
            base = f'/sys/class/drm/{name}/device'
            try:
                with open(base + '/vendor') as fh:
                    if fh.read().strip() != '0x1002':
                        continue
                with open(base + '/numa_node') as fh:
                    nodes.append(int(fh.read().strip()))
            except (OSError, ValueError):
                continue
        if not nodes:
            return None
        node = nodes[local % len(nodes)]
        if node < 0:
            return None
        with open(f'/sys/devices/system/node/node{node}/cpulist') as fh:
            cpus = set()
            for part in fh.read().strip().split(','):
                lo, _, hi = part.partition('-')
                cpus.update(range(int(lo), int(hi or lo) + 1))
        return cpus or None
    except OSError:
        return None


def quiet_l3_domains(count, avoid=(), sample_s=0.25, near=None):
    """The `count` least busy L3 domains that share no CPU with `avoid` (worker processes of the throughput modes: one
    domain each, none on the domain this process has pinned itself to).  Fewer if the box has fewer."""
    try:
        domains = l3_domains_by_load(range(os.cpu_count() or 1), sample_s, near=near, avoid=avoid)
        return [dom for _, dom in sorted(domains, key=lambda d: d[0])[:count]]
    except (OSError, ValueError, KeyError, IndexError):
        return []


def pin_to_l3_domain(local, ranks=1):
    """Keep this rank's host threads (search driver + the native noise / chain / spectral threads, about eight) on the
    logical CPUs of ONE L3 domain: noise tapes (about 1 MB each) are handed from thread to thread.  The domain is the
    quietest one this rank may claim (quietest_l3_domain); failing that, the domain of CPU 8 r (8 cores x 2 SMT threads
    on the EPYC hosts of this pool)."""
    if os.environ.get('FOKL_BENCH_PIN', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        want = None
        if os.environ.get('FOKL_BENCH_PIN', '1') != 'fixed':
            want = quietest_l3_domain(local, ranks)
        if not want:
            allowed = set(os.sched_getaffinity(0))
            want = l3_domain_of(8 * local) or sorted(set(range(8 * local, 8 * local + 8)) & allowed)
        if want and len(want) >= 2:
            os.sched_setaffinity(0, want)
            return want
    except OSError:
        pass
    return None


def KERNEL_SLOTS():
    """(name in the JSON line, timing slot of the library) of every kernel a fit launches."""
    from fokl_gpy_amd import _capi
    return (('basis_build', _capi.K_BASIS), ('gram', _capi.K_GRAM), ('gram_mfma', _capi.K_GRAM_MFMA),
            ('gram_reduce', _capi.K_GRAM_REDUCE), ('resid', _capi.K_RESID), ('resid_matrix_free', _capi.K_RESID_MF))


def kernel_report(kern, n, m, cfg):
    """Per-kernel roofline readings from the HIP-event totals `kern` (name -> ms, launches, bytes, flops, ideal_ms).
    -> (kernels, dominant kernel's `roofline` object, device milliseconds)"""
    def roof(name, bound):
        k = kern[name]
        if k['launches'] == 0 or k['ms'] <= 0:
            return None
        avg_ms = k['ms'] / k['launches']
        if bound == 'hbm':
            achieved = k['bytes'] / k['launches'] / (avg_ms * 1e-3) / 1e9
            return dict(kernel=name, bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=achieved / HBM_PEAK_GBS, traffic=None, launches=k['launches'], avg_ms=avg_ms,
                        total_ms=k['ms'], algorithmic_bytes_per_launch=k['bytes'] / k['launches'])
        achieved = k['flops'] / k['launches'] / (avg_ms * 1e-3) / 1e12
        return dict(kernel=name, bound='mfma', achieved=achieved, peak=FP64_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                    frac=achieved / FP64_MFMA_PEAK_TFLOPS, traffic=None, launches=k['launches'], avg_ms=avg_ms,
                    total_ms=k['ms'])

    # Gram launches are booked by the library under the roof that binds each of them (fokl_hip.h: FOKL_K_GRAM for
    # 8 N distinct-columns / HBM peak > 2 N nr nc / fp64 peak, FOKL_K_GRAM_MFMA otherwise): two kernels for the roofline
    kernels = {'basis_build': roof('basis_build', 'hbm'), 'gram': roof('gram', 'hbm'),
               'gram_mfma': roof('gram_mfma', 'mfma'), 'gram_reduce': roof('gram_reduce', 'hbm'),
               'resid': roof('resid', 'hbm'),
               'resid_matrix_free': roof('resid_matrix_free', 'hbm')}
    mf = kernels['resid_matrix_free']
    if mf:
        # the matrix-free residual pass trades the column reads for fp64 vector arithmetic: re-forming the columns
        # costs more time at the fp64 peak (FMA = 2 flops; most of its operations are separately-rounded multiplies
        # and adds, i.e. half of that at best) than its 8 N (M_used + 1) bytes cost at the HBM peak
        k = kern['resid_matrix_free']
        tf = k['flops'] / k['launches'] / (mf['avg_ms'] * 1e-3) / 1e12
        mf.update(fp64_valu_tflops=tf, fp64_valu_frac=tf / FP64_MFMA_PEAK_TFLOPS,
                  roofline_frac=k['ideal_ms'] / k['ms'])
        if k['flops'] / (FP64_MFMA_PEAK_TFLOPS * 1e12) > k['bytes'] / (HBM_PEAK_GBS * 1e9):
            mf.update(bound='valu-fp64', achieved=tf, peak=FP64_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                      frac=tf / FP64_MFMA_PEAK_TFLOPS, hbm_achieved_gbs=mf['achieved'], hbm_frac=mf['frac'])
    if kernels['gram_mfma']:
        kernels['gram_mfma'].update(
            algorithmic_flops_per_launch=kern['gram_mfma']['flops'] / kern['gram_mfma']['launches'],
            algorithmic_bytes_per_launch=kern['gram_mfma']['bytes'] / kern['gram_mfma']['launches'])
    both = [kern[name] for name in ('gram', 'gram_mfma') if kern[name]['launches']]
    if both:
        # all Gram launches of the run together, priced launch by launch: sum of max(bytes / HBM peak, flops / MFMA
        # peak) over the measured time
        ms = sum(k['ms'] for k in both)
        gram_all = dict(launches=sum(k['launches'] for k in both), total_ms=ms,
                        roofline_frac=sum(k['ideal_ms'] for k in both) / ms,
                        hbm_achieved_gbs=sum(k['bytes'] for k in both) / (ms * 1e-3) / 1e9,
                        mfma_achieved_tflops=sum(k['flops'] for k in both) / (ms * 1e-3) / 1e12)
        for name in ('gram', 'gram_mfma'):
            if kernels[name]:
                kernels[name]['all_gram_launches'] = gram_all
    # HBM traffic per launch: PMC counters cannot be read from inside this process; the figures come from the committed
    # rocprofv3 --pmc passes over this same command (profiles/pmc_r06.json, produced by tools/profile_r06.sh: separate
    # FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes) and are attached only when that file
    # was recorded for exactly this workload.
    traffic_source = None
    for cand in ('pmc_r06.json', 'pmc_r05.json', 'pmc_r04.json', 'pmc_r03.json', 'pmc_r02.json', 'pmc_r01.json'):
        pmc_path = os.path.join(ROOT, 'profiles', cand)
        if not os.path.exists(pmc_path):
            continue
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        if pmc.get('workload') in ({'rows': n, 'inputs': m}, {'rows': n, 'inputs': m, 'config': cfg}) and cfg == 2:
            for name, k in kernels.items():
                if k and name in pmc['kernels']:
                    k['traffic'] = pmc['kernels'][name]['hbm_bytes_per_launch']
            traffic_source = f'profiles/{cand}: rocprofv3 --pmc passes over this command, committed with the sources; ' \
                             f'not re-measured in this run'
            break
    dominant = max((k for k in kernels.values() if k), key=lambda k: k['total_ms'])
    dominant = dict(dominant, traffic_source=traffic_source)
    gpu_ms = sum(k['total_ms'] for k in kernels.values() if k)
    return kernels, dominant, gpu_ms


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of this one -- the same
    command line, RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would (no torch needed on the box; the ranks find each
    other through fokl_gpy_amd.dist's own rendezvous, which is keyed on MASTER_PORT and the run id).  This process never
    touches the GPU; rank 0 prints the JSON line.  -> exit code (the first failing rank's; the others are ended then)."""
    import socket
    import subprocess
    import time
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]                             # free now; only names this launch (dist._rendezvous_path)
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                TORCHELASTIC_RUN_ID=f'bench_{port}', TORCHELASTIC_RESTART_COUNT='0')
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    ranks = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    code = 0
    try:
        alive = list(ranks)
        while alive:
            for p in list(alive):
                rc = p.poll()
                if rc is None:
                    continue
                alive.remove(p)
                if rc != 0 and code == 0:
                    code = rc
                    for q in alive:                           # one rank failed: the others would wait for it for ever
                        q.terminate()
            time.sleep(0.05)
    except BaseException:
        for p in ranks:
            if p.poll() is None:
                p.kill()
        raise
    return code


def device_of(local):
    """HIP device of the rank with this LOCAL_RANK.  FOKL_BENCH_SHARE_GPU=1 (launcher rehearsals on a box with fewer GPUs
    than ranks) wraps the ranks round the devices there are -- RCCL then refuses to initialise (two ranks on one device) and
    independent fits carry on over the TCP control plane."""
    if os.environ.get('FOKL_BENCH_SHARE_GPU', '0') == '1':
        return local % max(1, visible_device_count())
    return local


def visible_device_count():
    """Number of GPUs of this box WITHOUT initialising the HIP runtime in this process (worker processes are started
    before the parent touches the GPU): FOKL_BENCH_DEVICES if set, the visible-devices lists, else the render nodes the
    amdgpu driver exposes."""
    if os.environ.get('FOKL_BENCH_DEVICES'):
        return int(os.environ['FOKL_BENCH_DEVICES'])
    for name in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES'):
        if os.environ.get(name):
            return len([v for v in os.environ[name].split(',') if v.strip()])
    try:
        return len([d for d in os.listdir('/dev/dri') if d.startswith('renderD')]) or 1
    except OSError:
        return 1


def bring_up_comm(ctx, rank, world, use_rccl, need_rccl):
    """-> (comm, description): RCCL when it comes up on every rank, else (independent fits only) the TCP control plane."""
    from fokl_gpy_amd import dist
    if not use_rccl:
        return dist.SingleComm(), 'single process'
    if world == 1:                                            # FOKL_BENCH_FORCE_RCCL=1: RCCL in a world of one
        return dist.RcclComm(ctx, rank, world), 'RCCL'
    return dist.bring_up(ctx, rank, world, need_rccl, log=lambda msg: print(f"bench.py: {msg}", file=sys.stderr))


def _host_reduced_backend():
    from fokl_gpy_amd import engine

    class HostReduced(engine.HipBackend):
        """Rehearsal only (ranks sharing one GPU, no RCCL): the row-sharded modes' sums over the ranks -- Gram blocks,
        residual moments, what fokl_gram(allreduce=1) / fokl_bic_resid(allreduce=1) do with ncclAllReduce on the device --
        are formed on the host over the control plane.  Same call order on every rank (the replicated search)."""

        def __init__(self, ctx, comm):
            super().__init__(ctx)
            self.comm, self._launched_reduced = comm, False

        def gram(self, row_slots, col_slots, allreduce=False):
            g = self.ctx.gram(row_slots, col_slots, 0, False)
            return self.comm.allreduce_sum(g) if allreduce else g

        def gram_launch(self, row_slots, col_slots, allreduce=False):
            self._launched_reduced = bool(allreduce)
            return self.ctx.gram_launch(row_slots, col_slots, False)

        def gram_fetch(self, shape):
            g = self.ctx.gram_fetch(shape)
            return self.comm.allreduce_sum(g) if self._launched_reduced else g

        def _moments(self, pair, allreduce):
            if not allreduce:
                return pair
            s = self.comm.allreduce_sum(np.array(pair, dtype=np.float64))
            return float(s[0]), float(s[1])

        def bic_resid(self, slots, betahat, allreduce=False):
            return self._moments(self.ctx.bic_resid(slots, betahat, False), allreduce)

        def bic_resid_fetch(self, allreduce=False):
            return self._moments(self.ctx.bic_resid_fetch(False), allreduce)

    return HostReduced


def HostReducedBackend(ctx, comm):
    return _host_reduced_backend()(ctx, comm)


def fits_worker(cfg, k, procs, local, unit_ids, rows, steps, warmup, start, done, gate=None, domain=None):
    """One of the worker PROCESSES of `--config 4 --procs P`: its own device context(s), host threads and L3 domain; fits
    its share of the rank's datasets back to back.  (Threads of one process share the interpreter lock of the Python
    drivers; processes do not: 13.7 / 21.9 / 32.8 / 40.6 fits/s with 1 / 2 / 3 / 4 of them on one MI355X.)"""
    if gate is not None and not gate.wait(timeout=900):     # started early, used late (or never: then just leave)
        return
    try:
        dom = domain or l3_domain_of(8 * (local * procs + k))
        if dom:
            os.sched_setaffinity(0, dom)
    except OSError:
        pass
    # host chains: the chain + finishing threads follow the recorder (a core's worth each), two spectral threads fit next
    # to them; device chains: those threads idle, the eigen-decompositions are what the driver waits for -- three
    device_chains = os.environ.get('FOKL_CHAIN', 'auto') != 'host'
    # host threads that wait for the GPU sleep instead of spinning: with several processes on one GPU a wait is long, and
    # what it burns is CPU of the shared quota (measured: 0.37 -> 0.31 CPU-seconds per fit with six processes)
    os.environ.setdefault('FOKL_SYNC', 'blocking')
    os.environ.setdefault('FOKL_SPIN', '0.05')            # ... and poll for tens of pauses, not thousands, before they sleep
    # tapes walked ahead of the decisions: 24 deep instead of 48 -- a fit alone takes the same time within the spread of the
    # boxes, 50 instead of 160 tapes per fit are walked for nothing, and that CPU is what the quota is short of
    # (tools/r05_spec.sh: 4 processes 854-913 k terms/s at 48, 952-981 k at 24, 961-965 k at 12, 709-944 k at 6)
    os.environ.setdefault('FOKL_SPECULATION', THROUGHPUT_SPECULATION)
    os.environ.setdefault('FOKL_WALK_HELPERS', '0')       # (helpers of the serial walk buy latency with CPU: not here)
    for name, val in (('FOKL_CHAIN_THREADS', '1'), ('FOKL_FINISH_THREADS', '1'),
                      ('FOKL_SPECTRAL_THREADS', '3' if device_chains or procs <= 2 else '2')):
        os.environ.setdefault(name, val)
    os.environ['FOKL_DEVICE'] = str(device_of(local))
    from fokl_gpy_amd import FoKLRoutines, _capi, engine
    fits = []
    prep_s = clean_s = upload_s = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for unit in unit_ids:
            x, y, spec = config_workload(cfg, unit, rows)
            kernel, phis, _ = kernel_and_phis(spec)
            model = FoKLRoutines.FoKL(kernel=kernel, phis=phis, UserWarnings=False, ConsoleOutput=False, **spec['fit'])
            model._backend_override = engine.HipBackend(_capi.DeviceContext(device_of(local)))
            t0 = time.perf_counter()
            model._prepare_fit(x, y, dict(clean=True))
            prep_s += time.perf_counter() - t0
            fits.append((model, spec))

        def one_pass(acc=None):
            for model, spec in fits:
                np.random.seed(spec['seed_fit'])
                model._search(model._backend_override, spec['rows'], spec['inputs'])
                model._backend_override.ctx.sync()
                if acc is not None:
                    for key in acc:
                        acc[key] += model.fit_stats.get(key, 0)

        for _ in range(warmup):
            one_pass()
        for model, _ in fits:
            model._backend_override.ctx.timing_enable(True)
            model._backend_override.ctx.timing_reset()
        acc = dict(terms_logical=0, terms_physical=0, gibbs_calls=0, pool_noise_s=0.0, pool_chain_s=0.0,
                   pool_finish_s=0.0, pool_spectral_s=0.0, t_eigh=0.0, t_resid=0.0, t_chain=0.0, noise_verdict_wait_s=0.0,
                   noise_queue_wait_s=0.0, seconds=0.0)
        start.wait()
        cpu0 = time.process_time()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_pass(acc)
        elapsed = time.perf_counter() - t0
        acc['cpu_s'] = time.process_time() - cpu0             # every thread of this worker
    kern = {}
    for name, kid in KERNEL_SLOTS():
        tot = dict(ms=0.0, launches=0, bytes=0.0, flops=0.0, ideal_ms=0.0)
        for model, _ in fits:
            t = model._backend_override.ctx.timing_get(kid)
            for key in tot:
                tot[key] += t[key]
        kern[name] = tot
    done.put(dict(worker=k, elapsed=elapsed, prep_s=prep_s, stats=acc, kern=kern))


def fits_with_worker_processes(args, cfg, rank, world, local, procs):
    """`--config 4` with `procs` worker processes per rank (throughput mode: many independent host-bound fits share one
    GPU).  This process only launches, synchronises and reports; it touches the GPU after the workers have been started
    (RCCL barrier / gather at N > 1, parity check and probes on rank 0)."""
    import multiprocessing as mp
    from fokl_gpy_amd import dist
    fits_per_step = args.fits_per_step or (8 if cfg == 4 else procs)
    units = [rank * fits_per_step + i for i in range(fits_per_step)]
    rows = args.rows or CONFIGS[cfg]['rows']
    ctx_mp = mp.get_context('spawn')                        # before this process has initialised the GPU
    start, done = ctx_mp.Barrier(procs + 1), ctx_mp.Queue()
    workers = [ctx_mp.Process(target=fits_worker, args=(cfg, k, procs, local, units[k::procs], rows, args.steps,
                                                            args.warmup, start, done)) for k in range(procs)]
    for w in workers:
        w.start()

    from fokl_gpy_amd import FoKLRoutines, _capi, engine
    use_rccl = world > 1 or os.environ.get('FOKL_BENCH_FORCE_RCCL', '0') == '1'
    backend = FoKLRoutines.device_backend(device_of(local))
    ctx = backend.ctx
    comm, comm_kind = bring_up_comm(ctx, rank, world, use_rccl, need_rccl=False)
    control = getattr(comm, 'control', None) or comm        # independent fits: barriers and gathers over the control plane
    control.barrier()
    start.wait(timeout=1800)                                # every worker has prepared and warmed up: go
    t0 = time.perf_counter()
    results = [done.get(timeout=3600) for _ in workers]
    control.barrier()
    elapsed = time.perf_counter() - t0
    for w in workers:
        w.join(60)

    kern = {}
    for name, _ in KERNEL_SLOTS():
        tot = dict(ms=0.0, launches=0, bytes=0.0, flops=0.0, ideal_ms=0.0)
        for r in results:
            for key in tot:
                tot[key] += r['kern'][name][key]
        kern[name] = tot
    host = {}
    for r in results:
        for key, val in r['stats'].items():
            host[key] = host.get(key, 0) + val
    logical, physical, calls = host.pop('terms_logical'), host.pop('terms_physical'), host.pop('gibbs_calls')

    # parity of the same fit this process can repeat on its own: unit `units[0]` against its golden
    parity_checked, parity = False, None
    x0, y0, spec0 = config_workload(cfg, units[0], rows)
    if rank == 0 and not args.no_parity:
        key = (cfg, units[0], rows)
        if key in GOLDENS and os.path.exists(os.path.join(ROOT, 'tests', 'golden', GOLDENS[key][0] + '.npz')):
            kernel, phis, _ = kernel_and_phis(spec0)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                model = FoKLRoutines.FoKL(kernel=kernel, phis=phis, UserWarnings=False, ConsoleOutput=False)
                model._backend_override = backend
                np.random.seed(spec0['seed_fit'])
                b0, m0, e0 = model.fit(x0, y0, clean=True, **spec0['fit'])
            parity = compare_with_golden(GOLDENS[key][0], model, b0, m0, e0, np.random.get_state())
            parity['workload'] = 'unit 0 of the timed fits, repeated by the reporting process after the timed region'
            parity_checked = True

    sustained = None
    if rank == 0 and not args.no_microbench:
        ctx.upload(np.zeros((64, 1)), np.zeros(64), 1, np.zeros(2), 1, 2)
        sustained = {'unit': 'GB/s and TFLOP/s', 'hbm_read_GBps': ctx.probe(0) / 1e9, 'hbm_write_GBps': ctx.probe(1) / 1e9,
                     'hbm_1_read_7_writes_GBps': ctx.probe(2) / 1e9, 'mfma_f64_TFLOPs': ctx.probe(3) / 1e12}

    gathered = control.allgather([elapsed, logical, physical, calls])
    if rank != 0:
        comm.close()
        return
    t_max = float(np.max(gathered[:, 0]))
    tot_logical, tot_physical = float(np.sum(gathered[:, 1])), float(np.sum(gathered[:, 2]))
    fits_total = world * fits_per_step * max(args.steps, 1)
    kernels, dominant, gpu_ms = kernel_report(kern, rows, spec0['inputs'], cfg)
    line = {
        'metric': 'candidate-terms/sec (basis build + Gibbs + BIC)',
        'value': tot_logical / t_max,
        'unit': 'candidate-terms/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * t_max / max(args.steps, 1),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': spec0['label'] + f', {fits_per_step} fits per rank and step dealt over {procs} worker '
                                                f'processes per GPU',
                   'config_index': cfg, 'rows': rows, 'inputs': spec0['inputs'],
                   'parallelism': (f'independent fits x{world} GPUs, ' if world > 1 else 'single GPU, ') +
                                  f'{procs} worker processes per GPU', 'collectives': comm_kind,
                   'terms_counted': 'logical (reference-equivalent, FoKLRoutines.py:1461)'},
        'value_physical': tot_physical / t_max,
        'what_is_counted': {'logical_terms_per_fit': tot_logical / fits_total,
                            'columns_built_on_the_device_per_fit': tot_physical / fits_total,
                            'value_logical_terms_per_s': tot_logical / t_max,
                            'value_physical_columns_per_s': tot_physical / t_max},
        'parity_checked': parity_checked,
        'parity': parity,
        'fits_per_s': fits_total / t_max,
        'terms_logical_per_fit': tot_logical / fits_total,
        'terms_physical_per_fit': tot_physical / fits_total,
        'gibbs_calls_per_fit': float(np.sum(gathered[:, 3])) / fits_total,
        'gpu_kernel_ms_per_step': gpu_ms / max(args.steps, 1),
        'host_prepare_s': sum(r['prep_s'] for r in results),
        'host_main_thread_s_per_step': {k: v / max(args.steps, 1) for k, v in host.items()},
        'worker_seconds': sorted(r['elapsed'] for r in results),
        'roofline': dominant,
        'kernels': kernels,
        'device_sustains': sustained,
    }
    if not args.no_cpu_baseline:
        line.update(cpu_baselines(x0, y0, spec0))
    comm.close()
    dist.flush_c_streams()
    sys.stderr.flush()
    print(json.dumps(line), flush=True)
    if parity_checked and not parity['ok']:
        print(f"bench.py: PARITY MISMATCH against {parity['golden']}: {parity}", file=sys.stderr)
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)      # (a fit is ~45 ms: the first two or three after start-up are 15 %
    ap.add_argument('--warmup', type=int, default=3)      # slower -- fresh mappings, cold caches -- and are left to the warm-up)
    ap.add_argument('--config', type=int, choices=sorted(CONFIGS), default=2,
                    help='BASELINE.json configs[i]; the metric is quoted on configs[2] (default)')
    ap.add_argument('--rows', type=int, default=None, help='override the configuration\'s row count')
    ap.add_argument('--inputs', type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument('--fits-per-step', type=int, default=None,
                    help='--config 4: independent fits per rank and step (default 8 = 64 fits over 8 GPUs)')
    ap.add_argument('--concurrent', type=int, default=None,
                    help='--config 4: fits of a step that run side by side on one GPU, each on its own stream and host '
                         'threads (a fit is bound by its serial random stream on the host and leaves the GPU idle most '
                         'of the time; measured on a 16-CPU quota: 15.4 / 16.4 / 18.5 / 17.1 fits/s at 1 / 2 / 3 / 4 -- the driver '
                         'threads share one Python interpreter lock); default 3 with a budget of 16 or more CPUs, 2 from 12, '
                         'else 1')
    ap.add_argument('--procs', type=int, default=None,
                    help='worker PROCESSES per rank, each fitting its share of the rank\'s datasets of a step on its own '
                         'device context, host threads and L3 domain (--config 4 default: one per 4 CPUs of the budget, at most '
                         '4; other configurations: opt-in, a step is then that many independent fits)')
    ap.add_argument('--mode', choices=('fits', 'rows', 'candidates', 'hybrid'), default=None,
                    help="N > 1, see the module docstring; default: candidates for --config 3, fits otherwise")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-microbench', action='store_true',
                    help='skip the back-to-back kernel launches after the timed region (used for profiler runs)')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-throughput', action='store_true',
                    help='skip the secondary measurement of the default run (several fits sharing the GPU)')
    ap.add_argument('--launch-check', action='store_true', help=argparse.SUPPRESS)    # tests: bring the ranks up, nothing else
    args = ap.parse_args()

    from fokl_gpy_amd import dist
    rank, world, local = dist.env_rank_world()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started by hand: this process becomes the launcher of its own ranks (before anything here touches the GPU)
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it without WORLD_SIZE (it launches its own "
                  f"ranks) or with a launcher that starts {args.gpus} of them", file=sys.stderr)
            sys.exit(2)
    if args.launch_check:
        # (no GPU involved: the ranks of the launch meet over the control plane and say who they are)
        tcp = dist.TcpComm(rank, world)
        seen = tcp.allgather([float(rank), float(local)]) if world > 1 else np.array([[0.0, 0.0]])
        tcp.barrier()
        tcp.close()
        if rank == 0:
            print(json.dumps({'launch_check': [int(v) for v in seen[:, 0]], 'world': world,
                              'local_world': int(os.environ.get('LOCAL_WORLD_SIZE', '1'))}), flush=True)
        return
    os.environ['FOKL_DEVICE'] = str(device_of(local))
    from fokl_gpy_amd import FoKLRoutines, _capi, engine
    cfg = args.config
    # Secondary measurement of the default run (one GPU, configs[2]): how many candidate terms per second the same GPU
    # delivers when several independent fits of the configuration share it (a fit leaves it idle three quarters of the
    # time).  The worker processes have to be started before this process initialises the GPU; they sleep until the
    # main measurement is over.  `value` stays the one-fit-at-a-time figure.
    side_job = None
    # this process's L3 domain is chosen first, the worker processes' domains next (all the others to choose from: the
    # affinity mask is put back for that, and the workers inherit the full mask), then this process pins itself
    early_pin, full_mask = None, None
    if cfg != 4 and not (args.procs and args.procs > 1):
        try:
            full_mask = os.sched_getaffinity(0)
        except (AttributeError, OSError):
            full_mask = None
        early_pin = pin_to_l3_domain(local, int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))
        if early_pin is not None and full_mask is not None:
            os.sched_setaffinity(0, full_mask)
    if (cfg == 2 and world == 1 and rank == 0 and not args.procs and not args.no_throughput and not args.mode
            and os.environ.get('FOKL_BENCH_FORCE_RCCL', '0') != '1'):
        # worker processes next to this one, which fits along as one more (six processes on the GPU at most, this one
        # included): one fitting process per ~3.9 CPUs of the budget.  A configs[2] fit costs 0.14-0.16 CPU-seconds in ~40 ms
        # when it gets the CPUs it asks for: four processes use 15 of a 16-CPU quota and are never throttled (996 k terms/s,
        # 41 ms per fit per process); a fifth and sixth push the container over its quota in most scheduler periods, and a
        # period that ends throttled stops every thread of every fit (651 k / 886 k, 69-79 ms per fit:
        # profiles/throughput_quota_r05.txt).  With host chains a fit costs twice the CPU: one process per 6 CPUs.
        per_proc = 6.0 if os.environ.get('FOKL_CHAIN', 'auto') == 'host' else 3.9
        side_procs = int(os.environ.get('FOKL_BENCH_SIDE_PROCS', max(1, min(6, int(engine._cpu_budget() / per_proc))) - 1))
        if side_procs >= 1:
            try:
                import multiprocessing as mp
                ctx_mp = mp.get_context('spawn')
                gate, start_b, done_q = ctx_mp.Event(), ctx_mp.Barrier(side_procs + 1), ctx_mp.Queue()
                side_domains = quiet_l3_domains(side_procs, avoid=early_pin, near=gpu_numa_cpus(local))
                side_domains += [None] * (side_procs - len(side_domains))
                side_workers = [ctx_mp.Process(target=fits_worker, daemon=True,
                                               args=(cfg, k + 1, side_procs + 1, local, [k + 1],
                                                     args.rows or CONFIGS[cfg]['rows'], SIDE_FITS, 1, start_b, done_q, gate,
                                                     side_domains[k]))
                                for k in range(side_procs)]
                for w in side_workers:
                    w.start()
                side_job = (side_procs, gate, start_b, done_q, side_workers)
            except Exception as exc:                          # never let the extra cost the main measurement
                print(f"bench.py: throughput side measurement not started: {exc}", file=sys.stderr)
                side_job = None
    if early_pin is not None:
        try:
            os.sched_setaffinity(0, early_pin)              # the workers are on their way: now this process's own domain
        except OSError:
            pass
    if cfg == 4:
        procs = args.procs if args.procs else max(1, min(4, int(engine._cpu_budget() // 4)))
        procs = max(1, min(procs, args.fits_per_step or 8))
        if procs > 1 and not args.concurrent:
            return fits_with_worker_processes(args, cfg, rank, world, local, procs)
    elif args.procs and args.procs > 1 and (args.mode or 'fits') == 'fits':
        # opt-in for the other configurations: a step is then `procs` independent fits of the configuration (datasets
        # unit 0 .. procs - 1 of the rank) side by side on the GPU
        return fits_with_worker_processes(args, cfg, rank, world, local, args.procs)
    concurrent = 1
    if cfg == 4:
        budget = engine._cpu_budget()
        concurrent = args.concurrent if args.concurrent else (3 if budget >= 16 else 2 if budget >= 12 else 1)
        concurrent = max(1, min(concurrent, args.fits_per_step or 8))
    domains = None
    if concurrent > 1:
        # every concurrent fit gets an L3 domain of its own (rank r takes domains r * concurrent ...) and a smaller
        # thread plan; the process itself is not pinned
        domains = [l3_domain_of(8 * (local * concurrent + k)) for k in range(concurrent)]
        pinned = [d for d in domains]
        for name, val in (('FOKL_CHAIN_THREADS', '1'), ('FOKL_FINISH_THREADS', '2'), ('FOKL_SPECTRAL_THREADS', '2')):
            os.environ.setdefault(name, val)
    else:
        pinned = early_pin if early_pin is not None else pin_to_l3_domain(
            local, int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))
    # configs[3] at N > 1: rows AND candidates sharded (the device work and the eigen-decompositions are both divided by N);
    # on one GPU the two splits are empty and the mode is the plain fit through the candidate-exchange code path
    mode = args.mode or (('hybrid' if world > 1 else 'candidates') if cfg == 3 else 'fits')
    one_fit_for_all = mode in ('rows', 'candidates', 'hybrid')
    fits_per_step = (args.fits_per_step or 8) if cfg == 4 else 1
    if one_fit_for_all and fits_per_step != 1:
        print("bench.py: --config 4 runs in --mode fits", file=sys.stderr)
        sys.exit(2)

    # FOKL_BENCH_FORCE_RCCL=1 takes the RCCL bootstrap + collectives also in a world of one (launcher smoke test)
    use_rccl = world > 1 or os.environ.get('FOKL_BENCH_FORCE_RCCL', '0') == '1'
    backends = [FoKLRoutines.device_backend(device_of(local))]          # raises without libfokl_hip.so / a gfx950 device
    for _ in range(fits_per_step - 1):                        # one resident dataset (and stream) per fit of a step
        backends.append(engine.HipBackend(_capi.DeviceContext(device_of(local))))
    ctx = backends[0].ctx
    # launcher rehearsals with the ranks on one GPU (FOKL_BENCH_SHARE_GPU=1; RCCL refuses two ranks on a device):
    # FOKL_BENCH_SHARDED_OVER_TCP=1 lets the sharded modes carry their exchange steps over the control plane instead
    over_tcp = os.environ.get('FOKL_BENCH_SHARDED_OVER_TCP', '0') == '1'
    comm, comm_kind = bring_up_comm(ctx, rank, world, use_rccl, need_rccl=one_fit_for_all and not over_tcp)
    if one_fit_for_all and world > 1 and comm_kind != 'RCCL':
        comm_kind = 'TCP control plane carries the exchange steps (rehearsal: FOKL_BENCH_SHARDED_OVER_TCP=1)'
        backends = [HostReducedBackend(b.ctx, comm) for b in backends]
    # barriers and the gathers of timing figures go over the TCP control plane when there is one (N > 1): a launch of
    # independent fits then runs no RCCL collective at all, RCCL carries the data path of the sharded modes only
    control = getattr(comm, 'control', None) or comm

    units = [0] if one_fit_for_all else [rank * fits_per_step + i for i in range(fits_per_step)]
    fits = []                                                 # (model, backend, x, y, spec, n_local)
    prep_s = clean_s = upload_s = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for unit, backend in zip(units, backends):
            x, y, spec = config_workload(cfg, unit, args.rows)
            if args.inputs and args.inputs != spec['inputs']:
                x, y = make_workload(spec['seed'], spec['rows'], args.inputs)
                spec['inputs'] = args.inputs
            kernel, phis, _ = kernel_and_phis(spec)
            hypers = {}
            clean_kw = dict(clean=True)
            n_local = spec['rows']
            if mode in ('rows', 'hybrid'):
                # rank r holds rows [lo, hi) of the one dataset; the data-driven defaults of b / btau (FR:1322-1348)
                # need the global mean and variance of y: one all-gather.  A shard must not be rescaled by its own
                # min / max: clean's normalisation (FR:436-437) is applied here with the bounds of the WHOLE dataset (one
                # more all-gather) -- the inputs of the single-process fit, and of the golden, bit for bit (round 6: the
                # raw U[0,1) columns, 1e-6 from their min-max form, left the sharded fit 5e-7 from the golden's BICs)
                lo, hi = dist.shard_range(spec['rows'], rank, world)
                x, y = x[lo:hi], y[lo:hi]
                n_local = hi - lo
                ends = control.allgather([float(v) for v in x.min(axis=0)] + [float(v) for v in x.max(axis=0)])
                lows, highs = ends[:, :x.shape[1]].min(axis=0), ends[:, x.shape[1]:].max(axis=0)
                x = np.ascontiguousarray(x, dtype=np.float64).copy()
                for k in range(x.shape[1]):
                    x[:, k] = (x[:, k] - lows[k]) / (highs[k] - lows[k])
                mom = control.allgather([n_local, float(np.sum(y)), float(np.sum(y * y))])
                mean = float(np.sum(mom[:, 1]) / spec['rows'])
                var = float(np.sum(mom[:, 2]) / spec['rows'] - mean * mean)
                hypers = dict(b=var * (4 + 1), btau=abs(mean) / var * (4 + 1))
                clean_kw = dict(clean=True, normalize=False)
            model = FoKLRoutines.FoKL(kernel=kernel, phis=phis, UserWarnings=False, ConsoleOutput=False, **hypers,
                                      **spec['fit'])
            model._backend_override = backend
            t0 = time.perf_counter()
            model._prepare_fit(x, y, clean_kw)                # format, normalise, defaults, H2D upload (untimed)
            prep_s += time.perf_counter() - t0
            clean_s += model.prepare_stats['clean_s']
            upload_s += model.prepare_stats['upload_s']
            fits.append((model, backend, x, y, spec, n_local))
    spec0 = fits[0][4]
    n, m = spec0['rows'], spec0['inputs']

    def one_fit_on_thread(k):
        model, backend, _, _, spec, n_local = fits[k]
        model._search(backend, n_local, m, rng_state=np.random.RandomState(spec['seed_fit']).get_state())
        backend.ctx.sync()
        return model.fit_stats

    executor = None
    if concurrent > 1:
        import threading
        from concurrent.futures import ThreadPoolExecutor
        slot_lock, next_slot = threading.Lock(), [0]

        def pin_worker():
            with slot_lock:
                k = next_slot[0]
                next_slot[0] += 1
            if domains[k % concurrent]:
                try:
                    os.sched_setaffinity(0, domains[k % concurrent])
                except OSError:
                    pass

        executor = ThreadPoolExecutor(concurrent, initializer=pin_worker)

    def one_step():
        if executor is not None:
            return list(executor.map(one_fit_on_thread, range(len(fits))))
        stats = []
        for model, backend, _, _, spec, n_local in fits:
            np.random.seed(spec['seed_fit'])
            if mode == 'rows':
                model._search(backend, n_local, m, n_global=spec['rows'], row_sharded=True)
            elif mode == 'hybrid':
                model._search(backend, n_local, m, n_global=spec['rows'], row_sharded=True, comm=comm,
                              candidate_sharded=use_rccl)
            elif mode == 'candidates' and use_rccl:
                model._search(backend, n_local, m, comm=comm, candidate_sharded=True)
            else:
                model._search(backend, n_local, m)
            backend.ctx.sync()
            stats.append(model.fit_stats)
        return stats

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for _ in range(args.warmup):
            one_step()
        for _, backend, *_ in fits:
            backend.ctx.timing_enable(True)
            backend.ctx.timing_reset()
        control.barrier()
        ctx.sync()
        cpu0 = time.process_time()                            # all threads of this process (pool, dispatcher, driver)
        from fokl_gpy_amd import _capi as _capi_cpu
        kinds0, driver0 = _capi_cpu.thread_cpu_seconds(), time.thread_time()
        t0 = time.perf_counter()
        logical = physical = calls = 0
        host = dict(t_eigh=0.0, t_resid=0.0, t_chain=0.0, chains_materialised=0, pool_noise_s=0.0, pool_chain_s=0.0,
                    pool_finish_s=0.0, pool_spectral_s=0.0, tapes_rewound=0, forecasts_used=0, spectral_remote=0,
                    exchanges=0, chains_skipped=0, spectral_submitted=0, resid_matrix_free=0, bic_from_gram=0,
                    noise_queue_wait_s=0.0, noise_verdict_wait_s=0.0, device_chains=0, chains_fetched=0, guessed=0,
                    guess_waits=0, guesses_verified=0, searches_repeated=0, dchain_dispatch_s=0.0, dchain_kernel_s=0.0,
                    dchain_timed=0, t_final_verify=0.0,
                    t_final_draws=0.0, t_search_body=0.0, t_teardown=0.0, seconds=0.0, t_kill_loop=0.0, pool_bulk_s=0.0,
                    walker_wait_s=0.0, stream_segments=0, gamma_attempts_exact=0, tapes_wasted=0, tapes_materialised=0,
                    rows_chains=0, path_repredicted=0, t_pool_up=0.0, spectral_device=0, spectral_updated=0,
                    direct_tests=0, chains_cancelled=0, t_settle=0.0, t_head_start=0.0, t_pool_create=0.0)
        direct_max_rel = guess_max_dev = 0.0                  # (maxima, not sums: kept out of `host`)
        drivers = set()
        for _ in range(args.steps):
            for st in one_step():
                drivers.add(st.get('search_driver', 'python'))
                logical += st['terms_logical']
                physical += st['terms_physical']
                calls += st['gibbs_calls']
                for key in host:
                    host[key] += st.get(key, 0)
                direct_max_rel = max(direct_max_rel, st.get('direct_max_rel', 0.0))
                guess_max_dev = max(guess_max_dev, st.get('guess_max_dev', 0.0))
                for key, value in st.get('phases', {}).items():
                    host['phase_' + key] = host.get('phase_' + key, 0.0) + value
        ctx.sync()
        control.barrier()
        elapsed = time.perf_counter() - t0
        cpu_s = time.process_time() - cpu0
        # ... and by kind of thread (fokl_thread_cpu_seconds: a pool's threads are counted when their fit's pool ends; the
        # driver is this Python thread; `other` = the rest of the process: HIP runtime threads, fits on other threads)
        kinds1, driver1 = _capi_cpu.thread_cpu_seconds(), time.thread_time()
        cpu_by_thread = {k: (kinds1[k] - kinds0[k]) / max(args.steps, 1) for k in kinds1}
        cpu_by_thread['driver'] = (driver1 - driver0) / max(args.steps, 1)
        cpu_by_thread['other'] = cpu_s / max(args.steps, 1) - sum(cpu_by_thread.values())
        end_state = np.random.get_state()
        for _, backend, *_ in fits:
            backend.ctx.timing_enable(False)

    kern = {}
    for name, kid in KERNEL_SLOTS():
        tot = dict(ms=0.0, launches=0, bytes=0.0, flops=0.0, ideal_ms=0.0)
        for _, backend, *_ in fits:
            t = backend.ctx.timing_get(kid)
            for key in tot:
                tot[key] += t[key]
        kern[name] = tot

    # ---- parity of what was just timed (rank 0, first fit of the step) against the oracle's golden ----------------
    parity_checked, parity = False, None
    if rank == 0 and not args.no_parity and mode != 'rows':
        model0 = fits[0][0]
        key = (cfg, units[0], n)
        if key in GOLDENS and not GOLDENS[key][1] and not args.inputs and \
                os.path.exists(os.path.join(ROOT, 'tests', 'golden', GOLDENS[key][0] + '.npz')):
            state = end_state if fits_per_step == 1 else None
            if state is None:                                  # several fits per step: rerun the first one on its own
                np.random.seed(spec0['seed_fit'])
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    model0._search(fits[0][1], n, m)
                state = np.random.get_state()
            parity = compare_with_golden(GOLDENS[key][0], model0, model0.betas, model0.mtx, model0.evs, state)
            parity['workload'] = 'the timed fit itself (last step)'
            parity_checked = True
        elif cfg == 3 and not args.inputs and world == 1:
            # no oracle can run configs[3] with 2000 Gibbs iterations on 585-column models (O(P^3) products per
            # iteration): the golden is the benchmark's own dataset -- N = 1e6, stages capped at 3 -- with chains of
            # burnin 250 + draws 250 (round 4; 3.4 h of oracle -- long enough for the kill tests' Monte-Carlo statistics to
            # settle: it selects 68 terms where the 30 + 30 twin of round 3 selects 79), else that twin, or, should both
            # files be missing, the N = 1e5 variant -- fitted here, untimed
            big = GOLDENS[(3, 0, 1_000_000)]
            rows_checked = n if (n == 1_000_000 and os.path.exists(
                os.path.join(ROOT, 'tests', 'golden', big[0] + '.npz'))) else 100_000
            name, over = GOLDENS[(3, 0, rows_checked)]
            if os.path.exists(os.path.join(ROOT, 'tests', 'golden', name + '.npz')):
                xs, ys, sp = config_workload(3, 0, rows_checked)
                kernel, phis, _ = kernel_and_phis(sp)
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    side = FoKLRoutines.FoKL(kernel=kernel, phis=phis, UserWarnings=False, ConsoleOutput=False)
                    side._backend_override = engine.HipBackend(_capi.DeviceContext(device_of(local)))
                    np.random.seed(sp['seed_fit'])
                    sb, sm, se = side.fit(xs, ys, clean=True, **sp['fit'], **over)
                parity = compare_with_golden(name, side, sb, sm, se, np.random.get_state())
                parity['workload'] = f'configs[3] (stages capped at 3) at N={rows_checked}, burnin {over["burnin"]} + draws ' \
                                     f'{over["draws"]} (chains shortened: the oracle\'s O(P^3) products per Gibbs ' \
                                     f'iteration), fitted after the timed region'
                parity_checked = True
                side._backend_override.ctx.close()
    if cfg == 3 and world > 1 and one_fit_for_all and not args.inputs and not args.no_parity and n == 1_000_000:
        # N > 1, one fit for all ranks: the SHARDED fit itself once more with the golden's chain length (burnin 250 + draws
        # 250, see above), on the data every rank already holds -- compared on EVERY rank, the verdicts gathered
        name, over = GOLDENS[(3, 0, 1_000_000)]
        if os.path.exists(os.path.join(ROOT, 'tests', 'golden', name + '.npz')):
            model0, backend0, _, _, _, n_local0 = fits[0]
            kept = model0.burnin, model0.draws
            model0.burnin, model0.draws = over['burnin'], over['draws']
            np.random.seed(spec0['seed_fit'])
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                if mode == 'rows':
                    model0._search(backend0, n_local0, m, n_global=spec0['rows'], row_sharded=True)
                elif mode == 'hybrid':
                    model0._search(backend0, n_local0, m, n_global=spec0['rows'], row_sharded=True, comm=comm,
                                   candidate_sharded=True)
                else:
                    model0._search(backend0, n_local0, m, comm=comm, candidate_sharded=True)
            backend0.ctx.sync()
            mine = compare_with_golden(name, model0, model0.betas, model0.mtx, model0.evs, np.random.get_state())
            model0.burnin, model0.draws = kept
            verdicts = control.allgather([1.0 if mine.get('ok') else 0.0, float(mine.get('max_rel_bic', 1.0)),
                                          float(mine.get('max_draw_err_over_scale', 1.0))])
            if rank == 0:
                parity = mine
                parity['workload'] = (f'the {mode}-sharded fit of configs[3] over {world} ranks with burnin {over["burnin"]} + '
                                      f'draws {over["draws"]} (the golden\'s chain length), after the timed region')
                parity['ok_on_every_rank'] = bool(np.all(verdicts[:, 0] == 1.0))
                parity['max_rel_bic_by_rank'] = verdicts[:, 1].tolist()
                parity['max_draw_err_over_scale_by_rank'] = verdicts[:, 2].tolist()
                parity['ok'] = bool(parity.get('ok')) and parity['ok_on_every_rank']
                parity_checked = True

    # What a drop-in user sees: one `fit(inputs, data, clean=True)` call from host arrays to returned draws (formatting,
    # normalisation, defaults, H2D upload + transposition, the search) -- after the timed region, on a warm process,
    # rank 0, the same dataset and chain seed (so the same search).  Never `value`: the metric's denominator starts
    # after `clean` with the inputs resident in HBM.
    fit_call, issued_ratio = None, None
    if rank == 0 and mode == 'fits' and cfg != 4 and not args.inputs:
        model0, backend0, x0, y0, sp0, _ = fits[0]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            kernel, phis, _ = kernel_and_phis(sp0)
            user = FoKLRoutines.FoKL(kernel=kernel, phis=phis, UserWarnings=False, ConsoleOutput=False, **sp0['fit'])
            user._backend_override = backend0
            np.random.seed(sp0['seed_fit'])
            trace_path = os.path.join(os.environ.get('TMPDIR', '/tmp'), f'fokl_gram_trace_{os.getpid()}.txt')
            trace_was = os.environ.get('FOKL_GRAM_TRACE')
            if trace_was is None:                          # (one line per Gram launch of this untimed fit: shapes and classes)
                os.environ['FOKL_GRAM_TRACE'] = trace_path
                open(trace_path, 'w').close()
            t0 = time.perf_counter()
            user.fit(x0, y0, clean=True)
            backend0.ctx.sync()
            if trace_was is None:
                os.environ.pop('FOKL_GRAM_TRACE', None)
                try:
                    issued_ratio = mfma_flops_issued_over_algorithmic(open(trace_path).read().splitlines())
                    os.remove(trace_path)
                except Exception:                          # (a diagnostic: never the reason a bench line is lost)
                    issued_ratio = None
            fit_call = dict(fit_call_ms=1e3 * (time.perf_counter() - t0), clean_ms=1e3 * user.prepare_stats['clean_s'],
                            upload_ms=1e3 * user.prepare_stats['upload_s'], search_ms=1e3 * user.fit_stats['seconds'],
                            note='one FoKL.fit(inputs, data, clean=True) from host arrays, warm process, after the timed '
                                 'region; upload = H2D copy of inputs and data + transposition to structure-of-arrays')

    # Outside the timed region: the basis-build kernel back to back on the workload's three sub-stage shapes.
    # Inside a fit the GPU idles between launches (the fit is bound by the serial random stream on the host), so the
    # in-situ average above is taken at idle clocks; this is the same kernel at sustained clocks.
    hot = {}
    if rank == 0 and not args.no_microbench and m >= 2:
        ctx.timing_enable(True)
        ctx.reserve_slots(2 + 2 * m * (m - 1))
        for label, pattern in ((f'T={m} (1)', [1, 0]), (f'T={m * (m - 1) // 2} (1,1)', [1, 1]),
                               (f'T={m * (m - 1)} (2,1)', [2, 1])):
            if max(pattern) > len(fits[0][0].phis):
                continue
            terms = engine.distinct_arrangements(pattern + [0] * (m - 2)).astype(np.int32)
            slots = np.arange(2, 2 + terms.shape[0], dtype=np.int32)
            for _ in range(3):
                ctx.build_terms(terms, slots)
            ctx.sync()
            ctx.timing_reset()
            for _ in range(20):
                ctx.build_terms(terms, slots)
            t = ctx.timing_get(_capi.K_BASIS)
            gbs = t['bytes'] / (t['ms'] * 1e-3) / 1e9
            hot[label] = dict(avg_us=1e3 * t['ms'] / t['launches'], achieved=gbs, frac=gbs / HBM_PEAK_GBS)
        ctx.timing_enable(False)

    # Also outside the timed region: what this device sustains for the kernels' access mixes (trivial streaming kernels)
    # and for fp64 MFMA on register operands -- context for the roofline fractions, which stay against the spec peaks.
    sustained = None
    if rank == 0 and not args.no_microbench:
        sustained = {'unit': 'GB/s and TFLOP/s', 'hbm_read_GBps': ctx.probe(0) / 1e9, 'hbm_write_GBps': ctx.probe(1) / 1e9,
                     'hbm_1_read_7_writes_GBps': ctx.probe(2) / 1e9, 'mfma_f64_TFLOPs': ctx.probe(3) / 1e12}

    throughput_mode = None
    if side_job is not None:
        side_procs, gate, start_b, done_q, side_workers = side_job
        try:
            gate.set()
            start_b.wait(timeout=300)                       # all of them have uploaded and warmed up
            cg0 = cgroup_cpu()
            t_side = time.perf_counter()
            own_terms = 0
            own_speculation = os.environ.get('FOKL_SPECULATION')
            # this process fits along under the workers' settings where they are read per search (order book, thread plan;
            # the polling budgets are fixed when the library loads): the measurement ends when the slowest process has done
            # its fits, and with a single fit's thread plan (2 / 2 / 8 host threads against the workers' 1 / 1 / 3) that was
            # this one (48 ms per fit against the workers' 39-45)
            side_env = {'FOKL_SPECULATION': THROUGHPUT_SPECULATION, 'FOKL_CHAIN_THREADS': '1', 'FOKL_FINISH_THREADS': '1',
                        'FOKL_WALK_HELPERS': '0',
                        'FOKL_SPECTRAL_THREADS': '3' if os.environ.get('FOKL_CHAIN', 'auto') != 'host' or side_procs + 1 <= 2 else '2'}
            side_env = {k: v for k, v in side_env.items() if k not in os.environ}
            os.environ.update(side_env)
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    for _ in range(SIDE_FITS):              # this process fits along: one more fitting process on the GPU
                        own_terms += sum(st['terms_logical'] for st in one_step())
            finally:
                for k in side_env:
                    os.environ.pop(k, None)
            side_res = [done_q.get(timeout=300) for _ in side_workers]
            t_side = time.perf_counter() - t_side
            cg1 = cgroup_cpu()
            side_terms = own_terms + sum(r['stats']['terms_logical'] for r in side_res)
            side_fits = (side_procs + 1) * SIDE_FITS
            throughput_mode = dict(
                workload=f'{side_procs + 1} processes (this one and {side_procs} workers), each fitting its own configs[2] '
                         f'dataset (dataset seeds 12 .. {12 + side_procs}) {SIDE_FITS} times, back to back, on this one GPU',
                procs=side_procs + 1, value=side_terms / t_side, unit='candidate-terms/s', fits_per_s=side_fits / t_side,
                seconds=t_side, ms_per_fit_per_process=1e3 * t_side / SIDE_FITS,
                chain_mode=os.environ.get('FOKL_CHAIN', 'auto'),
                order_book_depth=int(own_speculation or THROUGHPUT_SPECULATION),   # FOKL_SPECULATION of every process
                worker_s_per_fit={key: sum(r['stats'].get(key, 0.0) for r in side_res) / max(1, side_procs * SIDE_FITS)
                                  for key in ('t_eigh', 't_chain', 'pool_noise_s', 'pool_spectral_s', 'noise_verdict_wait_s',
                                              'noise_queue_wait_s', 'seconds', 'cpu_s')},
                worker_elapsed_s=[r['elapsed'] for r in side_res],
                # the CPU controller over the measurement: CPUs' worth of time the container used against its quota, and how
                # many scheduler periods ended with its threads throttled (every thread of every fit stands still then)
                host_cpu=dict(quota_cpus=cg0.get('quota_cpus'),
                              cpus_used=(cg1.get('usage_usec', 0) - cg0.get('usage_usec', 0)) / 1e6 / t_side
                              if 'usage_usec' in cg1 else None,
                              periods=cg1.get('nr_periods', 0) - cg0.get('nr_periods', 0),
                              periods_throttled=cg1.get('nr_throttled', 0) - cg0.get('nr_throttled', 0),
                              throttled_s=(cg1.get('throttled_usec', 0) - cg0.get('throttled_usec', 0)) / 1e6))
        except Exception as exc:
            print(f"bench.py: throughput side measurement failed: {type(exc).__name__} {exc}", file=sys.stderr)
        finally:
            for w in side_workers:
                w.join(5)
                if w.is_alive():
                    w.terminate()

    # the figures of the timed region, over the control plane: safe before anything below can go wrong
    gathered = control.allgather([elapsed, logical, physical, calls])

    # N > 1, independent fits (the weak-scaling default of configs[2]): north_star's own split measured in the same run --
    # every rank uploads the SAME dataset (unit 0) and the ranks fit it together, candidate models dealt over the ranks with
    # one RCCL all-gather per window of candidates -- so that a scaling run records both curves.  Secondary: `value` stays the
    # replica figure, and the measurement runs on a helper thread under a deadline (FOKL_BENCH_SHARDED_TIMEOUT seconds from
    # the moment every rank is ready): these are the first RCCL collectives of the launch, and one that never returns must
    # cost the side figure, not the line.  The ranks agree over the control plane whether it finished everywhere; if not,
    # nothing touches the device or the communicator afterwards and the process leaves through os._exit.
    sharded_line, wedged = None, False
    # (launcher rehearsals on a box where RCCL cannot come up: FOKL_BENCH_SHARDED_OVER_TCP=1 runs the same measurement with
    # the windows gathered over the control plane; FOKL_BENCH_SHARDED_TEST_HANG=<rank> makes that rank never come back)
    joint_transport_ok = comm_kind == 'RCCL' or os.environ.get('FOKL_BENCH_SHARDED_OVER_TCP', '0') == '1'
    if world > 1 and mode == 'fits' and cfg in (1, 2) and joint_transport_ok and not args.no_throughput:
        import threading
        box = {}

        def joint_measurement():
            try:
                if os.environ.get('FOKL_BENCH_SHARDED_TEST_HANG', '') == str(rank):
                    time.sleep(3600)
                xs, ys, sp = config_workload(cfg, 0, args.rows)
                kernel, phis, _ = kernel_and_phis(sp)
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    joint = FoKLRoutines.FoKL(kernel=kernel, phis=phis, UserWarnings=False, ConsoleOutput=False, **sp['fit'])
                    joint._backend_override = backends[0]
                    joint._prepare_fit(xs, ys, dict(clean=True))

                    def joint_fit():
                        np.random.seed(sp['seed_fit'])
                        box['result'] = joint._search(backends[0], sp['rows'], sp['inputs'], comm=comm,
                                                      candidate_sharded=True)
                        backends[0].ctx.sync()
                        return joint.fit_stats

                    joint_fit()                             # warm-up
                    comm.barrier()
                    t0 = time.perf_counter()
                    terms = sum(joint_fit()['terms_logical'] for _ in range(max(1, args.steps)))
                    backends[0].ctx.sync()
                    comm.barrier()
                    box['seconds'] = time.perf_counter() - t0
                box.update(terms=terms, spectral_remote=int(joint.fit_stats.get('spectral_remote', 0)),
                           exchanges=int(joint.fit_stats.get('exchanges', 0)),
                           search_driver=joint.fit_stats.get('search_driver'),
                           gathers=int(joint.fit_stats.get('candidate_gathers', 0)),
                           gather_s=float(joint.fit_stats.get('t_candidate_gather', 0.0)),
                           substages=int(joint.fit_stats.get('substages', 0)))
                # the joint fit against the same golden as the timed fit (every rank checks its own copy of the result)
                key = (cfg, 0, sp['rows'])
                if key in GOLDENS and not GOLDENS[key][1] and not args.inputs and \
                        os.path.exists(os.path.join(ROOT, 'tests', 'golden', GOLDENS[key][0] + '.npz')):
                    jb, jm, je = box['result']
                    check = compare_with_golden(GOLDENS[key][0], joint, jb, jm, je, np.random.get_state())
                    box['parity'] = {k: check.get(k) for k in ('golden', 'ok', 'mtx_equal', 'gibbs_calls_equal',
                                                               'max_rel_bic', 'max_draw_err_over_scale',
                                                               'numpy_stream_equal')}
            except BaseException as exc:                    # noqa: BLE001 -- reported below, never costs the line
                box['error'] = f'{type(exc).__name__} {exc}'

        state_before = np.random.get_state()
        control.barrier()                                   # rank 0 arrives last (its side measurements above)
        worker = threading.Thread(target=joint_measurement, name='fokl-bench-joint', daemon=True)
        worker.start()
        worker.join(float(os.environ.get('FOKL_BENCH_SHARDED_TIMEOUT', '180')))
        finished = not worker.is_alive()                    # decided once
        if not finished:
            box.setdefault('error', 'did not finish before the deadline (an RCCL collective that never returned?)')
        parity_flag = -1.0 if 'parity' not in box else (1.0 if box['parity']['ok'] else 0.0)
        report = control.allgather([1.0 if finished else 0.0, 1.0 if 'error' in box else 0.0, box.get('seconds', 0.0),
                                    parity_flag])
        wedged = float(np.min(report[:, 0])) < 1.0
        if wedged or float(np.max(report[:, 1])) > 0.0:
            why = box.get('error', 'failed on another rank')
            print(f"bench.py: rank {rank}: candidate-sharded side measurement failed: {why}", file=sys.stderr)
            sharded_line = dict(mode='candidates', value=None, error=why)
        else:
            t_joint = float(np.max(report[:, 2]))
            sharded_line = dict(mode='candidates', value=box['terms'] / t_joint, unit='candidate-terms/s',
                                ms_per_step=1e3 * t_joint / max(1, args.steps), scaling='strong',
                                search_driver=box.get('search_driver'),
                                # native driver: one all-gather of the candidates' Gram rows per forward step (sub-stage);
                                # Python loop (FOKL_SEARCH_DIST=python): G2 jobs dealt over the ranks, gathered by windows
                                all_gathers_per_fit=box.get('gathers'), all_gather_s_per_fit=box.get('gather_s'),
                                substages_per_fit=box.get('substages'),
                                spectral_remote=box['spectral_remote'], exchanges=box['exchanges'],
                                parity=box.get('parity'),
                                parity_ok_on_every_rank=None if float(np.max(report[:, 3])) < 0.0
                                else bool(float(np.min(report[:, 3])) == 1.0),
                                transport='RCCL' if comm_kind == 'RCCL' else 'TCP control plane (launcher rehearsal)',
                                note='ONE fit of the unit-0 dataset by all ranks together, every rank repeating the search: '
                                     'the Gram rows of each forward step\'s candidate terms are computed one share per rank '
                                     'and all-gathered, one gather per forward step (north_star\'s split); after the timed '
                                     'region of the independent fits')
        if finished:
            np.random.set_state(state_before)

    def leave(code=0):
        """End of the process.  After a collective that never returned: no communicator or device teardown (they would
        wait for it), streams flushed, hard exit."""
        if wedged:
            control.close()
            sys.stdout.flush()
            sys.stderr.flush()
            dist.flush_c_streams()
            os._exit(code)
        comm.close()

    if rank != 0:
        leave(4 if wedged else 0)
        return
    t_max = float(np.max(gathered[:, 0]))
    if one_fit_for_all:                                     # every rank ran the same search: count it once
        tot_logical, tot_physical = float(gathered[0, 1]), float(gathered[0, 2])
    else:
        tot_logical = float(np.sum(gathered[:, 1]))
        tot_physical = float(np.sum(gathered[:, 2]))
    fits_total = (1 if one_fit_for_all else world) * fits_per_step * max(args.steps, 1)

    kernels, dominant, gpu_ms = kernel_report(kern, n, m, cfg)

    parallelism = 'single GPU'
    if world > 1:
        parallelism = {'fits': f'independent fits x{world}', 'rows': f'rows sharded x{world}, RCCL all-reduce of Gram '
                       f'blocks', 'candidates': f'candidate models sharded x{world}, RCCL all-gather of per-candidate '
                       f'BIC + spectral factors', 'hybrid': f'rows sharded x{world} (RCCL all-reduce of Gram blocks on the '
                       f'device) + candidate models dealt over the ranks (RCCL all-gather of BIC + spectral factors)'}[mode]
    line = {
        'metric': 'candidate-terms/sec (basis build + Gibbs + BIC)',
        'value': tot_logical / t_max,
        'unit': 'candidate-terms/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * t_max / max(args.steps, 1),
        'higher_is_better': True,
        'scaling': 'strong' if one_fit_for_all else 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': spec0['label'] + (f', {fits_per_step} fits per rank and step, {concurrent} at a time' if cfg == 4 else
                                                 ', one full forward-selection fit per step'),
                   'config_index': cfg, 'rows': n, 'inputs': m, 'parallelism': parallelism, 'collectives': comm_kind,
                   'terms_counted': 'logical (reference-equivalent, FoKLRoutines.py:1461)'},
        'value_physical': tot_physical / t_max,
        # both numbers wherever the headline stands.  `value` counts LOGICAL terms -- the columns the
        # reference would have built for this sequence of gibbs() calls; the device builds each candidate column
        # once per sub-stage (kill tests read sub-blocks of the sub-stage's Gram) and runs a chain only for the
        # accepted kill tests whose draws something looks at (the stream position of the others is still advanced)
        'what_is_counted': {
            'logical_terms_per_fit': tot_logical / fits_total,
            'columns_built_on_the_device_per_fit': tot_physical / fits_total,
            'value_logical_terms_per_s': tot_logical / t_max,
            'value_physical_columns_per_s': tot_physical / t_max,
            'kill_tests_per_fit': host.get('direct_tests', 0.0) / max(args.steps, 1) / max(fits_per_step, 1),
            'kill_test_chains_never_run_per_fit': host.get('chains_cancelled', 0.0) / max(args.steps, 1) / max(fits_per_step, 1),
        },
        'parity_checked': parity_checked,
        'parity': parity,
        'fits_per_s': fits_total / t_max,
        'terms_logical_per_fit': tot_logical / fits_total,
        'terms_physical_per_fit': tot_physical / fits_total,
        'gibbs_calls_per_fit': float(np.sum(gathered[:, 3])) / world / (fits_per_step * max(args.steps, 1)),
        'gpu_kernel_ms_per_step': gpu_ms / max(args.steps, 1),
        'host_prepare_s': prep_s,
        'clean_ms': 1e3 * clean_s / max(len(fits), 1),
        'upload_ms': 1e3 * upload_s / max(len(fits), 1),
        'fit_call': fit_call,
        'fit_call_ms': fit_call['fit_call_ms'] if fit_call else None,
        'host_main_thread_s_per_step': {k: v / max(args.steps, 1) for k, v in host.items()},
        'cpu_seconds_per_step': cpu_s / max(args.steps, 1),   # process CPU time (every thread) over the timed region
        'cpu_seconds_per_step_by_thread': cpu_by_thread,
        'chain_mode': os.environ.get('FOKL_CHAIN', 'auto'),
        # who ran the kill tests (csrc/fokl_search.cpp or engine.py's loop) and how the random stream reached the chains
        'search_driver': '+'.join(sorted(drivers)),
        # how the kill tests' BICs were decided (fokl_search_set_decide): tests decided from the downdated least-squares model
        # per step, the largest relative difference between such a BIC and the one the accepted model's eigenpairs gave
        # afterwards (every accepted test is checked; the search ends if one exceeds 1e-9), accepted models whose chain
        # never had to run
        'kill_decisions': {'mode': os.environ.get('FOKL_KILL_DECIDE', 'direct'),
                           'direct_tests_per_step': host['direct_tests'] / max(args.steps, 1),
                           'max_rel_difference_to_eigen_bic': direct_max_rel,
                           # second clause of FR:1670 guessed from the least-squares intercept (margin FOKL_GUESS_MARGIN, 2 %):
                           # how far the chains' mean intercepts turned out to lie from it, relative (every guess is checked)
                           'max_rel_distance_of_guessed_intercept_scale': guess_max_dev,
                           'chains_never_started_per_step': host['chains_cancelled'] / max(args.steps, 1),
                           'waiting_for_eigenpairs_s_per_step': host['t_settle'] / max(args.steps, 1)},
        'random_stream': {
            'walker_busy_s_per_step': host['pool_noise_s'] / max(args.steps, 1),
            'walker_waiting_for_bulk_s_per_step': host['walker_wait_s'] / max(args.steps, 1),
            'bulk_threads_cpu_s_per_step': host['pool_bulk_s'] / max(args.steps, 1),
            'segments_of_79872_doubles_per_step': host['stream_segments'] / max(args.steps, 1),
            'tapes_expanded_on_the_device_per_step': host['rows_chains'] / max(args.steps, 1),
            'tapes_materialised_on_the_host_after_all_per_step': host['tapes_materialised'] / max(args.steps, 1),
            'gamma_attempts_needing_libm_per_step': host['gamma_attempts_exact'] / max(args.steps, 1),
            'kill_test_path_repredicted_per_step': host['path_repredicted'] / max(args.steps, 1)},
        'cpu_pinning': pinned,
        # (`achieved` books the algorithmic 2 N nr nc of SURVEY 8(d); what the MFMA-bound launches issue on the matrix cores
        # is this fraction of it -- the symmetric part's lower tiles are never computed, ragged tiles are padded: measured on
        # the launches of the untimed fit_call pass)
        'roofline': dict(dominant, mfma_flops_issued_over_algorithmic=issued_ratio) if dominant.get('bound') == 'mfma'
        else dominant,
        'kernels': kernels,
        'basis_build_sustained': hot,
        'device_sustains': sustained,
        'throughput_mode': throughput_mode,
        'candidate_sharded': sharded_line,
        # G3 on the device: one wavefront per chain on streams of their own, concurrent with everything above and with
        # each other -- not on the context's stream, so not among `kernels`; per-kernel durations are in the committed
        # rocprofv3 summary (profiles/rocprof_r03_summary.md: ~1.0 ms per chain of 2000 iterations at up to 64 columns).
        # A chain is a recursion of 2000 dependent iterations: bound by dependent-instruction latency (three long fp64
        # operations + a cross-lane sum per iteration), neither by HBM nor by the matrix pipe; it occupies 1 of the
        # chip's ~8000 wavefront slots, which is why summed kernel durations say nothing about what bounds the device.
        'device_chains': {'chains_per_step': host['device_chains'] / max(args.steps, 1),
                          'guessed_decisions_per_step': host['guessed'] / max(args.steps, 1),
                          'guesses_confirmed_per_step': host['guesses_verified'] / max(args.steps, 1),
                          'searches_repeated': host['searches_repeated'],
                          'dispatcher_cpu_s_per_step': host['dchain_dispatch_s'] / max(args.steps, 1),
                          # by the recursion kernel's own clock, over the chains whose statistics the search read
                          'avg_chain_kernel_ms': 1e3 * host['dchain_kernel_s'] / max(host['dchain_timed'], 1),
                          'chains_timed_per_step': host['dchain_timed'] / max(args.steps, 1),
                          'bound': 'latency (serial recursion, one wavefront per chain)'},
    }
    if not args.no_cpu_baseline:
        line.update(cpu_baselines(fits[0][2], fits[0][3], spec0))
    if not wedged:
        comm.close()
    dist.flush_c_streams()
    sys.stderr.flush()
    print(json.dumps(line), flush=True)          # the ONE JSON line, last thing on stdout
    if parity_checked and not parity['ok']:
        print(f"bench.py: PARITY MISMATCH against {parity['golden']}: {parity}", file=sys.stderr)
        if wedged:
            leave(3)
        sys.exit(3)
    if wedged:
        leave(4)        # the line above is valid, the side measurement's collective never returned: not a clean exit


if __name__ == '__main__':
    main()
