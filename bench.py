#!/usr/bin/env python3
"""
Benchmark of the forward-selection hot path on MI355X -- BASELINE.json's metric:

    candidate-terms/sec (basis build + Gibbs + BIC), N = 1e6, M = 8

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One *step* = one complete forward-selection fit (FoKL.fit's search: every sub-stage's basis build, Gram,
Gibbs chains, BIC and kill tests) over the synthetic configs[2] workload -- N = 1e6 rows, M = 8 inputs,
Bernoulli-polynomial kernel, 2-way interactions, reference-default hyper-parameters (burnin 1000, draws 1000,
tolerance 3) -- with the normalised inputs already resident in HBM when the timed region starts.

`value` = candidate terms / second, where the numerator is the reference-equivalent (logical) count: the sum
over all gibbs evaluations of the columns the reference builds for that evaluation (FoKLRoutines.py:1461),
tallied per call by the search driver.  `terms_physical` is what the GPU really built.

N > 1: every rank fits its own dataset of the same shape (independent fits are the unit that shards without a
data-path collective, BASELINE configs[4]); per-rank time and term counts are exchanged with one RCCL
all-gather; value = all ranks' terms / max-over-ranks time ("weak" scaling).
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


def cpu_baseline(x, y, n_sample=60000):
    """
    The reference's algorithm on the host (oracle, kind "port"): the first two gibbs evaluations of the same
    workload (T = 8 main effects, then + 28 two-way terms) on the first `n_sample` rows with the reference's own
    per-element Python loop structure (FoKLRoutines.py:1446-1485), one core.  The reference's cost at this N is
    its O(N * T * M) X-build, so the rate is extrapolated linearly in N to the benchmark's row count.
    """
    from oracle import fokl_oracle as O
    from fokl_gpy_amd import getKernels
    phis = getKernels.bernoulli()
    n_full, m = x.shape
    xs = np.ascontiguousarray(x[:n_sample])
    ys = np.ascontiguousarray(y[:n_sample])[:, None]
    a = atau = 4
    b, btau = O.default_b_btau(ys, a, atau)
    dtd = ys.T.dot(ys)
    t1 = O.distinct_arrangements(O.deal_indvec(1, m, 2))
    t2 = O.distinct_arrangements(O.deal_indvec(2, m, 2))
    state = np.random.get_state()
    np.random.seed(12345)
    t0 = time.perf_counter()
    r1 = O.gibbs(xs, ys, phis, O.KERNEL_BERNOULLI, [], t1, a, b, atau, btau, 2000, None, xs, b / (1 + a),
                 btau / (1 + atau), dtd, build=O.build_columns_scalar)
    O.gibbs(xs, ys, phis, O.KERNEL_BERNOULLI, r1.X, np.vstack([t1, t2]), a, b, atau, btau, 2000, None, xs,
            b / (1 + a), btau / (1 + atau), dtd, build=O.build_columns_scalar)
    dt = time.perf_counter() - t0
    np.random.set_state(state)
    terms = t1.shape[0] + t2.shape[0]
    rate_sample = terms / dt
    return dict(value=rate_sample * n_sample / n_full, unit='candidate-terms/s', cores=1, kind='port',
                sample=f'oracle scalar path (reference loop structure), first 2 gibbs evaluations ({terms} terms, '
                       f'2000 draws each) on the first {n_sample} of {n_full} rows in {dt:.1f} s; '
                       f'rate scaled by {n_sample}/{n_full} (X-build is linear in N)',
                seconds=dt, terms=terms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--rows', type=int, default=1_000_000)
    ap.add_argument('--inputs', type=int, default=8)
    ap.add_argument('--mode', choices=('fits', 'rows'), default='fits',
                    help="N > 1: 'fits' = every rank fits its own dataset (weak scaling, the default and the driver's "
                         "contract); 'rows' = ONE dataset of --rows rows sharded over the ranks, Gram blocks and residual "
                         "moments all-reduced over RCCL inside the library (strong scaling; for fits that are device "
                         "bound, N >= 5e7)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-microbench', action='store_true',
                    help='skip the back-to-back basis-build launches after the timed region (used for profiler runs)')
    args = ap.parse_args()

    from fokl_gpy_amd import dist
    rank, world, local = dist.env_rank_world()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
            sys.exit(2)
    os.environ['FOKL_DEVICE'] = str(local)

    # Keep this rank's host threads (search driver + the native noise / chain / spectral threads, about ten) on the
    # logical CPUs of ONE L3 domain: noise tapes (about 1 MB each) are handed from thread to thread.  Rank r takes the
    # domain of CPU 8 r (8 cores x 2 SMT threads on the EPYC hosts of this pool).
    pinned = None
    if os.environ.get('FOKL_BENCH_PIN', '1') != '0' and hasattr(os, 'sched_setaffinity'):
        try:
            allowed = set(os.sched_getaffinity(0))
            want = set(range(8 * local, 8 * local + 8))
            try:
                with open(f'/sys/devices/system/cpu/cpu{8 * local}/cache/index3/shared_cpu_list') as fh:
                    want = set()
                    for part in fh.read().strip().split(','):
                        lo, _, hi = part.partition('-')
                        want.update(range(int(lo), int(hi or lo) + 1))
            except (OSError, ValueError):
                pass
            want = sorted(want & allowed)
            if len(want) >= 2:
                os.sched_setaffinity(0, want)
                pinned = want
        except OSError:
            pinned = None

    from fokl_gpy_amd import FoKLRoutines, _capi
    backend = FoKLRoutines.device_backend(local)          # raises without libfokl_hip.so / a gfx950 device
    ctx = backend.ctx
    # FOKL_BENCH_FORCE_RCCL=1 takes the RCCL bootstrap + collectives also in a world of one (launcher smoke test)
    use_rccl = world > 1 or os.environ.get('FOKL_BENCH_FORCE_RCCL', '0') == '1'
    comm = dist.RcclComm(ctx, rank, world) if use_rccl else dist.SingleComm()

    n, m = args.rows, args.inputs
    rows_mode = args.mode == 'rows'
    if rows_mode:
        # one dataset, rank r holds rows [lo, hi): the concatenation of the ranks' shards (seed 12, shard r).  The
        # data-driven defaults of b / btau (FR:1322-1348) need the global mean and variance of y: one all-gather.
        lo, hi = dist.shard_range(n, rank, world)
        x, y = make_workload((12, rank), hi - lo, m)
        mom = comm.allgather([hi - lo, float(np.sum(y)), float(np.sum(y * y))])
        mean = float(np.sum(mom[:, 1]) / n)
        var = float(np.sum(mom[:, 2]) / n - mean * mean)
        hypers = dict(b=var * (4 + 1), btau=abs(mean) / var * (4 + 1))
        seed_fit = 1000                                     # the sampler is replicated: same stream on every rank
    else:
        x, y = make_workload(12 + rank, n, m)
        hypers = {}
        seed_fit = 1000 + rank

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FoKLRoutines.FoKL(kernel='Bernoulli Polynomials', UserWarnings=False, ConsoleOutput=False, **hypers)
        t0 = time.perf_counter()
        # format, normalise, defaults, H2D upload (untimed); a shard must not be rescaled by its own min / max
        model._prepare_fit(x, y, dict(clean=True, normalize=False) if rows_mode else dict(clean=True))
        prep_s = time.perf_counter() - t0

        def one_step():
            np.random.seed(seed_fit)
            if rows_mode:
                model._search(backend, x.shape[0], m, n_global=n, row_sharded=True)
            else:
                model._search(backend, n, m)
            ctx.sync()
            return model.fit_stats

        for _ in range(args.warmup):
            one_step()

        ctx.timing_enable(True)
        ctx.timing_reset()
        comm.barrier()
        ctx.sync()
        t0 = time.perf_counter()
        logical = physical = calls = 0
        host = dict(t_eigh=0.0, t_resid=0.0, t_chain=0.0, chains_materialised=0, pool_noise_s=0.0, pool_chain_s=0.0,
                    pool_finish_s=0.0, pool_spectral_s=0.0, tapes_rewound=0, forecasts_used=0)
        for _ in range(args.steps):
            st = one_step()
            logical += st['terms_logical']
            physical += st['terms_physical']
            calls += st['gibbs_calls']
            for key in host:
                host[key] += st.get(key, 0)
        ctx.sync()
        comm.barrier()
        elapsed = time.perf_counter() - t0
        ctx.timing_enable(False)

    kern = {name: ctx.timing_get(kid) for name, kid in
            (('basis_build', _capi.K_BASIS), ('gram', _capi.K_GRAM), ('resid', _capi.K_RESID))}

    # Outside the timed region: the basis-build kernel back to back on the workload's three sub-stage shapes.
    # Inside a fit the GPU idles between launches (the fit is bound by the serial random stream on the host), so the
    # in-situ average above is taken at idle clocks; this is the same kernel at sustained clocks.
    hot = {}
    if rank == 0 and not args.no_microbench:
        from fokl_gpy_amd import engine
        ctx.timing_enable(True)
        ctx.reserve_slots(2 + 56)
        for label, pattern in (('T=8 (1)', [1, 0]), ('T=28 (1,1)', [1, 1]), ('T=56 (2,1)', [2, 1])):
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

    gathered = comm.allgather([elapsed, logical, physical, calls])
    if rank != 0:
        comm.close()
        return
    t_max = float(np.max(gathered[:, 0]))
    if rows_mode:                                           # every rank ran the same search on its rows: count it once
        tot_logical, tot_physical = float(gathered[0, 1]), float(gathered[0, 2])
    else:
        tot_logical = float(np.sum(gathered[:, 1]))
        tot_physical = float(np.sum(gathered[:, 2]))

    def roof(name, bound):
        k = kern[name]
        if k['launches'] == 0 or k['ms'] <= 0:
            return None
        avg_ms = k['ms'] / k['launches']
        if bound == 'hbm':
            achieved = k['bytes'] / k['launches'] / (avg_ms * 1e-3) / 1e9
            return dict(kernel=name, bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=achieved / HBM_PEAK_GBS, traffic=None, launches=k['launches'], avg_ms=avg_ms,
                        total_ms=k['ms'])
        achieved = k['flops'] / k['launches'] / (avg_ms * 1e-3) / 1e12
        return dict(kernel=name, bound='mfma', achieved=achieved, peak=FP64_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                    frac=achieved / FP64_MFMA_PEAK_TFLOPS, traffic=None, launches=k['launches'], avg_ms=avg_ms,
                    total_ms=k['ms'])

    kernels = {'basis_build': roof('basis_build', 'hbm'), 'gram': roof('gram', 'hbm'), 'resid': roof('resid', 'hbm')}
    if kernels['gram']:
        # the Gram kernel crosses the ridge (HBM bound below ~40 columns, fp64-MFMA bound above): give both readings
        gm = roof('gram', 'mfma')
        kernels['gram']['mfma_achieved_tflops'] = gm['achieved']
        kernels['gram']['mfma_frac'] = gm['frac']
        # ... and the launch-by-launch roofline: sum of max(bytes / HBM peak, flops / MFMA peak) over measured time
        kernels['gram']['roofline_frac'] = kern['gram']['ideal_ms'] / kern['gram']['ms']
    # HBM traffic per launch from the committed PMC passes of this same workload (profiles/pmc_r01.json, produced by
    # tools/profile_r01.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 note).
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_r01.json')
    if os.path.exists(pmc_path):
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        if pmc.get('workload') == {'rows': n, 'inputs': m}:
            for name, k in kernels.items():
                if k and name in pmc['kernels']:
                    k['traffic'] = pmc['kernels'][name]['hbm_bytes_per_launch']
                    k['algorithmic_bytes_per_launch'] = kern[name]['bytes'] / kern[name]['launches']
    dominant = max((k for k in kernels.values() if k), key=lambda k: k['total_ms'])
    gpu_ms = sum(k['total_ms'] for k in kernels.values() if k)

    line = {
        'metric': 'candidate-terms/sec (basis build + Gibbs + BIC)',
        'value': tot_logical / t_max,
        'unit': 'candidate-terms/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * t_max / max(args.steps, 1),
        'higher_is_better': True,
        'scaling': 'strong' if rows_mode else 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': f'configs[2]: synthetic N={n}, M={m}, Bernoulli Polynomials, 2-way interactions, '
                               f'burnin 1000 + draws 1000, one full forward-selection fit per step',
                   'rows': n, 'inputs': m, 'parallelism': (f'rows sharded x{world}, RCCL all-reduce of Gram blocks' if rows_mode else
                                   f'independent fits x{world}') if world > 1 else 'single GPU',
                   'terms_counted': 'logical (reference-equivalent, FoKLRoutines.py:1461)'},
        'terms_logical_per_step': tot_logical / (1 if rows_mode else world) / max(args.steps, 1),
        'terms_physical_per_step': tot_physical / (1 if rows_mode else world) / max(args.steps, 1),
        'gibbs_calls_per_step': float(np.sum(gathered[:, 3])) / world / max(args.steps, 1),
        'gpu_kernel_ms_per_step': gpu_ms / max(args.steps, 1),
        'host_prepare_s': prep_s,
        'host_main_thread_s_per_step': {k: v / max(args.steps, 1) for k, v in host.items()},
        'cpu_pinning': pinned,
        'roofline': dominant,
        'kernels': kernels,
        'basis_build_sustained': hot,
        'device_sustains': sustained,
    }
    if not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(x, y)
    comm.close()
    dist.flush_c_streams()
    sys.stderr.flush()
    print(json.dumps(line), flush=True)          # the ONE JSON line, last thing on stdout


if __name__ == '__main__':
    main()
