"""How many host-bound fits can one MI355X serve at once?  P worker PROCESSES (no shared interpreter lock), each with its
own device context, host threads and L3 domain, fit configs[4] units (N = 1e5, M = 8) back to back; fits per second
over all of them.  usage: throughput_procs.py P [fits per worker] [rows]"""
import multiprocessing as mp
import os, sys, time, warnings
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)


def worker(k, procs, fits, rows, start, done):
    import numpy as np
    import bench
    dom = bench.l3_domain_of(8 * k)
    if dom:
        os.sched_setaffinity(0, dom)
    os.environ.setdefault('FOKL_CHAIN_THREADS', '1')
    os.environ.setdefault('FOKL_FINISH_THREADS', '1')
    os.environ.setdefault('FOKL_SPECTRAL_THREADS', '2' if procs > 2 else '3')
    from fokl_gpy_amd import FoKLRoutines, _capi, engine
    models = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for u in range(2):
            x, y, spec = bench.config_workload(4, 10 * k + u, rows)
            model = FoKLRoutines.FoKL(kernel=1, UserWarnings=False, ConsoleOutput=False)
            model._backend_override = engine.HipBackend(_capi.DeviceContext(0))
            model._prepare_fit(x, y, dict(clean=True))
            models.append((model, spec))
        for model, spec in models:                                   # warm-up
            np.random.seed(spec['seed_fit']); model._search(model._backend_override, spec['rows'], spec['inputs'])
        start.wait()
        t0 = time.perf_counter()
        terms = 0
        for i in range(fits):
            model, spec = models[i % 2]
            np.random.seed(spec['seed_fit'])
            model._search(model._backend_override, spec['rows'], spec['inputs'])
            terms += model.fit_stats['terms_logical']
        done.put((k, time.perf_counter() - t0, terms))


if __name__ == '__main__':
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    fits = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
    ctx = mp.get_context('spawn')
    start, done = ctx.Barrier(procs + 1), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(k, procs, fits, rows, start, done)) for k in range(procs)]
    for p in ps:
        p.start()
    start.wait()
    t0 = time.perf_counter()
    res = [done.get() for _ in ps]
    wall = time.perf_counter() - t0
    for p in ps:
        p.join()
    terms = sum(r[2] for r in res)
    print(f"processes {procs}: {procs * fits / wall:.1f} fits/s, {terms / wall:,.0f} candidate terms/s "
          f"(per worker {', '.join(f'{fits / r[1]:.1f}' for r in sorted(res))} fits/s)", flush=True)
