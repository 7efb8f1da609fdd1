"""Latency and throughput of the device chain (fokl_dchain_*) against the host chain on the same tapes.

    python tools/chain_device_probe.py [draws]

Per model size: one chain start to finish (submit -> statistics back), the host chain on one thread, and 16 chains
submitted together (what the dispatcher overlaps on its streams)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
from fokl_gpy_amd import _capi  # noqa: E402


def main():
    draws = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    eng = _capi.DeviceChainEngine(0, slots=32)
    rng = np.random.default_rng(0)
    print(f"{'p1':>5} {'host chain ms':>14} {'device 1 chain ms':>18} {'device 16 chains ms':>20} {'per chain':>10} "
          f"{'max |dw|/scale':>15}")
    for p1 in (9, 37, 60, 100, 150, 300, 586):
        lamb = np.sort(rng.random(p1) * 1e5 + 1e-2)
        qty = rng.standard_normal(p1) * np.sqrt(lamb) * 3
        np.random.seed(1)
        stream = _capi.LegacyStream()
        pinned = os.environ.get('PROBE_PINNED', '1') == '1'
        need = _capi.NoiseTape.doubles_needed(p1, draws) + 8
        tapes = [_capi.record_noise_tape(_capi.NoiseTape(p1, draws, _capi.pinned_empty(need) if pinned else None),
                                         5e5 + p1 / 2, 4 + (p1 - 1) / 2, stream) for _ in range(16)]
        args = (lamb, qty, 900.0, 2.0, 5e5, 0.3, 0.9)
        t0 = time.perf_counter()
        want, _ = _capi.gibbs_chain_from_tape(*args, tapes[0])
        t_host = time.perf_counter() - t0
        for _ in range(2):                                               # warm: slot buffers allocated
            job = eng.submit(*args, tapes[0], stat_first=draws // 2, follow=False)
            job.wait()
            job.release()
        t0 = time.perf_counter()
        job = eng.submit(*args, tapes[0], stat_first=draws // 2, follow=False)
        job.wait()
        t_one = time.perf_counter() - t0
        w = job.fetch_w()
        job.release()
        err = np.max(np.abs(w - want) / np.max(np.abs(want), axis=0))
        for _ in range(2):
            jobs = [eng.submit(*args, t, stat_first=draws // 2, follow=False) for t in tapes]
            t0 = time.perf_counter()
            for j in jobs:
                j.wait()
            for j in jobs:
                j.release()
        jobs = [eng.submit(*args, t, stat_first=draws // 2, follow=False) for t in tapes]
        t0 = time.perf_counter()
        for j in jobs:
            j.wait()
        t_many = time.perf_counter() - t0
        for j in jobs:
            j.release()
        print(f"{p1:5d} {1e3 * t_host:14.3f} {1e3 * t_one:18.3f} {1e3 * t_many:20.3f} {1e3 * t_many / 16:10.3f} {err:15.2e}")
    print(eng.stats())
    eng.close()


if __name__ == '__main__':
    main()
