"""
The host side of one fit (round 3: split out of engine.py): CPU placement and thread plan, the buffer spares, the
pipeline of native threads that record noise tapes, finish them, run chains and diagonalise candidate models
(csrc/fokl_hostpool.cpp), the device-chain engine a search may use for its kill tests (csrc/fokl_chain_device.inc), and
the outcome objects through which the search looks at a model evaluation's draws.  The search itself -- the sequence of
decisions of FoKLRoutines.py 1602-1760 -- is engine.ForwardSelection.
"""
import os
import threading
import time

import numpy as np

from . import _capi

# FOKL_POOL_TRACE=<file>: the driver's side of the noise thread's trace (csrc/fokl_hostpool.cpp), same monotonic clock;
# tools/pool_trace.py lines the two up.  A diagnostic: nothing is recorded without the variable.
_TRACE_PATH = os.environ.get('FOKL_POOL_TRACE')
_trace_log = []


def _mark(tag, info=''):
    if _TRACE_PATH:
        _trace_log.append((time.monotonic_ns(), tag, info))


def _flush_marks():
    if _TRACE_PATH and _trace_log:
        with open(_TRACE_PATH, 'a') as f:
            for t, tag, info in _trace_log:
                f.write(f"driver {t} {tag} {info}\n")
        del _trace_log[:]


_CPU_LISTS = {}                                             # sysfs path -> frozenset: the topology is the machine's


def _cpu_list(path):
    cpus = _CPU_LISTS.get(path)
    if cpus is None:
        with open(path) as fh:
            text = fh.read().strip()
        found = set()
        for part in text.split(','):
            lo, _, hi = part.partition('-')
            found.update(range(int(lo), int(hi or lo) + 1))
        cpus = _CPU_LISTS[path] = frozenset(found)          # (every fit used to read two dozen of these files again)
    return set(cpus)


def _place_host_threads():
    """CPU placement for the threads of one fit.  Restricts the calling thread (and the threads it creates next) to the
    logical CPUs that share a last-level cache with the CPU it is running on, minus one physical core that is set
    aside for the random-stream thread.  Returns (previous affinity mask or None, logical CPU for the noise thread or
    -1); nothing is changed when the topology cannot be read or the mask is too small to gain anything."""
    try:
        import ctypes
        cpu = ctypes.CDLL(None).sched_getcpu()
        allowed = os.sched_getaffinity(0)
        domain = _cpu_list(f'/sys/devices/system/cpu/cpu{cpu}/cache/index3/shared_cpu_list') & allowed
        if len(domain) < 2:
            return None, -1
        noise_cpu = -1
        candidate = max(domain - {cpu})
        siblings = _cpu_list(f'/sys/devices/system/cpu/cpu{candidate}/topology/thread_siblings_list')
        rest = domain - siblings
        if cpu in rest and len(rest) >= 4:
            noise_cpu, domain = candidate, rest
        if domain == allowed:
            return None, noise_cpu
        os.sched_setaffinity(0, domain)
        return allowed, noise_cpu
    except (OSError, AttributeError, ValueError):
        return None, -1


def _second_domain(allowed, home):
    """Logical CPUs of another last-level-cache domain inside `allowed` (the one after `home`'s, by lowest CPU number), for
    the spectral threads; None when there is none."""
    try:
        seen, domains = set(), []
        for cpu in sorted(allowed):
            if cpu in seen:
                continue
            dom = _cpu_list(f'/sys/devices/system/cpu/cpu{cpu}/cache/index3/shared_cpu_list') & allowed
            seen |= dom
            domains.append(dom)
        others = [d for d in domains if not (d & home) and len(d) >= 4]
        if not others:
            return None
        after = [d for d in others if min(d) > min(home)]
        return (after or others)[0]
    except (OSError, ValueError):
        return None


def _spectral_cpus(allowed, home, count):
    picked = _other_cores(allowed, home, count)
    return set(picked) if picked is not None else None


_OTHER_CORES = {}                                           # (allowed, home, count) -> picked: the topology does not change


def _other_cores(allowed, home, count):
    key = (frozenset(allowed), frozenset(home), count)
    if key not in _OTHER_CORES:
        if len(_OTHER_CORES) > 64:
            _OTHER_CORES.clear()
        _OTHER_CORES[key] = _other_cores_uncached(allowed, home, count)
    picked = _OTHER_CORES[key]
    return list(picked) if picked is not None else None


def _other_cores_uncached(allowed, home, count):
    """One logical CPU of each of `count` physical cores outside `home`'s last-level-cache domain(s), in the order picked: domain
    by domain in CPU order after `home`: the eigen-decompositions are compute bound and share nothing -- two of them on the
    hardware threads of one core run at half speed each (EPYC 9575F: an L3 domain is FOUR cores; the eight spectral
    threads pinned to the eight logical CPUs of one domain took 0.42 ms per 66-column dsyevr against 0.21 alone).
    None when the topology cannot be read or offers fewer than `count` such cores."""
    try:
        seen, domains = set(), []
        for cpu in sorted(allowed):
            if cpu in seen:
                continue
            dom = _cpu_list(f'/sys/devices/system/cpu/cpu{cpu}/cache/index3/shared_cpu_list') & allowed
            seen |= dom
            domains.append(dom)
        others = [d for d in domains if not (d & home)]
        after = [d for d in others if min(d) > min(home)] + [d for d in others if min(d) < min(home)]
        picked, cores = [], set()
        for dom in after:
            for cpu in sorted(dom):
                core = frozenset(_cpu_list(f'/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list'))
                if core in cores:
                    continue
                cores.add(core)
                picked.append(cpu)
                if len(picked) == count:
                    return picked
        return None
    except (OSError, ValueError):
        return None


_QUOTA = []                                                 # [CPUs' worth of the cgroup quota or None], read once


def _cgroup_quota():
    if not _QUOTA:
        quota = None
        try:
            with open('/sys/fs/cgroup/cpu.max') as fh:                       # cgroup v2: "<quota|max> <period>"
                q, period = fh.read().split()
                if q != 'max':
                    quota = float(q) / float(period)
        except (OSError, ValueError):
            try:
                with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as fh:      # cgroup v1
                    q = float(fh.read())
                with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fh:
                    period = float(fh.read())
                if q > 0:
                    quota = q / period
            except (OSError, ValueError):
                pass
        _QUOTA.append(quota)
    return _QUOTA[0]


def _cpu_budget():
    """CPUs this process may keep busy: its affinity mask, capped by the cgroup CPU quota (a container that sees 256
    CPUs may be allowed 16 CPU-seconds per second) shared among the ranks of the node (LOCAL_WORLD_SIZE)."""
    try:
        budget = float(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        budget = float(os.cpu_count() or 2)
    quota = _cgroup_quota()
    if quota is not None:
        try:
            ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
        except ValueError:
            ranks = 1
        budget = min(budget, quota / ranks)
    return budget


def _walk_helpers_default():
    """Helper threads of the walk for a process that has the CPUs for them (a fit alone on a 16-CPU budget: two -- same-box
    sweep 0 / 1 / 2 / 3 / 4: 27.2 / 22.2 / 21.0 / 21.4 / 21.4 ms per configs[2] fit, +15 ms of CPU per helper); fits side by
    side are bounded by their CPU-seconds and get none."""
    return 2 if _cpu_budget() >= 12 else 0


def _bulk_threads(budget=None, wide_models=False):
    """Threads that produce the random stream ahead of the noise thread's walk (csrc/fokl_stream.cpp): the walk consumes a
    segment of 79 872 doubles in 10-30 us, a bulk thread makes one in 20-40 us (AVX-512).  FOKL_BULK_THREADS overrides."""
    if budget is None:
        budget = _cpu_budget()
    default = (3 if wide_models else 4) if budget >= 12 else (2 if budget >= 5 else 1)
    return max(1, int(os.environ.get('FOKL_BULK_THREADS', str(default))))


def _thread_plan(wide_models=False):
    """(chain, finish, spectral) thread counts of the host pipeline for the CPU budget of this process; the driver and
    the noise thread come on top.  FOKL_CHAIN_THREADS / FOKL_FINISH_THREADS / FOKL_SPECTRAL_THREADS override.
    wide_models: the search will evaluate models of hundreds of columns (configs[3]: 3-way terms over 16 inputs) -- an
    eigen-decomposition then costs tens of milliseconds and the fit waits for little else: the spectral threads get what
    the stream's bulk threads can spare."""
    budget = _cpu_budget()
    # Finishing threads follow the recorder block by block and spin while they wait for it, so each of them costs about
    # the recorder's own busy time in CPU whatever it computes.  With the vector log (default, fokl_vlog.cpp) one thread
    # finishes a tape in a third of the time it takes to record; with libm's scalar log (FOKL_FINISH_LOG=exact) it takes
    # three.  Measured on the GPU boxes (tools/thread_plan_sweep.sh, tools/cpu_budget_sweep.sh; chain + finish + spectral:
    # ms per configs[2] fit / pool CPU-seconds per fit).  16 CPUs: 2+3+3 107.8 / 0.51, 2+1+3 107.8 / 0.40, 1+1+3 106.9 /
    # 0.37, 2+0+3 121 / 0.34, 1+0+3 132 / 0.32;  8 CPUs: 1+1+3 115.6, 2+1+3 119.7, 2+1+4 120.0;  6 CPUs: 1+1+3 113.0,
    # 1+1+2 161.6;  5 CPUs: 1+1+3 118.9;  4 CPUs: 1+1+2 146.6, 1+0+2 149.7, 1+1+1 179.6, 1+1+3 208.6;  3 CPUs: 1+1+2
    # 189.8, 1+0+2 199.5, 1+1+1 201.6;  2 CPUs: 1+0+1 239.5, 1+0+2 245.7.  The eigen-decompositions are the throughput
    # item (0.13 CPU-seconds per fit): three spectral threads as soon as five CPUs are there.
    exact_log = os.environ.get('FOKL_FINISH_LOG', 'fast') == 'exact'
    # Round 4: the random stream left the serial thread (walker + bulk threads, _bulk_threads), kill tests' tapes are
    # expanded on the device and the kill-test loop is native and predicts its path -- what a fit waits for now is the
    # THROUGHPUT of the eigen-decompositions (0.17 CPU-seconds of dsyevr per configs[2] fit).  tools/env_sweep_r04.sh on the
    # GPU boxes (16 CPUs, chain + finish + spectral / ms per fit): 2+1+4 61.7, 2+1+6 58.6, 2+1+8 51-52, 2+1+10 53.6,
    # 2+1+12 53.5 (contention with the stream's four bulk threads); finish threads only serve the sub-stage models' tapes.
    if budget >= 12 and wide_models:
        plan = (2, 3 if exact_log else 1, 11)               # (kill tests' tapes are expanded on the device; 6 CPU-s of dsyevr)
    elif budget >= 12:
        plan = (2, 3 if exact_log else 2, 8)
    elif budget >= 8:
        plan = (1, 2 if exact_log else 1, 5)
    elif budget >= 5:
        plan = (1, 2 if exact_log else 1, 3)
    elif budget >= 4:
        plan = (1, 1, 2)
    elif budget >= 3:
        plan = (1, 0, 2)                                    # round 2: 154.9 ms per fit against 165.9 with a finishing thread
    else:
        plan = (1, 0, 1)
    names = ('FOKL_CHAIN_THREADS', 'FOKL_FINISH_THREADS', 'FOKL_SPECTRAL_THREADS')
    chain, finish, spectral = (int(os.environ.get(name, str(default))) for name, default in zip(names, plan))
    return max(1, chain), max(0, finish), max(1, spectral)       # the threaded search needs a chain and a spectral thread


class Misprediction(RuntimeError):
    """A kill-test decision that was taken from a guess (the intercept's posterior mean ~ its least-squares value) before
    the chain that decides it had run turned out wrong when the chain arrived.  Every guess is verified; the caller
    (FoKL._search) restores the random stream and repeats the search without guessing."""


class ModelSize(int):
    """Size of a tape on order that is meant for a sub-stage's MODEL (finished and chained on host threads: its
    statistics order the kill tests, latency matters) as opposed to a kill-test candidate (device chain)."""
    __slots__ = ()


# G3 on the device (csrc/fokl_chain_device.inc): one engine per process and device, kept from fit to fit (its slots'
# device buffers are allocated once).  FOKL_CHAIN = device | host | auto (default: device where an engine can be made).
_CHAIN_ENGINES = {}
_chain_engine_factory = None            # tests: a stand-in with the interface of _capi.DeviceChainEngine


def chain_engine_for(device):
    """The device-chain engine of HIP device `device`, or None (FOKL_CHAIN=host, no device, engine creation failed)."""
    mode = os.environ.get('FOKL_CHAIN', 'auto')
    if mode not in ('auto', 'device', 'host'):
        raise ValueError("FOKL_CHAIN must be auto, device or host")
    if mode == 'host':
        return None
    if _chain_engine_factory is not None:
        return _chain_engine_factory()
    if device is None:
        if mode == 'device':
            raise RuntimeError("FOKL_CHAIN=device: the backend has no HIP device")
        return None
    engine = _CHAIN_ENGINES.get(device)
    if engine is None:
        try:
            engine = _capi.DeviceChainEngine(device, int(os.environ.get('FOKL_DCHAIN_SLOTS', '128')))
        except _capi.FoklNativeError:
            if mode == 'device':
                raise
            return None
        _CHAIN_ENGINES[device] = engine
    return engine


# G2 on the device (csrc/fokl_spectral_device.inc): likewise one engine per process and device.  FOKL_EIGH = host (default)
# | device.  The device solver (Jacobi in LDS) takes the eigen-decompositions off the host entirely -- 0.31 -> 0.16-0.20
# CPU-seconds per headline fit -- but a decomposition takes 0.6 ms at 66 columns and 4 ms at 144 against LAPACK's 0.2-0.4 and
# 1.2 ms on a host thread, and the search waits for the first decompositions of every sub-stage: 79-100 ms per fit against
# 54 (profiles/eigh_device_r04.txt).  It is there for hosts with fewer CPUs than a fit's thread plan wants.
# Under FOKL_EIGH_SIGNS=lapack always the host's LAPACK -- Jacobi rotations have no LAPACK signs to keep.
_SPECTRAL_ENGINES = {}


def spectral_engine_for(device):
    mode = os.environ.get('FOKL_EIGH', 'host')
    if mode not in ('device', 'hybrid', 'host'):
        raise ValueError("FOKL_EIGH must be host, hybrid or device")
    if mode == 'host' or os.environ.get('FOKL_EIGH_SIGNS', 'canonical') == 'lapack':
        return None
    if device is None:
        raise RuntimeError("FOKL_EIGH=device: the backend has no HIP device")
    engine = _SPECTRAL_ENGINES.get(device)
    if engine is None:
        engine = _SPECTRAL_ENGINES[device] = _capi.DeviceSpectralEngine(device)
    return engine


def close_chain_engines():
    for engine in list(_CHAIN_ENGINES.values()) + list(_SPECTRAL_ENGINES.values()):
        engine.close()
    _CHAIN_ENGINES.clear()
    _SPECTRAL_ENGINES.clear()


import atexit                                               # noqa: E402  (engines own streams and page-locked memory)
atexit.register(close_chain_engines)


# Spare tape / draw buffers of the calling thread, by size class: they survive the fit that allocated them, so that the
# next fit on this thread does not page-fault a few hundred MB in again (FOKL_HOST_POOL_MB caps what is kept, default
# 2048; the count is per process and approximate when several threads fit at once -- it only bounds memory).
class _SpareAccount:
    doubles = 0
    limit = int(float(os.environ.get('FOKL_HOST_POOL_MB', '2048')) * 131072)


_SPARES = _SpareAccount()
_SPARE_LISTS = threading.local()


def _thread_spares():
    spares = getattr(_SPARE_LISTS, 'by_class', None)
    if spares is None:
        spares = _SPARE_LISTS.by_class = {}
    return spares


def drop_spare_buffers():
    """Give the calling thread's spare tape / draw buffers back to the allocator."""
    spares = _thread_spares()
    _SPARES.doubles = max(0, _SPARES.doubles - sum(raw.shape[0] for stack in spares.values() for raw in stack))
    spares.clear()


class HostPipeline:
    """
    The host threads of one fit (include/fokl_hip.h: fokl_pool_*), all native, none holding the GIL:

    * the noise thread owns the numpy-legacy random stream for the duration of the fit and records, strictly in
      request order, the noise tape of every model evaluation (per Gibbs iteration p1 standard normals and two
      standard gammas -- everything random in FR:1519-1548, none of it data dependent).  The stream is serial by
      definition, so this thread is the critical path of a fit at large N;
    * chain threads follow the tapes and form every candidate's draws in the eigenbasis (FR:1521-1548);
    * spectral threads diagonalise candidate XtX sub-blocks with scipy's own LAPACK (FR:1499-1504).  That work uses
      no random numbers, so the driver submits it ahead of time for both possible next kill-test models.

    The driver thread keeps what needs Python or the device: the sequential decisions, the K1 / K2 / K3 launches.
    """

    def __init__(self, stream, draws, comm=None, chain_engine=None, wide_models=False, device=None):
        self.stream, self.draws = stream, int(draws)
        self.dchain = chain_engine          # G3 of kill-test candidates on the device (None: host chain threads)
        self._pinned = chain_engine is not None and getattr(chain_engine, 'wants_pinned_tapes', True)
        # candidate sharding (see ShardedSpectralJob): G2 jobs are dealt over the ranks of `comm` in submission order
        # (FOKL_CANDIDATE_SHARD_FORCE=1: also in a world of one, so that a 1-GPU box takes the exchange path)
        forced = os.environ.get('FOKL_CANDIDATE_SHARD_FORCE', '0') == '1'
        self.comm = comm if comm is not None and (comm.world > 1 or forced) else None
        self._seq = 0
        self._windows = {}                  # window (seq // world) -> its jobs whose results have not travelled yet
        self.remote_results = self.exchanges = self.spectral_submitted = 0
        # Tapes (about 1 MB each) are produced by one thread and consumed by others: keep all of them on cores that
        # share an L3 for the duration of the fit, and give the noise thread -- the serial resource -- a physical core
        # to itself (measured on a 2 x 64-core EPYC host).  The native threads inherit the affinity set here; the
        # driver's is restored in close().  FOKL_PIN_L3=0 disables.
        self._saved_affinity, noise_cpu = None, -1
        if os.environ.get('FOKL_PIN_L3', '1') != '0':
            self._saved_affinity, noise_cpu = _place_host_threads()
        chain, finish, spectral = _thread_plan(wide_models)
        # With a device chain engine the stream's bulk threads leave every segment's pre-state in the engine's page-locked
        # ring: the device regenerates the segments a tape covers and expands the tape from its 32-byte rows
        # (fokl_dchain_submit_rows) -- the native search then never materialises a kill test's tape on the host.
        # FOKL_DCHAIN_ROWS=0: tapes are materialised by the finish threads and read over the bus, as in round 3.
        self.device_rows = False
        prestates = None
        if isinstance(chain_engine, _capi.DeviceChainEngine) and os.environ.get('FOKL_DCHAIN_ROWS', '1') != '0':
            try:
                prestates = chain_engine.prestate_ring()
            except _capi.FoklNativeError:
                prestates = None
        try:
            # The product that ends a derived decomposition (0.4 GFLOP at 585 columns: 3.6 of a 5.1 ms step on a host core)
            # on the device's matrix cores from FOKL_EIGH_DGEMM_FROM columns on: 2.6 ms a step at 585 columns, break-even near
            # 200 (tools/eigen_update_bench.py).  Opt-in (default 0: scipy's dgemm for every size): no BASELINE config derives
            # models that wide -- configs[3] defers the G2 of its 585-column models, which leaves them without a parent to
            # derive from (tools/r05_cfg3b.sh: 348-358 ms per fit either way).
            dgemm_from = int(os.environ.get('FOKL_EIGH_DGEMM_FROM', '0'))
            bulk = _bulk_threads(wide_models=wide_models)
            self.pool = _capi.HostPool(stream, chain, finish, spectral, noise_cpu, bulk_threads=bulk,
                                       prestates=prestates,
                                       device_dgemm=(device, dgemm_from) if device is not None and dgemm_from > 0 else None)
            if prestates is not None:
                chain_engine.bind(self.pool.stream_handle())
                self.device_rows = True
            # Helper threads for the serial walk of the random stream (csrc/fokl_stream.cpp walk_tape_crew): with the sub-stage
            # loop native the walk bounds a fit.  On physical cores of their OWN, outside this process's last-level-cache domain
            # -- next to the bulk threads, two to a core, their passes ran at half speed and bought nothing.  FOKL_WALK_HELPERS
            # overrides the count (0: the walking thread does everything), FOKL_WALK_CPUS=same leaves them unpinned.
            # (wide models -- configs[3] -- are bound by the PRODUCTION of the stream, a segment per microsecond of walking:
            # helpers there only spin, and cost the fit a fifth: 392 against 318 ms, same box)
            helpers = int(os.environ.get('FOKL_WALK_HELPERS', str(0 if wide_models else _walk_helpers_default())))
            if helpers > 0:
                cpus = None
                if self._saved_affinity is not None and os.environ.get('FOKL_WALK_CPUS', 'other') != 'same':
                    picked = _other_cores(self._saved_affinity, os.sched_getaffinity(0), spectral + helpers)
                    if picked is not None:
                        cpus = picked[-helpers:]              # (behind the cores the spectral threads get)
                try:
                    self.pool.set_walk_helpers(helpers, cpus)
                except _capi.FoklNativeError:
                    pass
            # The stream's bulk threads on physical cores of their own as well (round 6): since a segment's words go out
            # past the cache nothing ties them to this domain, where they shared three cores with the driver's, chain and finish
            # threads.  Only where the process has the CPUs to spread out (a fit alone); FOKL_BULK_CPUS=same leaves them here.
            if (self._saved_affinity is not None and _cpu_budget() >= 12 and os.environ.get('FOKL_BULK_CPUS', 'other') != 'same'):
                picked = _other_cores(self._saved_affinity, os.sched_getaffinity(0), spectral + max(0, helpers) + bulk)
                if picked is not None:
                    try:
                        self.pool.place_bulk_threads(picked[-bulk:])
                    except _capi.FoklNativeError:
                        pass
            # the eigen-decompositions share nothing with the stream's threads: on a CPU with several last-level-cache
            # domains they get the next one to themselves (FOKL_SPECTRAL_DOMAIN=0: they stay where the others are)
            if self._saved_affinity is not None and os.environ.get('FOKL_SPECTRAL_DOMAIN', '1') != '0':
                other = None
                if os.environ.get('FOKL_SPECTRAL_DOMAIN', '1') != 'domain' and spectral > 0:
                    other = _spectral_cpus(self._saved_affinity, os.sched_getaffinity(0), spectral)
                if other is None:
                    other = _second_domain(self._saved_affinity, os.sched_getaffinity(0))
                if other:
                    try:
                        self.pool.spectral_affinity(other)
                    except _capi.FoklNativeError:
                        pass
        except BaseException:
            self._restore_affinity()        # the caller may carry on in line: not pinned to one L3 domain
            raise
        # Every job names buffers the native threads read and write: they are kept here until the job has run, whether
        # or not the driver still cares about the result (a rejected candidate's tape is recorded all the same).
        self._live = []
        self._reap_at = 24
        # Tapes and draws are a few MB per model evaluation; fresh allocations would be page-faulted in by the noise
        # and chain threads (measured: a third of the tape time).  Buffers go round in 512 KB size classes instead, and
        # stay with the thread from one fit to the next (_thread_spares).
        self._spare = _thread_spares()

    CLASS_DOUBLES = 65536

    def _take(self, doubles):
        cls = -(-doubles // self.CLASS_DOUBLES)
        spare = self._spare.get(cls)
        if spare:
            _SPARES.doubles -= cls * self.CLASS_DOUBLES
            return spare.pop()
        if self._pinned:
            # page-locked: the device chains' H2D copies of a tape are then DMA transfers, not staged copies
            try:
                return _capi.pinned_empty(cls * self.CLASS_DOUBLES)
            except _capi.FoklNativeError:
                self._pinned = False
        return np.empty(cls * self.CLASS_DOUBLES, dtype=np.float64)

    def give(self, raw):
        if _SPARES.doubles + raw.shape[0] <= _SPARES.limit:
            _SPARES.doubles += raw.shape[0]
            self._spare.setdefault(raw.shape[0] // self.CLASS_DOUBLES, []).append(raw)

    def _reap(self):
        live = []
        for job in self._live:              # (a device chain's done() is a load from page-locked memory: no frontier needed,
            if job.done():                  # and with several streams chains do not complete in submission order)
                if job.recycle:
                    for entry in job.recycle:
                        self._retire_buffer(entry)
                    job.recycle = job.keep = None
            else:
                live.append(job)
        self._live = live

    def _retire_buffer(self, entry):
        """A buffer whose owner has run: back to the spares -- or, if a chain that was started ahead and then given up may
        still be reading it (entry = (buffer, that chain's job)), on to that job."""
        if isinstance(entry, tuple):
            raw, reader = entry
            if reader.done():
                self.give(raw)
            else:
                reader.recycle.append(raw)
        else:
            self.give(entry)

    def _hand_tape(self, noise_job, owner):
        """The tape's buffer passes from being `held` to the job that is its last reader (the noise job itself when no
        chain will read the tape)."""
        raw, noise_job.held = noise_job.held, None
        if raw is None:
            return
        reader = noise_job.co_reader
        entry = raw if reader is None or reader is owner else (raw, reader)
        if owner.recycle is None:
            if owner._h is None:            # has run and was reaped already
                self._retire_buffer(entry)
                return
            owner.recycle = []
        owner.recycle.append(entry)

    def _track(self, job):
        if len(self._live) >= self._reap_at:
            self._reap()
            # with chains in flight on the device (a millisecond or two each) the list does not empty: look again only
            # after another dozen jobs, not on every submission
            self._reap_at = max(24, len(self._live) + 12)
        self._live.append(job)
        return job

    def request(self, p1, astar, atau_star, tentative=False, finish=True):
        """Queue the tape of one model evaluation; the buffers exist at once and fill up in the background.
        tentative: recorded ahead of the decision that the evaluation happens -- the caller owes the job a
        ``resolve(True / False)`` (False rewinds the stream to where the tape began).
        finish: host threads complete the normals while the tape is recorded (tapes whose chain runs on the host); a
        tape meant for a device chain stays raw -- the device finishes it.  Either kind of chain takes either kind of
        tape, so a tape ordered for one role may serve the other."""
        raw = self._take(_capi.NoiseTape.doubles_needed(p1, self.draws))
        job = self.pool.submit_noise(_capi.NoiseTape(p1, self.draws, raw), astar, atau_star, tentative, finish=finish)
        # The tape's last reader is the chain job, which does not exist yet: until chain() or discard() the buffer is
        # only `held`, so that a _reap() between request and chain (the recorder may well be done by then) cannot hand
        # it out again -- chain() would then get the tape's own memory as its output buffer.
        job.held = raw
        return self._track(job)

    def abandon(self, noise_job):
        """A tape that is recorded (the stream must advance exactly as if the model had been sampled) but that no chain
        will read: its buffer goes back to the pool once the recorder is done with it."""
        self._hand_tape(noise_job, noise_job)

    def discard(self, noise_job):
        """A tentative tape that will not be used: rewind the stream to where it began; its buffer goes back to the
        pool once the recorder has let go of it."""
        noise_job.resolve(False)
        self._hand_tape(noise_job, noise_job)

    def spectral(self, gram, idx):
        """Queue G2 for the model made of columns idx of gram; may be called ahead of need (no random numbers).
        With a communicator the job runs on ONE rank -- every rank drives the same search and therefore submits the same
        jobs in the same order, job number s belongs to rank s % world -- and wait() brings the result to all."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.spectral_submitted += 1
        if self.comm is None:
            return self._track(self.pool.submit_spectral(gram, idx, gram.shape[0] - 1))
        seq, self._seq = self._seq, self._seq + 1
        local = None
        if seq % self.comm.world == self.comm.rank:
            local = self._track(self.pool.submit_spectral(gram, idx, gram.shape[0] - 1))
        job = ShardedSpectralJob(self, seq, idx.shape[0], local)
        self._windows.setdefault(seq // self.comm.world, []).append(job)
        return job

    def _exchange(self, window):
        """One all-gather for the jobs of `window` (world consecutive job numbers: one job per rank) that have been
        submitted and not exchanged yet: every rank contributes the result it owns -- lamb, Q'Xty, betahat, Q' and the
        residual moments behind the candidate's BIC, SpectralResult's buffer as it is -- padded to the largest model
        of the exchange.  Every rank calls this for the same window at the same point of the (replicated) search."""
        jobs = self._windows.pop(window)
        length = max(_capi.SpectralResult.doubles(job.p1) for job in jobs)
        send = np.zeros(length)
        failure = None
        for job in jobs:
            if job.local is not None:
                try:
                    res = job.local.wait()
                    send[:res._buf.shape[0]] = res._buf
                    job.result = res
                except _capi.FoklNativeError as exc:
                    # a rank that raised here alone would leave the others waiting in the collective for ever: ship
                    # NaNs, let every rank see them and fail together
                    failure = exc
                    send[:] = np.nan
        gathered = self.comm.allgather(send)
        self.exchanges += 1
        for job in jobs:
            owner = job.seq % self.comm.world
            if np.isnan(gathered[owner, 0]):
                raise RuntimeError(f"candidate-sharded search: rank {owner} could not diagonalise model {job.seq} "
                                   f"({job.p1} columns)") from failure
            if job.result is None:
                need = _capi.SpectralResult.doubles(job.p1)
                job.result = _capi.SpectralResult(job.p1, np.array(gathered[owner, :need]))
                self.remote_results += 1

    def chain(self, spec, b, btau, dtd, sigsqd0, tausqd0, noise_job):
        """Queue the draws of the model whose tape is noise_job's.  Returns (job, raw buffer holding w): the caller
        hands the buffer back (give, or job.recycle while the job runs) when nobody reads w any more."""
        tape = noise_job.result
        w_raw = self._take(tape.draws * tape.p1)
        job = self.pool.submit_chain(spec.lamb, spec.qty, b, btau, dtd, sigsqd0, tausqd0, tape, w_raw)
        job.recycle = []
        self._hand_tape(noise_job, job)                             # the chain job is the last reader of the tape
        return self._track(job), w_raw

    def chain_device(self, spec, b, btau, dtd, sigsqd0, tausqd0, noise_job, stat_first):
        """chain() on the device engine: -> the job (its draws stay in device memory until fetched), or None when the
        engine has no free slot (the caller then takes the host chain).  The mean of w over the rows from stat_first
        on comes back with the job -- what the kill tests look at."""
        try:
            job = self.dchain.submit(spec.lamb, spec.qty, b, btau, dtd, sigsqd0, tausqd0, noise_job.result, stat_first)
        except _capi.FoklNativeError as exc:
            if exc.code == -3:                                          # FOKL_ERR_STATE: every slot is alive
                return None
            raise
        job.recycle = []
        self._hand_tape(noise_job, job)                                 # the device job is the last reader of the tape
        return self._track(job)

    def chain_ahead(self, spec, b, btau, dtd, sigsqd0, tausqd0, noise_job):
        """chain() for a tape that is still on order (tentative, no verdict yet): the draws are under way when the
        decision comes that this model is evaluated -- adopt() -- or thrown away -- disown().  The tape stays `held` by
        its noise job until then."""
        tape = noise_job.result
        w_raw = self._take(tape.draws * tape.p1)
        job = self.pool.submit_chain(spec.lamb, spec.qty, b, btau, dtd, sigsqd0, tausqd0, tape, w_raw)
        job.recycle = []
        job.ignore_failure = True                                   # the tape may be sent back under it
        return self._track(job), w_raw

    def adopt(self, chain_job, noise_job):
        """The chain started ahead is the model's chain: from now on a failure counts, and it is the tape's last reader."""
        chain_job.ignore_failure = False
        self._hand_tape(noise_job, chain_job)

    def disown(self, chain_job, w_raw, noise_job):
        """A chain started ahead that nobody will look at: its buffer goes back when it has run, and the tape it reads is
        not reused before that."""
        if chain_job.done():
            self.give(w_raw)
        else:
            chain_job.recycle.append(w_raw)
            noise_job.co_reader = chain_job

    def close(self):
        """Run everything still queued (every requested tape advances the stream, used or not), stop the threads."""
        for job in self._live:
            if job.unresolved:              # only after an exception in the driver: never leave the noise thread waiting
                job.resolve(False)
        for job in self._live:
            if not (isinstance(job, _capi.DeviceChainJob) and job.done()):
                job.wait()
        _mark('jobs_drained')
        self._reap()                        # buffers of the jobs that have run now go back to the thread's spares
        self._live = []
        busy = self.pool.busy_seconds()
        if self.device_rows:
            self.dchain.bind(None)          # the stream goes with the pool
            self.device_rows = False
        self.pool.close()
        self._restore_affinity()
        return busy

    def _restore_affinity(self):
        if self._saved_affinity is not None:
            try:
                os.sched_setaffinity(0, self._saved_affinity)
            except OSError:
                pass
            self._saved_affinity = None


class ShardedSpectralJob:
    """G2 of one candidate model in a candidate-sharded search: computed by rank seq % world (`local` is that rank's
    pool job, None elsewhere); wait() returns the SpectralResult on every rank."""
    __slots__ = ('owner', 'seq', 'p1', 'local', 'result')

    def __init__(self, owner, seq, p1, local):
        self.owner, self.seq, self.p1, self.local, self.result = owner, seq, p1, local, None

    def wait(self):
        if self.result is None:
            self.owner._exchange(self.seq // self.owner.comm.world)
        return self.result


class GibbsOutcome:
    """One model evaluation.  The BIC is known at once; the draws arrive from a chain thread (chain arithmetic in the
    eigenbasis), betas = w Q' (FR:1528) is formed only for the columns somebody looks at."""
    __slots__ = ('lamb', 'qty', 'Qt', 'betahat', 'ev', 'idx', 'intercept_scale', 'siglik', '_jobs', '_owner', '_w',
                 '_betas', '_w_raw', '_chain_job', 'on_device', 'checks', '_release_wanted', '_dtd')

    def __init__(self, owner, spec, ev, idx, noise_job, chain_job, w_raw):
        self.lamb, self.qty, self.Qt, self.betahat = spec.lamb, spec.qty, spec.Qt, spec.betahat
        self.ev, self.idx = ev, idx
        self._jobs, self._owner = (noise_job, chain_job), owner
        self._w = self._betas = self.intercept_scale = None
        self.siglik = 0.0
        self._w_raw, self._chain_job = w_raw, chain_job
        self.on_device = isinstance(chain_job, _capi.DeviceChainJob)   # the draws live in device memory (dchain slot)
        self.checks = []                    # (|mean beta| of a proposal, decision taken from the guessed intercept scale)
        self._release_wanted = False
        self._dtd = None                    # y'y of the fit (device-chained outcomes: intercept_scale_on_host)

    def release(self):
        """Hand the buffer of w back to the pipeline's pool (directly, or through the chain job that is still writing
        it) -- or the device slot to the chain engine.  Called by the search when no decision can look at this model's
        draws any more; idempotent.  A device chain whose statistics still have to confirm guessed decisions keeps its
        slot until they have (ForwardSelection._verify)."""
        if self.on_device:
            if self.checks:
                self._release_wanted = True
                return
            self._w = self._jobs = None
            self._owner._release_device_job(self._chain_job)
            return
        raw, self._w_raw = self._w_raw, None
        host = self._owner.host
        if raw is None or host is None:
            return
        self._w = self._jobs = None
        if self._chain_job.done():
            host.give(raw)
        else:
            self._chain_job.recycle.append(raw)

    def intercept_scale_on_host(self, first_row):
        """|mean intercept draw| of a device-chained model NOW, from a chain run in line on the calling thread (a quarter
        of a millisecond; the device's answer is a few milliseconds away).  For the rare proposal that sits too close
        to the threshold to be decided from the least-squares guess.  The tape is still there: its buffer belongs to
        the device job until that has run."""
        o = self._owner
        noise_job, chain_job = self._jobs
        noise_job.wait()
        tape = noise_job.result
        chain = _capi.gibbs_chain_from_finished_tape if tape.finishing_requested else _capi.gibbs_chain_from_tape
        w, negative = chain(self.lamb, self.qty, o.b, o.btau, self._dtd, o.sigsqd0, o.tausqd0, tape, follow=True)
        if negative:
            raise RuntimeError("bstar < 0 inside the Gibbs chain (only possible with b <= 0): the noise tape "
                               "cannot reproduce the reference's skipped draw (FR:1538-1539)")
        return abs(float(np.mean(w[first_row:] @ self.Qt[:, 0])))

    def chain_ready(self):
        """The chain has run: looking at its results costs no wait."""
        return self._w is not None or self.intercept_scale is not None or self._chain_job.done()

    def mean_intercept_draw(self, first_row):
        """np.mean(betas[first_row:, 0]) -- from the mean of w the device chain brings along (mean w . Q[0, :]), or from
        the draws of a host chain."""
        if self.on_device:
            o = self._owner
            t0 = time.perf_counter()
            mean_w, negative = self._chain_job.wait()
            o.stats['t_chain'] += time.perf_counter() - t0
            o.stats['dchain_kernel_s'] += getattr(self._chain_job, 'kernel_seconds', 0.0)
            o.stats['dchain_timed'] += 1
            if negative[0]:
                raise RuntimeError("bstar < 0 inside the Gibbs chain (only possible with b <= 0): the noise tape "
                                   "cannot reproduce the reference's skipped draw (FR:1538-1539)")
            return float(mean_w @ self.Qt[:, 0])
        return float(np.mean(self.beta_columns(np.array([0]), first_row)[:, 0]))

    @property
    def Q(self):
        return self.Qt.T

    @property
    def w(self):
        if self._w is None:
            o = self._owner
            noise_job, chain_job = self._jobs
            t0 = time.perf_counter()
            if self.on_device:
                _, negative = chain_job.wait()
                self._w = chain_job.fetch_w()           # D2H of the draws: only models that are returned get here
                o.stats['chains_fetched'] += 1
            else:
                self._w, negative = chain_job.wait()
            noise_job.wait()
            o.stats['t_chain'] += time.perf_counter() - t0
            o.stats['chains_materialised'] += 1
            if negative[0]:
                raise RuntimeError("bstar < 0 inside the Gibbs chain (only possible with b <= 0): the noise tape "
                                   "cannot reproduce the reference's skipped draw (FR:1538-1539)")
            self._jobs = None
        return self._w

    @property
    def betas(self):
        if self._betas is None:
            self._betas = self.w @ self.Qt
        return self._betas

    def beta_columns(self, cols, first_row=0):
        """Draws first_row.. of a few coefficients only: w Q[cols, :]' -- the kill-test statistics look at the second
        half of the draws of the new terms and of the intercept, never at the whole draws x (P+1) matrix."""
        if self._betas is not None:
            return self._betas[first_row:, cols]
        return self.w[first_row:] @ self.Qt[:, cols]


class NativeSpectrum:
    """A G2 job of a NativeSearch (fokl_search_spectral): the interface of a pool job -- done(), wait() -> SpectralResult
    over the search's own buffer -- plus the handle `h`.  Owns one reference; keeps the Gram it was computed from alive."""
    __slots__ = ('_ns', 'h', '_gram', '_view')

    def __init__(self, ns, handle, gram):
        self._ns, self.h, self._gram, self._view = ns, handle, gram, None

    def done(self):
        return self._view is not None or self._ns.spectrum_done(self.h)

    def wait(self):
        if self._view is None:
            self._view = self._ns.spectrum_view(self.h)
        return self._view

    def __del__(self):
        try:
            if self.h:
                self._ns.spectrum_release(self.h)
        except Exception:                                              # interpreter shutdown
            pass
        self.h = None


class NativeOutcome:
    """One model evaluation of a NativeSearch (csrc/fokl_search.cpp): GibbsOutcome's interface over a native handle.  The
    spectrum, the draws and their buffers belong to the search; the arrays here are views that live as long as this
    object is not released / dropped."""
    __slots__ = ('_ns', 'h', '_spec', '_idx', '_p1', '_spectrum_at', '_idx_at', 'ev', 'siglik', '_w', '_betas', '_scale',
                 '_owner')

    def __init__(self, owner, ns, handle):
        self._owner, self._ns, self.h = owner, ns, handle
        view = ns.outcome_info(handle)
        # (the views of the spectrum are formed when somebody reads them: most outcomes of a search are only ever asked
        # for their BIC and their handle -- 28 per configs[2] fit, 23 us each)
        self._p1, self._spectrum_at, self._idx_at = view.p1, view.spectrum, view.idx
        self._spec = self._idx = None
        self.ev, self.siglik = view.ev, view.siglik
        self._w = self._betas = self._scale = None

    def _spectrum(self):
        if self._spec is None:
            import ctypes
            buf = np.ctypeslib.as_array(
                (ctypes.c_double * _capi.SpectralResult.doubles(self._p1)).from_address(self._spectrum_at))
            self._spec = _capi.SpectralResult(self._p1, buf)
        return self._spec

    lamb = property(lambda self: self._spectrum().lamb)
    qty = property(lambda self: self._spectrum().qty)
    Qt = property(lambda self: self._spectrum().Qt)
    betahat = property(lambda self: self._spectrum().betahat)

    @property
    def idx(self):
        if self._idx is None:
            import ctypes
            self._idx = np.array(np.ctypeslib.as_array((ctypes.c_int32 * self._p1).from_address(self._idx_at)))
        return self._idx

    @property
    def on_device(self):
        # (a kill test decided before its G2 had run has no chain yet: asked when somebody wants to know)
        return bool(self._ns.outcome_info(self.h).on_device)

    @property
    def intercept_scale(self):
        if self._scale is None:
            v = self._ns.outcome_info(self.h).intercept_scale
            if v == v:
                self._scale = v
        return self._scale

    @intercept_scale.setter
    def intercept_scale(self, value):
        self._scale = value

    def chain_ready(self):
        return self._w is not None or self._ns.outcome_chain_ready(self.h)

    def release(self):
        self._w = None
        if self.h:
            self._ns.outcome_release(self.h)

    def __del__(self):
        try:
            if self.h:
                self._ns.outcome_drop(self.h)
        except Exception:                                              # interpreter shutdown
            pass
        self.h = None

    @property
    def Q(self):
        return self.Qt.T

    @property
    def w(self):
        if self._w is None:
            self._w = self._ns.outcome_draws(self.h, self.lamb.shape[0])     # (the wait is counted by the native side)
        return self._w

    @property
    def betas(self):
        if self._betas is None:
            self._betas = self.w @ self.Qt
        return self._betas

    def beta_columns(self, cols, first_row=0):
        if self._betas is not None:
            return self._betas[first_row:, cols]
        return self.w[first_row:] @ self.Qt[:, cols]


class EagerOutcome:
    """Model evaluation whose chain ran in line (b <= 0: bstar < 0 may skip draws, so no tape can be recorded ahead)."""
    __slots__ = ('w', 'Q', 'betahat', 'ev', 'idx', 'intercept_scale', '_betas')

    def __init__(self, w, Q, betahat, ev, idx):
        self.w, self.Q, self.betahat, self.ev, self.idx = w, Q, betahat, ev, idx
        self._betas = self.intercept_scale = None

    @property
    def betas(self):
        if self._betas is None:
            self._betas = self.w @ self.Q.T
        return self._betas

    def beta_columns(self, cols, first_row=0):
        if self._betas is not None:
            return self._betas[first_row:, cols]
        return self.w[first_row:] @ self.Q[cols, :].T
