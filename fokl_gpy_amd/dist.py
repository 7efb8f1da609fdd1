"""
Multi-GPU plumbing: one process per GPU, RCCL over xGMI through the C ABI (``fokl_comm_*``).

The forward-selection path has exactly two exchange steps, both tiny:
  * all-gather of per-candidate BIC values when kill-test proposals (or whole independent fits) are sharded
    over ranks;
  * all-reduce(sum) of Gram blocks / residual moments when *rows* are sharded over ranks.

``RcclComm`` drives them on the GPU; ``TcpComm`` is the control plane (barriers, a few timing figures) and the data path
of launcher rehearsals.  No PyTorch anywhere in this package: the stand-in with the same interface over
``torch.distributed``'s gloo backend, with which the N > 1 host logic is tested on CPU, is test scaffolding
(tests/dist_worker.py).
"""
import os

import numpy as np


def env_rank_world():
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    return rank, world, local


def flush_c_streams():
    """Flush C stdio buffers (librccl writes its banner there) so that they cannot land after later Python output."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


class SingleComm:
    """World of one: every collective is the identity."""
    rank, world = 0, 1

    def allgather(self, values):
        return np.asarray(values, dtype=np.float64)[None, :].copy()

    def allreduce_sum(self, values):
        return np.array(values, dtype=np.float64, copy=True)

    def barrier(self):
        pass

    def close(self):
        pass


def _rendezvous_path(tag='fokl'):
    """Where rank 0 of THIS launch advertises itself: named after what every rank of one launch shares whatever the
    launcher (MASTER_PORT -- bound by the launcher's own store, so unique among live launches of the node -- the
    elastic run id and restart count), never after process ids."""
    import tempfile
    port = os.environ.get('MASTER_PORT', '0')
    launch = os.environ.get('TORCHELASTIC_RUN_ID', 'none')
    restart = os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')
    safe = ''.join(c if c.isalnum() or c in '-_' else '_' for c in f'{port}_{launch}_{restart}')
    return os.path.join(tempfile.gettempdir(), f'{tag}_rccl_rdzv_{safe}.txt')


def _exchange_unique_id(rank, world, make_id, tag='fokl', timeout_s=300.0):
    """
    Rank 0 creates the 128-byte RCCL id and serves it to the other ranks over a loop-back / MASTER_ADDR TCP socket on
    an OS-assigned port; the port number travels through a small file (_rendezvous_path) that rank 0 creates with
    O_EXCL and mode 0600 after unlinking whatever an earlier, crashed launch may have left there.  A reader accepts
    the file only if it belongs to its own uid, and gets the id only from a LIVE rank 0 (a stale file points at a
    closed port: connection refused, read the file again), so neither a restart of the worker group nor a leftover
    or planted file can hand out a dead id.  Returns (id, path).

    Deliberately NOT torch.distributed's TCPStore: importing torch loads its private copies of the HIP runtime and
    of librccl into the process, and RCCL then initialises against the wrong runtime ("unhandled cuda error").
    """
    import socket
    import struct
    import time
    path = _rendezvous_path(tag)
    host = os.environ.get('MASTER_ADDR', '127.0.0.1') if world > 1 else '127.0.0.1'
    deadline = time.monotonic() + timeout_s
    if rank == 0:
        uid = bytes(make_id())
        if world == 1:
            return uid, None
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        try:
            try:
                srv.bind((host, 0))
            except OSError:
                srv.bind(('127.0.0.1', 0))                   # a name that does not resolve: loop-back, never 0.0.0.0
            srv.listen(world)
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
            with os.fdopen(fd, 'w') as fh:
                fh.write(f'{srv.getsockname()[1]}\n')
            served = set()
            while len(served) < world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError(f"rank 0: only {len(served)} of {world - 1} ranks asked for the RCCL id "
                                       f"within {timeout_s:.0f} s")
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = conn.recv(8, socket.MSG_WAITALL)
                        peer = struct.unpack('<i', hello[4:])[0] if len(hello) == 8 and hello[:4] == b'FOKL' else 0
                        if 1 <= peer < world:                 # strays, retries with another number: not counted
                            conn.sendall(uid)
                            served.add(peer)                  # only once the id has gone out
                    except OSError:
                        pass
        finally:
            srv.close()
            try:
                os.unlink(path)
            except OSError:
                pass
        return uid, path
    while True:
        try:
            st = os.stat(path)
            if st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0:
                with open(path) as fh:
                    port = int(fh.read().strip() or 0)
                with socket.create_connection((host, port), timeout=5.0) as conn:
                    conn.sendall(b'FOKL' + struct.pack('<i', rank))
                    uid = b''
                    while len(uid) < 128:
                        part = conn.recv(128 - len(uid))
                        if not part:
                            break
                        uid += part
                if len(uid) == 128:
                    return uid, path
        except (FileNotFoundError, ValueError, ConnectionError, socket.timeout, OSError):
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no RCCL id from rank 0 via {path} after {timeout_s:.0f} s")
        time.sleep(0.02)


class RcclComm:
    """RCCL communicator attached to a ``_capi.DeviceContext`` (backend "nccl" == RCCL on ROCm).

    ``RcclComm(ctx, rank, world)`` initialises and attaches in one go.  ``RcclComm.initialise(...)`` only runs the
    (collective, possibly never-returning) ncclCommInitRank and touches no context: it is what a helper thread with a
    deadline calls; ``attach(ctx)`` on the main thread then hands the communicator to the context, ``drop()`` releases
    one that will not be used (bring_up)."""

    def __init__(self, ctx, rank, world, unique_id=None, _pending=None):
        self.rank, self.world = int(rank), int(world)
        self.ctx, self._id_file = None, None
        self.control = None                                   # bring_up: the TCP control plane that came up before RCCL
        self._pending = _pending
        if _pending is None:
            pending = RcclComm.initialise(ctx.device, rank, world, ctx.comm_unique_id, unique_id)
            self._pending, self._id_file = pending._pending, pending._id_file
            self.attach(ctx)

    @classmethod
    def initialise(cls, device, rank, world, make_id, unique_id=None):
        """-> an RcclComm that owns a communicator but no context yet."""
        id_file = None
        if unique_id is None:
            unique_id, id_file = _exchange_unique_id(int(rank), int(world), make_id)
        # librccl prints a version banner on C stdout during init; benchmark drivers parse stdout, so route file
        # descriptor 1 to stderr for the duration of the call
        import sys
        from . import _capi
        sys.stdout.flush()
        flush_c_streams()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            comm = _capi.DeviceContext.comm_init_detached(device, unique_id, rank, world)   # collective
            flush_c_streams()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        self = cls(None, rank, world, _pending=comm)
        self._id_file = id_file
        return self

    def attach(self, ctx):
        ctx.comm_adopt(self._pending, self.rank, self.world)
        self.ctx, self._pending = ctx, None
        return self

    def drop(self):
        """Release a communicator that was never attached."""
        if self._pending is not None:
            from . import _capi
            _capi.DeviceContext.comm_release_detached(self._pending)
            self._pending = None

    def allgather(self, values):
        return self.ctx.allgather(np.asarray(values, dtype=np.float64).reshape(-1), self.world)

    def allreduce_sum(self, values):
        return self.ctx.allreduce_sum(values)

    def barrier(self):
        self.ctx.allreduce_sum(np.zeros(1))
        self.ctx.sync()

    def close(self, destroy=True):
        """destroy=False: leave the communicator alone (a collective on it never returned: ncclCommDestroy would wait)."""
        if self.control is not None:
            self.control.close()
            self.control = None
        if self.ctx is not None and destroy:
            self.ctx.comm_destroy()
        self.ctx = None
        if destroy:
            self.drop()
        flush_c_streams()


class TcpComm:
    """
    Control-plane collectives of a launch over plain TCP sockets (rank 0 is the hub): barrier and all-gather of a few
    doubles, nothing else.  The benchmark brings it up before RCCL so that (a) every rank learns whether RCCL came up
    on ALL ranks before anybody calls an RCCL collective, and (b) a launch of independent fits -- which has no data-path
    exchange at all -- can still be timed and reported if RCCL cannot initialise on the node.
    """

    def __init__(self, rank, world, tag='fokl_tcp', timeout_s=300.0):
        import socket
        import struct
        import time
        self.rank, self.world = int(rank), int(world)
        self._struct, self._peers, self._hub = struct, [], None
        if self.world == 1:
            return
        path = _rendezvous_path(tag)
        host = os.environ.get('MASTER_ADDR', '127.0.0.1')
        deadline = time.monotonic() + timeout_s
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                try:
                    srv.bind((host, 0))
                except OSError:
                    srv.bind(('127.0.0.1', 0))
                srv.listen(self.world)
                try:
                    os.unlink(path)
                except FileNotFoundError:
                    pass
                fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
                with os.fdopen(fd, 'w') as fh:
                    fh.write(f'{srv.getsockname()[1]}\n')
                peers = {}
                while len(peers) < self.world - 1:
                    srv.settimeout(max(0.1, deadline - time.monotonic()))
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        raise TimeoutError(f"rank 0: only {len(peers)} of {self.world - 1} ranks joined within "
                                           f"{timeout_s:.0f} s")
                    conn.settimeout(10.0)
                    try:
                        hello = self._recv(conn, 8)
                    except (ConnectionError, OSError):        # a client that connects and goes away is not a rank
                        conn.close()
                        continue
                    peer = struct.unpack('<i', hello[4:])[0] if hello[:4] == b'FTCP' else 0
                    if 1 <= peer < self.world and peer not in peers:
                        conn.settimeout(timeout_s)
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        peers[peer] = conn
                    else:
                        conn.close()
                self._peers = [peers[r] for r in range(1, self.world)]
            finally:
                srv.close()
                try:
                    os.unlink(path)
                except OSError:
                    pass
            return
        while True:
            try:
                st = os.stat(path)
                if st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0:
                    with open(path) as fh:
                        port = int(fh.read().strip() or 0)
                    conn = socket.create_connection((host, port), timeout=5.0)
                    conn.settimeout(timeout_s)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.sendall(b'FTCP' + struct.pack('<i', self.rank))
                    self._hub = conn
                    return
            except (FileNotFoundError, ValueError, ConnectionError, socket.timeout, OSError):
                pass
            if time.monotonic() > deadline:
                raise TimeoutError(f"rank {self.rank}: rank 0 not reachable via {path} after {timeout_s:.0f} s")
            time.sleep(0.02)

    @staticmethod
    def _recv(conn, count):
        buf = b''
        while len(buf) < count:
            part = conn.recv(count - len(buf))
            if not part:
                raise ConnectionError("peer closed the control connection")
            buf += part
        return buf

    def allgather(self, values):
        v = np.asarray(values, dtype=np.float64).reshape(-1)
        if self.world == 1:
            return v[None, :].copy()
        nbytes = v.shape[0] * 8
        if self.rank == 0:
            rows = [v] + [np.frombuffer(self._recv(c, nbytes), dtype=np.float64) for c in self._peers]
            out = np.stack(rows, axis=0)
            blob = out.tobytes()
            for c in self._peers:
                c.sendall(blob)
            return out
        self._hub.sendall(v.tobytes())
        return np.frombuffer(self._recv(self._hub, nbytes * self.world), dtype=np.float64).reshape(self.world, -1).copy()

    def allreduce_sum(self, values):
        v = np.asarray(values, dtype=np.float64)
        return self.allgather(v.reshape(-1)).sum(axis=0).reshape(v.shape)

    def barrier(self):
        self.allgather([0.0])

    def close(self):
        for c in self._peers + ([self._hub] if self._hub is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._hub = [], None


def bring_up(ctx, rank, world, need_rccl, timeout_s=180.0, log=None):
    """
    Communicator for a benchmark / driver process of a `world`-rank launch.  The TCP control plane comes up first; RCCL
    is then initialised on a helper thread with a deadline, and the ranks agree over TCP whether it came up everywhere.
    -> (comm, description).  If it did not: a launch that needs RCCL on its data path (`need_rccl`: rows or candidates
    sharded over ranks) raises on every rank; a launch of independent fits carries on over the control plane alone.
    """
    if world == 1:
        return SingleComm(), 'single process'
    import threading
    tcp = TcpComm(rank, world)
    box = {}
    device = getattr(ctx, 'device', 0)
    initialise = getattr(ctx, 'rccl_initialise', None)        # tests: a stand-in context brings its own (test_dist_tcp)
    if initialise is None:
        initialise = lambda: RcclComm.initialise(device, rank, world, ctx.comm_unique_id)   # noqa: E731

    def attempt():
        # touches nothing but `box`: if the deadline passes this thread is left behind for good, and whatever it does
        # when ncclCommInitRank finally returns (peers exiting, say) must not reach a context the main thread is using
        try:
            box['pending'] = initialise()
        except BaseException as exc:                          # noqa: BLE001 -- reported to every rank below
            box['error'] = f'{type(exc).__name__}: {exc}'

    stdout_fd = os.dup(1)                                     # initialise() parks fd 1 on stderr while librccl starts
    t = threading.Thread(target=attempt, name='fokl-rccl-init', daemon=True)
    t.start()
    t.join(timeout_s)
    timed_out = t.is_alive()                                  # decided first: a thread that finishes after this line
    pending = None if timed_out else box.get('pending')       # is treated as timed out, its communicator never adopted
    if timed_out:
        box.setdefault('error', f'RCCL initialisation did not return within {timeout_s:.0f} s')
        os.dup2(stdout_fd, 1)                                 # the stuck thread will never restore it
    os.close(stdout_fd)
    ok = 1.0 if pending is not None else 0.0
    everywhere = float(np.min(tcp.allgather([ok])[:, 0])) == 1.0
    if everywhere:
        comm = pending.attach(ctx)
        # the control plane stays up next to RCCL: barriers and the gathers of a few timing figures need no device
        # collective, and a launch whose data path has no exchange must not depend on one (RcclComm.close closes it)
        comm.control = tcp
        return comm, 'RCCL'
    if pending is not None:
        pending.drop()                                        # came up here but not everywhere: nothing stays attached
    why = box.get('error', 'RCCL failed on another rank')
    if log is not None:
        log(f"rank {rank}: RCCL not available ({why}); control-plane collectives over TCP")
    if need_rccl:
        tcp.close()
        raise RuntimeError(f"this mode exchanges data through RCCL, which did not come up: {why}")
    return tcp, f'TCP control plane only (RCCL did not come up: {why})'


def shard_range(count, rank, world):
    """Contiguous block partition of ``count`` units: the slice owned by ``rank``."""
    base, extra = divmod(int(count), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
